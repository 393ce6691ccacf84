#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 700 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider -x -s \
  -k "remainder_split or attention_split or gemm_dec_packed or packed_decode or bf16 or full_size or teacher or bench_workload or baseline_sizes or continuous or rope_attention" > gpurun_out/r2e_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/r2e_tests.log
grep -E "split vs whole|passed|failed|Error|error" gpurun_out/r2e_tests.log | tail -8
for r in 1 2; do
echo "A split=1"; timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline --no-parity-mode 2>/dev/null | tail -1 | cut -c1-200
echo "B split=0"; CTTS_ATT_SPLIT=0 timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline --no-parity-mode 2>/dev/null | tail -1 | cut -c1-200
done | tee gpurun_out/r2e_ab_split.log
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r2e -o r2e -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode > $R/gpurun_out/r2e_rocprof.log 2>&1
find /tmp/prof_r2e -name "*kernel_stats*.csv" -exec cp {} $R/gpurun_out/r2e_kernel_stats.csv \;
head -7 $R/gpurun_out/r2e_kernel_stats.csv | cut -c1-150
