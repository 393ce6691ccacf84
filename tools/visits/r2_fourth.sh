#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider -x \
  -k "gemm_dec_packed or packed_decode or bf16 or full_size or teacher or bench_workload or baseline_sizes" > gpurun_out/r2d_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/r2d_tests.log
tail -3 gpurun_out/r2d_tests.log
timeout 200 python tools/dec_phase_probe.py 64 2>&1 | grep -v amdgpu.ids > gpurun_out/r2d_dec_phase_probe.log
cat gpurun_out/r2d_dec_phase_probe.log
for r in 1 2; do timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline --no-parity-mode 2>/dev/null | tail -1 | cut -c1-200; done | tee gpurun_out/r2d_bench_quick.log
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r2d -o r2d -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode > $R/gpurun_out/r2d_rocprof.log 2>&1
find /tmp/prof_r2d -name "*kernel_stats*.csv" -exec cp {} $R/gpurun_out/r2d_kernel_stats.csv \;
head -6 $R/gpurun_out/r2d_kernel_stats.csv | cut -c1-140
