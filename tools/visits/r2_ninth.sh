#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider -x -s \
  -k "gemm_x3p or lds_dma or dwconv or codec_vs or decode_window or decode_to_wavs or big_tile" > gpurun_out/r2i_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/r2i_tests.log
grep -E "LDS-DMA|passed|failed|Error|error|assert" gpurun_out/r2i_tests.log | tail -12
for r in 1 2; do
echo "A x3p"; timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline --no-parity-mode 2>/dev/null | tail -1 | cut -c1-200
echo "B tiles"; CTTS_X3P_MIN_ROWS=0 timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline --no-parity-mode 2>/dev/null | tail -1 | cut -c1-200
done | tee gpurun_out/r2i_ab_x3p.log
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r2i -o r2i -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode > $R/gpurun_out/r2i_rocprof.log 2>&1
find /tmp/prof_r2i -name "*kernel_stats*.csv" -exec cp {} $R/gpurun_out/r2i_kernel_stats.csv \;
grep -E "x3p|bf16x3|dwconv|attention_k<unsigned short, 4" $R/gpurun_out/r2i_kernel_stats.csv | cut -c1-160
