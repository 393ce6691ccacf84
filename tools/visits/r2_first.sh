#!/bin/bash
# round-2 first GPU visit: new parity tests, then packed-vs-row-major decode A/B, then a kernel-stats profile
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider -s \
  -k "gemm_dec_packed or big_tile or baseline_sizes or teacher_forced or stream_chunks or packed_decode or embed_and_final or rope_attention or token_ids_bit_exact or bf16_mode_runs" \
  > gpurun_out/r2a_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/r2a_tests.log
tail -5 gpurun_out/r2a_tests.log
{
for r in 1 2; do
echo "A packed"; python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline 2>/dev/null | tail -1 | cut -c1-200
echo "B row-major"; CTTS_DEC_PACKED=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline 2>/dev/null | tail -1 | cut -c1-200
done
} > gpurun_out/r2a_ab.log 2>&1
cat gpurun_out/r2a_ab.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r2a -o r2a -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs > $R/gpurun_out/r2a_rocprof.log 2>&1
find /tmp/prof_r2a -name "*kernel_stats*.csv" -exec cp {} $R/gpurun_out/r2a_kernel_stats.csv \;
head -12 $R/gpurun_out/r2a_kernel_stats.csv | cut -c1-200
