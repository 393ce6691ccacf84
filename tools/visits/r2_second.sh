#!/bin/bash
# round-2 second GPU visit: full GPU suite, the new bench line, prefetch A/B, kernel stats with prefetch
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
lscpu | grep -E "Model name|^CPU\(s\)" > gpurun_out/r2b_env.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s -x > gpurun_out/r2b_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/r2b_tests.log
tail -4 gpurun_out/r2b_tests.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/r2b_bench.log 2>&1
echo "bench exit $?" >> gpurun_out/r2b_bench.log
tail -2 gpurun_out/r2b_bench.log | cut -c1-1500
{
for r in 1 2; do
echo "A prefetch=0"; python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline --no-parity-mode 2>/dev/null | tail -1 | cut -c1-200
echo "B prefetch=1"; CTTS_PREFETCH=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline --no-parity-mode 2>/dev/null | tail -1 | cut -c1-200
done
} > gpurun_out/r2b_ab.log 2>&1
cat gpurun_out/r2b_ab.log
cd /tmp
CTTS_PREFETCH=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r2b -o r2b -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode > $R/gpurun_out/r2b_rocprof.log 2>&1
find /tmp/prof_r2b -name "*kernel_stats*.csv" -exec cp {} $R/gpurun_out/r2b_kernel_stats_prefetch.csv \;
head -8 $R/gpurun_out/r2b_kernel_stats_prefetch.csv | cut -c1-160
