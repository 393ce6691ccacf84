#!/bin/bash
# visit 15: where does the f32 parity mode's decode step go?  kernel trace of the C3 workload in f32 mode
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_f32 -o f32 --output-format csv -- python $R/bench.py --dtype f32 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode > $R/gpurun_out/r2o_bench_f32.log 2>&1
tail -1 $R/gpurun_out/r2o_bench_f32.log | cut -c1-300
f=$(find /tmp/prof_f32 -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r2o_f32_kernel_stats.csv; head -30 "$f" | cut -c1-220
