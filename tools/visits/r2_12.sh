#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
{ timeout 100 python tools/c2_run.py 2>/dev/null | tail -1
CTTS_ATT_SPLIT=1 timeout 100 python tools/c2_run.py 2>/dev/null | tail -1
timeout 100 python tools/c2_run.py 2>/dev/null | tail -1; } | tee gpurun_out/r2l_c2.log
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -o c2 -- python $R/tools/c2_run.py 1 > $R/gpurun_out/r2l_c2_rocprof.log 2>&1
find /tmp/prof_c2 -name "*kernel_stats*.csv" -exec cp {} $R/gpurun_out/r2l_c2_kernel_stats.csv \;
head -14 $R/gpurun_out/r2l_c2_kernel_stats.csv | cut -c1-130
