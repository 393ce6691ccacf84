#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
for M in 2 8 15; do
echo "=== CTTS_PF=$M" >> gpurun_out/r3t_pf_crash.log
CTTS_PF=$M timeout 120 python tools/c2_run.py 1 > gpurun_out/r3t_tmp.out 2> gpurun_out/r3t_tmp.err; echo "exit $?" >> gpurun_out/r3t_pf_crash.log
tail -3 gpurun_out/r3t_tmp.out | cut -c1-300 >> gpurun_out/r3t_pf_crash.log; grep -v amdgpu.ids gpurun_out/r3t_tmp.err | head -12 | cut -c1-300 >> gpurun_out/r3t_pf_crash.log
done
cat gpurun_out/r3t_pf_crash.log
