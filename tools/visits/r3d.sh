#!/bin/bash
# round 3, visit d: is the C5 slowdown code or environment?  the round-2 tree (30e661d) under the same probe on today's box
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r3d
echo "=== round-2 tree (30e661d)" > gpurun_out/${T}_c5_old_vs_new.log
(cd _old_r2 && timeout 200 python tools/c5_probe.py 2>&1 | grep -v amdgpu.ids) >> gpurun_out/${T}_c5_old_vs_new.log
echo "=== current tree" >> gpurun_out/${T}_c5_old_vs_new.log
timeout 200 python tools/c5_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/${T}_c5_old_vs_new.log
cut -c1-700 gpurun_out/${T}_c5_old_vs_new.log
