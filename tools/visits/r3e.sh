#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/c5_sched_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3e_c5_sched_probe.log
echo "=== GPU_MAX_HW_QUEUES=8" >> gpurun_out/r3e_c5_sched_probe.log
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/c5_sched_probe.py 2>&1 | grep -v amdgpu.ids | head -2 >> gpurun_out/r3e_c5_sched_probe.log
cut -c1-400 gpurun_out/r3e_c5_sched_probe.log
