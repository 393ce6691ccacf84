#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/stall_probe.py 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r3h_stall_probe.log
cat gpurun_out/r3h_stall_probe.log
rocm-smi --showclocks --showperflevel 2>/dev/null | head -30 >> gpurun_out/r3h_stall_probe.log
uname -r >> gpurun_out/r3h_stall_probe.log; cat /sys/module/amdgpu/version 2>/dev/null >> gpurun_out/r3h_stall_probe.log
tail -12 gpurun_out/r3h_stall_probe.log
