#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/kernarg_probe.hip -o /tmp/kernarg_probe 2>&1 | tail -3
timeout 120 /tmp/kernarg_probe > gpurun_out/r3u_kernarg_probe.log 2>&1; cat gpurun_out/r3u_kernarg_probe.log
echo "--- HIP_FORCE_DEV_KERNARG=0" >> gpurun_out/r3u_kernarg_probe.log; HIP_FORCE_DEV_KERNARG=0 timeout 120 /tmp/kernarg_probe 2>&1 | head -8 | tee -a gpurun_out/r3u_kernarg_probe.log
echo "--- HIP_FORCE_DEV_KERNARG=1" >> gpurun_out/r3u_kernarg_probe.log; HIP_FORCE_DEV_KERNARG=1 timeout 120 /tmp/kernarg_probe 2>&1 | head -8 | tee -a gpurun_out/r3u_kernarg_probe.log
