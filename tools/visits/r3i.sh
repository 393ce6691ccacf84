#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "=== HSA_ENABLE_SDMA=0" > gpurun_out/r3i_stall_probe_sdma.log
HSA_ENABLE_SDMA=0 timeout 300 python tools/stall_probe.py 2>&1 | grep -v "amdgpu.ids" | head -2 >> gpurun_out/r3i_stall_probe_sdma.log
HSA_ENABLE_SDMA=0 timeout 200 python tools/c5_probe.py 2>&1 | grep -v amdgpu.ids | head -1 | cut -c1-600 >> gpurun_out/r3i_stall_probe_sdma.log
echo "=== default" >> gpurun_out/r3i_stall_probe_sdma.log
timeout 300 python tools/stall_probe.py 2>&1 | grep -v "amdgpu.ids" | head -2 >> gpurun_out/r3i_stall_probe_sdma.log
cat gpurun_out/r3i_stall_probe_sdma.log
