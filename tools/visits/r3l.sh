#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/hostcopy_probe.py 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r3l_hostcopy_probe.log; cat gpurun_out/r3l_hostcopy_probe.log
