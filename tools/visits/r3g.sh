#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python tools/c5_yield_probe.py 2>&1 | grep -v "amdgpu.ids\|incomplete" > gpurun_out/r3g_c5_yield_probe.log
cut -c1-1500 gpurun_out/r3g_c5_yield_probe.log
timeout 200 python tools/c5_probe.py 2>&1 | grep -v amdgpu.ids | head -1 | cut -c1-700
