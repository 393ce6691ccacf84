#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python tools/sample_phase_probe.py 2>&1 | grep -v "amdgpu.ids\|incomplete" > gpurun_out/r3n_sample_phase_probe.log; cat gpurun_out/r3n_sample_phase_probe.log
