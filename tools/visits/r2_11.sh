#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 120 python tools/x3p_phase_probe.py 2>&1 | grep "^N=" | tee gpurun_out/r2k_x3p_phase_probe.log
