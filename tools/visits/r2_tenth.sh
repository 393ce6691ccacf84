#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 0 1 2 1 0; do echo "CTTS_X3P_VAR=$v"; CTTS_X3P_VAR=$v timeout 120 python tools/x3p_probe.py 2>&1 | grep "^M="; done | tee gpurun_out/r2j_x3p_probe.log
