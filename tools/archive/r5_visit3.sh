#!/bin/bash
# round 5, visit 3: full GPU suite (DPP attention default, persistent grid opt-in, BASELINE-size codec goldens), split-bf16 sensitivity probe
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/r5c_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/r5c_tests.log; tail -4 gpurun_out/r5c_tests.log
grep -n "codec\[" gpurun_out/r5c_tests.log | tail -12
timeout 600 python tools/x3_sensitivity_probe.py 16 20 24 > gpurun_out/r5c_x3_sensitivity.log 2>&1
cat gpurun_out/r5c_x3_sensitivity.log | tail -5
