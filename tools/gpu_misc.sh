#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/misc_tests.log 2>&1; tail -3 gpurun_out/misc_tests.log
timeout 900 python tools/configs_run.py 2>/dev/null | tee gpurun_out/configs.log
