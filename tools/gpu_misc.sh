#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider -k "facade or stream" > gpurun_out/misc_tests.log 2>&1; tail -3 gpurun_out/misc_tests.log
timeout 900 python tools/configs_run.py 2>/dev/null | tee gpurun_out/configs.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ttfs 2>/dev/null | tail -1 | cut -c 700-1500
