#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider -x -s \
  -k "decode_window or chunked_prefill or stream or facade or full_size or packed_decode" > gpurun_out/r2f_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/r2f_tests.log
grep -E "passed|failed|Error|error" gpurun_out/r2f_tests.log | tail -6
timeout 400 python tools/configs_run.py 2>&1 | grep "^{" > gpurun_out/r2f_configs_C1_C2_C5.log
cat gpurun_out/r2f_configs_C1_C2_C5.log
