#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider -x -s \
  -k "prefill_mfma or long_audio_prompt or chunked_prefill or rope_attention or qkv_rope" > gpurun_out/r2h_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/r2h_tests.log
grep -E "long audio|passed|failed|Error|error|assert" gpurun_out/r2h_tests.log | tail -12
