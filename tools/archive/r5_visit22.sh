#!/bin/bash
# round 5, visit 22: the acoustic decoder on the short-length reference goldens (codec.npz s1x1 .. s3x33), both dense-layer modes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider -s -k "codec_vs_reference_golden or decode_window" > gpurun_out/r5ad_tests_codec_short.log 2>&1
grep "^codec\[\|passed\|failed\|Error" gpurun_out/r5ad_tests_codec_short.log | tail -30
