#!/bin/bash
# round 5, visit 21: the sampling kernel on the 36 seeded random reference-generated cases (sampling.npz rnd00..rnd35)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "sample_vs_reference_golden" > gpurun_out/r5ac_tests_sampling.log 2>&1
grep -v "^E    +" gpurun_out/r5ac_tests_sampling.log | tail -25
