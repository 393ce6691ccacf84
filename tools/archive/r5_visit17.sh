#!/bin/bash
# round 5, visit 17: the sampling-parameter-space goldens + the 160-utterance batch (generate_params.npz) on the GPU
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider -k "parameter_space or tail_shard or wide_batch or stream_yields or unexpected_end or random_sweep or decode_to_wavs_padding or default_max_new_token" > gpurun_out/r5v_tests_params.log 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r5v_tests_params.log | tail -40
