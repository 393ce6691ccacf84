#!/bin/bash
# codec GEMM iteration: kernel parity tests, phase probe, codec pass time.  The probe needs a second build of the library with
# -DCTTS_X3_PROBE (all csrc/*.hip compiled with that define, linked as chattts_amd/csrc/libchattts_amd_probe.so).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
T=${1:-x3}
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q -x -k "gemm or codec or decode_to_wavs" -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1; tail -2 gpurun_out/${T}_tests.log
CTTS_LIB=$PWD/chattts_amd/csrc/libchattts_amd_probe.so python tools/x3_phase_probe.py 2>&1 | grep "^N" | tee gpurun_out/${T}_phase.log
python tools/codec_probe.py 2>&1 | grep "codec ms" | tee gpurun_out/${T}_codec.log
CTTS_X3_TILE=256 python tools/codec_probe.py 2>&1 | grep "codec ms" | sed 's/^/single-buffer: /' | tee -a gpurun_out/${T}_codec.log
