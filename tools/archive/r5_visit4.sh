#!/bin/bash
# round 5, visit 4: back end on the device (float_to_int16, Chat pcm16), split-bf16 ARITHMETIC emulated in the f32 decode kernels (emux3 build)
# against every reference golden, cold-start legs of bench.py
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_backend.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r5d_tests_backend.log 2>&1
echo "pytest exit $?" >> gpurun_out/r5d_tests_backend.log; tail -5 gpurun_out/r5d_tests_backend.log
EMU=$PWD/chattts_amd/csrc/libchattts_amd_emux3.so
{
echo "== emux3 build: bench workload (85,752 draws) vs the reference golden"
CTTS_LIB=$EMU timeout 600 python tools/x3_sensitivity_probe.py 24 2>&1 | grep -v amdgpu.ids
echo "== emux3 build: the reference-generated e2e goldens of the f32 parity mode"
CTTS_LIB=$EMU timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=line -p no:cacheprovider -k "bit_exact or bench_workload or refine or text_ids or stream_chunks or continuous" 2>&1 | tail -15
} > gpurun_out/r5d_x3_emulation.log 2>&1
tail -25 gpurun_out/r5d_x3_emulation.log
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode --no-configs --no-slot-pool --no-ids-check --no-roofline > gpurun_out/r5d_bench_cold.log 2>&1
tail -1 gpurun_out/r5d_bench_cold.log | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print({k:j.get(k) for k in ('value','ttfs_ms_p50','ttfs_ms_cold','ttfs_ms_cold_prewarmed','ttfs_cold')})"
