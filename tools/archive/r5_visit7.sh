#!/bin/bash
# round 5, visit 7: split-bf16 parity decode, tuned (one round of loads per workgroup, 1/rms from partial sums of squares, RoPE factors from the step's table)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "dec32x" -x > gpurun_out/r5h_tests_kernels.log 2>&1
echo "pytest exit $?" >> gpurun_out/r5h_tests_kernels.log; tail -6 gpurun_out/r5h_tests_kernels.log
timeout 300 python tools/x3_debug_probe.py 2>&1 | grep -v "amdgpu.ids\|incomplete" > gpurun_out/r5h_x3_probe.log; grep "layers 20" gpurun_out/r5h_x3_probe.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider -k "bit_exact or bench_workload or text or stream_chunks or continuous or final_norm" > gpurun_out/r5h_tests_e2e.log 2>&1
echo "pytest exit $?" >> gpurun_out/r5h_tests_e2e.log; tail -8 gpurun_out/r5h_tests_e2e.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ttfs --no-configs --no-slot-pool --no-ids-check --no-bf16-parity --no-roofline --parity-steps 3"
{
for cfg in "X=1" "CTTS_D32X_MB_QKV=4 CTTS_D32X_MB_SILU=4" "CTTS_D32X_MB_QKV=2 CTTS_D32X_MB_SILU=2" "CTTS_D32X_MB_QKV=4" "CTTS_D32X_MB_SILU=2" "CTTS_D32X_MB_O=2 CTTS_D32X_MB_DOWN=2"; do
  echo "== $cfg"
  env $cfg timeout 300 $B 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); pm=j.get('parity_mode',{})
print('parity x3', pm.get('value'), pm.get('ids_match_reference'), 'step_ms', pm.get('decode_ms_per_gpt_step'), '| exact', (pm.get('exact_f32_mfma') or {}).get('value'))"
done
} > gpurun_out/r5h_ab_x3_mb.log 2>&1
cat gpurun_out/r5h_ab_x3_mb.log
