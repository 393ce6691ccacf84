#!/bin/bash
# round 5, visit 15: 8-wave K = 768 launches in the perf mode (QKV, gate/up) on the final tree; 8-wave variants of the split-bf16 parity kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "dec32x or gemm_dec or qkv_rope" > gpurun_out/r5s_tests_kernels.log 2>&1; tail -3 gpurun_out/r5s_tests_kernels.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider -k "bf16 or drift or packed or bounded" > gpurun_out/r5s_tests_e2e.log 2>&1; tail -4 gpurun_out/r5s_tests_e2e.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-ttfs --no-configs --no-slot-pool --no-ids-check --no-bf16-parity --no-roofline --parity-steps 3"
{
for r in 1 2; do
for cfg in "X=1" "CTTS_D32X_NW_QKV=8 CTTS_D32X_NW_SILU=8" "CTTS_D32X_NW_QKV=8 CTTS_D32X_NW_SILU=8 CTTS_D32X_NW_DOWN=8" "CTTS_D32X_NW_QKV=8 CTTS_D32X_NW_SILU=8 CTTS_D32X_NW_DOWN=8 CTTS_D32X_NW_O=8" "CTTS_D32X_NW_SILU=8"; do
  echo "== $cfg"
  env $cfg timeout 300 $B 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); pm=j.get('parity_mode',{})
print('bf16', j['value'], '| parity x3', pm.get('value'), pm.get('ids_match_reference'), 'step_ms', pm.get('decode_ms_per_gpt_step'))"
done
done
} > gpurun_out/r5s_ab_x3_nw.log 2>&1; cat gpurun_out/r5s_ab_x3_nw.log
