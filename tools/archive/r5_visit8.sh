#!/bin/bash
# round 5, visit 8: split-bf16 parity decode -- loads-per-round variants (builds u24 / uw6 / u24uw6) x rows per workgroup
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ttfs --no-configs --no-slot-pool --no-ids-check --no-bf16-parity --no-roofline --parity-steps 3"
L=$PWD/chattts_amd/csrc
{
for cfg in "X=1" "CTTS_LIB=$L/libchattts_amd_u24.so" "CTTS_LIB=$L/libchattts_amd_uw6.so CTTS_D32X_MB_QKV=2 CTTS_D32X_MB_SILU=2" "CTTS_LIB=$L/libchattts_amd_uw6.so CTTS_D32X_MB_QKV=4 CTTS_D32X_MB_SILU=4" "CTTS_LIB=$L/libchattts_amd_uw6.so CTTS_D32X_MB_QKV=1 CTTS_D32X_MB_SILU=2" "CTTS_LIB=$L/libchattts_amd_u24uw6.so CTTS_D32X_MB_QKV=2 CTTS_D32X_MB_SILU=2" "CTTS_LIB=$L/libchattts_amd_u24uw6.so CTTS_D32X_MB_QKV=1 CTTS_D32X_MB_SILU=2" "CTTS_LIB=$L/libchattts_amd_u24uw6.so CTTS_D32X_MB_QKV=2 CTTS_D32X_MB_SILU=4"; do
  echo "== $cfg"
  env $cfg timeout 300 $B 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); pm=j.get('parity_mode',{})
print('parity x3', pm.get('value'), pm.get('ids_match_reference'), 'step_ms', pm.get('decode_ms_per_gpt_step'), '| exact', (pm.get('exact_f32_mfma') or {}).get('value'))"
done
} > gpurun_out/r5i_ab_x3_rounds.log 2>&1
cat gpurun_out/r5i_ab_x3_rounds.log
