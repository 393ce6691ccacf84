#!/bin/bash
# round 5, visit 9: the attention merge on DPP / permlane swaps -- same bits as the LDS shuffles (two builds), A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/chattts_amd/csrc
{
echo "== default build (DPP + permlane swaps)"; python tools/att_bits_check.py 2>&1 | grep sha256
echo "== -DCTTS_ATT_DPP=0 build (xor shuffles through LDS)"; CTTS_LIB=$L/libchattts_amd_nodpp.so python tools/att_bits_check.py 2>&1 | grep sha256
} > gpurun_out/r5k_att_bits.log 2>&1; cat gpurun_out/r5k_att_bits.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "attention" > gpurun_out/r5k_tests_attention.log 2>&1; tail -2 gpurun_out/r5k_tests_attention.log
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-ttfs --no-configs --no-slot-pool --no-ids-check --no-bf16-parity --parity-steps 3"
{
for r in 1 2; do
for cfg in "CTTS_LIB=$L/libchattts_amd_nodpp.so" "X=1"; do
  echo "== $cfg"
  env $cfg timeout 300 $B 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
dk=j.get('decode_kernels',{}); ws=(j.get('roofline') or {}).get('whole_decode_step',{}); pm=j.get('parity_mode') or {}; pr=(pm.get('roofline') or {})
print('bf16',j['value'],'step_ms',ws.get('ms_per_step'),'att_us',(dk.get('attention') or {}).get('avg_launch_us'),'| f32',pm.get('value'),pm.get('ids_match_reference'),'step_ms',pm.get('decode_ms_per_gpt_step'),'att_us',pr.get('avg_launch_us'))"
done
done
} > gpurun_out/r5k_ab_att_merge.log 2>&1; cat gpurun_out/r5k_ab_att_merge.log
