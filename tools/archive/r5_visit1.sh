#!/bin/bash
# round 5, visit 1: persistent attention grid -- kernel parity test, e2e goldens, A/B bench over grid size / ring depth
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "attention" > gpurun_out/r5a_tests_attention.log 2>&1
echo "pytest exit $?" >> gpurun_out/r5a_tests_attention.log; tail -5 gpurun_out/r5a_tests_attention.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r5a_tests_e2e.log 2>&1
echo "pytest exit $?" >> gpurun_out/r5a_tests_e2e.log; tail -5 gpurun_out/r5a_tests_e2e.log
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-ttfs --no-configs --no-slot-pool --no-ids-check --no-bf16-parity --parity-steps 3"
{
for r in 1 2; do
for cfg in "CTTS_ATT_PERSIST=0" "CTTS_ATT_PERSIST=1 CTTS_ATT_D=4" "CTTS_ATT_PERSIST=1 CTTS_ATT_D=3" "CTTS_ATT_PERSIST=1 CTTS_ATT_D=2" "CTTS_ATT_PERSIST=1 CTTS_ATT_D=4 CTTS_ATT_G=512" "CTTS_ATT_PERSIST=1 CTTS_ATT_D=2 CTTS_ATT_G=512" "CTTS_ATT_PERSIST=1 CTTS_ATT_D=4 CTTS_ATT_G=128"; do
  echo "== $cfg"
  env $cfg timeout 300 $B 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
dk=j.get('decode_kernels',{})
ws=(j.get('roofline') or {}).get('whole_decode_step',{})
pm=j.get('parity_mode') or {}
pr=(pm.get('roofline') or {})
print('bf16 value',j['value'],'ms_per_pass',j['ms_per_step'],'step_ms',ws.get('ms_per_step'),'frac',ws.get('frac'),'att_us',(dk.get('attention') or {}).get('avg_launch_us'),'sum_us',ws.get('sum_kernel_us_per_step'))
print('f32 value',pm.get('value'),'ids_ok',pm.get('ids_match_reference'),'step_ms',pm.get('decode_ms_per_gpt_step'),'att_us',pr.get('avg_launch_us'))
"
done
done
} > gpurun_out/r5a_ab_persist.log 2>&1
tail -40 gpurun_out/r5a_ab_persist.log
