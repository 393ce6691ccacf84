#!/bin/bash
# round 5, visit 16: final wave counts of the split-bf16 kernels (down_proj on 8 vs 16 waves), kernel + e2e tests of the touched paths
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "dec32x" > gpurun_out/r5t_tests_kernels.log 2>&1; tail -3 gpurun_out/r5t_tests_kernels.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider -k "packed_decode or bit_exact or bench_workload" > gpurun_out/r5t_tests_e2e.log 2>&1; tail -4 gpurun_out/r5t_tests_e2e.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
grep "packed vs row-major" gpurun_out/r5t_tests_e2e.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ttfs --no-configs --no-slot-pool --no-ids-check --no-bf16-parity --no-roofline --parity-steps 5"
{
for r in 1 2 3; do
for cfg in "X=1" "CTTS_D32X_NW_DOWN=16" "CTTS_D32X_NW_DOWN=4"; do
  echo "== $cfg"
  env $cfg timeout 300 $B 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); pm=j.get('parity_mode',{})
print('bf16', j['value'], '| parity x3', pm.get('value'), pm.get('ids_match_reference'), 'step_ms', pm.get('decode_ms_per_gpt_step'))"
done
done
} > gpurun_out/r5t_ab_x3_down_nw.log 2>&1; cat gpurun_out/r5t_ab_x3_down_nw.log
