#!/bin/bash
# round 5, visit 12: decode chunks on graphs captured for a bound on the live rows -- bits, A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider -k "bounded_row or bit_exact or bench_workload or stream or continuous or interrupt" > gpurun_out/r5o_tests.log 2>&1; tail -6 gpurun_out/r5o_tests.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ttfs --no-configs --no-slot-pool --no-bf16-parity --parity-steps 3"
{
for r in 1 2; do
for cfg in "CTTS_GRAPH_ROWS=0" "CTTS_GRAPH_ROWS=1"; do
  echo "== $cfg"
  env $cfg timeout 300 $B 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
dk=j.get('decode_kernels',{}); ws=(j.get('roofline') or {}).get('whole_decode_step',{}); pm=j.get('parity_mode') or {}
print('bf16',j['value'],'ms',j['ms_per_step'],'step_ms',ws.get('ms_per_step'),'ids_check',j.get('ids_check',{}).get('graph_equals_eager'),'| f32',pm.get('value'),pm.get('ids_match_reference'),'step_ms',pm.get('decode_ms_per_gpt_step'))"
done
done
} > gpurun_out/r5o_ab_graph_rows.log 2>&1; cat gpurun_out/r5o_ab_graph_rows.log
