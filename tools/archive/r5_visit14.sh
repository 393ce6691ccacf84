#!/bin/bash
# round 5, visit 14: waves per workgroup of the K = 768 decode projections (bf16 perf mode): 4 (shipped) / 6 / 8 / 12
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/chattts_amd/csrc
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-ttfs --no-configs --no-slot-pool --no-ids-check --no-bf16-parity --no-parity-mode"
{
for r in 1 2; do
for v in "" _nw6 _nw8 _nw12; do
  echo "== build '$v'"
  CTTS_LIB=$L/libchattts_amd$v.so timeout 300 $B 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); dk=j.get('decode_kernels',{}); ws=(j.get('roofline') or {}).get('whole_decode_step',{})
print('bf16',j['value'],'step_ms',ws.get('ms_per_step'),{k:v['avg_launch_us'] for k,v in dk.items() if k in ('qkv_gemm','o_proj_gemm','gate_up_gemm','down_gemm','attention')})"
done
done
} > gpurun_out/r5r_ab_nw768.log 2>&1; cat gpurun_out/r5r_ab_nw768.log
