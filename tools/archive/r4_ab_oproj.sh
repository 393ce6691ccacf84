#!/bin/bash
# round 4: attention + o_proj in one launch (CTTS_ATT_OPROJ, default on) vs separate launches, interleaved; then the per-kernel numbers
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
F="--steps 3 --warmup 1 --no-cpu-baseline --no-ttfs --no-parity-mode --no-slot-pool --no-configs --no-bf16-parity"
pick() { python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=r.get('decode_kernels',{})
print(r['value'], r['ms_per_step'], r.get('roofline',{}).get('whole_decode_step',{}).get('ms_per_step'), {n:v['avg_launch_us'] for n,v in k.items()})
"; }
for r in 1 2; do
  echo "fused   :" $(python bench.py $F 2>/dev/null | pick)
  echo "separate:" $(CTTS_ATT_OPROJ=0 python bench.py $F 2>/dev/null | pick)
done
