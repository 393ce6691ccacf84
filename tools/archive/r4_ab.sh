#!/bin/bash
# round 4: interleaved A/B of environment switches of the same build.  usage: r4_ab.sh "NAME_A=ENV..." "NAME_B=ENV..." (ENV may be empty: "base=")
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
F="--steps 3 --warmup 1 --no-cpu-baseline --no-ttfs --no-parity-mode --no-slot-pool --no-configs --no-bf16-parity"
pick() { python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=r.get('decode_kernels',{})
print(r['value'], r['ms_per_step'], r.get('roofline',{}).get('whole_decode_step',{}).get('ms_per_step'), {n:v['avg_launch_us'] for n,v in k.items()})
"; }
for r in 1 2; do
  for spec in "$@"; do
    name="${spec%%=*}"; envs="${spec#*=}"
    echo "$name :" $(env $envs python bench.py $F 2>/dev/null | pick)
  done
done
