#!/bin/bash
# round 4: TTFS / streaming / batch-1 legs under environment switches.  usage: r4_ab_ttfs.sh "NAME=ENV..." ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
F="--steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode --no-slot-pool --no-bf16-parity --no-roofline --no-ids-check"
pick() { python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=r.get('configs',{})
print(r['value'], 'ttfs64', r.get('ttfs_ms_p50'), 'C5', c.get('C5',{}).get('ttfs_ms_p50'), c.get('C5',{}).get('total_ms_p50'), 'C2 wall', c.get('C2',{}).get('wall_ms'), c.get('C2',{}).get('gpt_ms'))
"; }
for r in 1 2; do
  for spec in "$@"; do
    name="${spec%%=*}"; envs="${spec#*=}"
    echo "$name :" $(env $envs python bench.py $F 2>/dev/null | pick)
  done
done
