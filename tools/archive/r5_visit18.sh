#!/bin/bash
# round 5, visit 18: the roofline line with the per-launch fit (duration = fixed + bytes / bandwidth) of the attention kernel, both modes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ttfs --no-configs --no-slot-pool --no-ids-check --no-bf16-parity --parity-steps 2 > gpurun_out/r5w_bench_fit.log 2>&1
grep "^{" gpurun_out/r5w_bench_fit.log | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('bf16', j['value'], json.dumps(j['roofline'].get('fit')))
pm=j.get('parity_mode',{}); print('f32', pm.get('value'), json.dumps(pm.get('roofline',{}).get('fit')))"
tail -3 gpurun_out/r5w_bench_fit.log | cut -c1-300
