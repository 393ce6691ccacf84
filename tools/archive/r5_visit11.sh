#!/bin/bash
# round 5, visit 11: planes-output attention test; decode lanes re-measured on today's kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "persistent_grid" > gpurun_out/r5m_tests.log 2>&1; tail -3 gpurun_out/r5m_tests.log
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-ttfs --no-configs --no-slot-pool --no-ids-check --no-bf16-parity --no-parity-mode --no-roofline"
{
for r in 1 2; do
for lanes in 1 2 4; do
  echo "== lanes $lanes"
  timeout 300 $B --lanes $lanes 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('bf16', j['value'], 'ms', j['ms_per_step'])"
done
done
} > gpurun_out/r5m_lanes.log 2>&1; cat gpurun_out/r5m_lanes.log
