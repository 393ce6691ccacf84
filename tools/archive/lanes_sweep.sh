#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for L in 1 2 4 8 16; do
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --lanes $L 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lanes',d['config']['lanes_per_gpu'],'value',d['value'],'ms',d['ms_per_step'])" | tee -a gpurun_out/lanes_sweep.log
done
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -5 | tee -a gpurun_out/lanes_sweep.log
