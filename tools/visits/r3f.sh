#!/bin/bash
# round 3, visit f: polled waits (chattts_amd/_sync.py) vs blocking waits on the streaming config and the bench
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r3f
for E in "CTTS_SPIN_WAIT=1" "CTTS_SPIN_WAIT=0" "CTTS_SPIN_WAIT=0 HSA_ENABLE_INTERRUPT=0"; do
  echo "=== $E" >> gpurun_out/${T}_spin_ab.log
  env $E timeout 200 python tools/c5_probe.py 2>&1 | grep -v amdgpu.ids | head -1 >> gpurun_out/${T}_spin_ab.log
  env $E timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity-mode --no-bf16-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], 'ttfs', d['ttfs_ms_p50'])" >> gpurun_out/${T}_spin_ab.log
done
cut -c1-900 gpurun_out/${T}_spin_ab.log
timeout 300 python tools/configs_run.py 2>&1 | grep -v "amdgpu.ids\|incomplete" > gpurun_out/${T}_configs.log; cut -c1-400 gpurun_out/${T}_configs.log
timeout 200 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -p no:cacheprovider -k "stream or continuous or slot_pool or unseeded or device_generator or interrupt" 2>&1 | tail -3
