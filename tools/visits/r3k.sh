#!/bin/bash
# round 3, visit k: polls and the final D2H as shader copies into pinned memory -- the streaming config and the bench again
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r3k
timeout 200 python tools/c5_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-700 > gpurun_out/${T}_c5_probe.log; cat gpurun_out/${T}_c5_probe.log
timeout 200 python tools/stall_probe.py 2>&1 | grep -v amdgpu.ids | head -2
timeout 300 python tools/configs_run.py 2>&1 | grep -v "amdgpu.ids\|incomplete" > gpurun_out/${T}_configs.log; cut -c1-400 gpurun_out/${T}_configs.log
for E in "X=1" "CTTS_D2H_SHADER=0"; do
env $E timeout 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --no-parity-mode --no-bf16-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], 'ttfs', d['ttfs_ms_p50'])" | tee -a gpurun_out/${T}_bench_quick.log
done
timeout 400 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
