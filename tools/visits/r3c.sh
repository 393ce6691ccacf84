#!/bin/bash
# round 3, visit c: what slowed BASELINE config C5 (streaming, batch 16) down; the sampling kernel after the penalty-stage fix
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r3c
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "sample or device_generator" > gpurun_out/${T}_tests_sample.log 2>&1
tail -3 gpurun_out/${T}_tests_sample.log
for E in "X=1" "CTTS_DEC_A_EARLY=0" "CTTS_GRAPH_STEPS=1" "CTTS_FNORM_FUSE=0" "CTTS_ORDER=0"; do
  echo "=== $E" >> gpurun_out/${T}_c5_probe.log
  env $E timeout 200 python tools/c5_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/${T}_c5_probe.log
done
cat gpurun_out/${T}_c5_probe.log | cut -c1-900
Q="--steps 3 --warmup 1 --no-cpu-baseline --no-ttfs --no-parity-mode --no-bf16-parity"
timeout 200 python bench.py $Q 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k: v['avg_launch_us'] for k, v in d['decode_kernels'].items()})" | tee gpurun_out/${T}_bench_quick.log
