#!/bin/bash
# round 3, visit p: software pipelining across batches (acoustic decode of batch i overlapping the generation of batch i+1)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r3p
timeout 300 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x --tb=short -p no:cacheprovider -k "pipelined or decode_to_wavs or chat_facade or stream_chunks" 2>&1 | tail -3
Q="--steps 6 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline --no-parity-mode --no-bf16-parity"
for rep in 1 2; do
for X in "--pipeline" ""; do
  echo "== bench $X" >> gpurun_out/${T}_pipeline_ab.log
  timeout 200 python bench.py $Q $X 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['pipelined'])" >> gpurun_out/${T}_pipeline_ab.log 2>&1
done
done
cat gpurun_out/${T}_pipeline_ab.log
