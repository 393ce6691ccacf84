#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r3w
Q="--steps 4 --warmup 1 --no-cpu-baseline --no-ttfs --no-parity-mode --no-bf16-parity"
ab() { L=$1; shift; echo "== $L" >> gpurun_out/${T}_ab.log
  env "$@" timeout 200 python bench.py $Q 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k: v['avg_launch_us'] for k, v in d['decode_kernels'].items()})" >> gpurun_out/${T}_ab.log 2>&1; }
for rep in 1 2; do
ab "4 waves per unit" CTTS_ATT_NW_PACKED=4

ab "2 waves per unit" CTTS_ATT_NW_PACKED=2
done
cat gpurun_out/${T}_ab.log
