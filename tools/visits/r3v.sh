#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r3v
Q="--steps 4 --warmup 1 --no-cpu-baseline --no-ttfs --no-parity-mode --no-bf16-parity"
ab() { L=$1; shift; echo "== $L" >> gpurun_out/${T}_ab.log
  env "$@" timeout 200 python bench.py $Q 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k: v['avg_launch_us'] for k, v in d['decode_kernels'].items()})" >> gpurun_out/${T}_ab.log 2>&1; }
for rep in 1 2; do
ab "CTTS_PF=0 (off)" CTTS_PF=0
ab "CTTS_PF=1 (QKV -> o_proj)" CTTS_PF=1
ab "CTTS_PF=16 (QKV -> gate/up)" CTTS_PF=16
ab "CTTS_PF=17 (QKV -> o_proj + gate/up)" CTTS_PF=17
done
cat gpurun_out/${T}_ab.log
