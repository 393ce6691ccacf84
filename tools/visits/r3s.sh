#!/bin/bash
# round 3, visit s: cross-kernel weight prefetch (CTTS_PF bit mask: 1 QKV->o_proj, 2 attention->gate/up, 4 gate/up->down, 8 gate/up->next QKV/heads)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r3s
timeout 300 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider -k "bf16 or gemm_dec_packed or attention or full_size or packed_decode" 2>&1 | tail -2
Q="--steps 4 --warmup 1 --no-cpu-baseline --no-ttfs --no-parity-mode --no-bf16-parity"
ab() { L=$1; shift; echo "== $L" >> gpurun_out/${T}_ab.log
  env "$@" timeout 200 python bench.py $Q 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k: v['avg_launch_us'] for k, v in d['decode_kernels'].items()})" >> gpurun_out/${T}_ab.log 2>&1; }
for rep in 1 2; do
ab "CTTS_PF=0 (off)" CTTS_PF=0
ab "CTTS_PF=15 (all)" CTTS_PF=15
ab "CTTS_PF=1" CTTS_PF=1
ab "CTTS_PF=2" CTTS_PF=2
ab "CTTS_PF=4" CTTS_PF=4
ab "CTTS_PF=8" CTTS_PF=8
ab "CTTS_PF=13 (no attention prefetch)" CTTS_PF=13
done
cat gpurun_out/${T}_ab.log
