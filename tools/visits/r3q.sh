#!/bin/bash
# round 3, visit q: acoustic decoder's point-wise GEMM with two k blocks per barrier (CTTS_X3P_VAR=4); decode weights of the first N
# layers with plain loads (CTTS_W_TEMPORAL_LAYERS)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r3q
for V in 1 4; do echo "== CTTS_X3P_VAR=$V" >> gpurun_out/${T}_x3p_probe.log; CTTS_X3P_VAR=$V timeout 120 python tools/x3p_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-110 >> gpurun_out/${T}_x3p_probe.log; done
cat gpurun_out/${T}_x3p_probe.log
CTTS_X3P_VAR=4 timeout 300 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider -k "codec or x3p" 2>&1 | tail -2
Q="--steps 4 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline --no-parity-mode --no-bf16-parity"
ab() { L=$1; shift; echo "== $L" >> gpurun_out/${T}_ab.log
  env "$@" timeout 200 python bench.py $Q 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/${T}_ab.log 2>&1; }
for rep in 1 2; do
ab "base" X=1
ab "CTTS_X3P_VAR=4" CTTS_X3P_VAR=4
ab "CTTS_W_TEMPORAL_LAYERS=8" CTTS_W_TEMPORAL_LAYERS=8
ab "CTTS_W_TEMPORAL_LAYERS=13" CTTS_W_TEMPORAL_LAYERS=13
ab "CTTS_W_TEMPORAL_LAYERS=20" CTTS_W_TEMPORAL_LAYERS=20
done
cat gpurun_out/${T}_ab.log
