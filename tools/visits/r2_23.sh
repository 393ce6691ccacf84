#!/bin/bash
# visit 23: decode32 -- first activation round requested before the RMSNorm prologue: A/B on one box
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x -k "dec32" 2>&1 | tail -1
B="python $R/bench.py --dtype f32 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode"
for kv in "CTTS_D32_A_EARLY=0" "CTTS_D32_A_EARLY=1" "CTTS_D32_A_EARLY=0" "CTTS_D32_A_EARLY=1"; do
  echo "$kv: $(env $kv timeout 200 $B 2>/dev/null | tail -1 | cut -c60-140)"
done | tee gpurun_out/r2w_d32_a_early_ab.log
