#!/bin/bash
# visit 25: decode32 rows-per-workgroup again, with the early activation round in place
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
B="python $R/bench.py --dtype f32 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode"
for kv in "X=0" "CTTS_D32_MB_SILU=2" "CTTS_D32_MB_QKV=2" "CTTS_D32_MB_DOWN=2" "X=1"; do
  echo "$kv: $(env $kv timeout 200 $B 2>/dev/null | tail -1 | cut -c60-140)"
done | tee gpurun_out/r2x_d32_mb_sweep2.log
