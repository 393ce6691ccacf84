#!/bin/bash
# visit 24: f32 attention with 8 waves per unit (A/B; NOT bit-identical to 4 waves: the key split over waves changes the merge order)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
B="python $R/bench.py --dtype f32 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode"
for kv in "CTTS_ATT32_NW=4" "CTTS_ATT32_NW=8" "CTTS_ATT32_NW=4" "CTTS_ATT32_NW=8"; do
  echo "$kv: $(env $kv timeout 200 $B 2>/dev/null | tail -1 | cut -c60-140)"
done | tee gpurun_out/r2x_att32_nw_ab.log
