#!/bin/bash
# visit 17: kernel trace of the f32 parity mode on packed operands + rows-per-workgroup sweep
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
B="python $R/bench.py --dtype f32 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_f32 -o f32 --output-format csv -- $B > /dev/null 2>&1
f=$(find /tmp/prof_f32 -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r2q_f32_packed_kernel_stats.csv; head -12 "$f" | cut -c1-200
cd $R
for kv in "CTTS_D32_MB_QKV=1" "CTTS_D32_MB_QKV=4" "CTTS_D32_MB_SILU=1" "CTTS_D32_MB_SILU=4" "CTTS_D32_MB_O=2" "CTTS_D32_MB_DOWN=2" "X=0"; do
  echo "$kv: $(env $kv timeout 200 $B 2>/dev/null | tail -1 | cut -c60-140)"
done | tee gpurun_out/r2q_f32_mb_sweep.log
