#!/bin/bash
# visit 18: batched RMS prologue in decode32 (pinned arithmetic): bit-identity + goldens + f32 bench + trace
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x -k "dec32 or gemm_skinny" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -x -k "f32 or golden or bit or exact or parity or baseline or bench" 2>&1 | tail -2
B="python $R/bench.py --dtype f32 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode"
for kv in "X=0" "CTTS_D32_MB_QKV=2" "CTTS_D32_MB_SILU=2"; do
  echo "$kv: $(env $kv timeout 200 $B 2>/dev/null | tail -1 | cut -c60-140)"
done | tee gpurun_out/r2r_f32_bench.log
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_f32 -o f32 --output-format csv -- $B > /dev/null 2>&1
f=$(find /tmp/prof_f32 -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r2r_f32_packed_kernel_stats.csv; head -8 "$f" | cut -c1-200
