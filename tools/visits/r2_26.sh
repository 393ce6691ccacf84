#!/bin/bash
# visit 26: decode32 RMSNorm launches with the statistics taken from the MFMA fragments (no re-read of the residual rows)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x -k "dec32" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -x -k "f32 or golden or bit or exact or parity or baseline or bench" 2>&1 | tail -2
B="python $R/bench.py --dtype f32 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode"
for kv in "CTTS_D32_RMS16=0" "CTTS_D32_RMS16=1" "CTTS_D32_RMS16=0" "CTTS_D32_RMS16=1"; do
  echo "$kv: $(env $kv timeout 200 $B 2>/dev/null | tail -1 | cut -c60-140)"
done | tee gpurun_out/r2x_d32_rms16_ab.log
