#!/bin/bash
# visit 22: bf16 decode, 16-row workgroups (o / down) requesting their activation tile before the live-row count: A/B on one box
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x -k "gemm_dec_packed" 2>&1 | tail -1
timeout 300 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -x -k "packed_decode or teacher" 2>&1 | tail -1
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode"
for kv in "CTTS_DEC_A_EARLY=0" "CTTS_DEC_A_EARLY=1" "CTTS_DEC_A_EARLY=0" "CTTS_DEC_A_EARLY=1"; do
  echo "$kv: $(env $kv timeout 200 $B 2>/dev/null | tail -1 | cut -c60-140)"
done | tee gpurun_out/r2v_a_early_ab.log
