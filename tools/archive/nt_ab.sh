#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{
echo "W nt=1"; python tools/gemm_chain_probe.py 2>/dev/null | grep cold
echo "W nt=0"; CTTS_W_NT=0 python tools/gemm_chain_probe.py 2>/dev/null | grep cold
echo "bench nt=1"; python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline 2>/dev/null | tail -1 | cut -c1-160
echo "bench W nt=0"; CTTS_W_NT=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline 2>/dev/null | tail -1 | cut -c1-160
} | tee gpurun_out/nt_ab.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "rope_attention or gemm_fast or qkv_rope" 2>&1 | tail -2
