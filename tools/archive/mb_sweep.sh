#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{
echo "default"; python tools/gemm_chain_probe.py 2>/dev/null | grep cold
for MB in 1 2 4; do
  echo "all MB=$MB"
  CTTS_MB_STORE=$MB CTTS_MB_RES=$MB CTTS_MB_SILU=$MB CTTS_MB_DOWN=$MB python tools/gemm_chain_probe.py 2>/dev/null | grep cold
done
} | tee gpurun_out/mb_sweep.log
