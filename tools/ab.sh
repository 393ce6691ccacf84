#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{
echo "skip_finished=1"; python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline 2>/dev/null | tail -1 | cut -c1-160
echo "skip_finished=0"; CTTS_SKIP_FINISHED=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline 2>/dev/null | tail -1 | cut -c1-160
} | tee gpurun_out/skip_ab.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
