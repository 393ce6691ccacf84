#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q -x -p no:cacheprovider > gpurun_out/prefill_tests.log 2>&1; tail -2 gpurun_out/prefill_tests.log
{
for r in 1 2; do
echo "tiled"; python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ttfs_ms_p50'])"
echo "gemm_fast"; CTTS_PREFILL_TILED=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ttfs_ms_p50'])"
done
} | tee gpurun_out/prefill_ab.log
