#!/bin/bash
# visit 29: heads GEMM grid padded to a multiple of 8 tiles (one XCD per weight tile): tests, heads PMC traffic, bench
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x -k "dec32" 2>&1 | tail -1
timeout 300 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -x -k "token_ids_bit_exact or packed or refine_text_mode" 2>&1 | tail -1
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ttfs --no-parity-mode"
timeout 200 $B 2>/dev/null | tail -1 > gpurun_out/r2x_heads_xcd_bench.log; grep -o '"value": [0-9.]*' gpurun_out/r2x_heads_xcd_bench.log | head -1; grep -o '"heads_gemm": {[^}]*' gpurun_out/r2x_heads_xcd_bench.log
