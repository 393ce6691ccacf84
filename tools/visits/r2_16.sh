#!/bin/bash
# visit 16: f32 parity mode on fragment-packed operands (decode32.hip): bit-identity tests, then the f32 goldens, then bench f32
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x -k "dec32 or gemm_skinny" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -4
timeout 300 python bench.py --dtype f32 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode 2>/dev/null | tail -1 | cut -c1-400 | tee gpurun_out/r2p_bench_f32.log
