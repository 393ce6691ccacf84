#!/bin/bash
# visit 27: heads GEMM on packed f32 operands (both modes): bit-identity tests, A/B
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x -k "dec32" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -x -k "f32 or golden or bit or exact or parity or baseline or bench or text or packed" 2>&1 | tail -2
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode"
for r in 1 2 3; do echo "bf16: $(timeout 200 $B 2>/dev/null | tail -1 | cut -c60-140)"; done | tee gpurun_out/r2x_heads_packed_bench.log
echo "f32: $(timeout 200 $B --dtype f32 2>/dev/null | tail -1 | cut -c60-140)" | tee -a gpurun_out/r2x_heads_packed_bench.log
