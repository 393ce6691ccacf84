#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -x -k "stream or facade or decode_to_wavs or full_size or lds_dma" 2>&1 | tail -2
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-ttfs > gpurun_out/r2m_torchrun_world1.log 2>&1
tail -1 gpurun_out/r2m_torchrun_world1.log | cut -c1-400
for r in 1 2 3; do timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline --no-parity-mode 2>/dev/null | tail -1 | cut -c1-180; done | tee gpurun_out/r2m_bench_quick.log
