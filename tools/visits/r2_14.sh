#!/bin/bash
# visit 14: grid-barrier vs launch-boundary probe; SlotPool on device-side compaction (parity test + throughput probe)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 120 tools/grid_barrier.bin 2>&1 | tee gpurun_out/r2n_grid_barrier.log
timeout 400 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -x -k "continuous or facade" 2>&1 | tail -3 | tee gpurun_out/r2n_slotpool_tests.log
timeout 300 python tools/serving_probe.py 2>&1 | tail -2 | tee gpurun_out/r2n_serving_probe.log
