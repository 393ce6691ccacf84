#!/bin/bash
# documentation runs: BASELINE configs C1/C2/C5, the f32 parity-mode bench line, the continuous-batching probe
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
python tools/configs_run.py > gpurun_out/configs_run.log 2>&1; tail -3 gpurun_out/configs_run.log | cut -c1-300
python bench.py --dtype f32 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-330 | tee gpurun_out/bench_f32.log
python tools/serving_probe.py > gpurun_out/serving_probe.log 2>&1; tail -4 gpurun_out/serving_probe.log | cut -c1-250
