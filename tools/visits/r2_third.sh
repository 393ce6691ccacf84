#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python tools/dec_phase_probe.py 64 > gpurun_out/r2c_dec_phase_probe.log 2>&1
timeout 100 python tools/dec_phase_probe.py 40 >> gpurun_out/r2c_dec_phase_probe.log 2>&1
cat gpurun_out/r2c_dec_phase_probe.log | grep -v amdgpu.ids
timeout 420 python bench.py --steps 3 --warmup 1 > gpurun_out/r2c_bench.log 2> gpurun_out/r2c_bench.err
echo "bench exit $?" >> gpurun_out/r2c_bench.log
cat gpurun_out/r2c_bench.err | grep -v amdgpu.ids | tail -12
tail -2 gpurun_out/r2c_bench.log | cut -c1-3000
