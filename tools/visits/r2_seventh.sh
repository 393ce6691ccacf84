#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python tools/step_floor_probe.py 2>&1 | grep -v "amdgpu.ids\|incomplete" > gpurun_out/r2g_step_floor.log
cat gpurun_out/r2g_step_floor.log
Q="--steps 3 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline --no-parity-mode"
run() { echo "$1"; shift; env "$@" timeout 200 python bench.py $Q $EXTRA 2>/dev/null | tail -1 | cut -c1-160; }
{
run "base"        X=1
run "MB_O=2"      CTTS_DEC_MB_O=2
run "MB_O=4"      CTTS_DEC_MB_O=4
run "MB_DOWN=2"   CTTS_DEC_MB_DOWN=2
run "MB_QKV=2"    CTTS_DEC_MB_QKV=2
run "MB_SILU=2"   CTTS_DEC_MB_SILU=2
run "base"        X=1
EXTRA="--lanes 2" run "lanes=2" X=1
EXTRA="--lanes 2" run "lanes=2" X=1
} > gpurun_out/r2g_knobs.log 2>&1
cat gpurun_out/r2g_knobs.log
