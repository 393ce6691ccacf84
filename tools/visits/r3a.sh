#!/bin/bash
# round 3, visit a: full GPU suite on the new tree, the full bench line, A/B of the compaction order / attention residency bound /
# multi-step graph, the launch floor with and without the dependent load, eager-vs-graph kernel durations under rocprofv3.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r3a
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/${T}_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/${T}_tests.log
tail -5 gpurun_out/${T}_tests.log
grep -h "bf16 vs f32, teacher-forced\|bf16 teacher-forced" gpurun_out/${T}_tests.log
timeout 420 python bench.py --steps 5 --warmup 2 > gpurun_out/${T}_bench.log 2> gpurun_out/${T}_bench.err
echo "bench exit $?" >> gpurun_out/${T}_bench.log
grep -v amdgpu.ids gpurun_out/${T}_bench.err | tail -8
tail -2 gpurun_out/${T}_bench.log | cut -c1-6000
Q="--steps 3 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline --no-parity-mode --no-bf16-parity"
ab() { # label, env...
  L=$1; shift
  echo "== $L" >> gpurun_out/${T}_ab.log
  env "$@" timeout 200 python bench.py $Q 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/${T}_ab.log 2>&1
}
for rep in 1 2; do
ab "base (order on, no residency bound)" X=1
ab "CTTS_ORDER=0" CTTS_ORDER=0
ab "CTTS_ATT_LDS=65536 (2 workgroups per CU)" CTTS_ATT_LDS=65536
ab "CTTS_ATT_LDS=65536 CTTS_ORDER=0" CTTS_ATT_LDS=65536 CTTS_ORDER=0
ab "CTTS_ATT_LDS=40000 (4 per CU)" CTTS_ATT_LDS=40000
ab "CTTS_ATT_LDS=98304 (1 per CU)" CTTS_ATT_LDS=98304
ab "CTTS_GRAPH_STEPS=16" CTTS_GRAPH_STEPS=16
done
cat gpurun_out/${T}_ab.log
# per-kernel events with and without the system-scope fence on the stop event
for F in 0 1; do
  echo "== CTTS_PROF_SYSFENCE=$F" >> gpurun_out/${T}_evfence.log
  CTTS_PROF_SYSFENCE=$F timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ttfs --no-parity-mode --no-bf16-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: v['avg_launch_us'] for k, v in d['decode_kernels'].items()})" >> gpurun_out/${T}_evfence.log 2>&1
done
cat gpurun_out/${T}_evfence.log
# launch floor: normal build (kernels exit after ONE dependent load) vs probe build (kernels exit without touching memory)
timeout 200 python tools/step_floor_probe.py > gpurun_out/${T}_step_floor.log 2>&1
echo "-- probe build (CTTS_PROBE_EXIT: no memory access at all)" >> gpurun_out/${T}_step_floor.log
CTTS_LIB=$R/chattts_amd/csrc/libchattts_amd_probe.so timeout 200 python tools/step_floor_probe.py >> gpurun_out/${T}_step_floor.log 2>&1
grep -v amdgpu.ids gpurun_out/${T}_step_floor.log
# rocprofv3 kernel trace: graph replay vs eager launches of the same pass
cd /tmp
for M in graph eager; do
  X=""; [ $M = eager ] && X="--no-graph"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${T}_$M -o ${T}_$M -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode --no-bf16-parity $X > $R/gpurun_out/${T}_rocprof_$M.log 2>&1
  F=$(find /tmp/prof_${T}_$M -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp $F $R/gpurun_out/${T}_kernel_stats_$M.csv && head -8 $F | cut -c1-160
done
cd "$R"
