#!/bin/bash
# Every GPU-box visit of round 3, one function per visit: `gpurun -- 'bash tools/visits/round3.sh r3a'`.
# Kept as the record of what each profiles/r3* file was measured with (the one-shot scripts of round 2 are in the history:
# git show 30e661d --stat -- tools/visits).  Everything lands in gpurun_out/; what is cited was copied to profiles/.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp

# round 3, visit a: full GPU suite on the new tree, the full bench line, A/B of the compaction order / attention residency bound /
# multi-step graph, the launch floor with and without the dependent load, eager-vs-graph kernel durations under rocprofv3.
# per-kernel events with and without the system-scope fence on the stop event
# launch floor: normal build (kernels exit after ONE dependent load) vs probe build (kernels exit without touching memory)
# rocprofv3 kernel trace: graph replay vs eager launches of the same pass
r3a() {
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
  for F in 0 1; do
    echo "== CTTS_PROF_SYSFENCE=$F" >> gpurun_out/${T}_evfence.log
    CTTS_PROF_SYSFENCE=$F timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ttfs --no-parity-mode --no-bf16-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: v['avg_launch_us'] for k, v in d['decode_kernels'].items()})" >> gpurun_out/${T}_evfence.log 2>&1
  done
  cat gpurun_out/${T}_evfence.log
  timeout 200 python tools/step_floor_probe.py > gpurun_out/${T}_step_floor.log 2>&1
  echo "-- probe build (CTTS_PROBE_EXIT: no memory access at all)" >> gpurun_out/${T}_step_floor.log
  CTTS_LIB=$R/chattts_amd/csrc/libchattts_amd_probe.so timeout 200 python tools/step_floor_probe.py >> gpurun_out/${T}_step_floor.log 2>&1
  grep -v amdgpu.ids gpurun_out/${T}_step_floor.log
  cd /tmp
  for M in graph eager; do
    X=""; [ $M = eager ] && X="--no-graph"
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${T}_$M -o ${T}_$M -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode --no-bf16-parity $X > $R/gpurun_out/${T}_rocprof_$M.log 2>&1
    F=$(find /tmp/prof_${T}_$M -name "*kernel_stats.csv" | head -1)
    [ -n "$F" ] && cp $F $R/gpurun_out/${T}_kernel_stats_$M.csv && head -8 $F | cut -c1-160
  done
  cd "$R"
}

# round 3, visit b: the new sampling kernel / fused final-norm + heads / small-batch path / device generator -- full suite, bench,
# BASELINE configs C1 C2 C5 with A/B of the small-batch knobs, unseeded probe, kernel stats and PMC passes (bf16 and f32 mode).
# kernel stats (graph replay) of the bench command, bf16 and f32 mode; PMC passes of both
r3b() {
  T=r3b
  timeout 700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s -x > gpurun_out/${T}_tests.log 2>&1
  echo "pytest exit $?" >> gpurun_out/${T}_tests.log
  tail -25 gpurun_out/${T}_tests.log | cut -c1-400
  grep -h "device generator chi2" gpurun_out/${T}_tests.log
  timeout 420 python bench.py --steps 5 --warmup 2 > gpurun_out/${T}_bench.log 2> gpurun_out/${T}_bench.err
  echo "bench exit $?" >> gpurun_out/${T}_bench.log
  grep -v amdgpu.ids gpurun_out/${T}_bench.err | tail -8
  tail -2 gpurun_out/${T}_bench.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('value', d['value'], 'ms', d['ms_per_step'], 'parity', d.get('parity_mode', {}).get('value'), d.get('parity_mode', {}).get('ids_match_reference'))
        print('roofline', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'step', d['roofline']['whole_decode_step'])
        print({k: v['avg_launch_us'] for k, v in d['decode_kernels'].items()})
        print('bf16_parity', {k: d['bf16_parity'][k] for k in ('token_agreement', 'worst_rel_hidden_err', 'worst_abs_dlogit')})
    else:
        print(l[:300])
  "
  Q="--steps 3 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline --no-parity-mode --no-bf16-parity"
  ab() { L=$1; shift; echo "== $L" >> gpurun_out/${T}_ab.log
    env "$@" timeout 200 python bench.py $Q 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/${T}_ab.log 2>&1; }
  for rep in 1 2; do
  ab "base" X=1
  ab "CTTS_FNORM_FUSE=0 (separate final norm launch)" CTTS_FNORM_FUSE=0
  ab "CTTS_GRAPH_STEPS=16" CTTS_GRAPH_STEPS=16
  ab "CTTS_GRAPH_STEPS=8" CTTS_GRAPH_STEPS=8
  done
  cat gpurun_out/${T}_ab.log
  timeout 300 python tools/configs_run.py > gpurun_out/${T}_configs.log 2>&1
  grep -v amdgpu.ids gpurun_out/${T}_configs.log | cut -c1-400
  for E in "CTTS_DEC_A_EARLY=0" "CTTS_ATT_SMALL_M=0" "CTTS_DEC_A_EARLY=0 CTTS_ATT_SMALL_M=0 CTTS_FNORM_FUSE=0"; do
    echo "== C2 with $E" >> gpurun_out/${T}_c2_ab.log
    env $E timeout 120 python tools/c2_run.py 3 2>/dev/null | tail -1 >> gpurun_out/${T}_c2_ab.log
  done
  echo "== C2 default" >> gpurun_out/${T}_c2_ab.log; timeout 120 python tools/c2_run.py 3 2>/dev/null | tail -1 >> gpurun_out/${T}_c2_ab.log
  cat gpurun_out/${T}_c2_ab.log | cut -c1-300
  timeout 200 python tools/unseeded_probe.py > gpurun_out/${T}_unseeded_probe.log 2>&1; tail -1 gpurun_out/${T}_unseeded_probe.log
  cd /tmp
  for D in bf16 f32; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${T}_$D -o ${T}_$D -- python $R/bench.py --dtype $D --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode --no-bf16-parity > $R/gpurun_out/${T}_rocprof_$D.log 2>&1
    F=$(find /tmp/prof_${T}_$D -name "*kernel_stats.csv" | head -1)
    [ -n "$F" ] && cp $F $R/gpurun_out/${T}_kernel_stats_$D.csv
    for C in FETCH_SIZE WRITE_SIZE; do
    CTTS_SYNC_POLL=1 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_${T}_${D}_$C -o ${T}_$C -- python $R/bench.py --dtype $D --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode --no-bf16-parity > $R/gpurun_out/${T}_pmc_${D}_$C.log 2>&1
    done
    python $R/tools/pmc_summary.py /tmp/pmc_${T}_${D}_FETCH_SIZE /tmp/pmc_${T}_${D}_WRITE_SIZE $R/gpurun_out/${T}_pmc_traffic.json > $R/gpurun_out/${T}_pmc_summary_$D.txt 2>&1
    head -12 $R/gpurun_out/${T}_pmc_summary_$D.txt | cut -c1-150
  done
  cd "$R"
}

# round 3, visit c: what slowed BASELINE config C5 (streaming, batch 16) down; the sampling kernel after the penalty-stage fix
r3c() {
  T=r3c
  timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "sample or device_generator" > gpurun_out/${T}_tests_sample.log 2>&1
  tail -3 gpurun_out/${T}_tests_sample.log
  for E in "X=1" "CTTS_DEC_A_EARLY=0" "CTTS_GRAPH_STEPS=1" "CTTS_FNORM_FUSE=0" "CTTS_ORDER=0"; do
    echo "=== $E" >> gpurun_out/${T}_c5_probe.log
    env $E timeout 200 python tools/c5_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/${T}_c5_probe.log
  done
  cat gpurun_out/${T}_c5_probe.log | cut -c1-900
  Q="--steps 3 --warmup 1 --no-cpu-baseline --no-ttfs --no-parity-mode --no-bf16-parity"
  timeout 200 python bench.py $Q 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k: v['avg_launch_us'] for k, v in d['decode_kernels'].items()})" | tee gpurun_out/${T}_bench_quick.log
}

# round 3, visit d: is the C5 slowdown code or environment?  the round-2 tree (30e661d) under the same probe on today's box
r3d() {
  T=r3d
  echo "=== round-2 tree (30e661d)" > gpurun_out/${T}_c5_old_vs_new.log
  (cd _old_r2 && timeout 200 python tools/c5_probe.py 2>&1 | grep -v amdgpu.ids) >> gpurun_out/${T}_c5_old_vs_new.log
  echo "=== current tree" >> gpurun_out/${T}_c5_old_vs_new.log
  timeout 200 python tools/c5_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/${T}_c5_old_vs_new.log
  cut -c1-700 gpurun_out/${T}_c5_old_vs_new.log
}


r3e() {
  timeout 300 python tools/c5_sched_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3e_c5_sched_probe.log
  echo "=== GPU_MAX_HW_QUEUES=8" >> gpurun_out/r3e_c5_sched_probe.log
  GPU_MAX_HW_QUEUES=8 timeout 300 python tools/c5_sched_probe.py 2>&1 | grep -v amdgpu.ids | head -2 >> gpurun_out/r3e_c5_sched_probe.log
  cut -c1-400 gpurun_out/r3e_c5_sched_probe.log
}

# round 3, visit f: polled waits (chattts_amd/_sync.py) vs blocking waits on the streaming config and the bench
r3f() {
  T=r3f
  for E in "CTTS_SPIN_WAIT=1" "CTTS_SPIN_WAIT=0" "CTTS_SPIN_WAIT=0 HSA_ENABLE_INTERRUPT=0"; do
    echo "=== $E" >> gpurun_out/${T}_spin_ab.log
    env $E timeout 200 python tools/c5_probe.py 2>&1 | grep -v amdgpu.ids | head -1 >> gpurun_out/${T}_spin_ab.log
    env $E timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity-mode --no-bf16-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], 'ttfs', d['ttfs_ms_p50'])" >> gpurun_out/${T}_spin_ab.log
  done
  cut -c1-900 gpurun_out/${T}_spin_ab.log
  timeout 300 python tools/configs_run.py 2>&1 | grep -v "amdgpu.ids\|incomplete" > gpurun_out/${T}_configs.log; cut -c1-400 gpurun_out/${T}_configs.log
  timeout 200 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -p no:cacheprovider -k "stream or continuous or slot_pool or unseeded or device_generator or interrupt" 2>&1 | tail -3
}


r3g() {
  timeout 200 python tools/c5_yield_probe.py 2>&1 | grep -v "amdgpu.ids\|incomplete" > gpurun_out/r3g_c5_yield_probe.log
  cut -c1-1500 gpurun_out/r3g_c5_yield_probe.log
  timeout 200 python tools/c5_probe.py 2>&1 | grep -v amdgpu.ids | head -1 | cut -c1-700
}


r3h() {
  timeout 300 python tools/stall_probe.py 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r3h_stall_probe.log
  cat gpurun_out/r3h_stall_probe.log
  rocm-smi --showclocks --showperflevel 2>/dev/null | head -30 >> gpurun_out/r3h_stall_probe.log
  uname -r >> gpurun_out/r3h_stall_probe.log; cat /sys/module/amdgpu/version 2>/dev/null >> gpurun_out/r3h_stall_probe.log
  tail -12 gpurun_out/r3h_stall_probe.log
}


r3i() {
  echo "=== HSA_ENABLE_SDMA=0" > gpurun_out/r3i_stall_probe_sdma.log
  HSA_ENABLE_SDMA=0 timeout 300 python tools/stall_probe.py 2>&1 | grep -v "amdgpu.ids" | head -2 >> gpurun_out/r3i_stall_probe_sdma.log
  HSA_ENABLE_SDMA=0 timeout 200 python tools/c5_probe.py 2>&1 | grep -v amdgpu.ids | head -1 | cut -c1-600 >> gpurun_out/r3i_stall_probe_sdma.log
  echo "=== default" >> gpurun_out/r3i_stall_probe_sdma.log
  timeout 300 python tools/stall_probe.py 2>&1 | grep -v "amdgpu.ids" | head -2 >> gpurun_out/r3i_stall_probe_sdma.log
  cat gpurun_out/r3i_stall_probe_sdma.log
}


r3j() {
  timeout 300 python tools/d2h_probe.py 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r3j_d2h_probe.log; cat gpurun_out/r3j_d2h_probe.log
}

# round 3, visit k: polls and the final D2H as shader copies into pinned memory -- the streaming config and the bench again
r3k() {
  T=r3k
  timeout 200 python tools/c5_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-700 > gpurun_out/${T}_c5_probe.log; cat gpurun_out/${T}_c5_probe.log
  timeout 200 python tools/stall_probe.py 2>&1 | grep -v amdgpu.ids | head -2
  timeout 300 python tools/configs_run.py 2>&1 | grep -v "amdgpu.ids\|incomplete" > gpurun_out/${T}_configs.log; cut -c1-400 gpurun_out/${T}_configs.log
  for E in "X=1" "CTTS_D2H_SHADER=0"; do
  env $E timeout 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --no-parity-mode --no-bf16-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], 'ttfs', d['ttfs_ms_p50'])" | tee -a gpurun_out/${T}_bench_quick.log
  done
  timeout 400 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
}


r3l() {
  timeout 300 python tools/hostcopy_probe.py 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r3l_hostcopy_probe.log; cat gpurun_out/r3l_hostcopy_probe.log
}

# round 3, visit m: host copy out of the staging buffer by memcpy (no OpenMP pool) -- streaming config, bench, full suite
r3m() {
  T=r3m
  timeout 200 python tools/c5_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-700 > gpurun_out/${T}_c5_probe.log; head -1 gpurun_out/${T}_c5_probe.log
  timeout 300 python tools/configs_run.py 2>&1 | grep -v "amdgpu.ids\|incomplete" > gpurun_out/${T}_configs.log; cut -c1-400 gpurun_out/${T}_configs.log
  timeout 420 python bench.py --steps 5 --warmup 2 > gpurun_out/${T}_bench.log 2> gpurun_out/${T}_bench.err
  echo "bench exit $?" >> gpurun_out/${T}_bench.log
  tail -2 gpurun_out/${T}_bench.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('value', d['value'], 'ms', d['ms_per_step'], 'parity', d['parity_mode']['value'], d['parity_mode']['ids_match_reference'], 'ttfs', d['ttfs_ms_p50'])
        print('roofline', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'step', d['roofline']['whole_decode_step']['ms_per_step'], d['roofline']['whole_decode_step']['frac'])
        print({k: v['avg_launch_us'] for k, v in d['decode_kernels'].items()})
        print('cpu', d['cpu_baseline'])
    else:
        print(l[:300])
  "
  timeout 700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/${T}_tests.log 2>&1; tail -3 gpurun_out/${T}_tests.log
}


r3n() {
  timeout 200 python tools/sample_phase_probe.py 2>&1 | grep -v "amdgpu.ids\|incomplete" > gpurun_out/r3n_sample_phase_probe.log; cat gpurun_out/r3n_sample_phase_probe.log
}


r3o() {
  T=r3o
  timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "sample or device_generator" > gpurun_out/${T}_tests_sample.log 2>&1; tail -2 gpurun_out/${T}_tests_sample.log
  timeout 200 python tools/sample_phase_probe.py 2>&1 | grep -v "amdgpu.ids\|incomplete" > gpurun_out/${T}_sample_phase_probe.log; cat gpurun_out/${T}_sample_phase_probe.log
  timeout 300 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x --tb=short -p no:cacheprovider -k "bit_exact or golden or bench_workload" 2>&1 | tail -2
  timeout 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-ttfs --no-parity-mode --no-bf16-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k: v['avg_launch_us'] for k, v in d['decode_kernels'].items()})" | tee gpurun_out/${T}_bench_quick.log
  timeout 200 python tools/serving_probe.py 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-300 | tee gpurun_out/${T}_serving_probe.log
}

# round 3, visit p: software pipelining across batches (acoustic decode of batch i overlapping the generation of batch i+1)
r3p() {
  T=r3p
  timeout 300 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x --tb=short -p no:cacheprovider -k "pipelined or decode_to_wavs or chat_facade or stream_chunks" 2>&1 | tail -3
  Q="--steps 6 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline --no-parity-mode --no-bf16-parity"
  for rep in 1 2; do
  for X in "--pipeline" ""; do
    echo "== bench $X" >> gpurun_out/${T}_pipeline_ab.log
    timeout 200 python bench.py $Q $X 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['pipelined'])" >> gpurun_out/${T}_pipeline_ab.log 2>&1
  done
  done
  cat gpurun_out/${T}_pipeline_ab.log
}

# round 3, visit q: acoustic decoder's point-wise GEMM with two k blocks per barrier (CTTS_X3P_VAR=4); decode weights of the first N
# layers with plain loads (CTTS_W_TEMPORAL_LAYERS)
r3q() {
  T=r3q
  for V in 1 4; do echo "== CTTS_X3P_VAR=$V" >> gpurun_out/${T}_x3p_probe.log; CTTS_X3P_VAR=$V timeout 120 python tools/x3p_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-110 >> gpurun_out/${T}_x3p_probe.log; done
  cat gpurun_out/${T}_x3p_probe.log
  CTTS_X3P_VAR=4 timeout 300 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider -k "codec or x3p" 2>&1 | tail -2
  Q="--steps 4 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline --no-parity-mode --no-bf16-parity"
  ab() { L=$1; shift; echo "== $L" >> gpurun_out/${T}_ab.log
    env "$@" timeout 200 python bench.py $Q 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/${T}_ab.log 2>&1; }
  for rep in 1 2; do
  ab "base" X=1
  ab "CTTS_X3P_VAR=4" CTTS_X3P_VAR=4
  ab "CTTS_W_TEMPORAL_LAYERS=8" CTTS_W_TEMPORAL_LAYERS=8
  ab "CTTS_W_TEMPORAL_LAYERS=13" CTTS_W_TEMPORAL_LAYERS=13
  ab "CTTS_W_TEMPORAL_LAYERS=20" CTTS_W_TEMPORAL_LAYERS=20
  done
  cat gpurun_out/${T}_ab.log
}

# round 3, visit r: validation of the tree -- full GPU suite, smoke, the driver's bench command, torchrun at world size 1, kernel stats
r3r() {
  T=r3r
  timeout 700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1; echo "pytest exit $?" >> gpurun_out/${T}_tests.log; tail -3 gpurun_out/${T}_tests.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; tail -1 gpurun_out/${T}_smoke.log
  timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench.log 2> gpurun_out/${T}_bench.err; echo "bench exit $?" >> gpurun_out/${T}_bench.log
  tail -2 gpurun_out/${T}_bench.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('value', d['value'], 'ms', d['ms_per_step'], 'parity', d['parity_mode']['value'], d['parity_mode']['ids_match_reference'], 'ttfs', d['ttfs_ms_p50'], 'pipelined', d.get('pipelined_queue'))
        print('roofline', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'step', d['roofline']['whole_decode_step']['ms_per_step'], d['roofline']['whole_decode_step']['frac'], 'f32 att', d['parity_mode']['roofline']['frac'])
        print({k: v['avg_launch_us'] for k, v in d['decode_kernels'].items()})
        print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'mfma', {k: (v['avg_launch_us'], v['frac']) for k, v in d['roofline_mfma'].items()})
    else:
        print(l[:300])
  "
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-ttfs > gpurun_out/${T}_torchrun_world1.log 2>&1
  echo "torchrun exit $?" >> gpurun_out/${T}_torchrun_world1.log; tail -3 gpurun_out/${T}_torchrun_world1.log | cut -c1-400
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${T} -o ${T} -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode --no-bf16-parity --no-slot-pool > $R/gpurun_out/${T}_rocprof.log 2>&1
  F=$(find /tmp/prof_${T} -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $R/gpurun_out/${T}_kernel_stats.csv && head -12 $F | cut -c1-120
  cd "$R"
}

# round 3, visit s: cross-kernel weight prefetch (CTTS_PF bit mask: 1 QKV->o_proj, 2 attention->gate/up, 4 gate/up->down, 8 gate/up->next QKV/heads)
r3s() {
  T=r3s
  true
  Q="--steps 4 --warmup 1 --no-cpu-baseline --no-ttfs --no-parity-mode --no-bf16-parity"
  ab() { L=$1; shift; echo "== $L" >> gpurun_out/${T}_ab.log
    env "$@" timeout 200 python bench.py $Q 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k: v['avg_launch_us'] for k, v in d['decode_kernels'].items()})" >> gpurun_out/${T}_ab.log 2>&1; }
  for rep in 1 2; do
  ab "CTTS_PF=0 (off)" CTTS_PF=0
  ab "CTTS_PF=15 (all)" CTTS_PF=15
  ab "CTTS_PF=1" CTTS_PF=1
  ab "CTTS_PF=2" CTTS_PF=2
  ab "CTTS_PF=4" CTTS_PF=4
  ab "CTTS_PF=8" CTTS_PF=8
  ab "CTTS_PF=13 (no attention prefetch)" CTTS_PF=13
  done
  cat gpurun_out/${T}_ab.log
}


r3t() {
  for M in 2 8 15; do
  echo "=== CTTS_PF=$M" >> gpurun_out/r3t_pf_crash.log
  CTTS_PF=$M timeout 120 python tools/c2_run.py 1 > gpurun_out/r3t_tmp.out 2> gpurun_out/r3t_tmp.err; echo "exit $?" >> gpurun_out/r3t_pf_crash.log
  tail -3 gpurun_out/r3t_tmp.out | cut -c1-300 >> gpurun_out/r3t_pf_crash.log; grep -v amdgpu.ids gpurun_out/r3t_tmp.err | head -12 | cut -c1-300 >> gpurun_out/r3t_pf_crash.log
  done
  cat gpurun_out/r3t_pf_crash.log
}


r3u() {
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/kernarg_probe.hip -o /tmp/kernarg_probe 2>&1 | tail -3
  timeout 120 /tmp/kernarg_probe > gpurun_out/r3u_kernarg_probe.log 2>&1; cat gpurun_out/r3u_kernarg_probe.log
  echo "--- HIP_FORCE_DEV_KERNARG=0" >> gpurun_out/r3u_kernarg_probe.log; HIP_FORCE_DEV_KERNARG=0 timeout 120 /tmp/kernarg_probe 2>&1 | head -8 | tee -a gpurun_out/r3u_kernarg_probe.log
  echo "--- HIP_FORCE_DEV_KERNARG=1" >> gpurun_out/r3u_kernarg_probe.log; HIP_FORCE_DEV_KERNARG=1 timeout 120 /tmp/kernarg_probe 2>&1 | head -8 | tee -a gpurun_out/r3u_kernarg_probe.log
}


r3v() {
  T=r3v
  Q="--steps 4 --warmup 1 --no-cpu-baseline --no-ttfs --no-parity-mode --no-bf16-parity"
  ab() { L=$1; shift; echo "== $L" >> gpurun_out/${T}_ab.log
    env "$@" timeout 200 python bench.py $Q 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k: v['avg_launch_us'] for k, v in d['decode_kernels'].items()})" >> gpurun_out/${T}_ab.log 2>&1; }
  for rep in 1 2; do
  ab "CTTS_PF=0 (off)" CTTS_PF=0
  ab "CTTS_PF=1 (QKV -> o_proj)" CTTS_PF=1
  ab "CTTS_PF=16 (QKV -> gate/up)" CTTS_PF=16
  ab "CTTS_PF=17 (QKV -> o_proj + gate/up)" CTTS_PF=17
  done
  cat gpurun_out/${T}_ab.log
}


r3w() {
  T=r3w
  Q="--steps 4 --warmup 1 --no-cpu-baseline --no-ttfs --no-parity-mode --no-bf16-parity"
  ab() { L=$1; shift; echo "== $L" >> gpurun_out/${T}_ab.log
    env "$@" timeout 200 python bench.py $Q 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k: v['avg_launch_us'] for k, v in d['decode_kernels'].items()})" >> gpurun_out/${T}_ab.log 2>&1; }
  for rep in 1 2; do
  ab "4 waves per unit" CTTS_ATT_NW_PACKED=4
  ab "2 waves per unit" CTTS_ATT_NW_PACKED=2
  done
  cat gpurun_out/${T}_ab.log
}


r3x() {
  timeout 200 python tools/attn_phase_probe.py 2>&1 | grep -v "amdgpu.ids\|incomplete" > gpurun_out/r3x_attn_phase_probe.log; cat gpurun_out/r3x_attn_phase_probe.log
  timeout 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-ttfs --no-parity-mode --no-bf16-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k: v['avg_launch_us'] for k, v in d['decode_kernels'].items()})"
}


r3y() {
  T=r3y
  timeout 200 python tools/attn_phase_probe.py 2>&1 | grep -v "amdgpu.ids\|incomplete" > gpurun_out/${T}_attn_phase_probe.log; head -16 gpurun_out/${T}_attn_phase_probe.log
  for rep in 1 2; do timeout 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-ttfs --no-bf16-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'parity', d['parity_mode']['value'], d['parity_mode']['ids_match_reference'], d['parity_mode']['roofline']['avg_launch_us'], {k: v['avg_launch_us'] for k, v in d['decode_kernels'].items()})" | tee -a gpurun_out/${T}_bench_quick.log; done
  timeout 600 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -3
}

# round 3, visit z: parity mode prefill on the packed f32 kernels
r3z() {
  T=r3z
  timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x --tb=short -p no:cacheprovider -k "f32 or bit_exact or golden or bench_workload or continuous or chunked or refine or stream" 2>&1 | tail -4
  for E in "X=1" "CTTS_PRE32_PACKED=0"; do
    env $E timeout 200 python bench.py --dtype f32 --steps 4 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$E', d['value'], d['ms_per_step'])" | tee -a gpurun_out/${T}_f32_prefill_ab.log
  done
  cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${T} -o ${T} -- python $R/bench.py --dtype f32 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs > $R/gpurun_out/${T}_rocprof.log 2>&1
  F=$(find /tmp/prof_${T} -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $R/gpurun_out/${T}_kernel_stats_f32.csv && head -16 $F | cut -c1-110
}

# round 3, final validation of the tree: the same as r3r (full suite, smoke, the driver's bench command, torchrun world 1, kernel stats)
r3ab() {
  T=r3ab
  timeout 700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1; echo "pytest exit $?" >> gpurun_out/${T}_tests.log; tail -3 gpurun_out/${T}_tests.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; tail -1 gpurun_out/${T}_smoke.log
  timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench.log 2> gpurun_out/${T}_bench.err; echo "bench exit $?" >> gpurun_out/${T}_bench.log
  tail -2 gpurun_out/${T}_bench.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('value', d['value'], 'ms', d['ms_per_step'], 'parity', d['parity_mode']['value'], d['parity_mode']['ids_match_reference'], 'ttfs', d['ttfs_ms_p50'], 'pipelined', d['pipelined_queue']['value']); print('pool', d.get('continuous_batching_queue', {}).get('value'), 'codec_parity', d.get('codec_parity', {}).get('wav_rms_diff'))
        print('roofline', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'step', d['roofline']['whole_decode_step']['ms_per_step'], d['roofline']['whole_decode_step']['frac'], 'f32 att', d['parity_mode']['roofline']['frac'], d['parity_mode']['decode_ms_per_gpt_step'])
        print({k: v['avg_launch_us'] for k, v in d['decode_kernels'].items()})
        print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'agree', d['bf16_parity']['token_agreement'])
    else:
        print(l[:300])
"
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-ttfs > gpurun_out/${T}_torchrun_world1.log 2>&1
  echo "torchrun exit $?" >> gpurun_out/${T}_torchrun_world1.log; tail -1 gpurun_out/${T}_torchrun_world1.log
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${T} -o ${T} -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode --no-bf16-parity --no-slot-pool > $R/gpurun_out/${T}_rocprof.log 2>&1
  F=$(find /tmp/prof_${T} -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $R/gpurun_out/${T}_kernel_stats.csv && head -6 $F | cut -c1-120
  cd "$R"
}

r3ac() {   # would a KV prefetch one layer ahead pay?  (tools/mall_probe.hip)
  T=r3ac
  hipcc --offload-arch=gfx950 -O3 tools/mall_probe.hip -o /tmp/mall_probe && timeout 120 /tmp/mall_probe > gpurun_out/${T}_mall_probe.log 2>&1
  cat gpurun_out/${T}_mall_probe.log
}

r3ad() {   # three KV blocks in flight per wave (CTTS_ATT_NBUF=3) vs two: parity first, then C3 and C2
  T=r3ad
  CTTS_ATT_NBUF=3 timeout 400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "attention or bit_exact or bench_workload or invariance" > gpurun_out/${T}_tests_nbuf3.log 2>&1; tail -2 gpurun_out/${T}_tests_nbuf3.log
  Q="--steps 4 --warmup 1 --no-cpu-baseline --no-ttfs --no-bf16-parity --parity-steps 2"
  ab() { L=$1; shift; echo "== $L" >> gpurun_out/${T}_ab.log
    env "$@" timeout 200 python bench.py $Q 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'parity', d['parity_mode']['value'], d['parity_mode']['ids_match_reference'], d['parity_mode']['roofline']['avg_launch_us'], {k: v['avg_launch_us'] for k, v in d['decode_kernels'].items()})" >> gpurun_out/${T}_ab.log 2>&1; }
  for rep in 1 2; do
  ab "2 blocks in flight" CTTS_ATT_NBUF=2
  ab "3 blocks in flight" CTTS_ATT_NBUF=3
  done
  for E in CTTS_ATT_NBUF=2 CTTS_ATT_NBUF=3 CTTS_ATT_NBUF=2 CTTS_ATT_NBUF=3; do
    echo "== C2 with $E" >> gpurun_out/${T}_ab.log
    env $E timeout 120 python tools/c2_run.py 3 2>/dev/null | tail -1 >> gpurun_out/${T}_ab.log
  done
  cut -c1-420 gpurun_out/${T}_ab.log
}

r3ae() {   # the perf mode's acoustic decoder on one fp16 MFMA per product (gemm="f16"): kernel + e2e tests, then A/B on C3
  T=r3ae
  timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "h1p or x3p or codec or decode_to_wavs or decode_window or stream" > gpurun_out/${T}_tests.log 2>&1; tail -5 gpurun_out/${T}_tests.log
  grep "codec f16 vs" gpurun_out/${T}_tests.log
  Q="--steps 4 --warmup 1 --no-cpu-baseline --no-ttfs --no-bf16-parity --parity-steps 2"
  ab() { L=$1; shift; echo "== $L" >> gpurun_out/${T}_ab.log
    timeout 300 python bench.py $Q "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'parity', d['parity_mode']['value'], d['parity_mode']['ids_match_reference'], 'codec_parity', d.get('codec_parity'), {k: (v['avg_launch_us'], v['frac']) for k, v in d['roofline_mfma'].items()})" >> gpurun_out/${T}_ab.log 2>&1; }
  for rep in 1 2; do
  ab "decoder bf16x3" --codec-gemm bf16x3
  ab "decoder f16" --codec-gemm f16
  done
  cut -c1-900 gpurun_out/${T}_ab.log
}

r3af() {   # gemm_h1p_k with fragment reads one k block ahead + the branch-free GELU: tests, GEMM timings, C3
  T=r3af
  timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "h1p or f16_mode" > gpurun_out/${T}_tests.log 2>&1; tail -3 gpurun_out/${T}_tests.log
  grep "codec f16 vs" gpurun_out/${T}_tests.log
  Q="--steps 4 --warmup 1 --no-cpu-baseline --no-ttfs --no-bf16-parity --parity-steps 1"
  for rep in 1 2; do
  timeout 300 python bench.py $Q 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'codec_parity', d['codec_parity']['wav_rms_diff'], {k: (v['avg_launch_us'], v['frac']) for k, v in d['roofline_mfma'].items()})" >> gpurun_out/${T}_ab.log 2>&1
  done
  cat gpurun_out/${T}_ab.log
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${T} -o ${T} -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode --no-bf16-parity --no-slot-pool > $R/gpurun_out/${T}_rocprof.log 2>&1
  F=$(find /tmp/prof_${T} -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $R/gpurun_out/${T}_kernel_stats.csv && grep -v "gemm_dec_k\|attention_k" $F | head -14 | cut -c1-150
  cd "$R"
}

r3ag() {   # where a 256 x 256 tile of gemm_h1p_k spends its time
  T=r3ag
  CTTS_H1P_PROBE=1 timeout 200 python tools/x3p_phase_probe.py --h1p 2>&1 | grep -v "amdgpu.ids\|incomplete" > gpurun_out/${T}_h1p_phase_probe.log; cat gpurun_out/${T}_h1p_phase_probe.log
}

r3ah() {   # residual epilogues without the load-after-store chain (gemm_x3p_k, gemm_h1p_k, the tiled kernels): tests, phase probe, C3, C5
  T=r3ah
  timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "h1p or x3p or codec or decode_to_wavs or decode_window or stream or gemm_tiled or dvae" > gpurun_out/${T}_tests.log 2>&1; tail -3 gpurun_out/${T}_tests.log
  CTTS_H1P_PROBE=1 timeout 200 python tools/x3p_phase_probe.py --h1p 2>&1 | grep -v "amdgpu.ids\|incomplete" > gpurun_out/${T}_h1p_phase_probe.log; cat gpurun_out/${T}_h1p_phase_probe.log
  Q="--steps 4 --warmup 1 --no-cpu-baseline --no-ttfs --no-bf16-parity --parity-steps 2"
  for rep in 1 2; do
  timeout 300 python bench.py $Q 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'parity', d['parity_mode']['value'], d['parity_mode']['ids_match_reference'], 'codec_parity', d['codec_parity']['wav_rms_diff'], {k: (v['avg_launch_us'], v['frac']) for k, v in d['roofline_mfma'].items()})" >> gpurun_out/${T}_ab.log 2>&1
  done
  cat gpurun_out/${T}_ab.log
  timeout 300 python tools/configs_run.py 2>&1 | grep -v "amdgpu.ids\|incomplete" > gpurun_out/${T}_configs.log; cut -c1-400 gpurun_out/${T}_configs.log
}

r3ai() {   # gemm_h1p_k with a 5-slot ring (160 KiB, four 32-wide k blocks in flight) vs the 2 x 2-slot stages: probe, tests, C3
  T=r3ai
  for RING in 4 5; do
    echo "== CTTS_H1P_RING=$RING" >> gpurun_out/${T}_h1p_ring.log
    CTTS_H1P_RING=$RING CTTS_H1P_PROBE=1 timeout 200 python tools/x3p_phase_probe.py --h1p 2>&1 | grep "^h1p" >> gpurun_out/${T}_h1p_ring.log
  done
  CTTS_H1P_RING=5 timeout 400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "h1p or f16_mode" > gpurun_out/${T}_tests_ring5.log 2>&1; tail -2 gpurun_out/${T}_tests_ring5.log
  Q="--steps 4 --warmup 1 --no-cpu-baseline --no-ttfs --no-bf16-parity --no-parity-mode"
  for rep in 1 2; do for RING in 4 5; do
    echo "== CTTS_H1P_RING=$RING" >> gpurun_out/${T}_h1p_ring.log
    CTTS_H1P_RING=$RING timeout 300 python bench.py $Q 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k: (v['avg_launch_us'], v['frac']) for k, v in d['roofline_mfma'].items()})" >> gpurun_out/${T}_h1p_ring.log 2>&1
  done; done
  cat gpurun_out/${T}_h1p_ring.log
}

r3aj() {   # sliding-window depthwise conv + LayerNorm for large batches: tests, A/B on C3, kernel times
  T=r3aj
  timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "dwconv or codec or f16_mode or decode_to_wavs" > gpurun_out/${T}_tests.log 2>&1; tail -4 gpurun_out/${T}_tests.log
  Q="--steps 4 --warmup 1 --no-cpu-baseline --no-ttfs --no-bf16-parity --no-parity-mode --no-roofline"
  for rep in 1 2; do for MINR in 0 12288; do
    echo "== CTTS_DWCONV_RUN_MIN_ROWS=$MINR" >> gpurun_out/${T}_ab.log
    CTTS_DWCONV_RUN_MIN_ROWS=$MINR timeout 300 python bench.py $Q 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/${T}_ab.log 2>&1
  done; done
  cat gpurun_out/${T}_ab.log
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${T} -o ${T} -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode --no-bf16-parity --no-slot-pool > $R/gpurun_out/${T}_rocprof.log 2>&1
  F=$(find /tmp/prof_${T} -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $R/gpurun_out/${T}_kernel_stats.csv && grep -v "gemm_dec_k\|attention_k" $F | head -12 | cut -c1-150
  cd "$R"
}

r3ak() {   # validation visit on the tree with the f16 decoder, the residual epilogues, the sliding-window depthwise conv and the one-copy outputs
  T=r3ak
  timeout 700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1; echo "pytest exit $?" >> gpurun_out/${T}_tests.log; tail -3 gpurun_out/${T}_tests.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; tail -1 gpurun_out/${T}_smoke.log
  timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench.log 2> gpurun_out/${T}_bench.err; echo "bench exit $?" >> gpurun_out/${T}_bench.log
  tail -2 gpurun_out/${T}_bench.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('value', d['value'], 'ms', d['ms_per_step'], 'parity', d['parity_mode']['value'], d['parity_mode']['ids_match_reference'], 'ttfs', d['ttfs_ms_p50'], 'pipelined', d['pipelined_queue']['value']); print('pool', d.get('continuous_batching_queue', {}).get('value'), 'codec_parity', d.get('codec_parity', {}).get('wav_rms_diff'))
        print('roofline', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'step', d['roofline']['whole_decode_step']['ms_per_step'], d['roofline']['whole_decode_step']['frac'], 'f32 att', d['parity_mode']['roofline']['frac'], d['parity_mode']['decode_ms_per_gpt_step'])
        print({k: v['avg_launch_us'] for k, v in d['decode_kernels'].items()})
        print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'agree', d['bf16_parity']['token_agreement'])
    else:
        print(l[:300])
"
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-ttfs > gpurun_out/${T}_torchrun_world1.log 2>&1
  echo "torchrun exit $?" >> gpurun_out/${T}_torchrun_world1.log; tail -1 gpurun_out/${T}_torchrun_world1.log
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${T} -o ${T} -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode --no-bf16-parity --no-slot-pool > $R/gpurun_out/${T}_rocprof.log 2>&1
  F=$(find /tmp/prof_${T} -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $R/gpurun_out/${T}_kernel_stats.csv && head -6 $F | cut -c1-120
  cd "$R"
}

r3al() {   # bus copy and staging-buffer copy-out overlapped in pieces: test, timeline, C3
  T=r3al
  timeout 300 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "to_host or rows_are_views or decode_to_wavs" > gpurun_out/${T}_tests.log 2>&1; tail -2 gpurun_out/${T}_tests.log
  for P in 0 1; do echo "== CTTS_D2H_PIPE=$P" >> gpurun_out/${T}_timeline.log; CTTS_D2H_PIPE=$P timeout 300 python tools/pass_timeline.py 2>&1 | grep -v amdgpu.ids | tail -2 >> gpurun_out/${T}_timeline.log; done
  cat gpurun_out/${T}_timeline.log
  Q="--steps 6 --warmup 2 --no-cpu-baseline --no-ttfs --no-bf16-parity --no-parity-mode --no-roofline"
  for rep in 1 2; do for P in 0 1; do
    echo "== CTTS_D2H_PIPE=$P" >> gpurun_out/${T}_ab.log
    CTTS_D2H_PIPE=$P timeout 300 python bench.py $Q 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/${T}_ab.log 2>&1
  done; done
  cat gpurun_out/${T}_ab.log
}

r3an() {   # SlotPool with one chunk running ahead: parity tests, then the continuous-batching leg of bench.py
  T=r3an
  timeout 400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "continuous or slot_pool or device_generator" > gpurun_out/${T}_tests.log 2>&1; tail -3 gpurun_out/${T}_tests.log
  for rep in 1 2; do
  timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ttfs --no-bf16-parity --no-parity-mode --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step']); print(d.get('continuous_batching_queue'))" >> gpurun_out/${T}_pool.log 2>&1
  done
  cat gpurun_out/${T}_pool.log
}

r3ap() {   # KV loads of the decode attention with and without the non-temporal hint (variant build: -DCTTS_KV_NT=0)
  T=r3ap
  Q="--steps 4 --warmup 1 --no-cpu-baseline --no-ttfs --no-bf16-parity --no-parity-mode --no-slot-pool"
  for rep in 1 2; do for V in nt plain; do
    echo "== KV loads: $V" >> gpurun_out/${T}_kv_nt_ab.log
    L=""; [ $V = plain ] && L="$R/chattts_amd/csrc/libchattts_amd_kvplain.so"
    CTTS_LIB=$L timeout 300 python bench.py $Q 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])" >> gpurun_out/${T}_kv_nt_ab.log 2>&1
  done; done
  cat gpurun_out/${T}_kv_nt_ab.log
}

r3aq() {   # validation visit: + SlotPool run-ahead, queue legs in bench.py, piped waveform D2H
  T=r3aq
  timeout 700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1; echo "pytest exit $?" >> gpurun_out/${T}_tests.log; tail -3 gpurun_out/${T}_tests.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; tail -1 gpurun_out/${T}_smoke.log
  timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench.log 2> gpurun_out/${T}_bench.err; echo "bench exit $?" >> gpurun_out/${T}_bench.log
  tail -2 gpurun_out/${T}_bench.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('value', d['value'], 'ms', d['ms_per_step'], 'parity', d['parity_mode']['value'], d['parity_mode']['ids_match_reference'], 'ttfs', d['ttfs_ms_p50'], 'pipelined', d['pipelined_queue']['value']); print('pool', d.get('continuous_batching_queue', {}).get('value'), 'codec_parity', d.get('codec_parity', {}).get('wav_rms_diff'))
        print('roofline', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'step', d['roofline']['whole_decode_step']['ms_per_step'], d['roofline']['whole_decode_step']['frac'], 'f32 att', d['parity_mode']['roofline']['frac'], d['parity_mode']['decode_ms_per_gpt_step'])
        print({k: v['avg_launch_us'] for k, v in d['decode_kernels'].items()})
        print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'agree', d['bf16_parity']['token_agreement'])
    else:
        print(l[:300])
"
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-ttfs > gpurun_out/${T}_torchrun_world1.log 2>&1
  echo "torchrun exit $?" >> gpurun_out/${T}_torchrun_world1.log; tail -1 gpurun_out/${T}_torchrun_world1.log
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${T} -o ${T} -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode --no-bf16-parity --no-slot-pool > $R/gpurun_out/${T}_rocprof.log 2>&1
  F=$(find /tmp/prof_${T} -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $R/gpurun_out/${T}_kernel_stats.csv && head -6 $F | cut -c1-120
  cd "$R"
}

r3as() {   # o_proj's spare CUs prefetch the layer's gate/up weights into the consumer XCD's L2 (CTTS_PF=32: the code of commit 23fb1f2, removed since)
  T=r3as
  CTTS_PF=32 timeout 300 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "packed_decode or bf16_parity or invariance or teacher" > gpurun_out/${T}_tests_pf32.log 2>&1; tail -2 gpurun_out/${T}_tests_pf32.log
  Q="--steps 4 --warmup 1 --no-cpu-baseline --no-ttfs --no-bf16-parity --no-parity-mode --no-slot-pool"
  ab() { L=$1; shift; echo "== $L" >> gpurun_out/${T}_ab.log
    env "$@" timeout 200 python bench.py $Q 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k: v['avg_launch_us'] for k, v in d['decode_kernels'].items()})" >> gpurun_out/${T}_ab.log 2>&1; }
  for rep in 1 2; do
  ab "off" CTTS_PF=0
  ab "o_proj + 2 rows of prefetch workgroups -> gate/up" CTTS_PF=32 CTTS_PF_ROWS=2
  ab "o_proj + 4 rows" CTTS_PF=32 CTTS_PF_ROWS=4
  done
  cat gpurun_out/${T}_ab.log
}

r3at() {   # validation visit on the final tree
  T=r3at
  timeout 700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1; echo "pytest exit $?" >> gpurun_out/${T}_tests.log; tail -3 gpurun_out/${T}_tests.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; tail -1 gpurun_out/${T}_smoke.log
  timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench.log 2> gpurun_out/${T}_bench.err; echo "bench exit $?" >> gpurun_out/${T}_bench.log
  tail -2 gpurun_out/${T}_bench.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('value', d['value'], 'ms', d['ms_per_step'], 'parity', d['parity_mode']['value'], d['parity_mode']['ids_match_reference'], 'ttfs', d['ttfs_ms_p50'], 'pipelined', d['pipelined_queue']['value']); print('pool', d.get('continuous_batching_queue', {}).get('value'), 'codec_parity', d.get('codec_parity', {}).get('wav_rms_diff'))
        print('roofline', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'step', d['roofline']['whole_decode_step']['ms_per_step'], d['roofline']['whole_decode_step']['frac'], 'f32 att', d['parity_mode']['roofline']['frac'], d['parity_mode']['decode_ms_per_gpt_step'])
        print({k: v['avg_launch_us'] for k, v in d['decode_kernels'].items()})
        print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'agree', d['bf16_parity']['token_agreement'])
    else:
        print(l[:300])
"
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-ttfs > gpurun_out/${T}_torchrun_world1.log 2>&1
  echo "torchrun exit $?" >> gpurun_out/${T}_torchrun_world1.log; tail -1 gpurun_out/${T}_torchrun_world1.log
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${T} -o ${T} -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode --no-bf16-parity --no-slot-pool > $R/gpurun_out/${T}_rocprof.log 2>&1
  F=$(find /tmp/prof_${T} -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $R/gpurun_out/${T}_kernel_stats.csv && head -6 $F | cut -c1-120
  cd "$R"
}

r3av() {   # PMC traffic of the final tree: --pmc FETCH_SIZE / WRITE_SIZE in separate kernel-trace-only passes, both modes
  T=r3av
  rm -f $R/gpurun_out/${T}_pmc_traffic.json
  cd /tmp
  for D in bf16 f32; do
    for C in FETCH_SIZE WRITE_SIZE; do
    CTTS_SYNC_POLL=1 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_${T}_${D}_$C -o ${T}_$C -- python $R/bench.py --dtype $D --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode --no-bf16-parity --no-slot-pool > $R/gpurun_out/${T}_pmc_${D}_$C.log 2>&1
    done
    python $R/tools/pmc_summary.py /tmp/pmc_${T}_${D}_FETCH_SIZE /tmp/pmc_${T}_${D}_WRITE_SIZE $R/gpurun_out/${T}_pmc_traffic.json > $R/gpurun_out/${T}_pmc_summary_$D.txt 2>&1
    head -14 $R/gpurun_out/${T}_pmc_summary_$D.txt | cut -c1-150
  done
  cd "$R"
}

if declare -F "$1" > /dev/null; then "$1"; else echo "usage: round3.sh <visit>   (one of: $(declare -F | awk '{print $3}' | tr '\n' ' '))"; exit 2; fi
