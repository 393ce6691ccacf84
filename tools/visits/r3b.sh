#!/bin/bash
# round 3, visit b: the new sampling kernel / fused final-norm + heads / small-batch path / device generator -- full suite, bench,
# BASELINE configs C1 C2 C5 with A/B of the small-batch knobs, unseeded probe, kernel stats and PMC passes (bf16 and f32 mode).
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
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
# kernel stats (graph replay) of the bench command, bf16 and f32 mode; PMC passes of both
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
