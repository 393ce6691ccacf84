#!/bin/bash
# round 3, visit z: parity mode prefill on the packed f32 kernels
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r3z
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x --tb=short -p no:cacheprovider -k "f32 or bit_exact or golden or bench_workload or continuous or chunked or refine or stream" 2>&1 | tail -4
for E in "X=1" "CTTS_PRE32_PACKED=0"; do
  env $E timeout 200 python bench.py --dtype f32 --steps 4 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$E', d['value'], d['ms_per_step'])" | tee -a gpurun_out/${T}_f32_prefill_ab.log
done
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${T} -o ${T} -- python $R/bench.py --dtype f32 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs > $R/gpurun_out/${T}_rocprof.log 2>&1
F=$(find /tmp/prof_${T} -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $R/gpurun_out/${T}_kernel_stats_f32.csv && head -16 $F | cut -c1-110
