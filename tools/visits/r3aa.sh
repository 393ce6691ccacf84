#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r3aa
timeout 300 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_kernels.py tests/test_gpu_dvae.py -m gpu -q -x --tb=short -p no:cacheprovider -k "codec or dwconv or dvae or decode_to_wavs or stream_chunks or decode_window" 2>&1 | tail -3
for rep in 1 2; do timeout 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline --no-parity-mode --no-bf16-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a gpurun_out/${T}_bench_quick.log; done
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${T} -o ${T} -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode --no-bf16-parity > $R/gpurun_out/${T}_rocprof.log 2>&1
F=$(find /tmp/prof_${T} -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $R/gpurun_out/${T}_kernel_stats.csv && grep -i "dwconv\|layernorm\|istft\|x3p\|tiled" $F | cut -c1-150
