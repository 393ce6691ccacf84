#!/bin/bash
# round 3, visit r: validation of the tree -- full GPU suite, smoke, the driver's bench command, torchrun at world size 1, kernel stats
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
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
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${T} -o ${T} -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode --no-bf16-parity > $R/gpurun_out/${T}_rocprof.log 2>&1
F=$(find /tmp/prof_${T} -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $R/gpurun_out/${T}_kernel_stats.csv && head -12 $F | cut -c1-120
cd "$R"
