#!/bin/bash
# round 3, visit m: host copy out of the staging buffer by memcpy (no OpenMP pool) -- streaming config, bench, full suite
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
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
