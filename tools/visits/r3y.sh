#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r3y
timeout 200 python tools/attn_phase_probe.py 2>&1 | grep -v "amdgpu.ids\|incomplete" > gpurun_out/${T}_attn_phase_probe.log; head -16 gpurun_out/${T}_attn_phase_probe.log
for rep in 1 2; do timeout 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-ttfs --no-bf16-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'parity', d['parity_mode']['value'], d['parity_mode']['ids_match_reference'], d['parity_mode']['roofline']['avg_launch_us'], {k: v['avg_launch_us'] for k, v in d['decode_kernels'].items()})" | tee -a gpurun_out/${T}_bench_quick.log; done
timeout 600 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -3
