#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python tools/attn_phase_probe.py 2>&1 | grep -v "amdgpu.ids\|incomplete" > gpurun_out/r3x_attn_phase_probe.log; cat gpurun_out/r3x_attn_phase_probe.log
timeout 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-ttfs --no-parity-mode --no-bf16-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k: v['avg_launch_us'] for k, v in d['decode_kernels'].items()})"
