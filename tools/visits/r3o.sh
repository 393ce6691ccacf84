#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r3o
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "sample or device_generator" > gpurun_out/${T}_tests_sample.log 2>&1; tail -2 gpurun_out/${T}_tests_sample.log
timeout 200 python tools/sample_phase_probe.py 2>&1 | grep -v "amdgpu.ids\|incomplete" > gpurun_out/${T}_sample_phase_probe.log; cat gpurun_out/${T}_sample_phase_probe.log
timeout 300 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x --tb=short -p no:cacheprovider -k "bit_exact or golden or bench_workload" 2>&1 | tail -2
timeout 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-ttfs --no-parity-mode --no-bf16-parity 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k: v['avg_launch_us'] for k, v in d['decode_kernels'].items()})" | tee gpurun_out/${T}_bench_quick.log
timeout 200 python tools/serving_probe.py 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-300 | tee gpurun_out/${T}_serving_probe.log
