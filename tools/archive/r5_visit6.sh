#!/bin/bash
# round 5, visit 6: split-bf16 parity decode after the fix (the row-major fallback loop also ran) -- debug probe, every e2e golden, bench parity leg
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/x3_debug_probe.py 2>&1 | grep -v "amdgpu.ids\|incomplete" > gpurun_out/r5g_x3_probe.log; cat gpurun_out/r5g_x3_probe.log | tail -30
timeout 1500 python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r5g_tests_e2e.log 2>&1
echo "pytest exit $?" >> gpurun_out/r5g_tests_e2e.log; tail -15 gpurun_out/r5g_tests_e2e.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ttfs --no-configs --no-slot-pool --no-ids-check --no-bf16-parity --parity-steps 3 > gpurun_out/r5g_bench.log 2>&1
tail -1 gpurun_out/r5g_bench.log | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); pm=j.get('parity_mode',{})
print('bf16', j['value'], 'parity x3', pm.get('value'), pm.get('ids_match_reference'), pm.get('decode_ms_per_gpt_step'), 'exact', pm.get('exact_f32_mfma'))"
