#!/bin/bash
# round 5, visit 5: split-bf16 parity-mode decode kernels (decode32x.hip) -- kernel test, every e2e golden in both arithmetics, bench parity leg
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "dec32x or attention_decode_persistent" -x > gpurun_out/r5e_tests_kernels.log 2>&1
echo "pytest exit $?" >> gpurun_out/r5e_tests_kernels.log; tail -12 gpurun_out/r5e_tests_kernels.log
timeout 1500 python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r5e_tests_e2e.log 2>&1
echo "pytest exit $?" >> gpurun_out/r5e_tests_e2e.log; tail -15 gpurun_out/r5e_tests_e2e.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ttfs --no-configs --no-slot-pool --no-ids-check --no-bf16-parity --parity-steps 3 > gpurun_out/r5e_bench.log 2>&1
tail -1 gpurun_out/r5e_bench.log | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); pm=j.get('parity_mode',{})
print('bf16', j['value'], 'parity x3', pm.get('value'), pm.get('ids_match_reference'), pm.get('decode_ms_per_gpt_step'), 'exact', pm.get('exact_f32_mfma'))"
