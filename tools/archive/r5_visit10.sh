#!/bin/bash
# round 5, visit 10: torchrun world 1 (bounded C-ABI broadcast check), `bench.py --gpus 2` on a 1-GPU box, Chat.warm test, pcm16 leg
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_backend.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r5l_tests_backend.log 2>&1; tail -3 gpurun_out/r5l_tests_backend.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode --no-configs --no-slot-pool --no-ids-check > gpurun_out/r5l_torchrun_world1.log 2>&1
echo "torchrun exit $?" >> gpurun_out/r5l_torchrun_world1.log
grep "^{" gpurun_out/r5l_torchrun_world1.log | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('world1 value', j['value'], 'ranks', j.get('ranks'))"
timeout 300 python bench.py --gpus 2 --steps 1 --warmup 0 > gpurun_out/r5l_gpus2_on_1gpu.out 2> gpurun_out/r5l_gpus2_on_1gpu.log; echo "exit $?" >> gpurun_out/r5l_gpus2_on_1gpu.log
echo "stdout bytes: $(wc -c < gpurun_out/r5l_gpus2_on_1gpu.out)"; grep -E "needs GPU|NO result|exit" gpurun_out/r5l_gpus2_on_1gpu.log | head -5
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-ttfs --no-configs --no-slot-pool --no-ids-check --no-bf16-parity --parity-steps 2 > gpurun_out/r5l_bench_pcm16.log 2>&1
grep "^{" gpurun_out/r5l_bench_pcm16.log | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('value', j['value'], 'pcm16', j.get('pcm16_output'))"
