#!/bin/bash
# round 5, visit 20: 2 / 3 / 4 heads per decode-attention workgroup (attention_hpw_k): A/B on the bench (bits: tests/test_gpu_kernels.py persistent_grid)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ttfs --no-configs --no-slot-pool --no-bf16-parity --parity-steps 2"
for cfg in "CTTS_ATT_HPW=1" "CTTS_ATT_HPW=3" "CTTS_ATT_HPW=4" "CTTS_ATT_HPW=2" "CTTS_ATT_HPW=1" "CTTS_ATT_HPW=3"; do
  echo "== $cfg"
  env $cfg timeout 300 $B 2>/dev/null | tail -1
done > gpurun_out/r5ab_ab_hpw.jsonl 2>&1
grep -c "^{" gpurun_out/r5ab_ab_hpw.jsonl
