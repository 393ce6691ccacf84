#!/bin/bash
# round 5, visit 13: VERDICT r4 item 8 -- batch 1 (C2) with 8 instead of 4 waves per workgroup on the K = 768 launches (each wave requests 3 chunks
# instead of 6: twice the requesters per launch), and the same build on the C3 bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/chattts_amd/csrc
{
for r in 1 2; do
echo "== C2 default (4 waves)"; python tools/c2_run.py 5 2>/dev/null | tail -1 | cut -c1-160
echo "== C2 nw8 build"; CTTS_LIB=$L/libchattts_amd_nw8.so python tools/c2_run.py 5 2>/dev/null | tail -1 | cut -c1-160
done
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-ttfs --no-configs --no-slot-pool --no-ids-check --no-bf16-parity --no-parity-mode --no-roofline"
echo "== C3 default"; $B 2>/dev/null | tail -1 | cut -c1-140
echo "== C3 nw8 build"; CTTS_LIB=$L/libchattts_amd_nw8.so $B 2>/dev/null | tail -1 | cut -c1-140
} > gpurun_out/r5q_c2_nw8.log 2>&1; cat gpurun_out/r5q_c2_nw8.log
