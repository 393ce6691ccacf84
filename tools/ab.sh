#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{
echo "att NW=8"; CTTS_ATT_NW=8 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline 2>/dev/null | tail -1 | cut -c1-160
echo "att NW=4"; python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline 2>/dev/null | tail -1 | cut -c1-160
} | tee gpurun_out/att_nw_ab.log
