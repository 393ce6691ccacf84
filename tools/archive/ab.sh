#!/bin/bash
# A/B of two builds of the same ABI: libchattts_amd.so (current tree) vs $1 (another .so), interleaved twice.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
OTHER=${1:?usage: ab.sh <path of the other libchattts_amd build, relative to the repo>}
{
for r in 1 2; do
echo "A current"; python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline 2>/dev/null | tail -1 | cut -c1-170
echo "B $OTHER"; CTTS_LIB=$PWD/$OTHER python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ttfs --no-roofline 2>/dev/null | tail -1 | cut -c1-170
done
} | tee gpurun_out/ab.log
