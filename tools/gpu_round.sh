#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof kernel stats.  Everything lands in gpurun_out/.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
TAG=${1:-r1}
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -5 > gpurun_out/${TAG}_env.log
lscpu | grep -E "Model name|^CPU\(s\)" >> gpurun_out/${TAG}_env.log
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${TAG}_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_tests.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/${TAG}_smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/${TAG}_smoke.log
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/${TAG}_bench.log 2>&1
echo "bench exit $?" >> gpurun_out/${TAG}_bench.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG} -o ${TAG} -- python ${GRAFT_REPO_ROOT:-/root/repo}/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > ${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/${TAG}_rocprof.log 2>&1
echo "rocprof exit $?" >> ${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/${TAG}_rocprof.log
find /tmp/prof_${TAG} -name "*stats*" -exec cp {} ${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/ \; 2>/dev/null
ls -la /tmp/prof_${TAG} >> ${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/${TAG}_rocprof.log 2>&1
tail -5 ${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/${TAG}_tests.log
tail -3 ${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/${TAG}_bench.log
