#!/bin/bash
# One GPU-box visit.  usage: gpu_round.sh TAG "stage stage ..."   stages: tests smoke bench prof pmc
# Everything lands in gpurun_out/ (merged back by gpurun).
set +e
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out
TAG=${1:-r1}
STAGES=${2:-"tests smoke bench prof"}
TESTSEL=${3:-tests}
export TMPDIR=/tmp
lscpu | grep -E "Model name|^CPU\(s\)" > gpurun_out/${TAG}_env.log
for S in $STAGES; do
  case $S in
    tests)
      timeout 900 python -m pytest $TESTSEL -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/${TAG}_tests.log 2>&1
      echo "pytest exit $?" >> gpurun_out/${TAG}_tests.log
      tail -4 gpurun_out/${TAG}_tests.log ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1
      echo "smoke exit $?" >> gpurun_out/${TAG}_smoke.log
      tail -2 gpurun_out/${TAG}_smoke.log ;;
    bench)
      timeout 420 python bench.py --steps 3 --warmup 1 > gpurun_out/${TAG}_bench.log 2>&1
      echo "bench exit $?" >> gpurun_out/${TAG}_bench.log
      tail -3 gpurun_out/${TAG}_bench.log ;;
    prof)
      cd /tmp
      timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o ${TAG} -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode --no-configs --no-slot-pool --no-ids-check > $R/gpurun_out/${TAG}_rocprof.log 2>&1
      echo "rocprof exit $?" >> $R/gpurun_out/${TAG}_rocprof.log
      find /tmp/prof_${TAG} -name "*stats*.csv" -exec cp {} $R/gpurun_out/ \;
      find /tmp/prof_${TAG} -type f >> $R/gpurun_out/${TAG}_rocprof.log
      cd "$R" ;;
    pmc)
      cd /tmp
      for C in FETCH_SIZE WRITE_SIZE; do
        timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_${TAG}_$C -o ${TAG}_$C -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode --no-configs --no-slot-pool --no-ids-check > $R/gpurun_out/${TAG}_pmc_$C.log 2>&1
        if ! ls /tmp/pmc_${TAG}_$C/*counter_collection.csv > /dev/null 2>&1; then   # the profiler crashed once under the pinned-memory polls: plain polls
          rm -rf /tmp/pmc_${TAG}_$C
          CTTS_SYNC_POLL=1 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_${TAG}_$C -o ${TAG}_$C -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode --no-configs --no-slot-pool --no-ids-check > $R/gpurun_out/${TAG}_pmc_${C}_retry.log 2>&1
        fi
      done
      python $R/tools/pmc_summary.py /tmp/pmc_${TAG}_FETCH_SIZE /tmp/pmc_${TAG}_WRITE_SIZE $R/gpurun_out/${TAG}_pmc_traffic.json > $R/gpurun_out/${TAG}_pmc_summary.txt 2>&1
      cd "$R" ;;
  esac
done
