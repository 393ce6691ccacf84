#!/bin/bash
# round 5 evidence run: full GPU suite, smoke, the driver's bench command, rocprofv3 kernel stats (bf16 leg and the f32 parity leg), PMC traffic
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r5j}
STAGES=${2:-"tests smoke bench prof prof32 pmc pmc32"}
NOLEGS="--no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode --no-configs --no-slot-pool --no-ids-check"
for S in $STAGES; do
  case $S in
    tests)
      timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/${TAG}_tests.log 2>&1
      echo "pytest exit $?" >> gpurun_out/${TAG}_tests.log; grep -E "passed|failed|pytest exit" gpurun_out/${TAG}_tests.log | tail -3 ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/${TAG}_smoke.log; tail -2 gpurun_out/${TAG}_smoke.log ;;
    bench)
      timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/${TAG}_bench.log 2>&1; echo "bench exit $?" >> gpurun_out/${TAG}_bench.log
      grep "^{" gpurun_out/${TAG}_bench.log | tail -1 | cut -c1-600 ;;
    prof)
      cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o ${TAG} -- python $R/bench.py --steps 1 --warmup 0 $NOLEGS > $R/gpurun_out/${TAG}_rocprof.log 2>&1
      f=$(find /tmp/prof_${TAG} -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/${TAG}_kernel_stats.csv; cd $R; head -8 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-160 ;;
    prof32)
      cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof32_${TAG} -o ${TAG}32 -- python $R/bench.py --dtype f32 --steps 1 --warmup 0 $NOLEGS > $R/gpurun_out/${TAG}_rocprof_f32.log 2>&1
      f=$(find /tmp/prof32_${TAG} -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/${TAG}_kernel_stats_f32.csv; cd $R; head -8 gpurun_out/${TAG}_kernel_stats_f32.csv | cut -c1-160 ;;
    pmc|pmc32)
      cd /tmp
      D=""; SUF=bf16; [ $S = pmc32 ] && { D="--dtype f32"; SUF=f32; }
      for C in FETCH_SIZE WRITE_SIZE; do
        CTTS_SYNC_POLL=1 timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_${TAG}_${SUF}_$C -o ${TAG}_$C -- python $R/bench.py $D --steps 1 --warmup 0 $NOLEGS > $R/gpurun_out/${TAG}_pmc_${SUF}_$C.log 2>&1
      done
      python $R/tools/pmc_summary.py /tmp/pmc_${TAG}_${SUF}_FETCH_SIZE /tmp/pmc_${TAG}_${SUF}_WRITE_SIZE $R/gpurun_out/${TAG}_pmc_traffic.json > $R/gpurun_out/${TAG}_pmc_summary_${SUF}.txt 2>&1
      cd $R; head -12 gpurun_out/${TAG}_pmc_summary_${SUF}.txt | cut -c1-150 ;;
  esac
done
