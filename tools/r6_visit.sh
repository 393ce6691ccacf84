#!/bin/bash
# round 6: one GPU-box visit.  usage: r6_visit.sh TAG "stage ..."   stages: tests testsel:<expr> smoke bound bench benchd prof pmc
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6a}
STAGES=${2:-"tests smoke bench"}
NOLEGS="--no-projection --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode --no-configs --no-slot-pool --no-ids-check --no-bf16-mode --no-refine-text"
for S in $STAGES; do
  case $S in
    tests)
      timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${TAG}_tests.log 2>&1
      echo "pytest exit $?" >> gpurun_out/${TAG}_tests.log; grep -E "^FAILED|^ERROR|passed|failed|pytest exit" gpurun_out/${TAG}_tests.log | tail -25 ;;
    testsel:*)
      timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "$(echo "${S#testsel:}" | tr '+' ' ')" > gpurun_out/${TAG}_testsel.log 2>&1
      echo "pytest exit $?" >> gpurun_out/${TAG}_testsel.log; grep -E "^FAILED|^ERROR|passed|failed|pytest exit" gpurun_out/${TAG}_testsel.log | tail -25 ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/${TAG}_smoke.log; tail -2 gpurun_out/${TAG}_smoke.log ;;
    bound)
      timeout 900 python tools/x3_logit_bound.py > gpurun_out/${TAG}_x3_logit_bound.log 2>&1; echo "exit $?" >> gpurun_out/${TAG}_x3_logit_bound.log; tail -40 gpurun_out/${TAG}_x3_logit_bound.log ;;
    bench)
      timeout 1200 python bench.py --steps 3 --warmup 1 > gpurun_out/${TAG}_bench.log 2>&1; echo "bench exit $?" >> gpurun_out/${TAG}_bench.log
      grep "^{" gpurun_out/${TAG}_bench.log | tail -1 | cut -c1-1500; tail -3 gpurun_out/${TAG}_bench.log | cut -c1-300 ;;
    benchd)   # the driver's command
      timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_benchd.log 2>&1; echo "bench exit $?" >> gpurun_out/${TAG}_benchd.log
      grep "^{" gpurun_out/${TAG}_benchd.log | tail -1 | cut -c1-1500 ;;
    benchq)   # headline leg only
      timeout 600 python bench.py --steps 5 --warmup 2 $NOLEGS > gpurun_out/${TAG}_benchq.log 2>&1; echo "bench exit $?" >> gpurun_out/${TAG}_benchq.log
      grep "^{" gpurun_out/${TAG}_benchq.log | tail -1 | cut -c1-400 ;;
    foldtests) # e2e ids / hidden-state goldens with the embedding fold on
      CTTS_EMBED_FOLD=1 timeout 1800 python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -p no:cacheprovider -k "bit_exact or bench_workload or teacher or certificate or shard or interrupt or stream" > gpurun_out/${TAG}_foldtests.log 2>&1
      echo "pytest exit $?" >> gpurun_out/${TAG}_foldtests.log; grep -E "^FAILED|^ERROR|passed|failed|pytest exit" gpurun_out/${TAG}_foldtests.log | tail -12 ;;
    ab:*)     # ab:<ENV=VAL>: the headline leg (10 timed passes) and the bf16 leg with / without one environment switch, alternating
      KV="${S#ab:}"
      for rep in 1 2; do for on in 1 0; do
        if [ $on = 1 ]; then export "$KV"; else unset "${KV%%=*}"; fi
        for D in f32x3 bf16; do
          timeout 600 python bench.py --dtype $D --steps 10 --warmup 2 $NOLEGS 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$KV on=$on $D', d['value'], d['ms_per_step'])" | tee -a gpurun_out/${TAG}_ab.log
        done
      done; done; unset "${KV%%=*}" ;;
    ttfs0)    # time to first sample with the prompt pass over all rows
      CTTS_PRE_COMPACT=0 timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-parity-mode --no-configs --no-slot-pool --no-ids-check --no-bf16-mode --no-refine-text > gpurun_out/${TAG}_ttfs0.log 2>&1
      grep "^{" gpurun_out/${TAG}_ttfs0.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no compaction', {k: d.get(k) for k in ('value','ttfs_ms_p50','ttfs_ms_cold','ttfs_ms_cold_prewarmed')})" ;;
    torchrun1) # the N-rank code path on one GPU: torch.distributed.run, RCCL world 1, dist.infer_sharded inside the clock
      timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_torchrun1.log 2>&1
      echo "exit $?" >> gpurun_out/${TAG}_torchrun1.log; grep "^{" gpurun_out/${TAG}_torchrun1.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('value','n_gpus','data_parallel_entry')}, d.get('ranks',{}).get('weight_broadcast',{}).get('ms'), d['ids_check'])"; tail -2 gpurun_out/${TAG}_torchrun1.log | cut -c1-200 ;;
    benchqc0) # headline leg only, prompt pass over all B * T rows (A/B of the valid-token compaction)
      CTTS_PRE_COMPACT=0 timeout 600 python bench.py --steps 5 --warmup 2 $NOLEGS > gpurun_out/${TAG}_benchqc0.log 2>&1; echo "bench exit $?" >> gpurun_out/${TAG}_benchqc0.log
      grep "^{" gpurun_out/${TAG}_benchqc0.log | tail -1 | cut -c1-400 ;;
    benchq0)  # headline leg only, prompt pass on the f32 MFMA kernels (A/B of prefill32x.hip)
      CTTS_PRE_X3=0 timeout 600 python bench.py --steps 5 --warmup 2 $NOLEGS > gpurun_out/${TAG}_benchq0.log 2>&1; echo "bench exit $?" >> gpurun_out/${TAG}_benchq0.log
      grep "^{" gpurun_out/${TAG}_benchq0.log | tail -1 | cut -c1-400 ;;
    ttfs)     # time to first sample, headline engine
      timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-parity-mode --no-configs --no-slot-pool --no-ids-check --no-bf16-mode --no-refine-text > gpurun_out/${TAG}_ttfs.log 2>&1
      grep "^{" gpurun_out/${TAG}_ttfs.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('value','ttfs_ms_p50','ttfs_ms_cold','ttfs_ms_cold_prewarmed')})" ;;
    benchfb)  # with the exact fallback inside the timed passes
      timeout 600 python bench.py --steps 3 --warmup 1 --exact-fallback $NOLEGS > gpurun_out/${TAG}_benchfb.log 2>&1; echo "bench exit $?" >> gpurun_out/${TAG}_benchfb.log
      grep "^{" gpurun_out/${TAG}_benchfb.log | tail -1 | cut -c1-1500 ;;
    prof|prof16)
      D=""; SUF=f32x3; [ $S = prof16 ] && { D="--dtype bf16"; SUF=bf16; }
      cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_$SUF -o ${TAG} -- python $R/bench.py $D --steps 1 --warmup 0 $NOLEGS > $R/gpurun_out/${TAG}_rocprof_$SUF.log 2>&1
      f=$(find /tmp/prof_${TAG}_$SUF -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/${TAG}_kernel_stats_$SUF.csv; cd $R; head -8 gpurun_out/${TAG}_kernel_stats_$SUF.csv | cut -c1-160 ;;
    pmc|pmc16)
      cd /tmp
      D=""; SUF=f32; [ $S = pmc16 ] && { D="--dtype bf16"; SUF=bf16; }
      for C in FETCH_SIZE WRITE_SIZE; do
        CTTS_SYNC_POLL=1 timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_${TAG}_${SUF}_$C -o ${TAG}_$C -- python $R/bench.py $D --steps 1 --warmup 0 $NOLEGS > $R/gpurun_out/${TAG}_pmc_${SUF}_$C.log 2>&1
      done
      python $R/tools/pmc_summary.py /tmp/pmc_${TAG}_${SUF}_FETCH_SIZE /tmp/pmc_${TAG}_${SUF}_WRITE_SIZE $R/gpurun_out/${TAG}_pmc_traffic.json > $R/gpurun_out/${TAG}_pmc_summary_${SUF}.txt 2>&1
      cd $R; head -12 gpurun_out/${TAG}_pmc_summary_${SUF}.txt | cut -c1-150 ;;
    proftext)  # rocprofv3 kernel trace of the refine-text legs
      cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_text -o ${TAG} -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode --no-configs --no-slot-pool --no-ids-check --no-bf16-mode > $R/gpurun_out/${TAG}_rocprof_text.log 2>&1
      f=$(find /tmp/prof_${TAG}_text -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/${TAG}_kernel_stats_text.csv; cd $R; grep -E "sample_text|fnorm16|embed_text|Name" gpurun_out/${TAG}_kernel_stats_text.csv | cut -c1-200 ;;
    share2)   # the N = 2 path on ONE GPU (functional, not a measurement): two ranks share GPU 0 and meet over gloo
      timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --share-gpu --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-ttfs --no-configs --no-slot-pool --no-bf16-mode --no-refine-text > gpurun_out/${TAG}_share2.log 2>&1
      echo "exit $?" >> gpurun_out/${TAG}_share2.log; grep "^{" gpurun_out/${TAG}_share2.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('value','n_gpus','data_parallel_entry','shared_gpu_debug')}, d['ids_check'])"; tail -3 gpurun_out/${TAG}_share2.log | cut -c1-300 ;;
    mlpab)    # the ConvNeXt MLP as two launches / as one (tools/mlp_ab.py) + the kernel-level bit-identity test
      timeout 600 python tools/mlp_ab.py > gpurun_out/${TAG}_mlp_ab.log 2>&1; echo "exit $?" >> gpurun_out/${TAG}_mlp_ab.log; tail -12 gpurun_out/${TAG}_mlp_ab.log | cut -c1-260 ;;
    mlpprobe) # in-kernel phase stamps of the fused MLP
      timeout 300 python tools/mlp_phase_probe.py > gpurun_out/${TAG}_mlp_phase.log 2>&1; INTER=1536 M=9216 timeout 300 python tools/mlp_phase_probe.py >> gpurun_out/${TAG}_mlp_phase.log 2>&1; tail -22 gpurun_out/${TAG}_mlp_phase.log | cut -c1-200 ;;
    mlpipos)  # fused MLP: where in the ring step a wave issues its refill (CTTS_MLP_IPOS 0 by SIMD pair / 1 wave % 4 / 2 all behind the barrier)
      for m in 0 1 2 0 1 2; do echo "CTTS_MLP_IPOS=$m" >> gpurun_out/${TAG}_mlp_ipos.log; CTTS_MLP_IPOS=$m timeout 300 python tools/mlp_ab.py 2>&1 | grep "M=65536\|M=32768" | cut -c1-200 >> gpurun_out/${TAG}_mlp_ipos.log; done; cat gpurun_out/${TAG}_mlp_ipos.log ;;
    proj)     # the weak-scaling projection leg alone
      timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode --no-configs --no-slot-pool --no-bf16-mode --no-refine-text > gpurun_out/${TAG}_proj.log 2>&1; echo "exit $?" >> gpurun_out/${TAG}_proj.log
      grep "^{" gpurun_out/${TAG}_proj.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], json.dumps(d.get('weak_scaling_projection'))[:1500])"; tail -2 gpurun_out/${TAG}_proj.log | cut -c1-300 ;;
    pmcdw)    # per-dispatch WRITE_SIZE / FETCH_SIZE of the depthwise-conv + LayerNorm launches of one pass (DVAE: 12 x dilation 2, then Vocos: 8 x dilation 1)
      cd /tmp
      for C in WRITE_SIZE FETCH_SIZE; do
        CTTS_SYNC_POLL=1 timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmcdw_${TAG}_$C -o ${TAG}_$C -- python $R/bench.py --steps 1 --warmup 0 $NOLEGS > $R/gpurun_out/${TAG}_pmcdw_$C.log 2>&1
        python - "$C" "/tmp/pmcdw_${TAG}_$C" <<'P' >> $R/gpurun_out/${TAG}_pmcdw.txt
import csv, glob, sys
c, d = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "dwconv_ln" in r["Kernel_Name"] and r["Counter_Name"] == c]
print(c, "KiB per dwconv_ln dispatch, in launch order:", [round(v) for v in vals])
P
      done
      cd $R; cat gpurun_out/${TAG}_pmcdw.txt | cut -c1-600 ;;
    dwab)     # depthwise conv + LayerNorm: transposed plane writes (default) vs isolated 16-byte stores (CTTS_DWCONV_SEQ=0): kernel stats of one pass each
      for m in 1 0; do
        cd /tmp; CTTS_DWCONV_SEQ=$m timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dw_${TAG}_$m -o ${TAG} -- python $R/bench.py --steps 1 --warmup 0 $NOLEGS > $R/gpurun_out/${TAG}_dwab_$m.log 2>&1
        f=$(find /tmp/dw_${TAG}_$m -name "*kernel_stats.csv" | head -1); echo "CTTS_DWCONV_SEQ=$m" >> $R/gpurun_out/${TAG}_dwab.txt; grep "dwconv\|gemm_x3p" $f | cut -c1-60,150-260 >> $R/gpurun_out/${TAG}_dwab.txt; cd $R
      done; cat gpurun_out/${TAG}_dwab.txt ;;
    cumask)   # queue of batches (--pipeline: decode of batch i on the side stream under the generation of batch i + 1): side stream unconfined / on n CUs
      for spec in "" "32" "64" "32:8" "16" "" "32" "64:4"; do
        for D in f32x3 bf16; do
          CTTS_CODEC_CUS="$spec" timeout 600 python bench.py --pipeline --dtype $D --steps 8 --warmup 2 $NOLEGS 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('CTTS_CODEC_CUS=[$spec] $D pipelined', d['value'], d['ms_per_step'])" | tee -a gpurun_out/${TAG}_cumask.log
        done
      done ;;
    cumask2)  # as cumask, the generator confined to the complement of the decoder's CUs
      for spec in "" "32:8" "64:4" "32" "16:16" "32:8"; do
        for D in f32x3 bf16; do
          CTTS_GPT_COMPLEMENT=1 CTTS_CODEC_CUS="$spec" timeout 600 python bench.py --pipeline --dtype $D --steps 8 --warmup 2 $NOLEGS 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('complement CTTS_CODEC_CUS=[$spec] $D pipelined', d['value'], d['ms_per_step'])" | tee -a gpurun_out/${TAG}_cumask2.log
        done
      done ;;
    reftext)  # refine-text legs + a kernel trace of them
      timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs --no-parity-mode --no-configs --no-slot-pool --no-ids-check --no-bf16-mode > gpurun_out/${TAG}_reftext.log 2>&1
      grep "^{" gpurun_out/${TAG}_reftext.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d['configs']['refine_text'], indent=1))" | head -80 ;;
  esac
done
