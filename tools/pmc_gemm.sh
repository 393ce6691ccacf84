#!/bin/bash
# PMC counters of the acoustic decoder's point-wise GEMMs alone (tools/x3p_ab.py): L2 hit rate and memory-side traffic per launch
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd /tmp; export TMPDIR=/tmp
TAG=$1; shift
for C in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  N=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pg_${TAG}_$N -o $N -- python $R/tools/x3p_ab.py "$@" > /tmp/pg_${TAG}_$N.log 2>&1
done
python - "$TAG" <<'P' | tee $R/gpurun_out/${TAG}_pmc_gemm.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"/tmp/pg_{sys.argv[1]}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_" not in r["Kernel_Name"]:
            continue
        acc[(r["Kernel_Name"][:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    d = {c: sum(v) / len(v) for c, v in acc[k].items()}
    hit = d.get("TCC_HIT_sum", 0) / max(1.0, d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0))
    print(k[0].ljust(62), "grid", k[1].rjust(8), " ".join(f"{c}={v:.4g}" for c, v in sorted(d.items())), f"L2 hit rate {hit:.3f}")
P
