#!/bin/bash
# one rocprofv3 --pmc pass of the C3 bench with the given counters; prints per-kernel means
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd /tmp; export TMPDIR=/tmp
T=$1; shift
timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$T -o $T -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-ttfs > $R/gpurun_out/${T}_run.log 2>&1
python - "$T" <<'P' > $R/gpurun_out/${T}_counters.txt
import csv, glob, sys, collections
f = glob.glob(f"/tmp/pmc_{sys.argv[1]}/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:70]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in sorted(acc, key=lambda k: -sum(acc[k].values()))[:14]:
    print(k.ljust(72), "  ".join(f"{c}={acc[k][c] / cnt[(k, c)]:.0f} (n={cnt[(k, c)]})" for c in sorted(acc[k])))
P
cat $R/gpurun_out/${T}_counters.txt
