#!/bin/bash
# round 5, visit 19: where the continuous-batching leg's time goes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
{
timeout 300 python tools/slot_pool_probe.py 2>&1 | grep "^POLL"
timeout 300 python tools/slot_pool_probe.py --no-codec 2>&1 | grep "^POLL"
timeout 300 python tools/slot_pool_probe.py --poll 16 2>&1 | grep "^POLL"
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pool -o pool -- python $R/tools/slot_pool_probe.py --no-codec 2>&1 | grep "^POLL"
f=$(find /tmp/prof_pool -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/r5aa_pool_kernel_stats.csv; cd $R
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r5aa_pool_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("GPU kernel time total (2 pool passes + setup) ms", round(tot/1e6,1))
for r in rows[:28]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {int(r['Calls']):7d} {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:100]}")
PY
} > gpurun_out/r5aa_pool_probe.log 2>&1
cat gpurun_out/r5aa_pool_probe.log
