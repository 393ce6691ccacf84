#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R; mkdir -p gpurun_out
python tools/codec_probe.py 2>/dev/null | tail -1 | tee gpurun_out/codec_pmc.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d /tmp/pmc_codec -o codec -- python $R/tools/codec_probe.py > /dev/null 2>&1
python - <<'PY' | tee -a $R/gpurun_out/codec_pmc.log
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob('/tmp/pmc_codec/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:64]
        a = acc[k][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k, d in acc.items():
    if 'bf16x3_k<4, 4' in k or 'bf16x3_k<5, 4' in k:
        print(k, {c: round(v[1] / v[0]) for c, v in d.items()})
PY
