"""Summarise rocprofv3 --pmc CSV runs: per kernel name, dispatch count and mean counter value.
usage: pmc_summary.py <dir_FETCH_SIZE> <dir_WRITE_SIZE> <out.json>
FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE reads exactly half of the bytes a
wide coalesced stream fetches (MI355X_MICROARCH.md, HBM section), so read bytes = 2 * FETCH_SIZE * 1024."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def collect(root, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                a = acc[row.get("Kernel_Name", "?")]
                a[0] += 1
                a[1] += float(row.get("Counter_Value", 0) or 0)
    return acc


def short(name):
    """rocprofv3 kernel name -> bench.py tag.  gemm_dec_k<MBT, NW, SCALE, EPI> (decode.hip, the packed decode projections):
    EPI 3 = QKV+RoPE, 2 = SiLU gate/up, 1 = residual (NW 4: o_proj, K = 768; NW 8/16: down_proj, K = 3072);
    gemm_fast_k<MB, NW, SCALE, EPI> are the row-major kernels (prefill below 256 rows, CTTS_DEC_PACKED=0)."""
    import re
    if "attention_k<float" in name:
        return "attention_f32"      # parity mode (f32 KV cache): bench.py's parity_mode.roofline
    if "attention_k" in name:
        return "attention"
    m = re.search(r"gemm_(dec|fast)_k<(\d+), (\d+), (true|false), (\d+)(?:, \d+)?>", name)   # (<MBT, NW, SCALE, EPI[, U]>: U since round 5)
    if m:
        nw, epi = int(m.group(3)), int(m.group(5))
        if epi == 3:
            return "qkv_gemm"
        if epi == 2:
            return "gate_up_gemm"
        if epi == 1:
            return "o_proj_gemm" if nw == 4 else "down_gemm"
    m = re.search(r"gemm_dec32x_k<(\d+), (\d+), (true|false), (\d+)(?:, \d+)?>", name)
    if m:                           # parity mode, split-bf16 decode projections (decode32x.hip): <MBT, K, RMS, EPI[, NW]>
        kt, epi = int(m.group(2)), int(m.group(4))
        return {100: "qkv_gemm_f32", 2: "gate_up_gemm_f32"}.get(epi, "o_proj_gemm_f32" if kt == 768 else "down_gemm_f32")
    if "gemm_dec32_fnorm16_k" in name or "gemm_dec32_m16_k<768, 0>" in name or "gemm_skinny_k<float" in name:
        return "heads_gemm"         # fused final-norm + heads | packed heads | row-major heads
    if "sample_k" in name:
        return "sample"
    if "dwconv_ln_run_k" in name:
        return "dwconv_ln_run"      # large batches: a wave walks 36 frames (codec.hip)
    if "dwconv_ln_k" in name:
        return "dwconv_ln"
    m = re.search(r"gemm_h1p_k<(\d+)", name)
    if m:
        return "codec_pwconv1_h1p" if int(m.group(1)) == 0 else "codec_pwconv2_h1p"   # the perf mode's decoder GEMMs (codec_gemm.hip)
    m = re.search(r"gemm_x3p_k<(\d+)", name)
    if m:
        return "codec_pwconv1_x3p" if int(m.group(1)) == 0 else "codec_pwconv2_x3p"
    return None


def main(d_fetch, d_write, out):
    fe, wr = collect(d_fetch, "FETCH_SIZE"), collect(d_write, "WRITE_SIZE")
    res = {}
    if os.path.exists(out):      # a second call (e.g. the f32 passes) ADDS its kernels to the file
        with open(out) as fh:
            res = json.load(fh)
    print("kernel, dispatches, mean FETCH_SIZE KiB, mean WRITE_SIZE KiB, HBM bytes/launch (2*FETCH+WRITE)")
    for k, (n, tot) in sorted(fe.items(), key=lambda kv: -kv[1][1]):
        f = tot / max(n, 1)
        w = wr.get(k, [0, 0.0])
        wv = w[1] / max(w[0], 1)
        hbm = int((2 * f + wv) * 1024)
        print(f"{k[:80]:80s} {n:7d} {f:12.1f} {wv:12.1f} {hbm:14d}")
        tag = short(k)
        if tag and (tag not in res or n > res[tag]["dispatches"]):
            res[tag] = {"dispatches": n, "fetch_kib_mean": round(f, 2), "write_kib_mean": round(wv, 2), "hbm_bytes_per_launch": hbm}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:4])
