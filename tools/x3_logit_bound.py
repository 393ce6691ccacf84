"""Measures what the parity certificate's bound rests on (GptEngine.REL_ERR_X3): the distance between the split-bf16 ("f32x3") and
the exact f32 ("f32") decode arithmetic on the bench workload, both engines TEACHER-FORCED on the reference's own token stream
(tests/golden/bench_c3.npz) so that every step of every row is compared under the same history.

Prints: max / rms |dlogit| (raw and relative to the head's logit scale), the teacher-forced token agreement of the two samplers, and the
distribution of the per-draw decision margins (float64 restatement, oracle/sampling_np.decision_margin, on the exact engine's logits):
how many of the 85,752 draws sit below 2 * eps / temperature for the measured and the stated eps -- i.e. what a certificate at that bound
flags.  Run on the GPU box:  python tools/x3_logit_bound.py > gpurun_out/r6_x3_logit_bound.log
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from chattts_amd import engine as E, rng  # noqa: E402
from chattts_amd import weights as W  # noqa: E402
from chattts_amd.config import GPT  # noqa: E402
from oracle import sampling_np  # noqa: E402  (test infrastructure: this is a measurement tool, not the product path)


def main():
    dev = torch.device("cuda:0")
    sds = W.synthetic_all()
    wl = bench.shard_workload(64, 1, 0, 128, 512)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "bench_c3.npz"))
    stop = wl["stop_all"]
    max_new = int(stop.max()) + 1
    lens_g, rows_g, teacher = bench.teacher_from_golden(gold, max_new)
    ids_t, mask_t, tm_t = torch.from_numpy(wl["ids"]), torch.from_numpy(wl["mask"]), torch.from_numpy(wl["tmask"])
    warpers, procs = E.gen_logits(625, 0.7, 20, 1.05)
    temp = torch.tensor([0.3] * 4)
    res = {}
    for dt in ("f32", "f32x3"):
        eng = E.GptEngine(sds["gpt"], sds["embed"], dev, dtype=dt, certify=False, exact_fallback=False)
        emb = eng.embed_prompt(ids_t.to(dev), tm_t.to(dev))
        out = None
        for out in eng.generate(emb, ids_t.to(dev), temp, 625, mask_t, max_new, 0, (*procs, *warpers), return_hidden=True, manual_seed=42,
                                stop_at=torch.from_numpy(wl["stop"]), total_rows=256, teacher_ids=torch.from_numpy(teacher), return_sampled=True):
            pass
        res[dt] = ([h.cpu().numpy() for h in out.hiddens], [t.cpu().numpy() for t in eng.last_sampled])
        scale = eng.logit_scale[False]
        del eng
        torch.cuda.empty_cache()
    heads = bench.generate_heads(sds["embed"]).astype(np.float64)
    q = rng.ExpDraws(256, 626, 42).step(0).numpy()
    pt = rng.penalty_table(1.05).numpy()
    dl_max, dl_sq, n_l, agree, total = 0.0, 0.0, 0, 0, 0
    margins = []
    for b in range(64):
        h32, hx = res["f32"][0][b].astype(np.float64), res["f32x3"][0][b].astype(np.float64)
        n = h32.shape[0]
        l32 = h32 @ heads.T                      # [n, 2504]
        d = (hx - h32) @ heads.T
        dl_max = max(dl_max, float(np.abs(d).max()))
        dl_sq += float((d ** 2).sum()); n_l += d.size
        s32, sx = res["f32"][1][b], res["f32x3"][1][b]
        agree += int((s32 == sx).sum()); total += s32.size
        # per-draw margins on the exact engine's logits: step i of row b has history = the reference's tokens before it
        ids_b = rows_g[b]
        for i in range(n):
            lg = l32[i].reshape(4, 626).astype(np.float32)
            hist = ids_b[max(0, i - 16): i].T.reshape(4, -1) if i else np.zeros((4, 0), np.int64)
            m = sampling_np.decision_margin(lg, hist.astype(np.int64), q[4 * b: 4 * b + 4], temperature=np.full(4, 0.3, np.float32), top_p=0.7, top_k=20,
                                            pow_table=pt, max_input_ids=625, mask_eos=bool(i < stop[b]), row_offset=4 * b)
            margins.append(m)
    mg = np.concatenate(margins)
    rms = float(np.sqrt(dl_sq / n_l))
    out = {"draws": int(mg.size), "logit_scale": round(scale, 4), "max_abs_dlogit": dl_max, "rms_dlogit": rms,
           "max_rel_to_scale": dl_max / scale, "rms_rel_to_scale": rms / scale,
           "teacher_forced_sampler_agreement_x3_vs_exact": agree / total, "draws_that_differ": int(total - agree),
           "margin_min": float(mg.min()), "margin_percentiles_1e-4_1e-3_1e-2": [float(np.percentile(mg, p)) for p in (0.01, 0.1, 1.0)],
           "stated_REL_ERR_X3": E.GptEngine.REL_ERR_X3}
    for name, eps in (("measured_max", dl_max), ("stated", E.GptEngine.REL_ERR_X3 * scale), ("rms", rms)):
        thr = 2.0 * eps / 0.3
        per_utt = sum(1 for b in range(64) if np.concatenate(margins[sum(lens_g[:b]): sum(lens_g[:b + 1])]).min() < thr) if False else None
        out["draws_below_2eps_over_T[%s]" % name] = {"eps": eps, "threshold": thr, "draws": int((mg < thr).sum())}
    # per-utterance: how many utterances a certificate at the stated bound flags
    off = np.concatenate([[0], np.cumsum(lens_g)])
    thr = 2.0 * E.GptEngine.REL_ERR_X3 * scale / 0.3
    out["utterances_flagged_at_stated_bound"] = int(sum(1 for b in range(64) if np.concatenate(margins[off[b]: off[b + 1]]).min() < thr))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
