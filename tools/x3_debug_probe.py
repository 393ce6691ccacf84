"""debug: split-bf16 parity decode (decode32x.hip) vs the f32 MFMA kernels -- hidden-state distance per step for 1 / 2 / 20 layers, per-kernel times"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chattts_amd import engine as E, synth, weights as W  # noqa: E402

dev = torch.device("cuda:0")


def make(nl, exact):
    sds = {"gpt": W.synthetic_gpt(n_layers=nl), "embed": W.synthetic_embed()}
    return E.GptEngine(sds["gpt"], sds["embed"], dev, dtype="f32" if exact else "f32x3", exact_fallback=False)


def run(eng, B, steps, **kw):
    ids, mask, tmask = synth.make_prompts(B, 6, 11, seed=5)
    warpers, procs = E.gen_logits(625, 0.7, 20, 1.05)
    ids_t, mask_t = torch.from_numpy(ids), torch.from_numpy(mask)
    emb = eng.embed_prompt(ids_t, torch.from_numpy(tmask))
    out = list(eng.generate(emb, ids_t, torch.tensor([0.3] * 4), 625, mask_t, steps, steps, (*procs, *warpers), return_hidden=True,
                            manual_seed=11, **kw))[-1]
    return out


for nl in (1, 2, 20):
    ex, x3 = make(nl, True), make(nl, False)
    assert x3.x3 is not None and ex.x3 is None
    for B in (3, 64):
        a, b = run(ex, B, 6), run(x3, B, 6)
        hd = [float((p - q).abs().max()) for p, q in zip(a.hiddens, b.hiddens)]
        same = all(torch.equal(p, q) for p, q in zip(a.ids, b.ids))
        first = a.hiddens[0] - b.hiddens[0]
        print(f"layers {nl} B {B}: ids equal {same}; max |hidden diff| per row (first 4) {hd[:4]}; row 0 per step {[float(r.abs().max()) for r in first]}", flush=True)
    for tag, name in ((1, "qkv"), (3, "attention"), (4, "o_proj"), (5, "gate_up"), (6, "down"), (0, "embed"), (8, "heads"), (9, "sample")):
        t = {}
        for label, eng in (("exact", ex), ("x3", x3)):
            run(eng, 64, 12, use_graph=False, profile_tag=tag, profile_stride=1)
            n, ms = eng.last_stats.get("profile", (0, 0.0))
            t[label] = 1e3 * ms / max(1, n)
        print(f"   layers {nl}: {name:10s} exact {t['exact']:7.2f} us   x3 {t['x3']:7.2f} us", flush=True)
    del ex, x3
    torch.cuda.empty_cache()
