"""Prices VERDICT r4 item 4 (parity mode on split-bf16 projections) BEFORE any kernel is written: how sensitive are the bench
workload's 85,752 draws to an operand perturbation of split-bf16 size?

hi + lo bf16 planes hold an f32 value to 16-17 significant bits; the product hi*hi + hi*lo + lo*hi additionally drops lo*lo (2^-18).  Here
the EXISTING f32 parity engine runs the bench workload (and the C2 golden) with the Llama linear weights rounded to `bits` significant
bits -- weights only, i.e. about HALF the perturbation the real kernels would apply (they would round the activations the same way) --
and the ids are compared with the reference-generated golden (tests/golden/bench_c3.npz).  A flip here is a flip there.
usage: python tools/x3_sensitivity_probe.py [bits ...]   (default: 16 17 24)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from chattts_amd import engine as E, weights as W  # noqa: E402


def round_bits(w: torch.Tensor, bits: int) -> torch.Tensor:
    """round-to-nearest-even to `bits` significant bits (24 = unchanged)"""
    if bits >= 24:
        return w.clone()
    hi = w.to(torch.bfloat16).to(torch.float32)           # 8 bits
    r = w - hi
    if bits == 16:
        return hi + r.to(torch.bfloat16).to(torch.float32)   # hi + lo planes: 8 + 8 (+ sign of the residual)
    # generic: scale trick on the residual
    m, e = torch.frexp(w)
    q = torch.round(m * (1 << bits)) / (1 << bits)
    return torch.ldexp(q, e)


def main():
    dev = torch.device("cuda:0")
    bits_list = [int(a) for a in sys.argv[1:]] or [16, 17, 24]
    gold = np.load(os.path.join(ROOT, "tests", "golden", "bench_c3.npz"))
    wl = bench.shard_workload(64, 1, 0, 128, 512)
    ids_t, mask_t = torch.from_numpy(wl["ids"]), torch.from_numpy(wl["mask"])
    warpers, procs = E.gen_logits(625, 0.7, 20, 1.05)
    want = gold["ids"].astype(np.int64)
    lens = gold["lens"]
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    sds = W.synthetic_all()
    for bits in bits_list:
        sd = {k: (round_bits(v, bits) if (v.dim() == 2 and "proj" in k) else v) for k, v in sds["gpt"].items()}
        n_changed = sum(int((sd[k] != sds["gpt"][k]).sum()) for k in sd)
        eng = E.GptEngine(sd, sds["embed"], dev, dtype="f32")
        emb = eng.embed_prompt(ids_t, torch.from_numpy(wl["tmask"]))
        out = list(eng.generate(emb, ids_t, torch.tensor([0.3] * 4), 625, mask_t, int(wl["stop_all"].max()) + 1, 0, (*procs, *warpers),
                                return_hidden=True, manual_seed=42, stop_at=torch.from_numpy(wl["stop"])))[-1]
        rows = [t.cpu().numpy() for t in out.ids]
        first = []
        for b, r in enumerate(rows):
            w_ = want[starts[b]: starts[b] + lens[b]]
            n = min(len(r), len(w_))
            d = np.nonzero((r[:n] != w_[:n]).any(1))[0]
            first.append(int(d[0]) if len(d) else -1)
        bad = [(b, f) for b, f in enumerate(first) if f >= 0]
        print(f"weights rounded to {bits} significant bits ({n_changed} of 188.8 M values changed): sha256 match {bench.ids_digest(rows) == str(gold['sha256'])}; "
              f"{len(bad)} of 64 utterances diverge from the reference's ids; first divergent step per utterance: {bad[:16]}", flush=True)
        del eng
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
