"""Regression guard of the PERF mode's free-running token stream: the bf16 engine's own ids for case `b8`, stored as
tests/golden/bf16_guard.npz.  Not a parity fixture (the bf16 mode has no bit-exact bar): a deliberate change of the bf16 arithmetic moves
rows apart after some tens of steps and this file is regenerated; a BUG moves them apart at once.  GPU box:  python tools/make_bf16_guard.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chattts_amd import engine as E  # noqa: E402
from chattts_amd import weights as W  # noqa: E402
from oracle import cases  # noqa: E402


def run(eng, c, use_graph=True):
    ids, mask, tmask = cases.gen_inputs(c)
    ids_t, mask_t = torch.from_numpy(ids), torch.from_numpy(mask)
    emb = eng.embed_prompt(ids_t, torch.from_numpy(tmask))
    warpers, procs = E.gen_logits(625, c["top_P"], c["top_K"], c["rep"])
    out = list(eng.generate(emb, ids_t, torch.tensor(c["temperature"]), 625, mask_t, c["max_new"], c["min_new"], (*procs, *warpers),
                            return_hidden=False, manual_seed=c["manual_seed"], use_graph=use_graph))[-1]
    return [t.cpu().numpy() for t in out.ids]


if __name__ == "__main__":
    sds = W.synthetic_all()
    eng = E.GptEngine(sds["gpt"], sds["embed"], torch.device("cuda:0"), dtype="bf16")
    rows = run(eng, cases.GEN_CASES["b8"])
    out = os.path.join(ROOT, "gpurun_out" if os.environ.get("GRAFT_REPO_ROOT") else "tests/golden", "bf16_guard.npz")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    np.savez_compressed(out, lens=np.array([len(r) for r in rows]), ids=np.concatenate(rows, 0))
    print("wrote", out, [len(r) for r in rows])
