"""How often does the split-bf16 parity arithmetic ("f32x3") actually sample another token than the exact f32 kernels ("f32")?
Free-running generation of the same random batches on both engines (random prompts, seeds, sampling parameters in the range
`InferCodeParams` is used in), counted per utterance up to its first divergence: draws compared, divergences seen.  The parity
certificate's worst-case bound cannot certify long utterances; this is the empirical rate beside it.
Run on the GPU box:  python tools/x3_flip_rate.py [n_batches] > gpurun_out/r6_x3_flip_rate.log"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chattts_amd import engine as E, synth  # noqa: E402
from chattts_amd import weights as W  # noqa: E402


def main():
    n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda:0")
    sds = W.synthetic_all()
    ex = E.GptEngine(sds["gpt"], sds["embed"], dev, dtype="f32")
    x3 = E.GptEngine(sds["gpt"], sds["embed"], dev, dtype="f32x3", exact_fallback=False)
    rs = np.random.RandomState(2026)
    draws = flips = utts = 0
    min_margin_of_flipped, margins_all = [], []
    t0 = time.time()
    for it in range(n_batches):
        B = 160
        steps = int(rs.choice([96, 160, 256]))
        ids, mask, tmask = synth.make_prompts(B, 8, 40, seed=int(rs.randint(1 << 30)))
        temp = float(rs.choice([0.1, 0.3, 0.7]))
        top_p, top_k, rep = float(rs.choice([0.5, 0.7, 0.9])), int(rs.choice([10, 20, 50])), float(rs.choice([1.0, 1.05, 1.2]))
        seed = int(rs.randint(1 << 30))
        warpers, procs = E.gen_logits(625, top_p, top_k, rep)
        ids_t, mask_t = torch.from_numpy(ids), torch.from_numpy(mask)
        outs = []
        for eng in (ex, x3):
            emb = eng.embed_prompt(ids_t, torch.from_numpy(tmask))
            o = list(eng.generate(emb, ids_t, torch.tensor([temp] * 4), 625, mask_t, steps, steps, (*procs, *warpers), return_hidden=False,
                                  manual_seed=seed))[-1]
            outs.append([t.cpu().numpy() for t in o.ids])
        mg = x3.last_margins
        bound = x3.last_stats["margin_bound"]
        for b in range(B):
            a, c = outs[0][b], outs[1][b]
            n = min(len(a), len(c))
            neq = np.nonzero((a[:n] != c[:n]).any(1))[0]
            first = int(neq[0]) if len(neq) else n
            draws += 4 * (first + (1 if len(neq) else 0))
            utts += 1
            margins_all.append(float(mg[b]))
            if len(neq):
                flips += 1
                min_margin_of_flipped.append((float(mg[b]), float(bound)))
        print(f"batch {it}: B={B} steps={steps} T={temp} top_p={top_p} top_k={top_k} rep={rep}: cumulative {draws} draws, {flips} diverged utterances "
              f"of {utts}, {time.time() - t0:.0f}s", flush=True)
    m = np.array(margins_all)
    print(json.dumps({"draws_compared": draws, "utterances": utts, "utterances_that_diverged": flips, "rate_per_draw": flips / max(1, draws),
                      "expected_per_C3_pass_of_85752_draws": 85752 * flips / max(1, draws),
                      "diverged_utterances_margin_and_bound": min_margin_of_flipped,
                      "all_diverged_were_flagged_by_the_certificate": all(mm < bb for mm, bb in min_margin_of_flipped),
                      "utterance_min_margin_percentiles_1_10_50": [float(np.percentile(m, p)) for p in (1, 10, 50)]}, indent=1))


if __name__ == "__main__":
    main()
