import os, sys, time, cProfile, pstats
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chattts_amd import engine as E, synth, weights as W
dev = torch.device("cuda:0")
gpt = E.GptEngine(W.synthetic_gpt(n_layers=20), W.synthetic_embed(), dev, dtype="bf16")
ids, mask, tmask = synth.make_prompts(64, 16, 48, seed=0)
stop = torch.from_numpy(synth.make_stop_lengths(64, 128, 200, seed=0))
ids_t, mask_t = torch.from_numpy(ids), torch.from_numpy(mask)
emb = gpt.embed_prompt(ids_t, torch.from_numpy(tmask))
warpers, procs = E.gen_logits(625, 0.7, 20, 1.05)
def run(seed):
    return list(gpt.generate(emb, ids_t, torch.tensor([0.3] * 4), 625, mask_t, int(stop.max()) + 1, 0, (*procs, *warpers), return_hidden=True,
                             manual_seed=seed, stop_at=stop))[-1]
run(None)
pr = cProfile.Profile(); pr.enable(); run(None); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
