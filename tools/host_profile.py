"""cProfile of one C3 pass (generate + decode_to_wavs): where the host time goes (the GPU is busy throughout; what matters is
what sits before the first launch and between launches)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chattts_amd import engine as E, synth, weights as W  # noqa: E402

dev = torch.device("cuda:0")
sds = W.synthetic_all()
gpt = E.GptEngine(sds["gpt"], sds["embed"], dev, dtype="bf16")
codec = E.CodecEngine(sds["decoder"], sds["vocos"], dev)
ids, mask, tmask = synth.make_prompts(64, 16, 48, seed=0)
stop = torch.from_numpy(synth.make_stop_lengths(64, 128, 512, seed=0))
ids_t, mask_t, tm_t = torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(tmask)
warpers, procs = E.gen_logits(625, 0.7, 20, 1.05)
temp = torch.tensor([0.3] * 4)


def one():
    t0 = time.perf_counter()
    emb = gpt.embed_prompt(ids_t, tm_t)
    out = None
    first = None
    for out in gpt.generate(emb, ids_t.to(dev), temp, 625, mask_t, 513, 0, (*procs, *warpers), return_hidden=True, manual_seed=42, stop_at=stop):
        pass
    t1 = time.perf_counter()
    wav = codec.decode_to_wavs(out.hiddens)
    torch.cuda.synchronize()
    return t1 - t0, time.perf_counter() - t1


for _ in range(2):
    one()
pr = cProfile.Profile()
pr.enable()
a, b = one()
pr.disable()
print(f"generate {a * 1e3:.1f} ms, codec {b * 1e3:.1f} ms")
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
