"""Seeded vs unseeded (manual_seed=None: host draws fresh Exp(1) tensors every step, like the reference's global
CPU generator) generation speed on the C3 batch, GPT part only; and unseeded with the opt-in device generator (rng="device")."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chattts_amd import engine as E, synth, weights as W
dev = torch.device("cuda:0")
gpt = E.GptEngine(W.synthetic_gpt(), W.synthetic_embed(), dev, dtype="bf16")
ids, mask, tmask = synth.make_prompts(64, 16, 48, seed=0)
stop = torch.from_numpy(synth.make_stop_lengths(64, 128, 512, seed=0))
ids_t, mask_t = torch.from_numpy(ids), torch.from_numpy(mask)
emb = gpt.embed_prompt(ids_t, torch.from_numpy(tmask))
warpers, procs = E.gen_logits(625, 0.7, 20, 1.05)
def run(seed, rng="host"):
    torch.manual_seed(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = list(gpt.generate(emb, ids_t, torch.tensor([0.3] * 4), 625, mask_t, int(stop.max()) + 1, 0, (*procs, *warpers), return_hidden=True,
                            manual_seed=seed, stop_at=stop, rng=rng))[-1]
    torch.cuda.synchronize(); return time.perf_counter() - t0, sum(int(t.shape[0]) for t in out.ids)
run(42); run(None); run(None, "device")
a = min(run(42) for _ in range(3)); b = min(run(None) for _ in range(2)); c = min(run(None, "device") for _ in range(3))
print(json.dumps({"seeded_host_s": round(a[0], 3), "unseeded_host_s": round(b[0], 3), "unseeded_device_rng_s": round(c[0], 3), "tokens": a[1],
                  "unseeded_host_over_seeded": round(b[0] / a[0], 2), "unseeded_device_over_seeded": round(c[0] / a[0], 2)}))
