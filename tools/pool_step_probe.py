"""SlotPool's decode step against GptEngine.generate's on the SAME rows: 64 utterances, all forced to 384 tokens (no admission, no
retirement until the end) -- what a pool step costs beyond a generate step."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chattts_amd import engine as E, weights as W, synth  # noqa: E402
from chattts_amd.serving import SlotPool  # noqa: E402

dev = torch.device("cuda:0")
sds = W.synthetic_all()
gpt = E.GptEngine(sds["gpt"], sds["embed"], dev, dtype="bf16")
B, N = 64, 384
ids, mask, tmask = synth.make_prompts(B, 16, 48, seed=0)
ids_t, mask_t, tm_t = torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(tmask)
stop = torch.full((B,), N, dtype=torch.int32)
warpers, procs = E.gen_logits(625, 0.7, 20, 1.05)
emb = gpt.embed_prompt(ids_t, tm_t)
ids_d = ids_t.to(dev)


def gen():
    out = None
    for out in gpt.generate(emb, ids_d, torch.tensor([0.3] * 4), 625, mask_t, N + 1, 0, (*procs, *warpers), return_hidden=True, manual_seed=42, stop_at=stop):
        pass
    torch.cuda.synchronize(dev)
    return out


for _ in range(2):
    gen()
t0 = time.perf_counter(); gen(); tg = time.perf_counter() - t0
print(f"generate: {tg * 1e3:.1f} ms for {N} steps of {B} rows = {tg * 1e3 / N:.4f} ms per step (decode loop {gpt.last_stats['decode_ms'] / N:.4f})")

pool = SlotPool(gpt, slots=B, cap=48 + N + 2 + 2 * SlotPool.POLL, hid_cap=N + 8, manual_seed=42)


def run_pool():
    for b in range(B):
        m = mask_t[b].bool()
        pool.submit(b, ids_t[b][m], tm_t[b][m], max_new_token=N + 1, stop_at=N)
    s0 = pool.steps
    n = sum(int(i.shape[0]) for _, i, _ in pool.run())
    torch.cuda.synchronize(dev)
    return n, pool.steps - s0


run_pool()
t0 = time.perf_counter(); n, st = run_pool(); tp = time.perf_counter() - t0
print(f"pool:     {tp * 1e3:.1f} ms for {st} steps ({n} tokens) = {tp * 1e3 / st:.4f} ms per step")
