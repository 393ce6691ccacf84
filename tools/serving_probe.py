"""Throughput of the SlotPool (continuous batching) on a stream of C3-like requests vs one-batch-at-a-time generate()."""
import os, sys, time, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chattts_amd import engine as E, synth, weights as W
from chattts_amd.serving import SlotPool
dev = torch.device("cuda:0")
gpt = E.GptEngine(W.synthetic_gpt(), W.synthetic_embed(), dev, dtype="bf16")
NREQ, S = 256, 64
ids, mask, tmask = synth.make_prompts(NREQ, 16, 48, seed=0)
stop = synth.make_stop_lengths(NREQ, 128, 512, seed=0)
def run_pool():
    pool = SlotPool(gpt, slots=S, cap=640, hid_cap=520, manual_seed=42)
    for i in range(NREQ):
        t = int(mask[i].sum())
        pool.submit(i, ids[i, -t:], max_new_token=515, stop_at=int(stop[i]))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 0
    for rid, out_ids, hid in pool.run():
        assert out_ids.shape[0] == stop[rid]
        n += out_ids.shape[0]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    steps = pool.steps; pool.close()
    return n, dt, steps
def run_batches():
    warpers, procs = E.gen_logits(625, 0.7, 20, 1.05)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    for lo in range(0, NREQ, S):
        sl = slice(lo, lo + S)
        ids_t, mask_t = torch.from_numpy(ids[sl]), torch.from_numpy(mask[sl])
        emb = gpt.embed_prompt(ids_t, torch.from_numpy(tmask[sl]))
        out = list(gpt.generate(emb, ids_t, torch.tensor([0.3] * 4), 625, mask_t, int(stop[sl].max()) + 1, 0, (*procs, *warpers),
                                return_hidden=True, manual_seed=42, stop_at=torch.from_numpy(stop[sl])))[-1]
        n += sum(int(t.shape[0]) for t in out.ids)
    torch.cuda.synchronize(); return n, time.perf_counter() - t0
run_pool(); run_batches()
n1, t1, steps = run_pool(); n2, t2 = run_batches()
print(json.dumps({"requests": NREQ, "slots": S, "tokens": int(n1), "slot_pool_s": round(t1, 3), "slot_pool_tok_per_s": round(n1 / t1),
                  "slot_pool_decode_steps": steps, "batched_generate_s": round(t2, 3), "batched_tok_per_s": round(n2 / t2),
                  "speedup": round((n1 / t1) / (n2 / t2), 3)}))
