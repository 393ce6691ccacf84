"""Runs BASELINE.json configs C1, C2, C5 on one GPU and prints one JSON line each (C3 is bench.py itself,
C4 is C3 per GPU under torchrun).  Synthetic weights/prompts as in bench.py."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chattts_amd import engine as E, synth, weights as W  # noqa: E402
from chattts_amd.core import Chat, InferCodeParams  # noqa: E402

dev = torch.device("cuda:0")
sds = W.synthetic_all()


def audio_s(lens):
    return sum(256 * (2 * int(t) - 1) for t in lens if t > 0) / 24000.0


def timed(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), out


# ---- C1: one 16-token sentence, near-greedy (tests/#511.py parameters), f32 parity mode ----
chat32 = Chat()
chat32.load(state_dicts=sds, device=dev, dtype="f32")
ids, mask, tmask = synth.make_prompts(1, 16, 16, seed=0)
p1 = InferCodeParams(top_P=0.005, top_K=1, max_new_token=48, manual_seed=42, show_tqdm=False)
a = (torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(tmask))
t, wav = timed(lambda: chat32.infer_ids(*a, p1))
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "generate.npz"))
out = list(chat32.infer_code(*a, p1))[-1]
exact = bool(np.array_equal(out.ids[0].cpu().numpy(), g["c1.ids"]))
print(json.dumps({"config": "C1 1x16-token prompt, top_K=1/top_P=0.005, f32 parity mode, 48 tokens + decode", "wall_ms": round(t * 1e3, 2),
                  "audio_s": round(audio_s([48]), 3), "audio_s_per_s": round(audio_s([48]) / t, 1), "token_ids_bit_exact_vs_reference": exact}))
del chat32

# ---- C2: batch=1, 512 speech tokens, bf16 ----
chat = Chat()
chat.load(state_dicts=sds, device=dev, dtype="bf16")
ids, mask, tmask = synth.make_prompts(1, 32, 32, seed=1)
a = (torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(tmask))
p2 = InferCodeParams(max_new_token=513, manual_seed=42, show_tqdm=False)
stop = torch.tensor([512], dtype=torch.int32)
t_gpt, o = timed(lambda: list(chat.infer_code(*a, p2, stop_at=stop))[-1])
t_all, wav = timed(lambda: chat.infer_ids(*a, p2, stop_at=stop))
print(json.dumps({"config": "C2 batch=1, 512 speech tokens, bf16 (GPT decode + DVAE + Vocos)", "wall_ms": round(t_all * 1e3, 2),
                  "gpt_ms": round(t_gpt * 1e3, 2), "ms_per_decode_step": round(t_gpt * 1e3 / 513, 4), "audio_s": round(audio_s([512]), 2),
                  "audio_s_per_s": round(audio_s([512]) / t_all, 1), "wav_shape": list(wav.shape)}))

# ---- C5: streaming, batch=16, reference yield schedule (stream_batch 24, first audio after 72 tokens) ----
ids, mask, tmask = synth.make_prompts(16, 16, 48, seed=2)
stop16 = torch.from_numpy(synth.make_stop_lengths(16, 128, 512, seed=2))
a = (torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(tmask))
p5 = InferCodeParams(max_new_token=int(stop16.max()) + 1, manual_seed=42, show_tqdm=False)
for mode in ("incremental", "prefix"):
    chat.incremental_stream = mode == "incremental"
    ttfs, totals, nchunks = [], [], 0
    for rep in range(22):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        first = None
        nchunks = 0
        for chunk in chat.infer_ids_stream(*a, p5, stop_at=stop16):
            if first is None:
                first = time.perf_counter() - t0
            nchunks += 1
        totals.append(time.perf_counter() - t0)
        ttfs.append(first)
    what = ("token windows with halos per yield (O(n))" if mode == "incremental" else "whole-prefix re-decode per yield (the reference's O(n^2) schedule)")
    print(json.dumps({"config": "C5 streaming batch=16, mixed lengths 128..512, reference yield schedule; codec: " + what,
                      "samples": len(ttfs) - 2, "ttfs_ms_p50": round(1e3 * float(np.median(ttfs[2:])), 2),
                      "ttfs_ms_p90": round(1e3 * float(np.percentile(ttfs[2:], 90)), 2),
                      "total_ms_p50": round(1e3 * float(np.median(totals[2:])), 1),
                      "chunks": nchunks, "audio_s": round(audio_s(stop16.tolist()), 1),
                      "audio_s_per_s": round(audio_s(stop16.tolist()) / float(np.median(totals[2:])), 1)}))
