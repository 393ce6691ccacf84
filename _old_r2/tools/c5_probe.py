"""BASELINE config C5 (streaming, batch 16) under a microscope: per-repetition time to first sample / total, the generator alone
(chunks consumed without decoding them), and the session's reuse -- to find what a regression of the streaming path comes from."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chattts_amd import synth, weights as W  # noqa: E402
from chattts_amd.core import Chat, InferCodeParams  # noqa: E402
dev = torch.device("cuda:0")
chat = Chat()
chat.load(state_dicts=W.synthetic_all(), device=dev, dtype="bf16")
ids, mask, tmask = synth.make_prompts(16, 16, 48, seed=2)
stop16 = torch.from_numpy(synth.make_stop_lengths(16, 128, 512, seed=2))
a = (torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(tmask))
p5 = InferCodeParams(max_new_token=int(stop16.max()) + 1, manual_seed=42, show_tqdm=False)
rows = []
for rep in range(8):
    sess_before = id(chat.gpt._session) if chat.gpt._session is not None else None
    torch.cuda.synchronize(); t0 = time.perf_counter(); first = None; n = 0
    for chunk in chat.infer_ids_stream(*a, p5, stop_at=stop16):
        if first is None:
            first = time.perf_counter() - t0
        n += 1
    tot = time.perf_counter() - t0
    rows.append((round(first * 1e3, 1), round(tot * 1e3, 1), n, round(chat.gpt.last_stats.get("decode_ms", 0), 1), chat.gpt.last_stats.get("steps"),
                 sess_before == id(chat.gpt._session)))
print("stream+decode  (ttfs ms, total ms, chunks, generator decode-loop ms, steps, session reused):", rows)
rows = []
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); first = None; n = 0
    for out in chat.infer_code(*a, p5, stream=True, stop_at=stop16):
        if first is None:
            first = time.perf_counter() - t0
        n += 1
    torch.cuda.synchronize()
    rows.append((round(first * 1e3, 1), round((time.perf_counter() - t0) * 1e3, 1), n))
print("generator only, stream=True (first yield ms, total ms, yields):", rows)
rows = []
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for out in chat.infer_code(*a, p5, stream=False, stop_at=stop16):
        pass
    torch.cuda.synchronize()
    rows.append(round((time.perf_counter() - t0) * 1e3, 1))
print("generator only, stream=False (total ms):", rows, "env", {k: v for k, v in os.environ.items() if k.startswith("CTTS_")})
