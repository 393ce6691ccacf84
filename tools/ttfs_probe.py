"""Where the time to first sample goes (C3 batch, stream=True): prefill, graph build, 72 decode steps, prefix decode, D2H."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chattts_amd import engine as E, synth, weights as W, _lib  # noqa: E402
from chattts_amd.core import Chat, InferCodeParams  # noqa: E402

dev = torch.device("cuda:0")
sds = W.synthetic_all()
chat = Chat()
chat.load(state_dicts=sds, device=dev, dtype=os.environ.get("DTYPE", "bf16"))
ids, mask, tmask = synth.make_prompts(64, 16, 48, seed=0)
stop = torch.from_numpy(synth.make_stop_lengths(64, 128, 512, seed=0))
a = (torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(tmask))
p = InferCodeParams(max_new_token=513, manual_seed=42, show_tqdm=False)
lib = _lib.lib()
marks = {}
T0 = [0.0]
for name in ("ctts_gpt_prefill", "ctts_gpt_graph_build", "ctts_gpt_graph_launch", "ctts_gpt_create"):
    orig = getattr(lib, name)

    def wrap(*args, _o=orig, _n=name):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = _o(*args)
        torch.cuda.synchronize()
        marks[_n] = marks.get(_n, 0.0) + time.perf_counter() - t
        marks.setdefault(_n + ".first_at", t - T0[0])
        marks[_n + ".calls"] = marks.get(_n + ".calls", 0) + 1
        return r
    setattr(lib, name, wrap)
for it in range(4):
    marks.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    T0[0] = t0
    ys = []
    for out in chat.infer_code(*a, p, stream=True, stop_at=stop):
        torch.cuda.synchronize()
        ys.append(time.perf_counter() - t0)
        if len(ys) == 3:
            t1 = time.perf_counter()
            wav = chat.codec.decode_to_wavs(out.hiddens)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            w = wav.cpu().numpy()
            t3 = time.perf_counter()
            break
    print(f"iter {it}: yields at {[round(1e3 * y, 1) for y in ys]} ms; prefill {1e3 * marks.get('ctts_gpt_prefill', 0):.1f} ms, graph build "
          f"{1e3 * marks.get('ctts_gpt_graph_build', 0):.1f} ms, prefix decode (72 tok) {1e3 * (t2 - t1):.1f} ms, D2H+numpy {1e3 * (t3 - t2):.1f} ms, "
          f"total {1e3 * (t3 - t0):.1f} ms", flush=True)
    print("   ", {k: round(1e3 * v, 2) if isinstance(v, float) else v for k, v in marks.items()}, flush=True)
