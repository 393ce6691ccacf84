"""C5 (streaming, batch 16): how the per-yield acoustic decode should be scheduled against the generator's run-ahead chunk.
Variants: (a) caller stream, concurrent with the generator's stream (what round 2 shipped); (b) generator stream raised to high
priority; (c) the consumer's work on the generator's own stream (no concurrency); plus the cost of one window decode alone."""
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

def run(reps=6):
    rows = []
    for rep in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); first = None
        for chunk in chat.infer_ids_stream(*a, p5, stop_at=stop16):
            if first is None:
                first = time.perf_counter() - t0
        rows.append((round(first * 1e3, 1), round((time.perf_counter() - t0) * 1e3, 1)))
    return rows[1:]

print("(a) consumer on the caller's (default) stream:", run())
with torch.cuda.stream(chat.gpt.stream):
    print("(c) consumer on the generator's stream:", run())
side = torch.cuda.Stream(device=dev)
with torch.cuda.stream(side):
    print("(a') consumer on another non-default stream:", run())
hp = torch.cuda.Stream(device=dev, priority=-1)
chat.gpt.stream = hp
chat.gpt._lane_res = [(chat.gpt.handle, hp)]
chat.gpt._session = None
print("(b) generator on a high-priority stream, consumer on the default stream:", run())
# one window decode alone
out = None
for out in chat.infer_code(*a, p5, stream=False, stop_at=stop16):
    pass
torch.cuda.synchronize()
for lo, hi in ((0, 12000), (60000, 72000)):
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        w = chat.codec.to_host(chat.codec.decode_window(out.hiddens, lo, hi))
        ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    print(f"decode_window samples [{lo}, {hi}) alone, ms:", ts)
