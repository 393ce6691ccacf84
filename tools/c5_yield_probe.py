"""C5 (streaming, batch 16): where one yield's time goes on the host -- generator resume (the poll that returns the chunk + the
cumulative result copies), the window's acoustic decode enqueue, the wait for it + D2H."""
import os, sys, time
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
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rows, length, n = [], 0, 0
    it = chat.infer_code(*a, p5, stream=True, stop_at=stop16)
    while True:
        ta = time.perf_counter()
        try:
            result = next(it)
        except StopIteration:
            break
        tb = time.perf_counter()
        n += 1
        if n <= p5.pass_first_n_batches:
            continue
        wav_d = chat.codec.decode_window(result.hiddens, length, length + p5.stream_speed)
        tc = time.perf_counter()
        piece = chat.codec.to_host(wav_d)
        td = time.perf_counter()
        length += piece.shape[1]
        rows.append((round((tb - ta) * 1e3, 1), round((tc - tb) * 1e3, 1), round((td - tc) * 1e3, 1)))
    print(f"rep {rep}: total {1e3 * (time.perf_counter() - t0):.1f} ms; per yield (generator resume, decode enqueue, wait + D2H) ms:", rows)
