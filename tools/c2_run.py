"""BASELINE config C2 alone (batch 1, 512 speech tokens, bf16): for rocprofv3 --stats and quick A/B of knobs."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chattts_amd import synth, weights as W  # noqa: E402
from chattts_amd.core import Chat, InferCodeParams  # noqa: E402
dev = torch.device("cuda:0")
chat = Chat()
chat.load(state_dicts=W.synthetic_all(), device=dev, dtype="bf16")
ids, mask, tmask = synth.make_prompts(1, 32, 32, seed=1)
a = (torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(tmask))
p2 = InferCodeParams(max_new_token=513, manual_seed=42, show_tqdm=False)
stop = torch.tensor([512], dtype=torch.int32)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ts = []
for r in range(reps + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o = list(chat.infer_code(*a, p2, stop_at=stop))[-1]
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
t = float(np.median(ts[1:]))
print(json.dumps({"config": "C2 GPT part: batch 1, 512 tokens, bf16", "gpt_ms": round(t * 1e3, 2), "ms_per_decode_step": round(t * 1e3 / 513, 4),
                  "weights_only_hbm_frac": round(381.4e6 / (t / 513) / 8e12, 4), "env": {k: v for k, v in os.environ.items() if k.startswith("CTTS_")}}))
