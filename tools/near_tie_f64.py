"""The one utterance of the N = 2 global bench batch whose token stream leaves the reference's (global 32, step 364 -- in BOTH parity
arithmetics, profiles/r6D_w2_divergence.log), looked at in float64: the model is evaluated in double precision on the reference's own
tokens up to that step (HF LlamaModel.double(), one teacher-forced forward), the sampling step is restated in float64 on those logits, and
the step's decision margin is computed.  If the margin is of the order of a float32 ulp of the logits, which side a float32 engine falls on
is decided by the summation order of its dot products -- the reference's binary included.  CPU only (a measurement tool: it may use oracle/)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from chattts_amd import rng, weights as W  # noqa: E402
from oracle import generate_np, sampling_np, torch_port  # noqa: E402

world, utt = int(sys.argv[1]) if len(sys.argv) > 1 else 2, int(sys.argv[2]) if len(sys.argv) > 2 else 32
gold = np.load(os.path.join(ROOT, "tests", "golden", "bench_c3_w%d.npz" % world))
off = np.concatenate([[0], np.cumsum(gold["lens"].astype(np.int64))])
ref = gold["ids"][off[utt]: off[utt + 1]].astype(np.int64)                    # [T_b, 4] the reference's tokens
wl = bench.shard_workload(64, world, 0, 128, 512)
ids, mask, tmask = wl["ids_all"][utt: utt + 1], wl["mask_all"][utt: utt + 1], wl["tmask_all"][utt: utt + 1]
sds = W.synthetic_all()
esd = {k: v.numpy() for k, v in sds["embed"].items()}
torch.set_num_threads(4)
llama = torch_port.build_llama(sds["gpt"]).double()
emb_code = [sds["embed"][f"emb_code.{k}.weight"].double() for k in range(4)]
heads = torch.from_numpy(generate_np.fold_heads(esd)).double()              # [4 * 626, 768] weight-normed, folded in float32 like the engines do
q_all = rng.ExpDraws(64 * world * 4, 626, 42, rows=torch.arange(utt * 4, utt * 4 + 4)).step(0).numpy()
ptab = rng.penalty_table(1.05).numpy()
steps = [int(s) for s in sys.argv[3:]] or [364]
for step in steps:
    emb_p = torch.from_numpy(generate_np.embed_prompt(esd, ids, tmask)).double()
    tok = torch.from_numpy(ref[:step])[None]                                 # tokens 0 .. step-1 are the inputs of steps 1 .. step
    x = torch.cat([emb_p, sum(emb_code[k][tok[..., k]] for k in range(4))], 1)
    am = torch.cat([torch.from_numpy(mask).bool(), torch.ones((1, step), dtype=torch.bool)], 1)
    pos = (am.long().cumsum(-1) - 1).masked_fill(am == 0, 1)
    with torch.inference_mode():
        h = llama(inputs_embeds=x, attention_mask=am, position_ids=pos, use_cache=False).last_hidden_state[0, -1]
    logits64 = (heads @ h).reshape(4, 626).numpy()                               # [4, 626] float64
    hist = ref[:step].T.copy()                                               # [4, step]
    kw = dict(temperature=np.full(4, 0.3, np.float32), top_p=0.7, top_k=20, pow_table=ptab, max_input_ids=625, row_offset=utt * 4,
              mask_eos=np.full(4, step < int(wl["stop_all"][utt])))
    # the sampling step on the float64 logits rounded to float32 (what an ideal float32 engine would hand the sampler), and its margin
    idx = sampling_np.sample_step(logits64.astype(np.float32), hist, q_all, **kw)
    marg = sampling_np.decision_margin(logits64.astype(np.float32), hist, q_all, **kw)
    print(f"utterance {utt} step {step}: reference token row {ref[step].tolist() if step < len(ref) else None} | float64-logit sampling {idx.tolist()} | "
          f"decision margin per code book (tempered-logit units) {[float('%.3e' % m) for m in marg]} | logit scale {np.abs(logits64).max():.2f}, "
          f"float32 ulp at that scale / 0.3 = {np.spacing(np.float32(np.abs(logits64).max())) / 0.3:.2e}")
