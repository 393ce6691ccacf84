"""In-kernel phase stamps of sample_text_k inside the refine-text decode step (eager launches): thread 0 of every workgroup writes the
100 MHz realtime counter at 7 points (env CTTS_SAMPLE_DBG_PTR -> SampleArgs.dbg); dbg[7] = 1 when the counting fast path was taken."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dbg = torch.zeros((B, 8), dtype=torch.int64, device=dev)
os.environ["CTTS_SAMPLE_DBG_PTR"] = str(dbg.data_ptr())
from chattts_amd import _lib, engine as E, synth, weights as W  # noqa: E402
sds = W.synthetic_all()
gpt = E.GptEngine(sds["gpt"], sds["embed"], dev, dtype=os.environ.get("DTYPE", "f32x3"))
lib = _lib.lib()
warpers, procs = E.gen_logits(21178, 0.7, 20, 1.0)
ids, mask, tmask = synth.make_prompts(B, 16, 48, seed=3)
ids_t, mask_t = torch.from_numpy(ids), torch.from_numpy(mask)
emb = gpt.embed_prompt(ids_t, torch.from_numpy(tmask))
stop = torch.full((B,), 30, dtype=torch.int32)
list(gpt.generate(emb, ids_t, torch.tensor([0.7]), 21000, mask_t, 64, 0, (*procs, *warpers), infer_text=True, manual_seed=42, stop_at=stop, use_graph=False))
ln = gpt._session_text["lanes"][0]
with torch.cuda.stream(ln.st):
    ln.finish.zero_(); ln.stop_d.fill_(100000); ln.len_d.fill_(48 + 20)
torch.cuda.synchronize()
rows = []
import ctypes as C
for step in range(30):
    dbg.zero_()
    torch.cuda.synchronize()
    _lib.check(lib.ctts_gpt_decode_step(ln.handle, C.byref(ln.s), ln.st.cuda_stream), "step")
    torch.cuda.synchronize()
    rows.append(dbg.cpu().numpy().copy())
d = np.stack(rows[5:]).astype(np.float64)
print("fast path taken: %.2f of launches" % d[:, :, 7].mean())
d = d * 10.0
names = ["entry -> row known (n_active / row_map / len chain)", "-> logits row landed", "-> temper + max + sum exp + double mass", "-> kept set", "-> token drawn", "-> write-back"]
ph = np.diff(d[:, :, :7], axis=2)
for i, n in enumerate(names):
    print(f"{n:58s} median {np.median(ph[:, :, i]):7.0f} ns   p90 {np.percentile(ph[:, :, i], 90):7.0f} ns")
tot = d[:, :, 6] - d[:, :, 0]
print(f"workgroup entry -> exit: median {np.median(tot):.0f} ns; first entry -> last exit per launch: median {np.median(d[:, :, 6].max(1) - d[:, :, 0].min(1)):.0f} ns")
