"""In-kernel phase stamps of sample_k inside the real decode step (C3-like batch, graph replay): thread 0 of every workgroup writes
the 100 MHz realtime counter at 8 points (env CTTS_SAMPLE_DBG_PTR -> SampleArgs.dbg).  Prints the median over rows and steps of each
phase, and the spread of the workgroups' entry / exit."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
B = 64
dbg = torch.zeros((B, 8), dtype=torch.int64, device=dev)
os.environ["CTTS_SAMPLE_DBG_PTR"] = str(dbg.data_ptr())
from chattts_amd import _lib, engine as E, synth, weights as W  # noqa: E402
sds = W.synthetic_all()
gpt = E.GptEngine(sds["gpt"], sds["embed"], dev, dtype="bf16")
lib = _lib.lib()
warpers, procs = E.gen_logits(625, 0.7, 20, 1.05)
ids, mask, tmask = synth.make_prompts(B, 16, 48, seed=0)
ids_t, mask_t = torch.from_numpy(ids), torch.from_numpy(mask)
emb = gpt.embed_prompt(ids_t, torch.from_numpy(tmask))
stop = torch.full((B,), 40, dtype=torch.int32)
list(gpt.generate(emb, ids_t, torch.tensor([0.3] * 4), 625, mask_t, 600, 0, (*procs, *warpers), return_hidden=True, manual_seed=42, stop_at=stop))
ln = gpt._session["lanes"][0]
with torch.cuda.stream(ln.st):
    ln.finish.zero_(); ln.stop_d.fill_(100000); ln.len_d.fill_(48 + 40)
torch.cuda.synchronize()
rows = []
for step in range(60):
    dbg.zero_()
    torch.cuda.synchronize()
    _lib.check(lib.ctts_gpt_graph_launch(ln.handle, 1, ln.st.cuda_stream), "launch")
    torch.cuda.synchronize()
    rows.append(dbg.cpu().numpy().copy())
d = np.stack(rows[10:]).astype(np.float64) * 10.0   # ns
names = ["entry -> row known (desc load)", "-> all loads landed (logits, draws, history, table)", "-> temperature + penalty", "-> softmax statistics",
         "-> kept set (rank counting + decisions)", "-> final softmax + argmax(p/q)", "-> finish / write-back"]
ph = np.diff(d, axis=2)
for i, n in enumerate(names):
    print(f"{n:58s} median {np.median(ph[:, :, i]):7.0f} ns   p90 {np.percentile(ph[:, :, i], 90):7.0f} ns")
tot = d[:, :, 7] - d[:, :, 0]
print(f"workgroup entry -> exit: median {np.median(tot):.0f} ns; first entry -> last exit per launch: median {np.median(d[:, :, 7].max(1) - d[:, :, 0].min(1)):.0f} ns; "
      f"entry spread across the 64 workgroups: median {np.median(d[:, :, 0].max(1) - d[:, :, 0].min(1)):.0f} ns")
