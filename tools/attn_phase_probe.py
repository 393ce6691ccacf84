"""In-kernel phase stamps of the decode attention kernel inside the real step (graph replay of ONE step at a C3-like state: 45 live
utterances of 64, contexts 200..250 keys).  Thread 0 of every workgroup writes the 100 MHz realtime counter at: entry, row descriptor
known, first KV block requested, first block consumed, wave 0's keys done, exit.  The last layer's launch is what is read."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
B = 64
dbg = torch.zeros((12 * B, 8), dtype=torch.int64, device=dev)
os.environ["CTTS_ATT_DBG_PTR"] = str(dbg.data_ptr())
from chattts_amd import _lib, engine as E, synth, weights as W  # noqa: E402
sds = W.synthetic_all()
gpt = E.GptEngine(sds["gpt"], sds["embed"], dev, dtype="bf16")
lib = _lib.lib()
warpers, procs = E.gen_logits(625, 0.7, 20, 1.05)
ids, mask, tmask = synth.make_prompts(B, 16, 48, seed=0)
ids_t, mask_t = torch.from_numpy(ids), torch.from_numpy(mask)
emb = gpt.embed_prompt(ids_t, torch.from_numpy(tmask))
stop = torch.full((B,), 30, dtype=torch.int32)
list(gpt.generate(emb, ids_t, torch.tensor([0.3] * 4), 625, mask_t, 600, 0, (*procs, *warpers), return_hidden=True, manual_seed=42, stop_at=stop))
ln = gpt._session["lanes"][0]
for live, gen in ((45, 190), (64, 190), (45, 450), (20, 400)):
    with torch.cuda.stream(ln.st):
        ln.finish.zero_(); ln.finish[live:] = 1; ln.stop_d.fill_(100000); ln.len_d.fill_(48 + gen)
    torch.cuda.synchronize()
    rows = []
    for step in range(40):
        dbg.zero_(); torch.cuda.synchronize()
        _lib.check(lib.ctts_gpt_graph_launch(ln.handle, 1, ln.st.cuda_stream), "launch")
        torch.cuda.synchronize()
        with torch.cuda.stream(ln.st):
            ln.len_d.fill_(48 + gen)       # stay at the same context
        rows.append(dbg.cpu().numpy().copy())
    d = np.stack(rows[8:]).astype(np.float64)
    ok = d[:, :, 5] > 0                      # workgroups that ran to the end (live rows)
    t = d[:, :, :6] * 10.0                   # ns
    first = np.where(ok, t[:, :, 0], np.inf).min(1)
    last = np.where(ok, t[:, :, 5], 0).max(1)
    ph = np.diff(t, axis=2)
    names = ["entry -> descriptor", "-> q read, first block requested", "-> first block consumed", "-> wave 0's keys done", "-> merged, stored, exit"]
    print(f"--- {live} live rows, contexts {int(d[0, :, 6][ok[0]].min())}..{int(d[0, :, 6][ok[0]].max())} keys, {int(ok[0].sum())} workgroups with work")
    for i, n in enumerate(names):
        v = ph[:, :, i][ok]
        print(f"   {n:36s} median {np.median(v):6.0f} ns   p90 {np.percentile(v, 90):6.0f} ns   max {v.max():6.0f} ns")
    ent = np.where(ok, t[:, :, 0], np.nan)
    print(f"   workgroup entry -> exit median {np.median((t[:, :, 5] - t[:, :, 0])[ok]):.0f} ns; first entry -> last exit per launch: median {np.median(last - first):.0f} ns; "
          f"entry spread (first -> last workgroup start): median {np.median(np.nanmax(ent, 1) - np.nanmin(ent, 1)):.0f} ns")
