"""Launch-chain floor of the captured decode step: the same graph replayed (a) with every utterance finished -- every kernel
exits at entry, what remains is launch + dispatch -- and (b) live, for B in {1, 64}.  Prints microseconds per step."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chattts_amd import _lib, engine as E, synth, weights as W  # noqa: E402
dev = torch.device("cuda:0")
sds = W.synthetic_all()
gpt = E.GptEngine(sds["gpt"], sds["embed"], dev, dtype="bf16")
lib = _lib.lib()
warpers, procs = E.gen_logits(625, 0.7, 20, 1.05)
for B in (64, 1):
    ids, mask, tmask = synth.make_prompts(B, 32, 32, seed=1)
    ids_t, mask_t = torch.from_numpy(ids), torch.from_numpy(mask)
    emb = gpt.embed_prompt(ids_t, torch.from_numpy(tmask))
    stop = torch.full((B,), 8, dtype=torch.int32)
    list(gpt.generate(emb, ids_t, torch.tensor([0.3] * 4), 625, mask_t, 900, 0, (*procs, *warpers), return_hidden=True, manual_seed=42, stop_at=stop))
    ln = gpt._session["lanes"][0]
    st = ln.st
    def replay(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(lib.ctts_gpt_graph_launch(ln.handle, n, st.cuda_stream), "launch")
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6
    replay(50)
    floor = min(replay(300) for _ in range(3))
    with torch.cuda.stream(st):
        ln.finish.zero_(); ln.stop_d.fill_(100000)
    torch.cuda.synchronize()
    live = []
    for _ in range(2):
        with torch.cuda.stream(st):
            ln.len_d.fill_(32 + 8); ln.finish.zero_()
        live.append(replay(400))
    print(f"B={B:3d}: all finished (kernels exit at entry) {floor:7.1f} us/step = {floor / 104:5.2f} us/launch | live, contexts 40..440: {min(live):7.1f} us/step")
