"""The N = 2 global bench batch (128 utterances) against the reference's own run of it (tests/golden/bench_c3_w2.npz), shard by shard:
which utterances' token rows differ in each parity arithmetic, were they flagged by the certificate, does the exact fallback restore them."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from chattts_amd import engine as E, weights as W  # noqa: E402
dev = torch.device("cuda:0")
sds = W.synthetic_all()
world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
gold = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "bench_c3_w%d.npz" % world))
off = np.concatenate([[0], np.cumsum(gold["lens"].astype(np.int64))])
warpers, procs = E.gen_logits(625, 0.7, 20, 1.05)
for name, kw in (("f32x3", dict(dtype="f32x3", exact_fallback=False)), ("f32", dict(dtype="f32")), ("f32x3+fallback", dict(dtype="f32x3", exact_fallback=True))):
    eng = E.GptEngine(sds["gpt"], sds["embed"], dev, **kw)
    for rank in range(world):
        wl = bench.shard_workload(64, world, rank, 128, 512)
        ids_t, mask_t = torch.from_numpy(wl["ids"]), torch.from_numpy(wl["mask"])
        emb = eng.embed_prompt(ids_t, torch.from_numpy(wl["tmask"]))
        out = list(eng.generate(emb, ids_t, torch.tensor([0.3] * 4), 625, mask_t, int(wl["stop_all"].max()) + 1, 0, (*procs, *warpers), manual_seed=42,
                                stop_at=torch.from_numpy(wl["stop"]), row_offset=wl["row_offset"], row_ids=wl["row_ids"], total_rows=wl["total_rows"]))[-1]
        rows = [t.cpu().numpy() for t in out.ids]
        bad = []
        for j, (r, b) in enumerate(zip(rows, wl["sel"])):
            g = gold["ids"][off[b]: off[b + 1]].astype(np.int64)
            if not np.array_equal(r, g):
                first = int(np.argmax((r != g).any(1)))
                bad.append((int(b), j, first, len(g)))
        ls = eng.last_stats
        flagged = [int(wl["sel"][j]) for j in ls.get("uncertified_rows", [])]
        marg = getattr(eng, "last_margins", None)
        print(f"{name:15s} rank {rank}: differing utterances (global, local, first differing step, length) {bad} | flagged by the certificate {flagged} | "
              f"re-run exactly {[int(wl['sel'][j]) for j in ls.get('exact_rerun_rows', [])]} | bound {ls.get('margin_bound')}"
              + ("" if marg is None or not bad or len(marg) != len(rows) else f" | margins of the differing ones {[float(marg[j]) for _, j, _, _ in bad]}"), flush=True)
    del eng
    torch.cuda.empty_cache()
