"""Host-side timeline of ONE C3 pass (bench.py's workload): where the milliseconds outside the decode loop go.
Stages are separated by stream synchronisation, so the sum is a little above the un-instrumented pass."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chattts_amd import engine as E, weights as W  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
sds = W.synthetic_all()
gpt = E.GptEngine(sds["gpt"], sds["embed"], dev, dtype="bf16")
codec = E.CodecEngine(sds["decoder"], sds["vocos"], dev, gemm="f16")
wl = bench.shard_workload(64, 1, 0, 128, 512)
ids_t, mask_t, tm_t = torch.from_numpy(wl["ids"]), torch.from_numpy(wl["mask"]), torch.from_numpy(wl["tmask"])
stop_t = torch.from_numpy(wl["stop"])
warpers, procs = E.gen_logits(625, 0.7, 20, 1.05)
temp = torch.tensor([0.3] * 4)
max_new = int(wl["stop_all"].max()) + 1
emb = gpt.embed_prompt(ids_t, tm_t)
ids_d = ids_t.to(dev)
sync = lambda: torch.cuda.synchronize(dev)


def one(instr):
    t = [time.perf_counter()]
    out = None
    for out in gpt.generate(emb, ids_d, temp, 625, mask_t, max_new, 0, (*procs, *warpers), return_hidden=True, manual_seed=42, stop_at=stop_t,
                            row_offset=wl["row_offset"], total_rows=wl["total_rows"]):
        pass
    if instr: sync()
    t.append(time.perf_counter())
    wav_d = codec.decode_to_wavs(out.hiddens)
    if instr: sync()
    t.append(time.perf_counter())
    wav = codec.to_host(wav_d)
    t.append(time.perf_counter())
    return np.diff(t) * 1e3, gpt.last_stats.get("decode_ms", 0.0), gpt.last_stats.get("steps", 0)


for _ in range(2):
    one(False)
rows = []
for _ in range(5):
    d, dec, steps = one(True)
    rows.append((d[0], dec, d[0] - dec, d[1], d[2], d.sum()))
r = np.median(np.array(rows), 0)
print(f"instrumented pass (median of 5): generate {r[0]:.2f} ms (decode loop {r[1]:.2f} ms for {steps} steps; prefill + set-up + outputs {r[2]:.2f} ms) | "
      f"DVAE + Vocos {r[3]:.2f} ms | waveform D2H + host copy {r[4]:.2f} ms | sum {r[5]:.2f} ms")
rows = []
for _ in range(5):
    sync(); t0 = time.perf_counter(); one(False); rows.append(1e3 * (time.perf_counter() - t0))
print(f"un-instrumented pass: median {np.median(rows):.2f} ms")
