"""Intermittent ~60-90 ms stalls of device-to-host copies on this pool (round 3): which way of getting 768 KB (one streamed chunk of a
16-utterance batch) and 67 MB (the waveforms of a 64-utterance batch) to the host avoids them?  40 repetitions each, GPU busy with a
window decode before every copy like the streaming loop."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chattts_amd import _lib, engine as E, weights as W  # noqa: E402
dev = torch.device("cuda:0")
lib = _lib.lib()
sds = W.synthetic_all()
codec = E.CodecEngine(sds["decoder"], sds["vocos"], dev)
hid = [torch.randn(300, 768, device=dev) for _ in range(16)]

def stats(name, ts):
    ts = np.array(ts) * 1e3
    print(f"{name}: median {np.median(ts):.2f} ms, max {ts.max():.1f} ms, > 30 ms: {(ts > 30).sum()}/{len(ts)}")

for n_el, label, reps in ((16 * 12000, "768 KB", 40), (64 * 261888, "67 MB", 12)):
    src = torch.randn(n_el, device=dev)
    pin = torch.empty(n_el, dtype=torch.float32).pin_memory()
    def work():
        codec.decode_window(hid, 20000, 32000)
    variants = {
        "pinned copy_(non_blocking) + event poll": lambda: (pin.copy_(src, non_blocking=True), torch.cuda.current_stream().synchronize()),
        "pageable .cpu()": lambda: src.cpu(),
        "shader copy into pinned memory (ctts_copy_bytes)": lambda: (_lib.check(lib.ctts_copy_bytes(pin.data_ptr(), src.data_ptr(), n_el * 4, torch.cuda.current_stream().cuda_stream), "copy"), torch.cuda.current_stream().synchronize()),
    }
    for name, fn in variants.items():
        ts = []
        for _ in range(reps):
            work()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        stats(f"[{label}] {name}", ts)
    src2 = src.clone()
    _lib.check(lib.ctts_copy_bytes(pin.data_ptr(), src2.data_ptr(), n_el * 4, torch.cuda.current_stream().cuda_stream), "copy")
    torch.cuda.synchronize()
    assert torch.equal(pin, src2.cpu()), "shader copy mismatch"
# the same with NO preceding GPU work (idle device)
src = torch.randn(16 * 12000, device=dev); pin = torch.empty(16 * 12000).pin_memory()
ts = []
for _ in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter(); pin.copy_(src, non_blocking=True); torch.cuda.current_stream().synchronize(); ts.append(time.perf_counter() - t0)
stats("[768 KB] pinned copy, idle device", ts)
