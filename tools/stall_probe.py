"""Where do the intermittent ~65 ms stalls of the acoustic-decode path come from (profiles/r3g_c5_yield_probe.log)?  For a loop of
window decodes: host wall time vs GPU elapsed (events on the stream) per call, with and without the D2H, with and without idle gaps."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chattts_amd import engine as E, weights as W  # noqa: E402
dev = torch.device("cuda:0")
sds = W.synthetic_all()
codec = E.CodecEngine(sds["decoder"], sds["vocos"], dev)
hid = [torch.randn(300, 768, device=dev) for _ in range(16)]

def loop(name, n, fn, gap=0.0):
    host, gpu = [], []
    for i in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        if gap:
            time.sleep(gap)
        t0 = time.perf_counter()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        host.append((time.perf_counter() - t0) * 1e3)
        gpu.append(e0.elapsed_time(e1))
    host, gpu = np.array(host), np.array(gpu)
    print(f"{name}: host ms median {np.median(host):.1f} max {host.max():.1f} (> 30 ms: {(host > 30).sum()}/{n}); GPU ms median {np.median(gpu):.1f} "
          f"max {gpu.max():.1f} (> 30 ms: {(gpu > 30).sum()}/{n})")

loop("decode_window + to_host, back to back", 40, lambda: codec.to_host(codec.decode_window(hid, 20000, 32000)))
loop("decode_window only (no D2H)", 40, lambda: codec.decode_window(hid, 20000, 32000))
loop("decode_window only, 20 ms idle before each", 40, lambda: codec.decode_window(hid, 20000, 32000), gap=0.02)
loop("decode_window only, 100 ms idle before each", 20, lambda: codec.decode_window(hid, 20000, 32000), gap=0.1)
x = torch.randn(4096, 4096, device=dev)
loop("torch matmul 4096^3 x4 (no chattts kernels)", 40, lambda: [x @ x for _ in range(4)])
big = [torch.randn(512, 768, device=dev) for _ in range(64)]
loop("full decode_to_wavs of a 64 x 512 batch", 8, lambda: codec.decode_to_wavs(big))
