"""Probe: decode attention time per launch vs number of live rows (workgroup-count quantisation over 256 CUs) and context."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chattts_amd import _lib  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda:0")
L, MMAX, CMAX = 20, 64, 640
torch.manual_seed(0)
kc = (torch.randn(L, MMAX, 12, CMAX, 64, device=dev) * 0.5).bfloat16()
vc = (torch.randn(L, MMAX, 12, CMAX, 64, device=dev) * 0.5).bfloat16()
qkv = torch.randn(MMAX, 2304, device=dev)
out = torch.empty(MMAX, 768, device=dev)
kv_start = torch.zeros(MMAX, dtype=torch.int32, device=dev)


def bench(n, c):
    ln = torch.full((MMAX,), c, dtype=torch.int32, device=dev)
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    per = MMAX * 12 * CMAX * 64 * 2

    def fn(l, st):
        lib.ctts_k_attention(qkv.data_ptr(), kc.data_ptr() + l * per, vc.data_ptr() + l * per, 1, CMAX, out.data_ptr(), 1, ln.data_ptr(),
                             kv_start.data_ptr(), n, st)
    with torch.cuda.stream(s):
        fn(0, s.cuda_stream)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for i in range(100):
                fn(i % L, torch.cuda.current_stream().cuda_stream)
    ts = []
    for rep in range(5):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 10.0)
    t = min(ts[1:])
    mb = n * 12 * c * 256 / 1e6
    print(f"rows {n:3d} ctx {c:4d}: {t:6.2f} us/launch  {mb:6.1f} MB  {mb / t * 1e3 / 1e3:5.2f} TB/s  wgs {12 * n} ({12 * n / 256:.2f}/CU)", flush=True)


for c in (300, 500):
    for n in (8, 16, 21, 22, 32, 42, 43, 48, 53, 64):
        bench(n, c)
