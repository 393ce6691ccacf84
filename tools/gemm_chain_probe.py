"""Probe: per-launch time of the perf-mode projection kernels inside a captured graph, with weights
L2/MALL-hot (same layer every launch) vs cold (cycling 20 layers + a 1.5 GB flush between replays)."""
import ctypes as C
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chattts_amd import _lib  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda:0")
M = 64
L = 20
torch.manual_seed(0)
xb = torch.randn(M, 768, device=dev).bfloat16()
ssq = torch.rand(M, 48, device=dev) * 16
x32 = torch.randn(M, 768, device=dev)
act = torch.randn(M, 3072, device=dev).bfloat16()
qkv = torch.empty(M, 2304, device=dev)
Wqkv = [torch.randn(2304, 768, device=dev).bfloat16() * 0.02 for _ in range(L)]
Wo = [torch.randn(768, 768, device=dev).bfloat16() * 0.02 for _ in range(L)]
Wgu = [torch.randn(6144, 768, device=dev).bfloat16() * 0.02 for _ in range(L)]
Wd = [torch.randn(768, 3072, device=dev).bfloat16() * 0.02 for _ in range(L)]
flush = torch.empty(1536 * 1024 * 1024 // 4, device=dev)
xb2 = torch.empty_like(xb)
ssq2 = torch.empty_like(ssq)
actb = torch.empty(M, 3072, device=dev, dtype=torch.bfloat16)


def k_qkv(l, st):
    lib.ctts_k_gemm_fast(xb.data_ptr(), 768, Wqkv[l].data_ptr(), M, 2304, 768, ssq.data_ptr(), 1e-6, 0, qkv.data_ptr(), 2304, None, 0, None, st)


def k_o(l, st):
    lib.ctts_k_gemm_fast(xb.data_ptr(), 768, Wo[l].data_ptr(), M, 768, 768, None, 0.0, 1, x32.data_ptr(), 768, xb2.data_ptr(), 768, ssq2.data_ptr(), st)


def k_gu(l, st):
    lib.ctts_k_gemm_fast(xb.data_ptr(), 768, Wgu[l].data_ptr(), M, 3072, 768, ssq.data_ptr(), 1e-6, 2, None, 0, actb.data_ptr(), 3072, None, st)


def k_d(l, st):
    lib.ctts_k_gemm_fast(act.data_ptr(), 3072, Wd[l].data_ptr(), M, 768, 3072, None, 0.0, 1, x32.data_ptr(), 768, xb2.data_ptr(), 768, ssq2.data_ptr(), st)


def bench(name, fn, cold):
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        fn(0, s.cuda_stream)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for i in range(100):
                fn((i % L) if cold else 0, torch.cuda.current_stream().cuda_stream)
    ts = []
    for rep in range(6):
        if cold:
            flush.fill_(1.0)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 10.0)  # us per kernel (100 kernels)
    print(f"{name:10s} {'cold' if cold else 'hot ':4s}: {min(ts[1:]):6.2f} us/launch (min of 5), first {ts[0]:.2f}")


for name, fn in (("qkv", k_qkv), ("o_proj", k_o), ("gate_up", k_gu), ("down", k_d)):
    bench(name, fn, False)
    bench(name, fn, True)
