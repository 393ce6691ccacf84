"""Probe (round 4): would folding o_proj into gate/up ALGEBRAICALLY pay?  gate_up(x + attn Wo) = x Wgu' + attn (Wo Wgu'): one launch with
K = 1536 (activations [x | attn], weights [Wgu' | Wo Wgu']) instead of the o_proj launch + the gate/up launch -- twice the gate/up weight
bytes against one kernel boundary and one kernel body.  Times, in a replayed graph over 20 layers' cold weights (packed decode kernels,
64 rows): the pair (o_proj RES, K = 768) + (gate/up SILU, K = 768) against one (gate/up SILU, K = 1536) launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chattts_amd import _lib  # noqa: E402
from chattts_amd.engine import pack_frag  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda:0")
M, L = 64, 20
torch.manual_seed(0)


def packed_act(k):
    return pack_frag(torch.randn(M, k).bfloat16()).to(dev)


xp, aop, xcat = packed_act(768), packed_act(768), packed_act(1536)
ssq = (torch.rand(M, 48) * 16).to(dev)
x32 = torch.randn(M, 768, device=dev)
xp_out = torch.empty_like(xp)
ssq_out = torch.empty_like(ssq)
act = torch.empty(M * 3072, dtype=torch.bfloat16, device=dev)
na = torch.tensor([M], dtype=torch.int32, device=dev)
Wo = [pack_frag((torch.randn(768, 768) * 0.02).bfloat16()).to(dev) for _ in range(L)]
Wgu = [pack_frag((torch.randn(6144, 768) * 0.02).bfloat16()).to(dev) for _ in range(L)]
Wgu2 = [pack_frag((torch.randn(6144, 1536) * 0.02).bfloat16()).to(dev) for _ in range(L)]
flush = torch.empty(1536 * 1024 * 1024 // 4, device=dev)


def k_o(l, st):
    _lib.check(lib.ctts_k_gemm_dec(aop.data_ptr(), Wo[l].data_ptr(), M, 768, 768, na.data_ptr(), None, 0.0, 1, x32.data_ptr(), 768, xp_out.data_ptr(), 24,
                                   ssq_out.data_ptr(), 0, st), "o")


def k_gu(l, st):
    _lib.check(lib.ctts_k_gemm_dec(xp.data_ptr(), Wgu[l].data_ptr(), M, 3072, 768, na.data_ptr(), ssq.data_ptr(), 1e-6, 2, None, 0, act.data_ptr(), 96, None, 0, st), "gu")


def k_gu2(l, st, mb=0):
    _lib.check(lib.ctts_k_gemm_dec(xcat.data_ptr(), Wgu2[l].data_ptr(), M, 3072, 1536, na.data_ptr(), ssq.data_ptr(), 1e-6, 2, None, 0, act.data_ptr(), 96, None, mb, st), "gu2")


def bench(name, fns, per):
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        for f in fns:
            f(0, s.cuda_stream)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for i in range(100):
                for f in fns:
                    f(i % L, torch.cuda.current_stream().cuda_stream)
    ts = []
    for rep in range(6):
        flush.fill_(1.0)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 10.0)
    print(f"{name:44s}: {min(ts[1:]):6.2f} us per {per} (min of 5)")


bench("o_proj (K 768)", [k_o], "launch")
bench("gate/up (K 768)", [k_gu], "launch")
bench("o_proj + gate/up (two launches)", [k_o, k_gu], "pair")
bench("gate/up' (K 1536, [x | attn] x [Wgu' | Wo Wgu'])", [k_gu2], "launch")
for mb in (1, 2):
    bench(f"gate/up' (K 1536), {16 * mb}-row workgroups", [lambda l, st, mb=mb: k_gu2(l, st, mb)], "launch")
