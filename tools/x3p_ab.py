"""The acoustic decoder's point-wise GEMMs alone at the C3 pass's shapes (hip events, 20 launches each): the batch is decoded padded to its
longest utterance (core.py:525-533), so Vocos sees 64 x 2 x 512 = 65,536 frames (K 512 <-> 1536) and the DVAE decoder 32,768 (K 256 <-> 1024);
plus the bench's roofline shape, and a streaming window (16 utterances x 2 x 36 frames).  A/B knobs: CTTS_CODEC_TILE=256|128 (read at every
launch), CTTS_X3P_VAR (one process per value).  `--h1p`: the fp16-plane kernel instead."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chattts_amd import _lib  # noqa: E402
from chattts_amd.engine import pack_h1p, pack_x3p  # noqa: E402
dev = torch.device("cuda:0")
lib = _lib.lib()
h1p = "--h1p" in sys.argv
pack, fn = (pack_h1p, lib.ctts_k_gemm_h1p) if h1p else (pack_x3p, lib.ctts_k_gemm_x3p)


def timed(f, n=20):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = 0.0
for (M, N, K, epi, per_pass) in ((65536, 1536, 512, 0, 8), (65536, 512, 1536, 1, 8), (32768, 1024, 256, 0, 12), (32768, 256, 1024, 1, 12),
                                (65536, 2048, 512, 0, 0), (65536, 512, 2048, 1, 0), (1152, 1536, 512, 0, 0), (1152, 512, 1536, 1, 0),
                                (576, 1024, 256, 0, 0), (576, 256, 1024, 1, 0)):
    torch.manual_seed(0)
    Ap, Wp = pack(torch.randn(M, K)).to(dev), pack(torch.randn(N, K) / K ** 0.5).to(dev)
    bias, gam = torch.randn(N, device=dev), torch.rand(N, device=dev)
    C = torch.randn(M, N, device=dev)
    Cp = torch.empty(M * N * (1 if h1p else 2), dtype=torch.float16, device=dev)
    t = timed(lambda: fn(Ap.data_ptr(), Wp.data_ptr(), M, N, K, epi, bias.data_ptr(), gam.data_ptr(), C.data_ptr(), C.data_ptr(), Cp.data_ptr(), None))
    tot += per_pass * t
    print(f"{'h1p' if h1p else 'x3p'} tile={os.environ.get('CTTS_CODEC_TILE', 'auto')} M={M} N={N} K={K} epi={epi}: {t:8.1f} us = {2.0 * M * N * K / t * 1e-6:6.1f} TFLOP/s algorithmic"
          f" (x {per_pass} per C3 pass)")
print(f"sum over the C3 pass's 40 launches: {tot / 1e3:.2f} ms")
