"""Timing of the acoustic decoder's point-wise GEMM pair alone (hip events, 20 launches): the LDS-DMA staged kernel on pre-split
planes (csrc/codec_gemm.hip, variant = CTTS_X3P_VAR) vs the register-staged split-bf16 tiles (gemm.hip) at the bench's size."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chattts_amd import _lib  # noqa: E402
from chattts_amd.engine import pack_x3p, split_bf16  # noqa: E402
dev = torch.device("cuda:0")
lib = _lib.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (N, K, epi) in ((2048, 512, 0), (512, 2048, 1), (1536, 512, 0), (512, 1536, 1)):
    A = torch.randn(M, K)
    W = torch.randn(N, K) / K ** 0.5
    Ap, Wp = pack_x3p(A).to(dev), pack_x3p(W).to(dev)
    bias, gam = torch.randn(N, device=dev), torch.rand(N, device=dev)
    C = torch.randn(M, N, device=dev)
    Cp = torch.empty(M * N * 2, dtype=torch.bfloat16, device=dev)
    t_new = timed(lambda: lib.ctts_k_gemm_x3p(Ap.data_ptr(), Wp.data_ptr(), M, N, K, epi, bias.data_ptr(), gam.data_ptr(), C.data_ptr(), C.data_ptr(), Cp.data_ptr(), None))
    A_d, W_s = A.to(dev), split_bf16(W).to(dev)
    C2 = torch.empty(M, N, device=dev)
    oe = 4 if epi == 0 else 5
    t_old = timed(lambda: lib.ctts_k_gemm(2, A_d.data_ptr(), W_s.data_ptr(), C2.data_ptr(), M, N, K, K, N, 0, oe, None, 0.0, C2.data_ptr(), N, bias.data_ptr(), gam.data_ptr(), 1, 0, 0, 0, 1, None))
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K} epi={epi}: LDS-DMA planes {t_new:8.1f} us = {fl / t_new * 1e-6:6.1f} TFLOP/s f32-equiv ({3 * fl / t_new * 1e-6:6.0f} bf16 MFMA) | "
          f"register-staged tiles {t_old:8.1f} us = {fl / t_old * 1e-6:6.1f} TFLOP/s")
