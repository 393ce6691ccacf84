"""Probe: split-bf16 tiled GEMM time vs the row stride of A and C (power-of-two strides vs +64 floats)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chattts_amd import _lib  # noqa: E402
from chattts_amd.engine import split_bf16  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda:0")
M = 65536
torch.manual_seed(0)


def run(N, K, lda, ldc, epi=4):
    A = torch.randn(M, lda, device=dev)
    W = split_bf16(torch.randn(N, K) * 0.05).to(dev)
    Cm = torch.empty(M, ldc, device=dev)
    bias = torch.zeros(N, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def go():
        rc = lib.ctts_k_gemm(2, A.data_ptr(), W.data_ptr(), Cm.data_ptr(), M, N, K, lda, ldc, 0, epi, None, 0.0, None, 0, bias.data_ptr(), None,
                             1, 0, 0, 0, 1, st)
        assert rc == 0, lib.ctts_last_error()
    for _ in range(2):
        go()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        go()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 5 * 1e3
    print(f"N {N:5d} K {K:5d} lda {lda:5d} ldc {ldc:5d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s(f32-equiv)", flush=True)


for N, K in ((2048, 512), (512, 2048), (1536, 512), (512, 1536)):
    for pa, pc in ((0, 0), (64, 0), (0, 64), (64, 64), (32, 32), (16, 16)):
        run(N, K, K + pa, N + pc)
