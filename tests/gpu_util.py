"""Helpers for the `-m gpu` parity tests: thin wrappers that call single kernels through the C ABI."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from chattts_amd import _lib

DEV = torch.device("cuda:0")


def dev(x, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(x)) if not isinstance(x, torch.Tensor) else x
    if dtype is not None:
        t = t.to(dtype)
    return t.contiguous().to(DEV)


def bf16_round(x: np.ndarray) -> np.ndarray:
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def gemm(A, W, *, wt="f32", tiled=False, epi=0, norm_w=None, eps=1e-6, res=None, bias=None, gamma=None,
         taps=1, cin=0, frames=0, pad=0, dil=1, n_out=None):
    """A [M(or B*F), lda] f32, W [N(,2N),K].  Returns C [M, N] numpy."""
    lib = _lib.lib()
    A_d = dev(A, torch.float32)
    if tiled == 2:  # split-bf16 tiles: weights packed [2][N][Kp]
        from chattts_amd.engine import split_bf16
        Wt = torch.as_tensor(np.ascontiguousarray(W), dtype=torch.float32)
        W_d = split_bf16(Wt).to(DEV)
        K = Wt.shape[1]
        N = n_out if n_out is not None else Wt.shape[0]
    else:
        W_d = dev(W, torch.bfloat16 if wt == "bf16" else torch.float32)
        K = W_d.shape[1]
        N = n_out if n_out is not None else W_d.shape[0]
    M = A_d.shape[0]
    C_d = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
    nw = None if norm_w is None else dev(norm_w, torch.float32)
    r = None if res is None else dev(res, torch.float32)
    b = None if bias is None else dev(bias, torch.float32)
    g = None if gamma is None else dev(gamma, torch.float32)
    rc = lib.ctts_k_gemm(int(tiled), A_d.data_ptr(), W_d.data_ptr(), C_d.data_ptr(), M, N, K, A_d.shape[1], N,
                         _lib.BF16 if wt == "bf16" else _lib.F32, epi, _lib.ptr(nw), eps, _lib.ptr(r), N, _lib.ptr(b), _lib.ptr(g),
                         taps, cin, frames, pad, dil, None)
    _lib.check(rc, "ctts_k_gemm")
    torch.cuda.synchronize()
    return C_d.cpu().numpy()


def relerr(got: np.ndarray, ref: np.ndarray) -> float:
    return float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).max() / max(1e-30, np.abs(ref).max()))
