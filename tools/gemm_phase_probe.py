"""Phase stamps of the perf-mode projection kernels (100 MHz realtime counter, wave 0 of every workgroup)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
dbg = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
os.environ["CTTS_GEMM_DBG_PTR"] = str(dbg.data_ptr())
from chattts_amd import _lib  # noqa: E402
lib = _lib.lib()
M = 64
xb = torch.randn(M, 768, device=dev).bfloat16(); ssq = torch.rand(M, 48, device=dev) * 16
x32 = torch.randn(M, 768, device=dev); act = torch.randn(M, 3072, device=dev).bfloat16()
qkv = torch.empty(M, 2304, device=dev); xb2 = torch.empty_like(xb); ssq2 = torch.empty_like(ssq)
actb = torch.empty(M, 3072, device=dev, dtype=torch.bfloat16)
W = {"qkv": torch.randn(2304, 768, device=dev).bfloat16(), "o": torch.randn(768, 768, device=dev).bfloat16(),
     "gu": torch.randn(6144, 768, device=dev).bfloat16(), "d": torch.randn(768, 3072, device=dev).bfloat16()}
flush = torch.empty(1024 * 1024 * 1024 // 4, device=dev)
def run(name):
    if name == "qkv": lib.ctts_k_gemm_fast(xb.data_ptr(), 768, W["qkv"].data_ptr(), M, 2304, 768, ssq.data_ptr(), 1e-6, 0, qkv.data_ptr(), 2304, None, 0, None, None)
    if name == "o": lib.ctts_k_gemm_fast(xb.data_ptr(), 768, W["o"].data_ptr(), M, 768, 768, None, 0.0, 1, x32.data_ptr(), 768, xb2.data_ptr(), 768, ssq2.data_ptr(), None)
    if name == "gu": lib.ctts_k_gemm_fast(xb.data_ptr(), 768, W["gu"].data_ptr(), M, 3072, 768, ssq.data_ptr(), 1e-6, 2, None, 0, actb.data_ptr(), 3072, None, None)
    if name == "d": lib.ctts_k_gemm_fast(act.data_ptr(), 3072, W["d"].data_ptr(), M, 768, 3072, None, 0.0, 1, x32.data_ptr(), 768, xb2.data_ptr(), 768, ssq2.data_ptr(), None)
for name in ("o", "qkv", "gu", "d"):
    for cold in (False, True):
        for rep in range(3):
            if cold: flush.fill_(1.0)
            torch.cuda.synchronize(); dbg.zero_(); torch.cuda.synchronize()
            run(name); torch.cuda.synchronize()
        t = dbg.view(-1, 8).cpu().numpy()
        t = t[t[:, 0] > 0][:, :5].astype(np.float64) * 10.0  # ns
        t0 = t[:, 0].min()
        rel = t - t0
        print(f"{name:4s} {'cold' if cold else 'hot '} wgs {len(t):4d} | entry spread {rel[:,0].max():7.0f} ns | per-WG: issue {np.mean(t[:,1]-t[:,0]):6.0f}  loads+mfma {np.mean(t[:,2]-t[:,1]):6.0f}  reduce/sync {np.mean(t[:,3]-t[:,2]):6.0f}  epilogue {np.mean(t[:,4]-t[:,3]):6.0f} | last exit {rel[:,4].max():7.0f} ns")
