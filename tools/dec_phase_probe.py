"""Phase stamps of the packed decode projection kernels (csrc/decode.hip; 100 MHz realtime counter, thread 0 of every workgroup).
stamps: 0 kernel entry | 1 body entry (live-row count known) | 2 all loads of round 0 issued | 3 MFMAs done (= loads landed)
        | 4 after the LDS reduction barrier | 5 epilogue done"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
dbg = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
os.environ["CTTS_GEMM_DBG_PTR"] = str(dbg.data_ptr())
from chattts_amd import _lib  # noqa: E402
from chattts_amd.engine import pack_frag  # noqa: E402
lib = _lib.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Mp = (M + 15) // 16 * 16
xp = pack_frag(torch.randn(Mp, 768).bfloat16()).to(dev); ssq = torch.rand(M, 48, device=dev) * 16
x32 = torch.randn(M, 768, device=dev); actp = pack_frag(torch.randn(Mp, 3072).bfloat16()).to(dev)
xp2 = torch.empty_like(xp); ssq2 = torch.empty_like(ssq); actp2 = torch.empty_like(actp)
na = torch.tensor([M], dtype=torch.int32, device=dev)
W = {"o": pack_frag(torch.randn(768, 768).bfloat16()).to(dev), "gu": pack_frag(torch.randn(6144, 768).bfloat16()).to(dev),
     "d": pack_frag(torch.randn(768, 3072).bfloat16()).to(dev)}
flush = torch.empty(1024 * 1024 * 1024 // 4, device=dev)
def run(name):
    if name == "o": lib.ctts_k_gemm_dec(xp.data_ptr(), W["o"].data_ptr(), M, 768, 768, na.data_ptr(), None, 0.0, 1, x32.data_ptr(), 768, xp2.data_ptr(), 24, ssq2.data_ptr(), 0, None)
    if name == "gu": lib.ctts_k_gemm_dec(xp.data_ptr(), W["gu"].data_ptr(), M, 3072, 768, na.data_ptr(), ssq.data_ptr(), 1e-6, 2, None, 0, actp2.data_ptr(), 96, None, 0, None)
    if name == "d": lib.ctts_k_gemm_dec(actp.data_ptr(), W["d"].data_ptr(), M, 768, 3072, na.data_ptr(), None, 0.0, 1, x32.data_ptr(), 768, xp2.data_ptr(), 24, ssq2.data_ptr(), 0, None)
for name in ("o", "gu", "d"):
    for cold in (False, True):
        for rep in range(3):
            if cold:
                flush.fill_(1.0)
                # what the previous kernel of the step would have left near the caches: the activations
                xp.add_(0); actp.add_(0); ssq.add_(0); x32.add_(0)
            torch.cuda.synchronize(); dbg.zero_(); torch.cuda.synchronize()
            run(name); torch.cuda.synchronize()
        t = dbg.view(-1, 8).cpu().numpy()
        t = t[t[:, 0] > 0][:, :6].astype(np.float64) * 10.0  # ns
        rel = t - t[:, 0].min()
        d = lambda a, b: np.mean(t[:, a] - t[:, b])
        print(f"M={M} {name:3s} {'cold-W' if cold else 'hot   '} wgs {len(t):4d} | entry spread {rel[:,0].max():6.0f} ns | n_active {d(1,0):5.0f}  issue {d(2,1):5.0f}  "
              f"land+mfma {d(3,2):5.0f}  reduce {d(4,3):5.0f}  epilogue {d(5,4):5.0f} | last exit {rel[:,5].max():6.0f} ns")
