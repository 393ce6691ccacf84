"""Phase times inside gemm_x3p_k (probe variant CTTS_X3P_VAR=3): per 16-wide k block, wave 0 of every workgroup."""
import os, sys
os.environ["CTTS_X3P_VAR"] = "3"
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
dbg = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
os.environ["CTTS_X3_DBG_PTR"] = str(dbg.data_ptr())
from chattts_amd import _lib  # noqa: E402
from chattts_amd.engine import pack_x3p  # noqa: E402
lib = _lib.lib()
M = 65536
for (N, K, epi) in ((2048, 512, 0), (512, 2048, 1)):
    Ap, Wp = pack_x3p(torch.randn(M, K)).to(dev), pack_x3p(torch.randn(N, K) / K ** 0.5).to(dev)
    bias, gam = torch.randn(N, device=dev), torch.rand(N, device=dev)
    C = torch.randn(M, N, device=dev); Cp = torch.empty(M * N * 2, dtype=torch.bfloat16, device=dev)
    for rep in range(2):
        dbg.zero_(); torch.cuda.synchronize()
        lib.ctts_k_gemm_x3p(Ap.data_ptr(), Wp.data_ptr(), M, N, K, epi, bias.data_ptr(), gam.data_ptr(), C.data_ptr(), C.data_ptr(), Cp.data_ptr(), None)
        torch.cuda.synchronize()
    t = dbg.view(-1, 8).cpu().numpy().astype(np.float64)
    t = t[t[:, :5].sum(1) > 0][:, :5] * 10.0 / (K // 16)   # ns per k block
    m = t.mean(0)
    print(f"N={N} K={K}: per 16-wide k block (ns, mean over {len(t)} workgroups): DMA wait {m[0]:6.0f} | barrier {m[1]:6.0f} | DMA issue {m[2]:6.0f} | "
          f"fragment reads {m[3]:6.0f} | MFMAs {m[4]:6.0f} | sum {m.sum():6.0f}")

# ---- gemm_h1p_k (gemm_mode 2, one fp16 plane): per 64-wide k stage, wave 0 of every workgroup (CTTS_H1P_PROBE=1) ----
if "--h1p" in sys.argv:
    from chattts_amd.engine import pack_h1p  # noqa: E402
    for (N, K, epi) in ((2048, 512, 0), (512, 2048, 1)):
        Ap, Wp = pack_h1p(torch.randn(M, K)).to(dev), pack_h1p(torch.randn(N, K) / K ** 0.5).to(dev)
        bias, gam = torch.randn(N, device=dev), torch.rand(N, device=dev)
        C = torch.randn(M, N, device=dev); Cp = torch.empty(M * N, dtype=torch.float16, device=dev)
        for rep in range(2):
            dbg.zero_(); torch.cuda.synchronize()
            lib.ctts_k_gemm_h1p(Ap.data_ptr(), Wp.data_ptr(), M, N, K, epi, bias.data_ptr(), gam.data_ptr(), C.data_ptr(), C.data_ptr(), Cp.data_ptr(), None)
            torch.cuda.synchronize()
        t = dbg.view(-1, 8).cpu().numpy().astype(np.float64)
        t = t[t[:, 6] > 0] * 10.0      # ns
        st = K // 64
        m = t.mean(0)
        print(f"h1p N={N} K={K}: {len(t)} workgroups, whole tile {m[6] / 1e3:6.1f} us | prologue {m[0] / 1e3:5.2f} us | per 64-wide k stage (ns): multiply + reads {m[1] / st:6.0f} | "
              f"own DMA wait {m[2] / st:6.0f} | barrier {m[3] / st:6.0f} | DMA issue + first reads {m[4] / st:6.0f} | epilogue {m[5] / 1e3:5.2f} us")
