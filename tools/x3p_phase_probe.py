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
