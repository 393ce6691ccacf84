"""Phase probe of the split-bf16 GEMM tiles (needs a -DCTTS_X3_PROBE build of the library: CTTS_LIB=...): per-workgroup
accumulated time of: fetch issue, MFMA phase, barrier, stage (incl. the wait for the loads), barrier."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
dbg = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
os.environ["CTTS_X3_DBG_PTR"] = str(dbg.data_ptr())
from chattts_amd import _lib  # noqa: E402
from chattts_amd.engine import split_bf16  # noqa: E402

lib = _lib.lib()
M = 65536
for N, K in ((2048, 512), (512, 2048)):
    A = torch.randn(M, K, device=dev)
    W = split_bf16(torch.randn(N, K) * 0.05).to(dev)
    Cm = torch.empty(M, N, device=dev)
    bias = torch.zeros(N, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for rep in range(3):
        dbg.zero_()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        lib.ctts_k_gemm(2, A.data_ptr(), W.data_ptr(), Cm.data_ptr(), M, N, K, K, N, 0, 4, None, 0.0, None, 0, bias.data_ptr(), None, 1, 0, 0, 0, 1, st)
        b.record()
        torch.cuda.synchronize()
    tiles = (N // 256) * (M // 256)
    d = dbg.view(-1, 8)[: ((tiles + 7) // 8) * 8].cpu().double() * 10.0   # 100 MHz ticks -> ns
    d = d[d[:, 1] > 0]
    steps = (K + 31) // 32
    m = d.mean(0) / steps
    print(f"N {N} K {K}: {a.elapsed_time(b) * 1e3:.0f} us, {d.shape[0]} tiles, per k-step ns: fetch-issue {m[0]:.0f}  mfma {m[1]:.0f}  barrier {m[2]:.0f}  "
          f"stage(+wait) {m[3]:.0f}  barrier {m[4]:.0f}  sum {m[:5].sum():.0f}", flush=True)
