"""Prefill attention (one wave per query row) at long prompts: 8 rows per workgroup vs one (CTTS_ATT_PQR=1 in the environment)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chattts_amd import _lib  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda:0")
torch.manual_seed(0)
for B, T in ((64, 48), (16, 400), (64, 400)):
    cmax = T + 8
    kc = (torch.randn(B, 12, cmax, 64, device=dev) * 0.5).bfloat16()
    vc = (torch.randn(B, 12, cmax, 64, device=dev) * 0.5).bfloat16()
    qkv = torch.randn(B * T, 2304, device=dev)
    out = torch.empty(B * T, 768, device=dev)
    ks = torch.zeros(B, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def go():
        lib.ctts_k_attention(qkv.data_ptr(), kc.data_ptr(), vc.data_ptr(), 1, cmax, out.data_ptr(), T, None, ks.data_ptr(), B * T, st)
    for _ in range(2):
        go()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        go()
    b.record()
    torch.cuda.synchronize()
    print(f"PQR={os.environ.get('CTTS_ATT_PQR', '8')} B {B} T {T}: {a.elapsed_time(b) / 5 * 1e3:9.1f} us per layer", flush=True)
