"""sha256 of the decode attention's packed output on a seeded case, both modes -- run once per build of the library (CTTS_LIB) to show that a
change of the reduction MECHANISM (xor shuffles through LDS -> DPP / permlane swaps) left every bit where it was."""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chattts_amd import _lib  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda:0")
rs = np.random.RandomState(7)
B, nh, d, H, cmax, n_live = 64, 12, 64, 768, 560, 45
for mode, dt in (("bf16", torch.bfloat16), ("f32", torch.float32)):
    kc = torch.from_numpy(rs.standard_normal((B, nh, cmax, d)).astype(np.float32)).to(dt).to(dev)
    vc = torch.from_numpy(rs.standard_normal((B, nh, cmax, d)).astype(np.float32)).to(dt).to(dev)
    jlo = rs.randint(0, 30, size=n_live)
    slot = np.array([rs.randint(jlo[m] + 1, cmax) for m in range(n_live)])
    desc = np.zeros((B, 4), np.int32)
    desc[:, 0] = -1
    desc[:n_live, 0], desc[:n_live, 1], desc[:n_live, 2], desc[:n_live, 3] = rs.permutation(B)[:n_live], slot, slot - jlo, jlo
    qkv = torch.from_numpy(rs.standard_normal((B, 3 * H)).astype(np.float32)).to(dev)
    desc_d = torch.from_numpy(desc).to(dev)
    for persist in (0, 1):
        _lib.check(lib.ctts_k_attention_cfg(persist, 256, 4), "cfg")
        o = torch.zeros((B * H,), dtype=dt, device=dev)
        _lib.check(lib.ctts_k_attention_dec2(qkv.data_ptr(), kc.data_ptr(), vc.data_ptr(), _lib.BF16 if mode == "bf16" else _lib.F32, cmax, o.data_ptr(),
                                             desc_d.data_ptr(), None, 1, B, None), "attention_dec2")
        torch.cuda.synchronize()
        raw = o.view(torch.int16 if mode == "bf16" else torch.int32).cpu().numpy().tobytes()
        print(f"{mode} persist={persist} sha256 {hashlib.sha256(raw).hexdigest()}", flush=True)
