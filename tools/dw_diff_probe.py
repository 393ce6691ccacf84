"""Where `dwconv_ln_run_k` (sliding window, large batches) and `dwconv_ln_k` (one wave per frame) differ bit for bit: the probe that
found hipcc contracting the LayerNorm's `v += d * d` in one inlining and not in the other (round 3; now 0 differing elements)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from chattts_amd import _lib
lib = _lib.lib(); DEV = torch.device('cuda:0'); f32 = np.float32
def dev(a): return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
for dil, B, F in ((2, 13, 1000), (1, 200, 70), (1, 13, 1000)):
    rs = np.random.RandomState(dil + F); Cc = 512
    x = rs.standard_normal((B, F, Cc)).astype(f32); w = rs.standard_normal((Cc, 1, 7)).astype(f32) * 0.4
    b = rs.standard_normal(Cc).astype(f32) * 0.1; lw = (1 + 0.1 * rs.standard_normal(Cc)).astype(f32); lb = rs.standard_normal(Cc).astype(f32) * 0.1
    y = torch.empty((B, F, Cc), dtype=torch.float32, device=DEV)
    keep = [dev(x), dev(w[:, 0, :].T), dev(b), dev(lw), dev(lb)]
    lib.ctts_k_dwconv_ln(*[k.data_ptr() for k in keep], 1e-6, dil, y.data_ptr(), B, F, None)
    got = y.cpu().numpy()
    ys = torch.empty((3, F, Cc), dtype=torch.float32, device=DEV); xs = dev(x[:3])
    lib.ctts_k_dwconv_ln(xs.data_ptr(), *[k.data_ptr() for k in keep[1:]], 1e-6, dil, ys.data_ptr(), 3, F, None)
    small = ys.cpu().numpy()
    d = (small != got[:3])
    fr = np.unique(np.nonzero(d.any(2))[1])
    print(dil, B, F, 'differing elements', int(d.sum()), 'max abs', float(np.abs(small - got[:3]).max()), 'frames', fr[:40], len(fr))
