"""In-kernel phase times of the fused ConvNeXt MLP (mlp_fused_h1p_k<PROBE>): wave 0 of every workgroup accumulates 100 MHz stamps."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
M, inter = int(os.environ.get("M", 65536)), int(os.environ.get("INTER", 2048))
tiles = (M + 127) // 128
dbg = torch.zeros(tiles * 8, dtype=torch.int64, device=dev)
os.environ["CTTS_X3_DBG_PTR"] = str(dbg.data_ptr())
from chattts_amd import _lib  # noqa: E402
from chattts_amd.engine import pack_h1p  # noqa: E402
lib = _lib.lib()
torch.manual_seed(0)
Mp = (M + 255) // 256 * 256
A = torch.zeros(Mp, 512); A[:M] = torch.randn(M, 512)
Ap, W1p, W2p = pack_h1p(A).to(dev), pack_h1p(torch.randn(inter, 512) / 512 ** 0.5).to(dev), pack_h1p(torch.randn(512, inter) / inter ** 0.5).to(dev)
b1, b2, gam = torch.randn(inter, device=dev) * 0.1, torch.randn(512, device=dev) * 0.1, torch.rand(512, device=dev) * 0.1
C = torch.randn(M, 512, device=dev)
for _ in range(3):
    dbg.zero_()
    lib.ctts_k_mlp_fused(Ap.data_ptr(), W1p.data_ptr(), W2p.data_ptr(), M, inter, b1.data_ptr(), b2.data_ptr(), gam.data_ptr(), C.data_ptr(), 1, None)
    torch.cuda.synchronize()
d = dbg.view(tiles, 8).cpu().double() * 0.01   # us
names = ["DMA wait", "barrier", "DMA issue", "P1 reads+MFMA", "GELU + H writes", "P3 reads+MFMA", "epilogue", "whole tile"]
print(f"M={M} inter={inter}: {tiles} workgroups; mean / min / max us per workgroup (wave 0)")
for i, n in enumerate(names):
    print(f"  {n:18s} {d[:, i].mean():8.2f} {d[:, i].min():8.2f} {d[:, i].max():8.2f}")
