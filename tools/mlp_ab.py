"""One ConvNeXt MLP of the acoustic decoder (pwconv1 -> GELU -> pwconv2 -> gamma -> residual) as TWO launches (ctts_k_gemm_h1p x 2) and as ONE
(ctts_k_mlp_fused), hip events over 20 launches each, at the C3 pass's shapes (64 utterances padded to 2 x 512 frames = 65,536 rows: Vocos
inter 1536 x 8 blocks, DVAE decoder inter 2048 x 12 blocks) and at smaller batches.  Also checks that the two results are bit-identical."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chattts_amd import _lib  # noqa: E402
from chattts_amd.engine import pack_h1p  # noqa: E402
dev = torch.device("cuda:0")
lib = _lib.lib()


def timed(f, n=20):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot2 = tot1 = 0.0
for (M, inter, per_pass) in ((65536, 1536, 8), (65536, 2048, 12), (32768, 2048, 0), (49152, 1536, 0), (40960, 2048, 0), (9216, 1536, 0), (2304, 2048, 0)):
    torch.manual_seed(0)
    Mp = (M + 255) // 256 * 256
    A = torch.zeros(Mp, 512); A[:M] = torch.randn(M, 512)
    Ap, W1p, W2p = pack_h1p(A).to(dev), pack_h1p(torch.randn(inter, 512) / 512 ** 0.5).to(dev), pack_h1p(torch.randn(512, inter) / inter ** 0.5).to(dev)
    b1, b2, gam = torch.randn(inter, device=dev) * 0.1, torch.randn(512, device=dev) * 0.1, torch.rand(512, device=dev) * 0.1
    res = torch.randn(M, 512, device=dev)
    Hp = torch.empty(Mp * inter, dtype=torch.float16, device=dev)
    Ca, Cb = res.clone(), res.clone()

    def two(C):
        lib.ctts_k_gemm_h1p(Ap.data_ptr(), W1p.data_ptr(), M, inter, 512, 0, b1.data_ptr(), None, None, None, Hp.data_ptr(), None)
        lib.ctts_k_gemm_h1p(Hp.data_ptr(), W2p.data_ptr(), M, 512, inter, 1, b2.data_ptr(), gam.data_ptr(), C.data_ptr(), C.data_ptr(), None, None)

    def one(C):
        lib.ctts_k_mlp_fused(Ap.data_ptr(), W1p.data_ptr(), W2p.data_ptr(), M, inter, b1.data_ptr(), b2.data_ptr(), gam.data_ptr(), C.data_ptr(), 1, None)

    two(Ca); one(Cb); torch.cuda.synchronize()
    same = bool(torch.equal(Ca.view(torch.int32), Cb.view(torch.int32)))
    t2, t1 = timed(lambda: two(Ca)), timed(lambda: one(Cb))
    fl = 4.0 * M * inter * 512
    tot2 += per_pass * t2; tot1 += per_pass * t1
    print(f"M={M} inter={inter}: two launches {t2:8.1f} us ({fl / t2 * 1e-6:6.1f} TFLOP/s)   one launch {t1:8.1f} us ({fl / t1 * 1e-6:6.1f} TFLOP/s = "
          f"{fl / t1 * 1e-6 / 2500:.3f} of the dense fp16 peak)   bit-identical: {same}   (x {per_pass} per C3 pass)", flush=True)
print(f"sum over the C3 pass's 20 MLPs: two launches {tot2 / 1e3:.2f} ms, one launch {tot1 / 1e3:.2f} ms")
