"""Codec decode time of a streaming prefix (64 x 72 tokens) and of one short batch: 256x256 vs 128x128 tiles (CTTS_X3_TILE=128)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chattts_amd import engine as E, weights as W
dev = torch.device("cuda:0")
codec = E.CodecEngine(W.synthetic_decoder(), W.synthetic_vocos(), dev)
for B, T in ((64, 72), (64, 144), (16, 72), (4, 300)):
    hid = torch.randn(B, T, 768, device=dev)
    for _ in range(2):
        codec.vocos_decode(codec.dvae_decode(hid))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        codec.vocos_decode(codec.dvae_decode(hid))
    torch.cuda.synchronize()
    print(f"tile {os.environ.get('CTTS_X3_TILE', '256')}: B {B} T {T} ({B * T * 2} frames): {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms", flush=True)
