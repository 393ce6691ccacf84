"""Runs only the acoustic decoder on a C3-sized batch (64 x 512 tokens): target for rocprofv3 --pmc passes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chattts_amd import engine as E, weights as W
dev = torch.device("cuda:0")
codec = E.CodecEngine(W.synthetic_decoder(), W.synthetic_vocos(), dev, gemm=os.environ.get("CTTS_CODEC_GEMM", "bf16x3"))
hid = torch.randn(64, 512, 768, device=dev)
for _ in range(2):
    codec.vocos_decode(codec.dvae_decode(hid))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    wav = codec.vocos_decode(codec.dvae_decode(hid))
torch.cuda.synchronize()
print("codec ms per pass", (time.perf_counter() - t0) / 3 * 1e3, tuple(wav.shape))
