"""BUILD-CONTAINER ONLY.  tests/golden/dvae.npz: the full-DVAE path evaluated with the REAL reference
classes where they can be imported -- `DVAEDecoder` (both trunks) and `ConvNeXtBlock` from
/root/reference/ChatTTS/model/dvae.py, torch's own `Conv1d` / `GELU` composed exactly as `DVAE.__init__` does for
`downsample_conv` / `out_conv` (dvae.py:229-239), torch.stft for the spectrogram -- and with the numpy restatement
(oracle/dvae_np.py) for the two pieces whose packages are absent (torchaudio's framing call is torch.stft; the GFSQ
quantiser is restated, parity unpinned).

    python -m oracle.make_dvae_goldens
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn

from chattts_amd import weights as W
from oracle import dvae_np, ref_harness

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {"half_second": dict(seed=3, n=12000), "odd_frames": dict(seed=4, n=256 * 40 + 17), "short": dict(seed=5, n=2100)}


def test_wave(seed: int, n: int) -> np.ndarray:
    """a few decaying partials + noise: deterministic, speech-like level"""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 24000.0
    x = sum(a * np.sin(2 * np.pi * f * t + p) for a, f, p in zip(rng.uniform(0.02, 0.2, 6), rng.uniform(80, 4000, 6), rng.uniform(0, 6.28, 6)))
    return (x * np.exp(-1.5 * t) + 0.01 * rng.standard_normal(n)).astype(np.float32)


def build(sd):
    m = ref_harness.ref_modules()["dvae"]
    sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
    enc = m.DVAEDecoder(idim=512, odim=1024, hidden=256, n_layer=12, bn_dim=128)
    enc.load_state_dict(sub("encoder."))
    dec = m.DVAEDecoder(idim=512, odim=512, hidden=256, n_layer=12, bn_dim=128)
    dec.load_state_dict(sub("decoder."))
    down = nn.Sequential(nn.Conv1d(100, 512, 3, 1, 1), nn.GELU(), nn.Conv1d(512, 512, 4, 2, 1), nn.GELU())
    down.load_state_dict(sub("downsample_conv."))
    out_conv = nn.Conv1d(512, 100, 3, 1, 1, bias=False)
    out_conv.load_state_dict(sub("out_conv."))
    return enc.eval(), dec.eval(), down.eval(), out_conv.eval()


@torch.inference_mode()
def main():
    sd = W.synthetic_dvae()
    nsd = {k: v.numpy() for k, v in sd.items()}
    enc, dec, down, out_conv = build(sd)
    coef = sd["coef"]
    out = {}
    for name, c in CASES.items():
        wav = test_wave(**c)
        # MelSpectrogram(power=1): |stft| (center, reflect, periodic hann) -> fb; then log(clip) (dvae.py:200-206)
        spec = torch.stft(torch.from_numpy(wav), 1024, 256, 1024, sd["preprocessor_mel.mel_spec.spectrogram.window"], center=True,
                          pad_mode="reflect", return_complex=True).abs()                        # [513, F]
        mel = torch.log(torch.clip(sd["preprocessor_mel.mel_spec.mel_scale.fb"].T @ spec, min=1e-5))   # [100, F]
        x = down(torch.div(mel, coef.view(100, 1))).unsqueeze(0)                                 # dvae.py:267-270
        feat = enc(x)[0].T.contiguous().numpy()                                                  # [T, 1024]
        codes = dvae_np.gfsq_encode(nsd, feat)                                                   # restated quantiser
        vq = torch.from_numpy(dvae_np.gfsq_embed(nsd, codes)).T.unsqueeze(0)                     # [1, 1024, T]
        vq = vq.view(1, 2, 512, vq.size(2)).permute(0, 2, 3, 1).flatten(2)                       # dvae.py:281-287
        mel_out = torch.mul(out_conv(dec(vq)), coef)[0].T.contiguous().numpy()                   # [2T, 100]
        out[name + ".wav"] = wav
        out[name + ".logmel"] = mel.T.contiguous().numpy()
        out[name + ".feat"] = feat
        out[name + ".codes"] = codes.astype(np.int32)
        out[name + ".mel_out"] = mel_out
        # cross-check of the numpy restatement against the reference-class outputs (the oracle's own pin)
        o_mel = dvae_np.mel_features(wav, nsd["preprocessor_mel.mel_spec.spectrogram.window"], nsd["preprocessor_mel.mel_spec.mel_scale.fb"])
        o_feat = dvae_np.encoder_features(nsd, o_mel)
        o_out = dvae_np.dvae_decode_codes(nsd, codes[None])[0]
        print(f"{name}: F={mel.shape[1]} T={feat.shape[0]}  oracle-vs-reference  logmel {np.abs(o_mel - out[name + '.logmel']).max():.2e}"
              f"  feat {np.abs(o_feat - feat).max():.2e} (|feat| {np.abs(feat).mean():.2f})  mel_out {np.abs(o_out - mel_out).max():.2e}"
              f"  codes equal {np.array_equal(dvae_np.gfsq_encode(nsd, o_feat), codes)}")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "dvae.npz"), **out)
    with open(os.path.join(ROOT, "tests", "golden", "weights_fingerprint.txt")) as f:
        lines = [l for l in f.read().splitlines() if l and not l.startswith("dvae ")]
    lines.append("dvae " + W.fingerprint(sd))
    with open(os.path.join(ROOT, "tests", "golden", "weights_fingerprint.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
