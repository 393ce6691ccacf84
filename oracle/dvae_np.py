"""numpy restatement of the full DVAE: audio -> codes (`DVAE.forward(mode="encode")`,
`sample_audio`) and codes -> mel (`use_decoder=False` decode through the GFSQ codebook).

Reference: /root/reference/ChatTTS/model/dvae.py
  :175-206  MelSpectrogramFeatures  (torchaudio.transforms.MelSpectrogram(sr 24000, n_fft 1024, hop 256, n_mels 100,
            center, power=1) -> log(clip(mel, 1e-5)))
  :229-236  downsample_conv  Conv1d(100,512,3,1,1) GELU Conv1d(512,512,4,2,1) GELU
  :265-274  encode branch: mel / coef -> downsample -> encoder (DVAEDecoder 512->1024, hidden 256) -> GFSQ indices
  :69-128   GFSQ: GroupedResidualFSQ(dim 1024, levels (5,5,5,5), num_quantizers R=2, groups G=2);
            indices [G,B,T,R] -> [B,T,G*R] -> [B,G*R,T]; `_embed` is the inverse (get_output_from_indices)
  :276-297  decode branch: feat [B,1024,T] -> view(B,2,512,T).permute(0,2,3,1).flatten(2) -> decoder (512->512,
            hidden 256) -> out_conv -> * coef

PARITY UNPINNED for two third-party pieces that are neither vendored nor installed here (requirements.txt lists them
un-pinned):
  * `torchaudio.transforms.MelSpectrogram`: restated as torch.stft-equivalent framing (reflect pad n_fft/2, periodic
    hann, one-sided |rfft|) and the HTK mel filterbank (`melscale_fbanks`, norm=None).  The real checkpoint carries
    both buffers (`preprocessor_mel.mel_spec.spectrogram.window`, `.mel_scale.fb`), so only the framing is restated.
  * `vector_quantize_pytorch.GroupedResidualFSQ` / `ResidualFSQ` / `FSQ`: the published algorithm (Mentzer et al.
    2023, "Finite Scalar Quantization", and the library's residual wrapper) restated in `fsq_*` below:
    bound(z) = tanh(z + shift) * half_l - offset with half_l = (L-1)(1+1e-3)/2, round, / (L//2); index =
    sum((code * (L//2) + L//2) * basis), basis = cumprod([1, L...]); residual scales (L-1)^-r; the residual loop is
    seeded with bound(project_in(x)) (`bound_first`, the library's behaviour since 2024; False = the older loop).
The conv / ConvNeXt trunk IS pinned: tests/golden/dvae.npz holds outputs of the reference's own `DVAEDecoder` class
(oracle/make_dvae_goldens.py).  TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
from __future__ import annotations

import math

import numpy as np

from .codec_np import conv1d_cl, convnext_block, gelu

f32 = np.float32


# ---- mel front end ---------------------------------------------------------------------------------------------------
def hann_periodic(n: int) -> np.ndarray:
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)).astype(f32)


def melscale_fbanks(n_freqs: int = 513, f_min: float = 0.0, f_max: float = 12000.0, n_mels: int = 100,
                    sample_rate: int = 24000) -> np.ndarray:
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale="htk") -> [n_freqs, n_mels] float32."""
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = np.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up)).astype(f32)


def stft_mag(wav: np.ndarray, window: np.ndarray, n_fft: int = 1024, hop: int = 256) -> np.ndarray:
    """|torch.stft(center=True, pad_mode="reflect", onesided)|: wav [n] -> [F, n_fft/2+1], F = 1 + n // hop."""
    x = np.pad(wav.astype(np.float64), (n_fft // 2, n_fft // 2), mode="reflect")
    F = 1 + (x.size - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(F)[:, None]
    frames = x[idx] * window.astype(np.float64)
    return np.abs(np.fft.rfft(frames, axis=-1)).astype(f32)


def mel_features(wav: np.ndarray, window: np.ndarray, fb: np.ndarray) -> np.ndarray:
    """dvae.py:200-206 -> [F, 100] channels-last."""
    mel = stft_mag(wav, window) @ fb.astype(f32)
    return np.log(np.maximum(mel, f32(1e-5))).astype(f32)


# ---- FSQ -------------------------------------------------------------------------------------------------------------
def fsq_bound(z: np.ndarray, levels: np.ndarray, eps: float = 1e-3) -> np.ndarray:
    half_l = (levels - 1).astype(f32) * f32(1 + eps) / f32(2)
    offset = np.where(levels % 2 == 0, f32(0.5), f32(0.0)).astype(f32)
    shift = np.arctanh(offset / half_l).astype(f32)
    return (np.tanh(z + shift) * half_l - offset).astype(f32)


def fsq_quantize(z: np.ndarray, levels: np.ndarray):
    """-> (normalised codes in [-1, 1], flat index)"""
    half_w = (levels // 2).astype(f32)
    q = np.round(fsq_bound(z, levels))            # round half to even, like torch.round
    codes = (q / half_w).astype(f32)
    basis = np.cumprod(np.concatenate([[1], levels[:-1]])).astype(np.int64)
    idx = ((codes * half_w + half_w) * basis).sum(-1).astype(np.int32)
    return codes, idx


def fsq_codes_from_index(idx: np.ndarray, levels: np.ndarray) -> np.ndarray:
    basis = np.cumprod(np.concatenate([[1], levels[:-1]])).astype(np.int64)
    lv = (idx[..., None].astype(np.int64) // basis) % levels
    half_w = (levels // 2).astype(f32)
    return ((lv.astype(f32) - half_w) / half_w).astype(f32)


def _q(sd, g, name):
    return np.asarray(sd[f"vq_layer.quantizer.rvqs.{g}.{name}"], dtype=f32)


def gfsq_encode(sd: dict, x: np.ndarray, levels=(5, 5, 5, 5), G: int = 2, R: int = 2, bound_first: bool = True) -> np.ndarray:
    """x [..., 1024] -> indices [..., G*R] (order g-major, then r: dvae.py:108-114)."""
    lv = np.asarray(levels, dtype=np.int64)
    D = x.shape[-1] // G
    out = []
    for g in range(G):
        z = (x[..., g * D: (g + 1) * D].astype(f32) @ _q(sd, g, "project_in.weight").T + _q(sd, g, "project_in.bias")).astype(f32)
        residual = fsq_bound(z, lv) if bound_first else z
        for r in range(R):
            scale = ((lv - 1).astype(f32) ** f32(-r)).astype(f32)
            codes, idx = fsq_quantize((residual / scale).astype(f32), lv)
            residual = (residual - codes * scale).astype(f32)
            out.append(idx)
    return np.stack(out, -1)


def gfsq_round_margin(sd: dict, x: np.ndarray, levels=(5, 5, 5, 5), G: int = 2, R: int = 2, bound_first: bool = True) -> np.ndarray:
    """x [..., 1024] -> [..., G*R]: for every emitted index, the distance of the NEAREST of its pre-rounding FSQ coordinates from a
    rounding boundary (half-integer).  An index can differ between two float32 evaluations of the trunk only where this is within their
    rounding noise -- what tests/test_gpu_dvae.py demands of every code that differs from the golden."""
    lv = np.asarray(levels, dtype=np.int64)
    D = x.shape[-1] // G
    out = []
    for g in range(G):
        z = (x[..., g * D: (g + 1) * D].astype(f32) @ _q(sd, g, "project_in.weight").T + _q(sd, g, "project_in.bias")).astype(f32)
        residual = fsq_bound(z, lv) if bound_first else z
        for r in range(R):
            scale = ((lv - 1).astype(f32) ** f32(-r)).astype(f32)
            pre = fsq_bound((residual / scale).astype(f32), lv).astype(np.float64)
            out.append(np.abs(np.abs(pre - np.floor(pre)) - 0.5).min(-1))
            codes, _ = fsq_quantize((residual / scale).astype(f32), lv)
            residual = (residual - codes * scale).astype(f32)
    return np.stack(out, -1)


def gfsq_embed(sd: dict, idx: np.ndarray, levels=(5, 5, 5, 5), G: int = 2, R: int = 2) -> np.ndarray:
    """indices [..., G*R] -> features [..., 1024] (`get_output_from_indices`)."""
    lv = np.asarray(levels, dtype=np.int64)
    feats = []
    for g in range(G):
        acc = 0.0
        for r in range(R):
            scale = ((lv - 1).astype(f32) ** f32(-r)).astype(f32)
            acc = acc + fsq_codes_from_index(idx[..., g * R + r], lv) * scale
        feats.append((acc.astype(f32) @ _q(sd, g, "project_out.weight").T + _q(sd, g, "project_out.bias")).astype(f32))
    return np.concatenate(feats, -1)


# ---- trunk -----------------------------------------------------------------------------------------------------------
def conv1d_k4s2_cl(x: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    """nn.Conv1d(C, Cout, 4, stride 2, padding 1) on channels-last x [B,F,C] -> [B, (F-2)//2+1, Cout]."""
    B, F, C = x.shape
    Fo = (F + 2 - 4) // 2 + 1
    xp = np.zeros((B, F + 2, C), dtype=f32)
    xp[:, 1: F + 1] = x
    y = np.zeros((B, Fo, w.shape[0]), dtype=f32)
    for j in range(4):
        y += xp[:, j: j + 2 * Fo: 2] @ w[:, :, j].T
    return (y + b).astype(f32)


def dvae_trunk(sd: dict, prefix: str, x: np.ndarray) -> np.ndarray:
    """DVAEDecoder.forward (dvae.py:163-172) for `encoder.` / `decoder.`: conv_in (k3, GELU, k3) -> ConvNeXt blocks
    (k7, dilation 2) -> conv_out (k1, no bias)."""
    g = lambda k: np.asarray(sd[prefix + k], dtype=f32)
    x = gelu(conv1d_cl(x, g("conv_in.0.weight"), g("conv_in.0.bias"), pad=1))
    x = conv1d_cl(x, g("conv_in.2.weight"), g("conv_in.2.bias"), pad=1)
    n = 0
    while f"{prefix}decoder_block.{n}.weight" in sd:
        x = convnext_block(x, sd, f"{prefix}decoder_block.{n}.", "weight", dil=2)
        n += 1
    return conv1d_cl(x, g("conv_out.weight"), None, pad=0)


def encoder_features(sd: dict, mel: np.ndarray) -> np.ndarray:
    """log-mel [F,100] -> pre-quantiser features [T,1024] (dvae.py:266-272)."""
    g = lambda k: np.asarray(sd[k], dtype=f32)
    x = (mel / g("coef").reshape(1, -1)).astype(f32)[None]
    x = gelu(conv1d_cl(x, g("downsample_conv.0.weight"), g("downsample_conv.0.bias"), pad=1))
    x = gelu(conv1d_k4s2_cl(x, g("downsample_conv.2.weight"), g("downsample_conv.2.bias")))
    return dvae_trunk(sd, "encoder.", x)[0]


def dvae_encode(sd: dict, wav: np.ndarray, bound_first: bool = True) -> np.ndarray:
    """`DVAE.sample_audio(wav)` (dvae.py:299-303): wav [n] float32 -> codes [4, T] int32."""
    g = lambda k: np.asarray(sd[k], dtype=f32)
    mel = mel_features(wav, g("preprocessor_mel.mel_spec.spectrogram.window"), g("preprocessor_mel.mel_spec.mel_scale.fb"))
    return gfsq_encode(sd, encoder_features(sd, mel), bound_first=bound_first).T.copy()


def dvae_decode_codes(sd: dict, codes: np.ndarray) -> np.ndarray:
    """codes [B,T,4] (zero padded, core.py:525-533) -> mel [B,2T,100] channels-last (dvae.py:276-297)."""
    g = lambda k: np.asarray(sd[k], dtype=f32)
    B, T, _ = codes.shape
    x = gfsq_embed(sd, codes).reshape(B, 2 * T, 512)     # [B,T,1024] -> frame 2t+j = channels [512j, 512j+512)
    x = dvae_trunk(sd, "decoder.", x)
    x = conv1d_cl(x, g("out_conv.weight"), None, pad=1)
    return (x * g("coef").reshape(1, 1, -1)).astype(f32)
