"""numpy float32 restatement of the acoustic decoder: DVAE decode branch + Vocos decode.

DVAE: /root/reference/ChatTTS/model/dvae.py:276-297 (reshape, decoder, out_conv, coef),
      :163-172 (DVAEDecoder.forward), :46-66 (ConvNeXtBlock.forward), :145-161 (layer shapes).
Vocos: un-vendored third-party package `vocos` (requirements.txt:8, un-pinned; not installed in
      the build container).  Call sites /root/reference/ChatTTS/core.py:298-317,505-510; class
      paths/init args /root/reference/ChatTTS/config/config.py:74-121; head arithmetic restated
      in-tree at /root/reference/examples/onnx/exporter.py:392-405.  Published algorithm
      (vocos.models.VocosBackbone / vocos.modules.ConvNeXtBlock / vocos.heads.ISTFTHead /
      vocos.spectral_ops.ISTFT with padding="center" == torch.istft(center=True)) restated here.

Layout: everything is channels-last [B, F, C] (F = mel frames); the reference is [B, C, F].

TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
from __future__ import annotations

import math

import numpy as np

f32 = np.float32
_erf = np.vectorize(math.erf, otypes=[np.float64])


def gelu(x: np.ndarray) -> np.ndarray:
    """nn.GELU() default (approximate='none'): 0.5 x (1 + erf(x / sqrt 2))."""
    try:
        from scipy.special import erf
        e = erf(x.astype(np.float64) / math.sqrt(2.0))
    except Exception:  # pragma: no cover
        e = _erf(x.astype(np.float64) / math.sqrt(2.0))
    return (0.5 * x.astype(np.float64) * (1.0 + e)).astype(f32)


def layer_norm(x: np.ndarray, w: np.ndarray, b: np.ndarray, eps: float) -> np.ndarray:
    """nn.LayerNorm over the channel (last) axis, biased variance."""
    x64 = x.astype(np.float64)
    mu = x64.mean(-1, keepdims=True)
    var = ((x64 - mu) ** 2).mean(-1, keepdims=True)
    return (((x64 - mu) / np.sqrt(var + eps)) * w + b).astype(f32)


def conv1d_cl(x: np.ndarray, w: np.ndarray, b, pad: int, dil: int = 1) -> np.ndarray:
    """Dense nn.Conv1d on channels-last x [B,F,Cin]; w is the torch layout [Cout, Cin, k]."""
    B, F, Cin = x.shape
    Cout, _, k = w.shape
    xp = np.zeros((B, F + 2 * pad, Cin), dtype=f32)
    xp[:, pad: pad + F] = x
    y = np.zeros((B, F, Cout), dtype=f32)
    for j in range(k):
        y += xp[:, j * dil: j * dil + F] @ w[:, :, j].T
    if b is not None:
        y += b
    return y.astype(f32)


def dwconv1d_cl(x: np.ndarray, w: np.ndarray, b: np.ndarray, pad: int, dil: int) -> np.ndarray:
    """Depthwise nn.Conv1d(groups=C) on channels-last x; w is [C, 1, k]."""
    B, F, C = x.shape
    k = w.shape[-1]
    xp = np.zeros((B, F + 2 * pad, C), dtype=f32)
    xp[:, pad: pad + F] = x
    y = np.zeros((B, F, C), dtype=f32)
    for j in range(k):
        y += xp[:, j * dil: j * dil + F] * w[:, 0, j]
    return (y + b).astype(f32)


def convnext_block(x, sd, p, gamma_key, dil, eps=1e-6):
    """dvae.py:46-66 / vocos.modules.ConvNeXtBlock.forward:
    dwconv(k7, pad 3*dil, dilation dil) -> LayerNorm -> Linear -> GELU -> Linear -> *gamma -> +residual."""
    g = lambda k: np.asarray(sd[p + k], dtype=f32)
    y = dwconv1d_cl(x, g("dwconv.weight"), g("dwconv.bias"), pad=3 * dil, dil=dil)
    y = layer_norm(y, g("norm.weight"), g("norm.bias"), eps)
    y = gelu(y @ g("pwconv1.weight").T + g("pwconv1.bias"))
    y = y @ g("pwconv2.weight").T + g("pwconv2.bias")
    y = y * np.asarray(sd[p + gamma_key], dtype=f32)
    return (x + y).astype(f32)


def dvae_decode(sd: dict, hid: np.ndarray) -> np.ndarray:
    """hid [B, T, 768] (the per-token hidden states, i.e. `batch_result` of core.py:519-534 before
    its transpose) -> mel [B, 2T, 100] channels-last.

    dvae.py:281-287: (B,768,T).view(B,2,384,T).permute(0,2,3,1).flatten(2) puts hidden channels
    [0,384) of token t at frame 2t and [384,768) at frame 2t+1 -- in channels-last terms a plain
    reshape [B,T,768] -> [B,2T,384]."""
    g = lambda k: np.asarray(sd[k], dtype=f32)
    B, T, _ = hid.shape
    x = hid.astype(f32).reshape(B, 2 * T, 384)
    x = gelu(conv1d_cl(x, g("decoder.conv_in.0.weight"), g("decoder.conv_in.0.bias"), pad=1))  # dvae.py:145-147
    x = conv1d_cl(x, g("decoder.conv_in.2.weight"), g("decoder.conv_in.2.bias"), pad=1)  # dvae.py:148
    n = 0
    while f"decoder.decoder_block.{n}.weight" in sd:
        x = convnext_block(x, sd, f"decoder.decoder_block.{n}.", "weight", dil=2)  # dvae.py:150-160
        n += 1
    x = conv1d_cl(x, g("decoder.conv_out.weight"), None, pad=0)  # dvae.py:161
    x = conv1d_cl(x, g("out_conv.weight"), None, pad=1)  # dvae.py:239,289-293
    return (x * g("coef").reshape(1, 1, -1)).astype(f32)  # dvae.py:297


def vocos_backbone(sd: dict, mel: np.ndarray) -> np.ndarray:
    """vocos.models.VocosBackbone.forward: embed Conv1d(100->512,k7,p3) -> LayerNorm -> 8 ConvNeXt
    blocks (dilation 1, gamma) -> final LayerNorm.  mel [B,F,100] -> [B,F,512]."""
    g = lambda k: np.asarray(sd[k], dtype=f32)
    x = conv1d_cl(mel, g("backbone.embed.weight"), g("backbone.embed.bias"), pad=3)
    x = layer_norm(x, g("backbone.norm.weight"), g("backbone.norm.bias"), 1e-6)
    n = 0
    while f"backbone.convnext.{n}.gamma" in sd:
        x = convnext_block(x, sd, f"backbone.convnext.{n}.", "gamma", dil=1)
        n += 1
    return layer_norm(x, g("backbone.final_layer_norm.weight"), g("backbone.final_layer_norm.bias"), 1e-6)


def istft_center(spec: np.ndarray, window: np.ndarray, n_fft: int, hop: int, trim: int = None) -> np.ndarray:
    """torch.istft(spec, n_fft, hop, n_fft, window, center=True): per-frame irfft (1/n norm),
    x window, overlap-add, / window-envelope, trim n_fft/2 each side.  spec [B, F, n_fft/2+1] complex.
    `trim` = (n_fft - hop) / 2 gives Vocos' own "same"-padding ISTFT (vocos/spectral_ops.py; ChatTTS configures "center",
    config.py:83-121) -- used by the test that pins this head against transformers' port of it (Xcodec2ISTFTHead)."""
    B, F, _ = spec.shape
    frames = np.fft.irfft(spec.astype(np.complex128), n=n_fft, axis=-1) * window.astype(np.float64)
    total = n_fft + hop * (F - 1)
    y = np.zeros((B, total), dtype=np.float64)
    env = np.zeros(total, dtype=np.float64)
    w2 = window.astype(np.float64) ** 2
    for f in range(F):
        y[:, f * hop: f * hop + n_fft] += frames[:, f]
        env[f * hop: f * hop + n_fft] += w2
    t = n_fft // 2 if trim is None else int(trim)
    s, e = t, total - t
    return (y[:, s:e] / env[s:e]).astype(f32)


def vocos_head(sd: dict, x: np.ndarray, n_fft: int = 1024, hop: int = 256, trim: int = None) -> np.ndarray:
    """vocos.heads.ISTFTHead.forward (exporter.py:395-404): Linear(512->1026), chunk(mag, phase),
    mag = clip(exp(mag), max=1e2), S = mag (cos p + i sin p), ISTFT."""
    g = lambda k: np.asarray(sd[k], dtype=f32)
    y = x @ g("head.out.weight").T + g("head.out.bias")
    nb = n_fft // 2 + 1
    mag = np.minimum(np.exp(y[..., :nb]), f32(1e2))
    ph = y[..., nb:]
    spec = mag * (np.cos(ph) + 1j * np.sin(ph))
    return istft_center(spec, g("head.istft.window"), n_fft, hop, trim)


def vocos_decode(sd: dict, mel: np.ndarray) -> np.ndarray:
    """vocos.Vocos.decode: head(backbone(mel)).  mel [B,F,100] channels-last -> wav [B, hop*(F-1)]."""
    return vocos_head(sd, vocos_backbone(sd, mel))


def decode_to_wavs(dec_sd: dict, voc_sd: dict, hiddens: list) -> np.ndarray:
    """core.py:513-539 `_decode_to_wavs`: zero-pad the per-row hidden lists to the longest row
    (App. D-6), DVAE decode, Vocos decode -> [B, 256*(2*Tmax-1)] float32 (BEFORE the |x|>1e-5 strip
    of core.py:258-270)."""
    B = len(hiddens)
    Tmax = max(h.shape[0] for h in hiddens)
    batch = np.zeros((B, Tmax, hiddens[0].shape[1]), dtype=f32)
    for b, h in enumerate(hiddens):
        batch[b, : h.shape[0]] = h
    return vocos_decode(voc_sd, dvae_decode(dec_sd, batch))
