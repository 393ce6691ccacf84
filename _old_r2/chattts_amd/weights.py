"""Weight layout of the hot path: the reference's safetensors keys/shapes, a frozen synthetic
recipe that fills them (no real assets are reachable offline), and loaders.

Key names/shapes restate SURVEY.md App. B, i.e.
  * asset/gpt/model.safetensors  -- HF LlamaModel state dict (/root/reference/ChatTTS/model/gpt.py:75-78)
  * asset/Embed.safetensors      -- /root/reference/ChatTTS/model/embed.py:18-35
  * asset/Decoder.safetensors    -- /root/reference/ChatTTS/model/dvae.py:145-161,226,239
  * asset/Vocos.safetensors      -- vocos.Vocos state dict (un-vendored; /root/reference/ChatTTS/core.py:298-317)

The synthetic recipe follows SURVEY.md section 8(d): Llama linears N(0, 0.02^2), RMSNorm weights 1,
embeddings N(0,1), weight-norm direction v ~ N(0,1) with gain g = 4 (pre-temperature logit std ~ 4),
DVAE/Vocos conv+linear N(0, 1/fan_in), LayerNorm (1, 0), layer scale 1/n_layers, coef 1.
It is generated with torch's CPU generator, which is bit-reproducible for a fixed torch build
(checked by `fingerprint`, pinned in tests/golden/).
"""
from __future__ import annotations

import hashlib
import math
import os
from typing import Dict

import torch

from .config import GPT, DVAE, VOCOS, GptDims

StateDict = Dict[str, torch.Tensor]


def _normal(gen: torch.Generator, shape, std: float) -> torch.Tensor:
    return torch.empty(shape, dtype=torch.float32).normal_(0.0, std, generator=gen)


def synthetic_gpt(seed: int = 1234, n_layers: int = GPT.n_layers, std: float = 0.02) -> StateDict:
    g = torch.Generator().manual_seed(seed)
    H, I = GPT.hidden, GPT.inter
    sd: StateDict = {}
    for i in range(n_layers):
        p = f"layers.{i}."
        sd[p + "self_attn.q_proj.weight"] = _normal(g, (H, H), std)
        sd[p + "self_attn.k_proj.weight"] = _normal(g, (H, H), std)
        sd[p + "self_attn.v_proj.weight"] = _normal(g, (H, H), std)
        sd[p + "self_attn.o_proj.weight"] = _normal(g, (H, H), std)
        sd[p + "mlp.gate_proj.weight"] = _normal(g, (I, H), std)
        sd[p + "mlp.up_proj.weight"] = _normal(g, (I, H), std)
        sd[p + "mlp.down_proj.weight"] = _normal(g, (H, I), std)
        # RMSNorm gains: 1 + small jitter so that a kernel which forgets the gain is caught.
        sd[p + "input_layernorm.weight"] = 1.0 + _normal(g, (H,), 0.05)
        sd[p + "post_attention_layernorm.weight"] = 1.0 + _normal(g, (H,), 0.05)
    sd["norm.weight"] = 1.0 + _normal(g, (H,), 0.05)
    return sd


def synthetic_embed(seed: int = 1235, head_gain: float = 4.0) -> StateDict:
    g = torch.Generator().manual_seed(seed)
    H = GPT.hidden
    sd: StateDict = {}
    for k in range(GPT.n_vq):
        sd[f"emb_code.{k}.weight"] = _normal(g, (GPT.n_audio, H), 1.0)
    sd["emb_text.weight"] = _normal(g, (GPT.n_text, H), 1.0)
    sd["head_text.parametrizations.weight.original0"] = torch.full((GPT.n_text, 1), head_gain)
    sd["head_text.parametrizations.weight.original1"] = _normal(g, (GPT.n_text, H), 1.0)
    for k in range(GPT.n_vq):
        # per-row gains jittered around head_gain so the weight-norm fold is exercised
        sd[f"head_code.{k}.parametrizations.weight.original0"] = head_gain * (
            1.0 + _normal(g, (GPT.n_audio, 1), 0.1)
        )
        sd[f"head_code.{k}.parametrizations.weight.original1"] = _normal(g, (GPT.n_audio, H), 1.0)
    return sd


def synthetic_decoder(seed: int = 1236) -> StateDict:
    """Decoder-`DVAE` (`DVAE(decoder_config=config.decoder, dim=384)`, core.py:366-375)."""
    g = torch.Generator().manual_seed(seed)
    D = DVAE
    sd: StateDict = {"coef": 0.5 + torch.rand((1, D.n_mels, 1), generator=g)}
    sd["decoder.conv_in.0.weight"] = _normal(g, (D.bn_dim, D.idim, 3), (1.0 / (D.idim * 3)) ** 0.5)
    sd["decoder.conv_in.0.bias"] = _normal(g, (D.bn_dim,), 0.1)
    sd["decoder.conv_in.2.weight"] = _normal(g, (D.hidden, D.bn_dim, 3), (1.0 / (D.bn_dim * 3)) ** 0.5)
    sd["decoder.conv_in.2.bias"] = _normal(g, (D.hidden,), 0.1)
    for i in range(D.n_layers):
        p = f"decoder.decoder_block.{i}."
        sd[p + "weight"] = torch.full((D.hidden,), 1.0 / D.n_layers) * (1.0 + _normal(g, (D.hidden,), 0.1))
        sd[p + "dwconv.weight"] = _normal(g, (D.hidden, 1, D.kernel), (1.0 / D.kernel) ** 0.5)
        sd[p + "dwconv.bias"] = _normal(g, (D.hidden,), 0.1)
        sd[p + "norm.weight"] = 1.0 + _normal(g, (D.hidden,), 0.05)
        sd[p + "norm.bias"] = _normal(g, (D.hidden,), 0.05)
        sd[p + "pwconv1.weight"] = _normal(g, (D.hidden * 4, D.hidden), (1.0 / D.hidden) ** 0.5)
        sd[p + "pwconv1.bias"] = _normal(g, (D.hidden * 4,), 0.1)
        sd[p + "pwconv2.weight"] = _normal(g, (D.hidden, D.hidden * 4), (1.0 / (D.hidden * 4)) ** 0.5)
        sd[p + "pwconv2.bias"] = _normal(g, (D.hidden,), 0.1)
    sd["decoder.conv_out.weight"] = _normal(g, (D.odim, D.hidden, 1), (1.0 / D.hidden) ** 0.5)
    sd["out_conv.weight"] = _normal(g, (D.n_mels, D.odim, 3), (1.0 / (D.odim * 3)) ** 0.5)
    return sd


def synthetic_vocos(seed: int = 1237) -> StateDict:
    """vocos.Vocos(backbone=VocosBackbone(100,512,1536,8), head=ISTFTHead(512,1024,256,'center'))."""
    g = torch.Generator().manual_seed(seed)
    V = VOCOS
    sd: StateDict = {}
    sd["backbone.embed.weight"] = _normal(g, (V.dim, V.n_mels, 7), (1.0 / (V.n_mels * 7)) ** 0.5)
    sd["backbone.embed.bias"] = _normal(g, (V.dim,), 0.1)
    sd["backbone.norm.weight"] = 1.0 + _normal(g, (V.dim,), 0.05)
    sd["backbone.norm.bias"] = _normal(g, (V.dim,), 0.05)
    for i in range(V.n_layers):
        p = f"backbone.convnext.{i}."
        sd[p + "gamma"] = torch.full((V.dim,), 1.0 / V.n_layers) * (1.0 + _normal(g, (V.dim,), 0.1))
        sd[p + "dwconv.weight"] = _normal(g, (V.dim, 1, 7), (1.0 / 7) ** 0.5)
        sd[p + "dwconv.bias"] = _normal(g, (V.dim,), 0.1)
        sd[p + "norm.weight"] = 1.0 + _normal(g, (V.dim,), 0.05)
        sd[p + "norm.bias"] = _normal(g, (V.dim,), 0.05)
        sd[p + "pwconv1.weight"] = _normal(g, (V.inter, V.dim), (1.0 / V.dim) ** 0.5)
        sd[p + "pwconv1.bias"] = _normal(g, (V.inter,), 0.1)
        sd[p + "pwconv2.weight"] = _normal(g, (V.dim, V.inter), (1.0 / V.inter) ** 0.5)
        sd[p + "pwconv2.bias"] = _normal(g, (V.dim,), 0.1)
    sd["backbone.final_layer_norm.weight"] = 1.0 + _normal(g, (V.dim,), 0.05)
    sd["backbone.final_layer_norm.bias"] = _normal(g, (V.dim,), 0.05)
    # head: log-magnitude bias chosen so the waveform rms lands near 0.05-0.1 (speech-like level)
    sd["head.out.weight"] = _normal(g, (V.n_fft + 2, V.dim), 0.5 * (1.0 / V.dim) ** 0.5)
    bias = _normal(g, (V.n_fft + 2,), 0.1)
    bias[: V.n_fft // 2 + 1] += 0.5
    sd["head.out.bias"] = bias
    sd["head.istft.window"] = torch.hann_window(V.n_fft)
    return sd


def _trunk(g: torch.Generator, sd: StateDict, prefix: str, idim: int, odim: int, hidden: int, bn: int, n_layers: int) -> None:
    """`DVAEDecoder(idim, odim, n_layer, bn_dim, hidden)` (dvae.py:131-161), same init style as `synthetic_decoder`."""
    sd[prefix + "conv_in.0.weight"] = _normal(g, (bn, idim, 3), (1.0 / (idim * 3)) ** 0.5)
    sd[prefix + "conv_in.0.bias"] = _normal(g, (bn,), 0.1)
    sd[prefix + "conv_in.2.weight"] = _normal(g, (hidden, bn, 3), (1.0 / (bn * 3)) ** 0.5)
    sd[prefix + "conv_in.2.bias"] = _normal(g, (hidden,), 0.1)
    for i in range(n_layers):
        p = f"{prefix}decoder_block.{i}."
        sd[p + "weight"] = torch.full((hidden,), 1.0 / n_layers) * (1.0 + _normal(g, (hidden,), 0.1))
        sd[p + "dwconv.weight"] = _normal(g, (hidden, 1, 7), (1.0 / 7) ** 0.5)
        sd[p + "dwconv.bias"] = _normal(g, (hidden,), 0.1)
        sd[p + "norm.weight"] = 1.0 + _normal(g, (hidden,), 0.05)
        sd[p + "norm.bias"] = _normal(g, (hidden,), 0.05)
        sd[p + "pwconv1.weight"] = _normal(g, (hidden * 4, hidden), (1.0 / hidden) ** 0.5)
        sd[p + "pwconv1.bias"] = _normal(g, (hidden * 4,), 0.1)
        sd[p + "pwconv2.weight"] = _normal(g, (hidden, hidden * 4), (1.0 / (hidden * 4)) ** 0.5)
        sd[p + "pwconv2.bias"] = _normal(g, (hidden,), 0.1)
    sd[prefix + "conv_out.weight"] = _normal(g, (odim, hidden, 1), (1.0 / hidden) ** 0.5)


def mel_filterbank(n_freqs: int = 513, n_mels: int = 100, sample_rate: int = 24000) -> torch.Tensor:
    """torchaudio.functional.melscale_fbanks(n_freqs, 0, sr/2, n_mels, sr, norm=None, mel_scale="htk") -> [n_freqs, n_mels]:
    the `mel_scale.fb` buffer of the MelSpectrogram inside DVAE.preprocessor_mel (dvae.py:189-196)."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    to_mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)
    m_pts = torch.linspace(to_mel(0.0), to_mel(sample_rate / 2.0), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


def synthetic_dvae(seed: int = 1238, n_layers: int = 12) -> StateDict:
    """The full DVAE of `asset/DVAE.safetensors` (dvae.py:209-244 with config.py:31-47: encoder 512->1024 and decoder
    512->512 trunks of hidden 256 / bn 128, GroupedResidualFSQ(1024, levels 5^4, G=2, R=2), mel front end buffers).
    Quantiser keys follow vector_quantize_pytorch's module tree (`rvqs.{g}.project_in/out`)."""
    g = torch.Generator().manual_seed(seed)
    sd: StateDict = {"coef": 0.5 + torch.rand((1, 100, 1), generator=g)}
    sd["downsample_conv.0.weight"] = _normal(g, (512, 100, 3), (1.0 / 300) ** 0.5)
    sd["downsample_conv.0.bias"] = _normal(g, (512,), 0.1)
    sd["downsample_conv.2.weight"] = _normal(g, (512, 512, 4), (1.0 / 2048) ** 0.5)
    sd["downsample_conv.2.bias"] = _normal(g, (512,), 0.1)
    _trunk(g, sd, "encoder.", 512, 1024, 256, 128, n_layers)
    _trunk(g, sd, "decoder.", 512, 512, 256, 128, n_layers)
    sd["out_conv.weight"] = _normal(g, (100, 512, 3), (1.0 / 1536) ** 0.5)
    for grp in range(2):
        p = f"vq_layer.quantizer.rvqs.{grp}."
        sd[p + "project_in.weight"] = _normal(g, (4, 512), 2.0 * (1.0 / 512) ** 0.5)   # z std ~ 2-3: all 5 levels get used
        sd[p + "project_in.bias"] = _normal(g, (4,), 0.2)
        sd[p + "project_out.weight"] = _normal(g, (512, 4), 0.5)
        sd[p + "project_out.bias"] = _normal(g, (512,), 0.1)
    sd["preprocessor_mel.mel_spec.spectrogram.window"] = torch.hann_window(1024)
    sd["preprocessor_mel.mel_spec.mel_scale.fb"] = mel_filterbank()
    return sd


def synthetic_all(n_layers: int = GPT.n_layers) -> Dict[str, StateDict]:
    return {
        "gpt": synthetic_gpt(n_layers=n_layers),
        "embed": synthetic_embed(),
        "decoder": synthetic_decoder(),
        "vocos": synthetic_vocos(),
    }


def fingerprint(sd: StateDict) -> str:
    """sha256 over (key, raw bytes) in sorted-key order -- pins the synthetic recipe."""
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].contiguous().numpy().tobytes())
    return h.hexdigest()


# --------------------------------------------------------------------------------------
# safetensors I/O in the reference's asset layout (config.py:4-11)
# --------------------------------------------------------------------------------------
ASSET_FILES = {
    "gpt": os.path.join("gpt", "model.safetensors"),
    "embed": "Embed.safetensors",
    "decoder": "Decoder.safetensors",
    "vocos": "Vocos.safetensors",
}
OPTIONAL_ASSET_FILES = {"dvae": "DVAE.safetensors"}   # full DVAE (encoder + GFSQ): speaker prompts and use_decoder=False only


def save_assets(root: str, sds: Dict[str, StateDict]) -> None:
    from safetensors.torch import save_file

    for name, rel in {**ASSET_FILES, **OPTIONAL_ASSET_FILES}.items():
        if name not in sds:
            continue
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        save_file({k: v.contiguous() for k, v in sds[name].items()}, path)


def load_assets(root: str) -> Dict[str, StateDict]:
    """Mirror of `load_safetensors` (/root/reference/ChatTTS/utils/io.py:19-25) over the four hot-path files."""
    from safetensors import safe_open

    out: Dict[str, StateDict] = {}
    for name, rel in {**ASSET_FILES, **OPTIONAL_ASSET_FILES}.items():
        if name in OPTIONAL_ASSET_FILES and not os.path.exists(os.path.join(root, rel)):
            continue
        sd: StateDict = {}
        with safe_open(os.path.join(root, rel), framework="pt") as f:
            for k in f.keys():
                kk = k[len("model."):] if (name == "gpt" and k.startswith("model.")) else k
                sd[kk] = f.get_tensor(k)
        sd.pop("embed_tokens.weight", None)  # gpt.py:78  (`del self.gpt.embed_tokens`)
        out[name] = sd
    return out


def gpt_layer_count(sd: StateDict) -> int:
    n = 0
    while f"layers.{n}.input_layernorm.weight" in sd:
        n += 1
    return n


def fold_weight_norm(g: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """W = g * v / ||v||_2 per output row (torch weight_norm dim=0; embed.py:23-35).

    The reference recomputes this every step under `P.cached()` (gpt.py:438); it is constant,
    so the engine folds it once at load.  Same op order as torch._weight_norm: v * (g / norm).
    """
    norm = torch.linalg.vector_norm(v, ord=2, dim=1, keepdim=True)
    return v * (g / norm)
