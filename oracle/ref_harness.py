"""Runs the REAL reference code (imported from /root/reference) on CPU with the synthetic weights.

BUILD-CONTAINER ONLY: /root/reference does not exist on the GPU box, so nothing in `-m gpu`
tests, smoke() or bench.py imports this module.  It exists to (a) validate the numpy oracle and
(b) generate tests/golden/*.npz (oracle/make_goldens.py).  The reference is imported, never copied.

Import recipe: `import ChatTTS` fails (vocos, pybase16384, torchaudio,
vector_quantize_pytorch, numba are not installed), so the package object is stubbed and the
needed submodules are imported individually; none of the stubbed deps is touched on the decode path.
Shim: transformers 5.15 `DynamicCache.get_max_cache_shape()` returns -1 where the reference
(written against >=4.41) expects None (gpt.py:190-200) -- patched harness-side only.
"""
from __future__ import annotations

import importlib
import sys
import types
from dataclasses import asdict

import numpy as np
import torch

REF = "/root/reference"


def available() -> bool:
    import os
    return os.path.isdir(os.path.join(REF, "ChatTTS"))


_mods = {}


def ref_modules():
    if _mods:
        return _mods
    import transformers  # noqa: F401  -- BEFORE stubbing torchaudio (its availability probe needs a real spec)
    from transformers import LlamaModel, LlamaConfig  # noqa: F401
    from transformers.generation import TopKLogitsWarper, TopPLogitsWarper  # noqa: F401
    for name, path in [("ChatTTS", f"{REF}/ChatTTS"), ("ChatTTS.model", f"{REF}/ChatTTS/model")]:
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [path]
            sys.modules[name] = m
    for n in ["pybase16384", "torchaudio", "torchaudio.transforms", "vector_quantize_pytorch"]:
        if n not in sys.modules:
            sys.modules[n] = types.ModuleType(n)
    sys.modules["vector_quantize_pytorch"].GroupedResidualFSQ = None
    sys.modules["torchaudio"].transforms = sys.modules["torchaudio.transforms"]
    for short in ["gpt", "embed", "processors", "dvae"]:
        _mods[short] = importlib.import_module(f"ChatTTS.model.{short}")
    _mods["config"] = importlib.import_module("ChatTTS.config.config")
    from transformers import DynamicCache

    if not getattr(DynamicCache, "_ctts_shim", False):
        orig = DynamicCache.get_max_cache_shape

        def shim(self, *a, **k):
            v = orig(self, *a, **k)
            return None if (v is None or v < 0) else v

        DynamicCache.get_max_cache_shape = shim
        DynamicCache._ctts_shim = True
    return _mods


def build_gpt(sds: dict):
    """Reference `Embed` + `GPT` with the synthetic state dicts loaded (core.py:336-358 minus disk I/O)."""
    m = ref_modules()
    from transformers import LlamaModel

    cfg = m["config"].Config()
    embed = m["embed"].Embed(cfg.embed.hidden_size, cfg.embed.num_audio_tokens, cfg.embed.num_text_tokens, cfg.embed.num_vq)
    embed.load_state_dict(sds["embed"])
    embed.eval()
    gcfg = asdict(cfg.gpt)
    n_layers = 0
    while f"layers.{n_layers}.input_layernorm.weight" in sds["gpt"]:
        n_layers += 1
    gcfg["num_hidden_layers"] = n_layers
    gpt = m["gpt"].GPT(gpt_config=gcfg, embed=embed)
    gpt.gpt = LlamaModel(gpt.llama_config)  # gpt.py:75 builds it from asset/gpt/config.json
    del gpt.gpt.embed_tokens  # gpt.py:78
    missing, unexpected = gpt.gpt.load_state_dict(sds["gpt"], strict=False)
    assert not [k for k in missing if "embed_tokens" not in k] and not unexpected, (missing, unexpected)
    gpt.gpt.eval()
    gpt.prepare(compile=False)
    return embed, gpt


def run_generate(embed, gpt, input_ids, attention_mask, text_mask, *, temperature, top_P, top_K,
                 repetition_penalty, max_new_token, min_new_token, manual_seed, extra_processors=(),
                 capture_logits=False, infer_text=False, eos_token=None, stream=False, stream_batch=24, yields=None):
    """`Chat._infer_code` from `gen_logits` on (core.py:580-658), minus tokenizer/speaker.  With `stream=True` every yield of
    gpt.py:579-589 is appended to the list `yields` as (ids per row, hidden rows per row) copies; the return value is the last one."""
    m = ref_modules()
    ids = torch.from_numpy(input_ids)
    am = torch.from_numpy(attention_mask)
    tm = torch.from_numpy(text_mask)
    num_code = (gpt.num_audio_tokens - 1) if not infer_text else gpt.num_text_tokens  # core.py:580 / :682-687
    if eos_token is None:
        eos_token = num_code
    warpers, procs = m["processors"].gen_logits(num_code=num_code, top_P=top_P, top_K=top_K,
                                                repetition_penalty=repetition_penalty)
    emb = embed(ids, tm)
    cap = []
    plist = [*procs, *warpers, *extra_processors]
    if capture_logits:
        def spy(tok, logits):  # first in the chain: sees logits / temperature
            cap.append(logits.clone().numpy())
            return logits
        plist = [spy] + plist
    out = None
    for out in gpt.generate(
        emb, ids, temperature=torch.tensor(temperature), eos_token=eos_token, attention_mask=am,
        max_new_token=max_new_token, min_new_token=min_new_token, logits_processors=tuple(plist),
        infer_text=infer_text, return_hidden=True, stream=stream, show_tqdm=False, ensure_non_empty=True,
        stream_batch=stream_batch, manual_seed=manual_seed,
    ):
        if yields is not None:
            yields.append(([r.clone().numpy() for r in out.ids], [int(h.shape[0]) for h in out.hiddens]))
    if out is None:
        return None, emb.numpy(), cap
    return out, emb.numpy(), cap


def build_decoder(sds: dict):
    m = ref_modules()
    cfg = m["config"].Config()
    dec = m["dvae"].DVAE(decoder_config=asdict(cfg.decoder), dim=cfg.decoder.idim)
    dec.load_state_dict(sds["decoder"])
    return dec.eval()


def torch_vocos_decode(sd: dict, mel_bcf: torch.Tensor) -> torch.Tensor:
    """Vocos.decode restated with torch ops (the `vocos` package is not installed): the pin for
    oracle/codec_np.vocos_decode.  mel [B,100,F] (reference layout) -> wav [B, 256(F-1)].  Lives in oracle/torch_port.py
    (which also travels to the GPU box for bench.py's CPU baseline)."""
    from . import torch_port
    return torch_port.vocos_decode(sd, mel_bcf)
