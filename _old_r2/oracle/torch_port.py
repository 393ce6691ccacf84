"""torch / MKL float32 restatement of the hot path, for bench.py's `cpu_baseline` leg ONLY.

TEST INFRASTRUCTURE (see oracle/__init__.py): nothing under chattts_amd/ imports this.  It travels to the GPU box
(unlike oracle/ref_harness.py it never touches /root/reference) and exists because BASELINE.md section 3 asks for the
same-box CPU number of the reference's OWN stack -- HF `LlamaModel` + `DynamicCache` under torch/MKL with every host
thread -- rather than of the numpy oracle.  Each function cites the reference lines it follows:

  * generate():   GPT.generate, /root/reference/ChatTTS/model/gpt.py:316-618 -- LlamaModel.forward with a DynamicCache
                  (:419-427), weight-normed heads (:438-454, embed.py:27-35), temperature (:487), repetition penalty
                  + transformers' TopP / TopK warpers in the order of core.py:649 (processors.py:18-58), EOS mask
                  (:494-495), softmax + torch.multinomial on a re-seeded CPU generator (:497-508), finish bookkeeping
                  (:512-577).  The same transformers classes the reference imports are used.
  * dvae_decode(): DVAE decode branch, dvae.py:276-297 / :163-172 / :46-66, with torch conv1d / layer_norm / linear.
  * vocos_decode(): vocos.Vocos.decode restated with torch ops (the `vocos` package is not installed anywhere here;
                  core.py:505-510, exporter.py:392-405).
Pinned: tests/test_oracle_vs_golden.py::test_torch_port_matches_goldens checks generate() against the reference's golden
token ids and the codec functions against the reference-class mel / wav goldens.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch


def build_llama(gpt_sd: dict):
    """LlamaModel(LlamaConfig(**Config.gpt)) of gpt.py:75-78 with the given state dict (embed_tokens deleted)."""
    from transformers import LlamaConfig, LlamaModel

    n_layers = 0
    while f"layers.{n_layers}.input_layernorm.weight" in gpt_sd:
        n_layers += 1
    cfg = LlamaConfig(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=n_layers,
                      use_cache=False, max_position_embeddings=4096, spk_emb_dim=192, spk_KL=False, num_audio_tokens=626,
                      num_text_tokens=21178, num_vq=4, vocab_size=32)   # config.py:50-63 (vocab_size: embed_tokens is deleted)
    with torch.device("cpu"):
        m = LlamaModel(cfg)
    del m.embed_tokens
    missing, unexpected = m.load_state_dict({k: v.float() for k, v in gpt_sd.items()}, strict=False)
    assert not [k for k in missing if "embed_tokens" not in k] and not unexpected, (missing, unexpected)
    return m.eval()


def fold_heads(embed_sd: dict) -> torch.Tensor:
    ws = []
    for k in range(4):
        g = embed_sd[f"head_code.{k}.parametrizations.weight.original0"].float()
        v = embed_sd[f"head_code.{k}.parametrizations.weight.original1"].float()
        ws.append(v * (g / v.norm(dim=1, keepdim=True)))
    return torch.cat(ws, 0)     # [2504, 768]


class RepPenalty:
    """Windowed repetition penalty (processors.py:6-35): a token seen n times in the last `past_window` steps has its
    score multiplied (negative scores) or divided (positive scores) by penalty**n.  As in the reference, the count matrix
    is zeroed from ROW `max_input_ids` on (processors.py:25-28 narrows dim 0), which never triggers at B*num_vq <= 625."""

    def __init__(self, penalty, max_input_ids=625, past_window=16):
        self.penalty, self.max_input_ids, self.past_window = penalty, max_input_ids, past_window

    def __call__(self, hist, scores):
        recent = hist[:, -self.past_window:]
        counts = torch.zeros(scores.shape, dtype=torch.int64)
        counts.scatter_add_(1, recent, torch.ones_like(recent))
        counts[self.max_input_ids:] = 0
        alpha = torch.pow(self.penalty, counts)
        return torch.where(scores < 0, scores * alpha, scores / alpha)


@torch.inference_mode()
def generate(llama, embed_sd: dict, emb: torch.Tensor, input_ids: torch.Tensor, attention_mask: torch.Tensor, *, temperature,
             top_P=0.7, top_K=20, repetition_penalty=1.05, max_new_token=32, min_new_token=0, manual_seed: Optional[int] = 42,
             eos=625, deadline: Optional[float] = None):
    """-> (ids [B, n, 4] int64 generated tokens, hiddens [B, n, 768], end_idx [B]) after `max_new_token` steps, when every row
    has finished, or when time.perf_counter() passes `deadline` (bench.py's bounded CPU sample; n = steps actually run)."""
    import time
    from transformers import DynamicCache
    from transformers.generation import TopKLogitsWarper, TopPLogitsWarper

    B, T, nvq = input_ids.shape
    heads = fold_heads(embed_sd)
    emb_code = [embed_sd[f"emb_code.{k}.weight"].float() for k in range(nvq)]
    procs = []
    if repetition_penalty is not None and repetition_penalty != 1:
        procs.append(RepPenalty(repetition_penalty))
    if top_P is not None:
        procs.append(TopPLogitsWarper(top_P, min_tokens_to_keep=3))
    if top_K is not None:
        procs.append(TopKLogitsWarper(top_K, min_tokens_to_keep=3))
    temp = torch.tensor(temperature, dtype=torch.float32).unsqueeze(0).expand(B, -1).contiguous().view(-1, 1)
    mask = torch.ones((B, T + max_new_token), dtype=torch.bool)
    mask[:, :T] = attention_mask.bool()
    buf = torch.zeros((B, T + max_new_token, nvq), dtype=torch.int64)
    buf[:, :T] = input_ids
    cache = DynamicCache()
    finish = torch.zeros(B, dtype=torch.bool)
    end_idx = torch.zeros(B, dtype=torch.int64)
    gen = torch.Generator()
    hiddens = []
    step_s = generate.step_seconds = []     # wall seconds of each step of the LAST call (step 0 = prefill), for bench.py
    x = emb.float()
    for i in range(max_new_token):
        t_step = time.perf_counter()
        if i > 0:
            last = buf[:, T + i - 1]
            x = sum(emb_code[k][last[:, k]] for k in range(nvq)).unsqueeze(1)          # gpt.py:403-415
        am = mask[:, : T + i]
        pos = (am.long().cumsum(-1) - 1).masked_fill(am == 0, 1)[:, -x.shape[1]:]      # gpt.py:234-241
        out = llama(inputs_embeds=x, attention_mask=am, position_ids=pos, past_key_values=cache, use_cache=True)
        h = out.last_hidden_state[:, -1].float()
        hiddens.append(h)
        logits = (h @ heads.T).view(B * nvq, -1)                                        # gpt.py:438-464
        hist = buf[:, T: T + i].permute(0, 2, 1).reshape(B * nvq, i)                    # gpt.py:466-475
        logits = logits / temp                                                          # gpt.py:487
        for p in procs:
            logits = p(hist, logits)
        if i < min_new_token:
            logits[:, eos] = -torch.inf
        scores = torch.softmax(logits, dim=-1)
        if manual_seed is None:
            idx = torch.multinomial(scores, 1)
        else:
            idx = torch.multinomial(scores, 1, generator=gen.manual_seed(manual_seed))   # gpt.py:501-508
        idx = idx.view(B, nvq)
        finish |= (idx == eos).any(1)
        buf[:, T + i] = idx
        end_idx += (~finish).long()
        step_s.append(time.perf_counter() - t_step)
        if bool(finish.all()) or (deadline is not None and time.perf_counter() > deadline):
            break
    return buf[:, T: T + len(hiddens)], torch.stack(hiddens, 1), end_idx


@torch.inference_mode()
def dvae_decode(sd: dict, hid: torch.Tensor) -> torch.Tensor:
    """hid [B, T, 768] -> mel [B, 100, 2T] (reference layout).  dvae.py:276-297."""
    F = torch.nn.functional
    B, T, _ = hid.shape
    x = hid.float().reshape(B, 2 * T, 384).transpose(1, 2)                 # dvae.py:281-287 in channels-first terms
    x = F.gelu(F.conv1d(x, sd["decoder.conv_in.0.weight"], sd["decoder.conv_in.0.bias"], padding=1))
    x = F.conv1d(x, sd["decoder.conv_in.2.weight"], sd["decoder.conv_in.2.bias"], padding=1)
    i = 0
    while f"decoder.decoder_block.{i}.weight" in sd:
        p = f"decoder.decoder_block.{i}."
        y = F.conv1d(x, sd[p + "dwconv.weight"], sd[p + "dwconv.bias"], padding=6, dilation=2, groups=x.shape[1]).transpose(1, 2)
        y = F.layer_norm(y, (y.shape[-1],), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)
        y = F.gelu(F.linear(y, sd[p + "pwconv1.weight"], sd[p + "pwconv1.bias"]))
        y = F.linear(y, sd[p + "pwconv2.weight"], sd[p + "pwconv2.bias"]) * sd[p + "weight"]
        x = x + y.transpose(1, 2)
        i += 1
    x = F.conv1d(x, sd["decoder.conv_out.weight"])
    x = F.conv1d(x, sd["out_conv.weight"], padding=1)
    return x * sd["coef"].reshape(1, -1, 1)


@torch.inference_mode()
def vocos_decode(sd: dict, mel_bcf: torch.Tensor) -> torch.Tensor:
    """Vocos.decode restated with torch ops.  mel [B,100,F] (reference layout) -> wav [B, 256(F-1)]."""
    F = torch.nn.functional
    x = F.conv1d(mel_bcf, sd["backbone.embed.weight"], sd["backbone.embed.bias"], padding=3)
    x = F.layer_norm(x.transpose(1, 2), (512,), sd["backbone.norm.weight"], sd["backbone.norm.bias"], 1e-6).transpose(1, 2)
    i = 0
    while f"backbone.convnext.{i}.gamma" in sd:
        p = f"backbone.convnext.{i}."
        r = x
        y = F.conv1d(x, sd[p + "dwconv.weight"], sd[p + "dwconv.bias"], padding=3, groups=512).transpose(1, 2)
        y = F.layer_norm(y, (512,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)
        y = F.gelu(F.linear(y, sd[p + "pwconv1.weight"], sd[p + "pwconv1.bias"]))
        y = F.linear(y, sd[p + "pwconv2.weight"], sd[p + "pwconv2.bias"]) * sd[p + "gamma"]
        x = r + y.transpose(1, 2)
        i += 1
    x = F.layer_norm(x.transpose(1, 2), (512,), sd["backbone.final_layer_norm.weight"], sd["backbone.final_layer_norm.bias"], 1e-6)
    y = F.linear(x, sd["head.out.weight"], sd["head.out.bias"]).transpose(1, 2)
    mag, p = y.chunk(2, dim=1)
    mag = torch.clip(torch.exp(mag), max=1e2)
    S = mag * (torch.cos(p) + 1j * torch.sin(p))
    return torch.istft(S, 1024, 256, 1024, sd["head.istft.window"], center=True)
