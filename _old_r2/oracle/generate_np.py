"""numpy restatement of `GPT.generate` (code mode, infer_text=False) --
/root/reference/ChatTTS/model/gpt.py:316-618, SURVEY.md App. A steps 1-13.

TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional

import numpy as np

from . import llama_np, sampling_np

f32 = np.float32


def fold_head_text(embed_sd: dict) -> np.ndarray:
    """embed.py:23-26: the weight-normed text head, folded: [n_text, H]."""
    g = np.asarray(embed_sd["head_text.parametrizations.weight.original0"], dtype=f32)
    v = np.asarray(embed_sd["head_text.parametrizations.weight.original1"], dtype=f32)
    nrm = np.sqrt(np.sum(v.astype(np.float64) ** 2, axis=1, keepdims=True)).astype(f32)
    return v * (g / nrm)


def fold_heads(embed_sd: dict, n_vq: int = 4) -> np.ndarray:
    """embed.py:27-35: W_k = g_k * v_k / ||v_k||_2 (row-wise), stacked to [n_vq*V, H] so that
    logits.reshape(B, n_vq, V) is the (b n) c layout of gpt.py:459-464."""
    ws = []
    for k in range(n_vq):
        g = np.asarray(embed_sd[f"head_code.{k}.parametrizations.weight.original0"], dtype=f32)
        v = np.asarray(embed_sd[f"head_code.{k}.parametrizations.weight.original1"], dtype=f32)
        nrm = np.sqrt(np.sum(v.astype(np.float64) ** 2, axis=1, keepdims=True)).astype(f32)
        ws.append(v * (g / nrm))
    return np.concatenate(ws, axis=0)


def embed_codes(embed_sd: dict, ids: np.ndarray, n_vq: int = 4) -> np.ndarray:
    """gpt.py:409-413: sum_k emb_code[k][ids[..., k]] (stack(...,3).sum(3): k-ordered f32 adds)."""
    out = None
    for k in range(n_vq):
        e = np.asarray(embed_sd[f"emb_code.{k}.weight"], dtype=f32)[ids[..., k]]
        out = e if out is None else out + e
    return out.astype(f32)


def embed_prompt(embed_sd: dict, input_ids: np.ndarray, text_mask: np.ndarray) -> np.ndarray:
    """embed.py:52-79: text positions use emb_text[id0]; code positions sum the 4 code tables."""
    et = np.asarray(embed_sd["emb_text.weight"], dtype=f32)
    emb = np.where(text_mask[..., None], et[input_ids[..., 0]], embed_codes(embed_sd, np.minimum(input_ids, 625)))
    return emb.astype(f32)


@dataclass
class GenResult:
    ids: List[np.ndarray]       # per row [T_b, n_vq] int64            (gpt.py:297-299)
    hiddens: List[np.ndarray]   # per row [T_b, H] f32                 (gpt.py:303-307)
    steps: int = 0
    logits: list = field(default_factory=list)  # optional per-step pre-processor logits [B*n_vq, V]


def generate(llama: llama_np.LlamaWeights, embed_sd: dict, heads: np.ndarray, emb: np.ndarray,
             input_ids: np.ndarray, attention_mask: np.ndarray, *, temperature: np.ndarray,
             draw_q: Callable[[int], np.ndarray], top_p: Optional[float] = 0.7, top_k: Optional[int] = 20,
             pow_table: Optional[np.ndarray] = None, max_new_token: int = 2048, min_new_token: int = 0,
             eos: int = 625, stop_at: Optional[np.ndarray] = None, row_offset: int = 0,
             keep_logits: bool = False, teacher_ids: Optional[np.ndarray] = None, infer_text: bool = False) -> GenResult:
    """Autoregressive loop.  `draw_q(step)` returns the Exp(1) tensor [B*n_vq, V] the reference's
    CPU generator would produce at that step (gpt.py:501-508; re-seeded every step when manual_seed
    is given, App. D-1).  `stop_at[b]` is the bench harness's length-forcing hook:
    EOS masked while gen < stop_at[b], forced once gen >= stop_at[b]; -1 disables.
    `teacher_ids` [B, n, n_vq] forces the sampled tokens (teacher forcing for logit-level checks).
    `infer_text=True` is the refine-text mode (gpt.py:406-407,439-440,477-485,519-525): `heads` is the folded
    text head [n_text, H], one sampling row per batch row, `temperature` has one entry, the sampled token is
    replicated over the n_vq slots and the next step embeds it with emb_text.
    """
    B, T, n_vq = input_ids.shape
    nr = 1 if infer_text else n_vq        # sampling rows per batch row
    V = heads.shape[0] // nr
    et = np.asarray(embed_sd["emb_text.weight"], dtype=f32) if infer_text else None
    kv_start = (T - attention_mask.astype(np.int64).sum(1)).astype(np.int64)
    assert all((attention_mask[b, : kv_start[b]] == 0).all() and (attention_mask[b, kv_start[b]:] != 0).all()
               for b in range(B)), "oracle supports left padding only (tokenizer.py:73-110)"
    cache = llama_np.KVCache(llama.n_layers, B, llama.n_heads, T + max_new_token, llama.head_dim)
    ids_buf = np.zeros((B, T + max_new_token, n_vq), dtype=np.int64)  # gpt.py:372-379
    ids_buf[:, :T] = input_ids
    finish = np.zeros(B, dtype=bool)
    end_idx = np.zeros(B, dtype=np.int64)
    temp_rows = np.broadcast_to(temperature.astype(f32)[None, :], (B, nr)).reshape(-1)  # gpt.py:350-355
    hiddens = []
    res = GenResult([], [])
    x = emb.astype(f32)
    for i in range(max_new_token):
        if i > 0:
            last = ids_buf[:, T + i - 1: T + i]
            x = et[last[..., 0]] if infer_text else embed_codes(embed_sd, last)  # gpt.py:403-415
        h = llama_np.forward(llama, x, cache, kv_start)[:, -1]  # gpt.py:419-436
        hiddens.append(h)
        logits = (h @ heads.T).reshape(B * nr, V).astype(f32)  # gpt.py:438-464
        if keep_logits:
            res.logits.append(logits.copy())
        history = ids_buf[:, T: T + i, :nr].transpose(0, 2, 1).reshape(B * nr, i)  # gpt.py:466-485
        if teacher_ids is not None:
            idx = teacher_ids[:, i].reshape(-1)
        else:
            mask_rows = np.full(B * nr, i < min_new_token)
            force_rows = None
            if stop_at is not None:  # harness-side extra "logits processor" (same hook on the HIP side)
                sa = np.repeat(stop_at, nr)
                mask_rows = mask_rows | ((sa >= 0) & (i < sa))
                force_rows = (sa >= 0) & (i >= sa)
            idx = sampling_np.sample_step(
                logits, history, draw_q(i), temperature=temp_rows, top_p=top_p, top_k=top_k,
                pow_table=pow_table, max_input_ids=V - 1, mask_eos=mask_rows, force_eos=force_rows,
                eos=eos, row_offset=row_offset)
        idx = idx.reshape(B, nr)
        finish |= (idx == eos).any(1)  # gpt.py:512-515 / :519-521
        if infer_text:
            idx = np.repeat(idx, n_vq, axis=1)  # gpt.py:522-525: expand over the n_vq slots
        ids_buf[:, T + i] = idx  # gpt.py:518
        if i == 0 and finish.any():
            # gpt.py:527-570: seeded run with a step-0 EOS yields nothing
            res.steps = 1
            return res
        end_idx += ~finish  # gpt.py:575-577
        res.steps = i + 1
        if finish.all():  # gpt.py:592
            break
    H = np.stack(hiddens, 1)
    res.ids = [ids_buf[b, T: T + end_idx[b]] for b in range(B)]  # gpt.py:297-299
    if infer_text:
        res.ids = [r[:, 0] for r in res.ids]  # gpt.py:300-301
    res.hiddens = [H[b, : end_idx[b]] for b in range(B)]  # gpt.py:303-307
    return res
