"""numpy restatement of the reference's per-step sampling chain.

Follows /root/reference/ChatTTS/model/gpt.py:487-508 (temperature, processors, EOS mask, softmax,
multinomial), /root/reference/ChatTTS/model/processors.py:18-35 (repetition penalty) and :38-58
(`gen_logits`: TopP(min_tokens_to_keep=3) then TopK(min_tokens_to_keep=3)); the two warpers live
in `transformers.generation.logits_process` (third-party, transformers>=4.41.1 per
requirements.txt:7; 5.15.0 in the build container) and are restated from their published
algorithm.  `torch.multinomial(p, 1, generator=g)` == argmax(p / q), q ~ Exp(1) drawn by
`empty_like(p).exponential_(1, g)`; q is an INPUT here, drawn with torch
on the host by the caller.

TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def softmax_f32(x: np.ndarray) -> np.ndarray:
    """ATen vec_host_softmax_lastdim: exp(x - max) * (1 / sum)."""
    m = np.max(x, axis=-1, keepdims=True)
    e = np.exp((x - m).astype(f32)).astype(f32)
    s = np.sum(e, axis=-1, keepdims=True, dtype=f32)
    return (e * (f32(1.0) / s)).astype(f32)


def repetition_penalty(history: np.ndarray, scores: np.ndarray, pow_table: np.ndarray,
                       max_input_ids: int, past_window: int, row_offset: int = 0) -> np.ndarray:
    """processors.py:18-35.  history [rows, n] int64 (generated tokens of each (b,k) row).
    pow_table[f] = torch.pow(penalty, f) as float32, f = 0..past_window.
    `row_offset` is the global index of row 0 (multi-GPU sharding keeps the reference's
    "rows >= max_input_ids get no penalty" quirk keyed on the GLOBAL row index, processors.py:24-27)."""
    rows, V = scores.shape
    if history.shape[1] > past_window:
        history = history[:, -past_window:]
    freq = np.zeros((rows, V), dtype=np.int64)
    for j in range(history.shape[1]):
        np.add.at(freq, (np.arange(rows), history[:, j]), 1)
    grow = np.arange(rows) + row_offset
    freq[grow >= max_input_ids] = 0
    alpha = pow_table[freq]
    return np.where(scores < 0, scores * alpha, scores / alpha).astype(f32)


def top_p_warp(scores: np.ndarray, top_p: float, min_keep: int) -> np.ndarray:
    """TopPLogitsWarper.__call__: ascending sort, softmax, cumsum (ATen CPU cumsum accumulates a
    float row in double and rounds each prefix to float), remove cum <= 1-top_p (threshold cast
    to float32 by type promotion), never remove the last `min_keep`, scatter back, fill -inf."""
    order = np.argsort(scores, axis=-1, kind="stable")
    srt = np.take_along_axis(scores, order, axis=-1)
    probs = softmax_f32(srt)
    cum = np.cumsum(probs.astype(np.float64), axis=-1).astype(f32)
    remove_sorted = cum <= f32(1.0 - top_p)
    remove_sorted[:, -min_keep:] = False
    remove = np.zeros_like(remove_sorted)
    np.put_along_axis(remove, order, remove_sorted, axis=-1)
    out = scores.copy()
    out[remove] = -np.inf
    return out


def top_k_warp(scores: np.ndarray, top_k: int, min_keep: int) -> np.ndarray:
    """TopKLogitsWarper.__call__: k = min(max(top_k, min_keep), V); remove scores < k-th largest."""
    k = min(max(top_k, min_keep), scores.shape[-1])
    kth = np.sort(scores, axis=-1)[:, -k][:, None]
    out = scores.copy()
    out[scores < kth] = -np.inf
    return out


def sample_step(logits: np.ndarray, history: np.ndarray, q: np.ndarray, *, temperature: np.ndarray,
                top_p, top_k, pow_table, max_input_ids: int, past_window: int = 16,
                mask_eos=False, force_eos=None, eos: int = 625, row_offset: int = 0,
                return_processed: bool = False):
    """One sampling step over rows = B*n_vq.

    logits [rows, V] f32 (row r = b*n_vq + k, gpt.py:459-464); history [rows, n] (gpt.py:466-475);
    temperature [rows] (gpt.py:350-355); q [rows, V] Exp(1) draws.
    mask_eos: bool or bool[rows] -- rows whose EOS logit is set to -inf after the processors
    (gpt.py:494-495 for i < min_new_token; per-row for the bench harness's `stop_at` hook).
    force_eos: optional bool[rows] -- rows whose token is forced to EOS (`stop_at` hook only).
    Returns idx [rows] int64 (and the processed logits if asked).
    """
    x = (logits / temperature[:, None].astype(f32)).astype(f32)  # gpt.py:487
    if pow_table is not None:
        x = repetition_penalty(history, x, pow_table, max_input_ids, past_window, row_offset)
    if top_p is not None:
        x = top_p_warp(x, top_p, 3)
    if top_k is not None:
        x = top_k_warp(x, top_k, 3)
    mrows = np.broadcast_to(np.asarray(mask_eos, dtype=bool), (x.shape[0],))
    if mrows.any():
        x = x.copy()
        x[mrows, eos] = -np.inf  # gpt.py:494-495
    p = softmax_f32(x)  # gpt.py:497
    idx = np.argmax((p / q).astype(f32), axis=-1).astype(np.int64)  # gpt.py:501-508
    if force_eos is not None:
        idx = np.where(force_eos, eos, idx).astype(np.int64)
    if return_processed:
        return idx, x
    return idx
