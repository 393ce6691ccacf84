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
    # row_offset: the global index of row 0 (a contiguous shard) or, as an array, the global index of every row (a shard that is not a
    # contiguous block of the caller's batch: chattts_amd.dist.deal_shards)
    grow = np.asarray(row_offset) if np.ndim(row_offset) else np.arange(rows) + row_offset
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


def decision_margin(logits: np.ndarray, history: np.ndarray, q: np.ndarray, *, temperature: np.ndarray, top_p, top_k, pow_table,
                    max_input_ids: int, past_window: int = 16, mask_eos=False, eos: int = 625, row_offset: int = 0) -> np.ndarray:
    """float64 restatement of the PARITY CERTIFICATE the sampling kernels compute (include/chattts_amd.h, ctts_gen_state.margin; not a
    reference feature -- the reference has one arithmetic): per sampling row, the smallest distance by which the step was decided, in
    units of the pre-penalty tempered logit:
      c_arg = log(r_best / r_second) of argmax(p / q) over the kept tokens;
      c_cut = value gap between the last kept and the first dropped token of the warpers' prefix (0 when ties with the k-th largest value
              extended the prefix);
      c_p   = |log(cum / (1 - top_p))| of the top-p test at the last kept rank (if it was tested: rank >= 3) and at the rank top-p
              removed first;
    divided by the largest factor the repetition penalty applies to a perturbation (alpha for negative, 1 / alpha for positive scores).
    Order inside the prefix: value descending, ties lowest index first (the kernels' order)."""
    rows, V = logits.shape
    x = (logits / temperature[:, None].astype(f32)).astype(f32)
    amp = np.ones(rows)
    if pow_table is not None:
        h = history[:, -past_window:] if history.shape[1] > past_window else history
        freq = np.zeros((rows, V), dtype=np.int64)
        for j in range(h.shape[1]):
            np.add.at(freq, (np.arange(rows), h[:, j]), 1)
        freq[(np.arange(rows) + row_offset) >= max_input_ids] = 0
        alpha = pow_table[freq].astype(np.float64)
        amp = np.where(x < 0, alpha, 1.0 / alpha).max(1)
        x = repetition_penalty(history, x, pow_table, max_input_ids, past_window, row_offset)
    mrows = np.broadcast_to(np.asarray(mask_eos, dtype=bool), (rows,))
    out = np.full(rows, np.inf)
    for r in range(rows):
        xr = x[r].astype(np.float64)
        order = np.lexsort((np.arange(V), -xr))          # value desc, index asc
        prob = softmax_f32(x[r][None])[0].astype(np.float64)
        cum = prob[order][::-1].cumsum()[::-1]           # ascending cumulative probability including the rank itself
        n, c_cut, c_p = V, np.inf, np.inf
        if top_p is not None or top_k is not None:
            kk = min(max(top_k, 3), V) if top_k is not None else V
            thr = float(f32(1.0 - top_p)) if top_p is not None else 0.0
            kth = xr[order[kk - 1]]
            n = V
            p_dropped = False
            for j in range(V):
                bad_p = top_p is not None and j >= 3 and f32(cum[j]) <= f32(thr)
                bad_k = top_k is not None and j >= kk and xr[order[j]] != kth
                if bad_p or bad_k:
                    n, p_dropped = j, bad_p
                    break
            if n < V:
                c_cut = 0.0 if (top_k is not None and n > kk) else xr[order[n - 1]] - xr[order[n]]
            if top_p is not None:
                if n - 1 >= 3:
                    c_p = abs(np.log(max(cum[n - 1], 1e-38) / thr)) if thr > 0 else np.inf
                if n < V and p_dropped:
                    c_p = min(c_p, abs(np.log(max(cum[n], 1e-38) / thr)) if thr > 0 else np.inf)
        kept = order[:n]
        live = kept[~((kept == eos) & mrows[r])] if mrows[r] else kept
        lr = xr[live] - np.log(q[r][live].astype(np.float64))
        srt = np.sort(lr)[::-1]
        c_arg = (srt[0] - srt[1]) if len(srt) > 1 else np.inf
        out[r] = min(c_arg, c_cut, c_p) / amp[r]
    return out
