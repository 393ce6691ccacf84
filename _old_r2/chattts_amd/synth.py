"""Synthetic prompts / workloads: no tokenizer assets or real checkpoints are
reachable offline, so prompts are seeded random token ids in the layout `Tokenizer.encode`
produces (/root/reference/ChatTTS/model/tokenizer.py:73-110): ids replicated over the 4 slots,
LEFT padded with id 0 / mask 0; text_mask = attention_mask (tokenizer.py:112), so pad slots are
embedded as code id 0 by `Embed.forward` (they are masked out of attention anyway)."""
from __future__ import annotations

import numpy as np

from .config import GPT


def make_prompts(batch: int, t_min: int, t_max: int, seed: int = 0):
    """-> input_ids [B,T,4] int64, attention_mask [B,T] bool, text_mask [B,T] bool (T = longest)."""
    rs = np.random.RandomState(seed)
    lens = rs.randint(t_min, t_max + 1, size=batch)
    T = int(lens.max())
    ids = np.zeros((batch, T, GPT.n_vq), dtype=np.int64)
    mask = np.zeros((batch, T), dtype=bool)
    for b in range(batch):
        tok = rs.randint(1, GPT.n_text, size=int(lens[b]))
        ids[b, T - lens[b]:, :] = tok[:, None]
        mask[b, T - lens[b]:] = True
    text_mask = mask.copy()  # tokenizer.py:112: text_mask = attention_mask.bool()
    return ids, mask, text_mask


def make_stop_lengths(batch: int, n_min: int, n_max: int, seed: int = 0) -> np.ndarray:
    """Per-row forced output lengths N_b ~ U{n_min..n_max} (random weights never emit EOS on cue)."""
    rs = np.random.RandomState(seed + 1000)
    return rs.randint(n_min, n_max + 1, size=batch).astype(np.int32)
