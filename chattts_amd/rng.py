"""Host-side random draws and constant tables for the sampling kernel.

The reference samples with `torch.multinomial(scores, 1, generator=G.manual_seed(seed))` on the CPU
(/root/reference/ChatTTS/model/gpt.py:497-508), which ATen evaluates as
`argmax(scores / empty_like(scores).exponential_(1, G))`.  The parity contract is the CPU stream
(SURVEY.md App. D-1), so the Exp(1) tensor is drawn here with the very same torch call and
uploaded; the kernel only does the divide + argmax.  Facts relied on (probed):
repeatable, thread-count independent, and a draw of shape (R, V) is a prefix of (R', V), R' > R --
so a shard holding global rows [r0, r1) takes rows r0..r1 of the full-batch draw.
"""
from __future__ import annotations

from typing import Optional

import torch


class ExpDraws:
    """q tensors for successive steps of one `generate` call."""

    def __init__(self, total_rows: int, vocab: int, manual_seed: Optional[int],
                 row_begin: int = 0, row_end: Optional[int] = None, rows: Optional[torch.Tensor] = None):
        """`rows` (int64 [n], optional): the sampling rows of the full-batch draw this object serves, in the caller's order, when
        they are not one contiguous block [row_begin, row_end) -- a length-balanced data-parallel shard (dist.py) or the exact re-run
        of the utterances a parity certificate flagged (engine.py)."""
        self.total_rows = total_rows
        self.vocab = vocab
        self.seed = manual_seed
        self.r0 = row_begin
        self.r1 = total_rows if row_end is None else row_end
        self.rows = None if rows is None else rows.to(torch.int64).reshape(-1)
        self.gen = torch.Generator(device="cpu")  # gpt.py:39
        self._const = None
        if manual_seed is not None:
            # gpt.py:504-507 re-seeds before EVERY step => the draw is the same at every step.
            self.gen.manual_seed(manual_seed)
            self._const = self._draw(self.gen)

    def _draw(self, gen) -> torch.Tensor:
        q = torch.empty((self.total_rows, self.vocab), dtype=torch.float32)
        q.exponential_(1, generator=gen)
        if self.rows is not None:
            return q[self.rows].contiguous()
        return q[self.r0: self.r1].contiguous()

    @property
    def constant(self) -> bool:
        return self._const is not None

    def step(self, i: int) -> torch.Tensor:
        """[rows, vocab] f32 CPU tensor for step i.  Must be called for i = 0, 1, 2, ... in order
        when unseeded (it advances torch's global CPU generator exactly like the reference)."""
        if self._const is not None:
            return self._const
        return self._draw(None)

    def step_into(self, i: int, dst: torch.Tensor) -> None:
        """Writes step i's draw into `dst` ([rows, vocab] float32, e.g. a pinned staging row).  When this object covers
        the whole batch the draw goes straight into `dst` (same generator call, no temporary, no copy)."""
        if self._const is not None:
            dst.copy_(self._const)
        elif self.rows is None and self.r0 == 0 and self.r1 == self.total_rows and dst.is_contiguous():
            dst.exponential_(1)
        else:
            dst.copy_(self._draw(None))

    def block(self, i0: int, n: int) -> torch.Tensor:
        """[n, rows, vocab] for steps i0 .. i0+n-1 (n == 1 when seeded)."""
        if self._const is not None:
            return self._const.unsqueeze(0)
        return torch.stack([self._draw(None) for _ in range(n)], 0)


def penalty_table(penalty: Optional[float], window: int = 16) -> Optional[torch.Tensor]:
    """alpha[f] = torch.pow(penalty, f) for f = 0..window, evaluated by torch itself so the kernel's
    table holds the reference's float32 values (/root/reference/ChatTTS/model/processors.py:29)."""
    if penalty is None or penalty == 1:
        return None
    return torch.pow(float(penalty), torch.arange(0, window + 1)).to(torch.float32)
