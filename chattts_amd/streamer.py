"""Streaming back end: turns the `[B, n]` float32 chunks of `Chat.infer(stream=True)` into an ordered sequence of
16-bit PCM blocks for ONE listener -- the behaviour of the reference's `ChatStreamer` (/root/reference/examples/cmd/stream.py:9-183),
re-stated (not imported: the GPU box has no /root/reference) and pinned byte for byte on that class fed the same chunks
(tests/golden/backend.npz, oracle/make_backend_goldens.py).

What the reference class does, in words:
  * chunks are collected until `base_block_size` samples per still-sounding utterance have arrived (:15-30, :81-95); a chunk with no
    sounding utterance (max |x| <= 1e-6 everywhere) is ignored;
  * a collected block is converted as a whole -- ONE peak over all its rows (`float_to_int16`, tools/audio/np.py:7-11) -- and appended
    to the per-utterance history (:96-102);
  * the listener hears utterance 0 first: its part of every block is emitted in 12000-sample pieces, silent pieces dropped (:104-109);
    the first block in which the current utterance is silent moves on to the next one, whose WHOLE history so far is emitted (:110-118);
    when the last utterance falls silent the loop stops (:121-122);
  * at the end: the block still being collected (:124-138), then the complete histories of the utterances not yet reached (:139-145).
Kept quirks: the leftover block is emitted for the current utterance only if that utterance sounds in it, and it enters the histories only
then; "PCM16" yields int16 arrays, "PCM16_byte" their little-endian bytes, None the float32 pieces."""
from __future__ import annotations

from typing import Iterable, Iterator, Optional

import numpy as np

from .audio import float_to_int16

SILENCE = 1e-6      # stream.py:59: a piece counts as sound when max |x| >= 1e-6
PIECE = 12000       # stream.py:65: samples per emitted piece (0.5 s at 24 kHz)


def sounds(x: np.ndarray) -> bool:
    return bool(x.size) and not (np.abs(x).max() < SILENCE)


class ChatStreamer:
    def __init__(self, base_block_size: int = 8000, product: str = "f64"):
        self.base_block_size = base_block_size
        self.product = product       # arithmetic of the PCM conversion: see audio.float_to_int16

    def _convert(self, block: np.ndarray, fmt: Optional[str]) -> np.ndarray:
        return float_to_int16(block, self.product) if fmt in ("PCM16", "PCM16_byte") else block

    @staticmethod
    def _emit(row: np.ndarray, fmt: Optional[str]) -> Iterator:
        for lo in range(0, row.shape[0], PIECE):
            piece = row[lo: lo + PIECE]
            if sounds(piece):
                yield piece.astype("<i2").tobytes() if fmt == "PCM16_byte" else piece

    def generate(self, streamchat: Iterable[np.ndarray], output_format: Optional[str] = None) -> Iterator:
        assert output_format in ("PCM16_byte", "PCM16", None)
        cur = 0                  # the utterance the listener is hearing
        held = None              # block under collection
        history = None           # converted blocks so far, [B, total]
        rows = 0
        tail, holding = None, False
        for chunk in streamchat:
            tail = chunk         # (the reference's loop variable: what its epilogue looks at, stream.py:124-128)
            rows = len(chunk)
            if chunk.shape[1] == 0:          # an empty chunk (the reference's np.max raises on it): ignored like a silent one
                continue
            n_sounding = int((np.abs(chunk).max(axis=1) > SILENCE).sum())
            if n_sounding == 0:
                continue
            block = chunk if held is None else np.concatenate([held, chunk], axis=1)
            holding = block.shape[0] * block.shape[1] < n_sounding * self.base_block_size
            tail = block
            if holding:
                held = block
                continue
            held = None
            block = self._convert(block, output_format)
            history = block if history is None else np.concatenate([history, block], axis=1)
            if sounds(block[cur]):
                yield from self._emit(block[cur], output_format)
            elif cur < rows - 1:
                cur += 1
                yield from self._emit(history[cur], output_format)
            else:
                break
        # what was still being collected when the stream ended -- as the reference has it: it looks at its loop variable, so silent chunks
        # behind the last collected one hide the collected block (nothing is emitted for it)
        if holding and tail is not None and tail.shape[1] > 0:
            block = self._convert(tail, output_format)
            if sounds(block[cur]):
                yield from self._emit(block[cur], output_format)
                history = block if history is None else np.concatenate([history, block], axis=1)
        if history is not None:
            for b in range(cur + 1, rows):     # utterances the listener never reached: everything they produced
                yield from self._emit(history[b], output_format)
