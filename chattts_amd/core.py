"""`Chat`-level seam of the hot path: mirror of `Chat._infer_code` / `Chat._decode_to_wavs`
(/root/reference/ChatTTS/core.py:513-662) over token ids.

The reference's text front end (normalizer, tokenizer, speaker prompt decoration -- SURVEY 2.1 rows
7-9) is host string work outside the accelerated path and needs assets that are not reachable
offline; this facade therefore starts where `_infer_code` has tokens: `input_ids [B,T,4]`,
`attention_mask [B,T]`, `text_mask [B,T]` exactly as `Tokenizer.encode` returns them
(tokenizer.py:36-126).  INTEGRATION.md shows the ~10-line patch that routes the reference's own
`Chat` through these two calls.
"""
from __future__ import annotations

import logging
from dataclasses import dataclass
from typing import Iterator, List, Optional

import numpy as np
import torch

from . import weights as W
from .config import GPT
from .engine import CodecEngine, Context, GenerationOutputs, GptEngine, gen_logits


@dataclass(repr=False, eq=False)
class RefineTextParams:           # core.py:182-193
    prompt: str = ""
    top_P: float = 0.7
    top_K: int = 20
    temperature: float = 0.7
    repetition_penalty: float = 1.0
    max_new_token: int = 384
    min_new_token: int = 0
    show_tqdm: bool = True
    ensure_non_empty: bool = True
    manual_seed: Optional[int] = None


@dataclass(repr=False, eq=False)
class InferCodeParams:            # core.py:195-206 (+ the RefineTextParams fields it inherits, :182-193)
    prompt: str = ""
    top_P: float = 0.7
    top_K: int = 20
    temperature: float = 0.3
    repetition_penalty: float = 1.05
    max_new_token: int = 2048
    min_new_token: int = 0
    show_tqdm: bool = True
    ensure_non_empty: bool = True
    manual_seed: Optional[int] = None
    spk_emb: Optional[str] = None
    spk_smp: Optional[str] = None
    txt_smp: Optional[str] = None
    stream_batch: int = 24
    stream_speed: int = 12000
    pass_first_n_batches: int = 2


class Chat:
    def __init__(self, logger=logging.getLogger("chattts_amd")):
        self.logger = logger
        self.context = Context()
        self.gpt: Optional[GptEngine] = None
        self.codec: Optional[CodecEngine] = None

    def has_loaded(self) -> bool:
        return self.gpt is not None and self.codec is not None

    def load(self, custom_path: Optional[str] = None, device: Optional[torch.device] = None, dtype: str = "bf16",
             state_dicts: Optional[dict] = None) -> bool:
        """`Chat.load(source="custom", custom_path=...)` (core.py:137-163) for the four hot-path asset
        files; `state_dicts` short-circuits disk I/O (synthetic weights)."""
        device = device or torch.device("cuda:0")
        sds = state_dicts if state_dicts is not None else W.load_assets(custom_path)
        self.gpt = GptEngine(sds["gpt"], sds["embed"], device, dtype=dtype, logger=self.logger)
        self.codec = CodecEngine(sds["decoder"], sds["vocos"], device)
        return True

    def unload(self):               # core.py:165-174
        self.gpt = None
        self.codec = None

    def interrupt(self):            # core.py:272-273
        self.context.set(True)

    def infer_code(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, text_mask: torch.Tensor,
                   params: InferCodeParams = InferCodeParams(), stream: bool = False, return_hidden: bool = True,
                   **shard_kw) -> Iterator[GenerationOutputs]:
        """`Chat._infer_code` from `gen_logits` on (core.py:580-658)."""
        assert self.has_loaded()
        # core.py:558-561: a scalar temperature is replicated over the 4 codebooks, a list is used as is
        temperature = torch.tensor(params.temperature if isinstance(params.temperature, list) else [params.temperature] * GPT.n_vq)
        warpers, procs = gen_logits(GPT.n_audio - 1, params.top_P, params.top_K, params.repetition_penalty)
        emb = self.gpt.embed_prompt(input_ids, text_mask)
        return self.gpt.generate(
            emb, input_ids, temperature, GPT.n_audio - 1, attention_mask, params.max_new_token, params.min_new_token,
            (*procs, *warpers), False, False, return_hidden, stream, params.show_tqdm, params.ensure_non_empty,
            params.stream_batch, params.manual_seed, self.context, **shard_kw)

    def refine_text_ids(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, text_mask: torch.Tensor, eos_token: int,
                        params: RefineTextParams = RefineTextParams(), **kw) -> GenerationOutputs:
        """`Chat._refine_text` from `gen_logits` on (core.py:682-751): the same generator in text mode
        (`infer_text=True`: text embedding, 21178-way text head, tokens replicated over the 4 slots); `eos_token`
        is `tokenizer.eos_token` ([Ebreak]).  Returns `GenerationOutputs` whose `ids[b]` is the 1-D refined token row."""
        assert self.has_loaded()
        warpers, procs = gen_logits(GPT.n_text, params.top_P, params.top_K, params.repetition_penalty)
        emb = self.gpt.embed_prompt(input_ids, text_mask)
        return next(self.gpt.generate(
            emb, input_ids, torch.tensor([params.temperature]), eos_token, attention_mask, params.max_new_token,
            params.min_new_token, (*procs, *warpers), True, False, False, False, params.show_tqdm, params.ensure_non_empty,
            24, params.manual_seed, self.context, **kw))

    def decode_to_wavs(self, hiddens: List[torch.Tensor]) -> np.ndarray:
        """`Chat._decode_to_wavs(result.hiddens, use_decoder=True)` (core.py:513-539) -> np.float32 [B, n]."""
        assert self.has_loaded()
        return self.codec.decode_to_wavs(hiddens).cpu().numpy()

    def infer_ids(self, input_ids, attention_mask, text_mask, params: InferCodeParams = InferCodeParams(), **kw) -> np.ndarray:
        """non-stream `Chat._infer` body for one batch (core.py:469-481, split_text=False, skip_refine_text=True),
        BEFORE the sample-level silence strip of core.py:258-270."""
        last = None
        for last in self.infer_code(input_ids, attention_mask, text_mask, params, stream=False, **kw):
            pass
        if last is None:
            return np.zeros((0,), np.float32)
        return self.decode_to_wavs(last.hiddens)

    def infer_ids_stream(self, input_ids, attention_mask, text_mask, params: InferCodeParams = InferCodeParams(), **kw):
        """stream=True body of `Chat._infer` for one batch (core.py:455-503): every `stream_batch` live steps the
        generator yields the cumulative result, the prefix is decoded again (the reference's O(n^2) schedule,
        App. D-10) and the next `stream_speed` samples are emitted; the first `pass_first_n_batches` yields are
        dropped (their decode is skipped here: its output is discarded by the reference, core.py:488-490);
        the tail is emitted with all-silent columns removed (core.py:500-503)."""
        length = 0
        pass_batch_count = 0
        wavs = None
        for result in self.infer_code(input_ids, attention_mask, text_mask, params, stream=True, **kw):
            pass_batch_count += 1
            if pass_batch_count <= params.pass_first_n_batches:
                wavs = None
                continue
            wavs = self.decode_to_wavs(result.hiddens)
            a = length
            b = min(a + params.stream_speed, wavs.shape[1])
            length = b
            yield wavs[:, a:b]
        if wavs is None and pass_batch_count > 0:
            wavs = self.decode_to_wavs(result.hiddens)
        if wavs is not None:
            new_wavs = wavs[:, length:]
            keep_cols = np.sum(np.abs(new_wavs) > 1e-5, axis=0) > 0
            yield new_wavs[:, keep_cols]

    def infer_tokens(self, input_ids, attention_mask, text_mask, params: InferCodeParams = InferCodeParams(),
                     stream: bool = False, split_text: bool = False, max_split_batch: int = 4, **kw):
        """`Chat.infer(..., skip_refine_text=True)` from the point where text has become tokens
        (core.py:208-270 + `_infer` :455-503): batches of `max_split_batch` rows when `split_text` (else one
        batch), then the sample-level silence strip of :258-268 (`wav[|wav| > 1e-5]`, also mid-utterance) and,
        with `split_text`, one concatenated waveform.  `stream=True` returns the chunk generator instead.
        `interrupt()` state is cleared first, like core.py:223."""
        self.context.set(False)
        B = int(input_ids.shape[0])
        if B == 0:
            return []
        if stream:
            return self.infer_ids_stream(input_ids, attention_mask, text_mask, params, **kw)
        step = max_split_batch if split_text else B
        thr = np.float32(1e-5)
        stripped = []
        for lo in range(0, B, step):
            sl = slice(lo, min(lo + step, B))
            kw_b = dict(kw)
            if "stop_at" in kw_b and kw_b["stop_at"] is not None:
                kw_b["stop_at"] = kw_b["stop_at"][sl]
            wavs = self.infer_ids(input_ids[sl], attention_mask[sl], text_mask[sl], params, **kw_b)
            for wav in wavs:
                stripped.append(wav[np.abs(wav) > thr])
        if split_text:
            return [np.concatenate(stripped)]
        return stripped
