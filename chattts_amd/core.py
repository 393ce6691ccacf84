"""`Chat`-level seam of the hot path: mirror of `Chat.infer` / `_infer` / `_infer_code` / `_refine_text` /
`_decode_to_wavs` (/root/reference/ChatTTS/core.py:208-270, 395-503, 513-751).

Two entry levels:
  * text level  -- `Chat.infer(text, ...)` with the reference's argument list.  Needs a tokenizer directory
    (`asset/tokenizer`, config.py:10) and, for speaker sampling, the `Config.spk_stat` string (config.py:132); the host
    front end lives in `chattts_amd.frontend`.
  * token level -- `infer_code` / `refine_text_ids` / `infer_ids` / `infer_ids_stream` / `infer_tokens` start where
    `_infer_code` has tensors: `input_ids [B,T,4]`, `attention_mask [B,T]`, `text_mask [B,T]` exactly as
    `Tokenizer.encode` returns them (tokenizer.py:36-126).  bench.py and the parity tests use this level (synthetic
    prompts; the trained tokenizer is not reachable offline).
INTEGRATION.md shows the ~10-line patch that routes the reference's own `Chat` through the engine instead.
"""
from __future__ import annotations

import logging
import os
import re
from dataclasses import dataclass
from typing import Iterator, List, Optional, Union

import numpy as np
import torch

from . import weights as W
from .audio import float_to_int16
from .config import GPT
from .dvae import DvaeEngine
from .engine import CodecEngine, Context, GenerationOutputs, GptEngine, gen_logits
from .frontend import Normalizer, Speaker, Tokenizer, apply_speaker


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
    prompt: str = "[speed_5]"
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
    RefineTextParams = RefineTextParams      # the reference nests the two dataclasses in `Chat` (core.py:182-206)
    InferCodeParams = InferCodeParams

    def __init__(self, logger=logging.getLogger("chattts_amd"), homophones_map: Optional[str] = None):
        """`homophones_map`: path of the reference package's `res/homophones_map.json` (core.py:39-42); without it the
        normalizer skips homophone replacement."""
        self.logger = logger
        self.context = Context()
        self.gpt: Optional[GptEngine] = None
        self.codec: Optional[CodecEngine] = None
        self.dvae: Optional[DvaeEngine] = None
        self.tokenizer: Optional[Tokenizer] = None
        self.speaker: Optional[Speaker] = None
        self.normalizer = Normalizer(homophones_map, logger)
        self.incremental_stream = True   # stream=True decodes token windows with halos, not the whole prefix per yield

    def has_loaded(self, use_decoder: bool = True) -> bool:
        """core.py:50-66: the decoder path needs `Decoder.safetensors`, the `use_decoder=False` path the full DVAE"""
        return self.gpt is not None and self.codec is not None and (use_decoder or self.dvae is not None)

    def load(self, source: str = "local", force_redownload: bool = False, compile: bool = False, custom_path: Optional[str] = None,
             device: Optional[torch.device] = None, coef=None, use_flash_attn: bool = False, use_vllm: bool = False,
             experimental: bool = False, *, dtype: str = "bf16", state_dicts: Optional[dict] = None,
             tokenizer: Union[None, str, Tokenizer] = None, spk_stat: Optional[str] = None, codec_gemm: Optional[str] = None,
             warm=None) -> bool:
        """`Chat.load` with the reference's positional parameters and defaults (core.py:137-148), for `source="local"` /
        `"custom"` (assets already on disk; there is no network path here, `"huggingface"` returns False): the four
        hot-path safetensors files under `custom_path` (default: the working directory, like the reference's "local"
        source) and `asset/tokenizer`.  Keyword-only extras of this engine: `dtype` ("bf16" perf mode | "f32" parity mode on float32 arithmetic
        throughout | "f32x3" parity mode with split-fp16 Llama projections (float32-class, faster), every call reports its decision margins -- GptEngine),
        `state_dicts` short-circuits disk I/O (synthetic weights); `tokenizer` is a directory or a `Tokenizer`;
        `spk_stat` is the reference's `Config.spk_stat` string (needed by `sample_random_speaker` only); `codec_gemm` picks the
        acoustic decoder's dense-layer arithmetic (`CodecEngine`: "f16" | "bf16x3" | "f32"; default: "f16" in perf mode --
        waveform within 2e-5 RMS of the f32-class decoder for the same hidden states -- and "bf16x3" in parity mode).
        `warm=(B, T)` or `(B, T, InferCodeParams)`: pay the first request's cold start at load time instead (see `Chat.warm`).
        `compile`, `use_flash_attn`, `use_vllm`, `experimental` select between the reference's torch back ends and
        have no meaning for this engine (accepted, ignored).  `coef` is accepted and has no effect, as in the reference:
        `DVAE.__init__` installs it (dvae.py:219-226) and `load_pretrained` then overwrites the buffer with the
        checkpoint's `coef` tensor (dvae.py:254-259)."""
        if source not in ("custom", "local"):
            self.logger.error("chattts_amd loads local assets only (source=%s)", source)
            return False
        device = device or torch.device("cuda:0")
        root = custom_path if custom_path is not None else os.getcwd()
        try:
            sds = state_dicts if state_dicts is not None else W.load_assets(root)
        except W.AssetError as e:
            self.logger.error("%s", e)
            return False
        try:
            for name in W.ASSET_FILES:      # key set / shapes / dtypes against SURVEY App. B: a diff, not a KeyError in the repacking
                W.validate_state_dict(name, sds[name])
        except (W.AssetError, KeyError) as e:   # the reference's load logs and returns False (core.py:131-135,384)
            self.logger.error("%s", e)
            return False
        self.gpt = GptEngine(sds["gpt"], sds["embed"], device, dtype=dtype, logger=self.logger, **sds.get("gpt_config", {}))
        self.codec = CodecEngine(sds["decoder"], sds["vocos"], device, gemm=codec_gemm or ("f16" if dtype == "bf16" else "bf16x3"))
        self.dvae = DvaeEngine(sds["dvae"], device) if "dvae" in sds else None
        self.device = device
        if tokenizer is None and state_dicts is None and os.path.isdir(os.path.join(root, "asset", "tokenizer")):
            tokenizer = os.path.join(root, "asset", "tokenizer")
        if tokenizer is not None:
            self.tokenizer = tokenizer if isinstance(tokenizer, Tokenizer) else Tokenizer(tokenizer)
        if spk_stat is not None:
            self.speaker = Speaker(GPT.hidden, spk_stat, torch.device("cpu"))
        if warm is not None:
            self.warm(*warm)
        return True

    def warm(self, batch: int, prompt_len: int, params: Optional["Chat.InferCodeParams"] = None, stream: bool = True, **kw) -> float:
        """Pre-warm the engine for one geometry -- `batch` utterances, prompts padded to `prompt_len` tokens, the sampling constants
        and `max_new_token` of `params` (default: InferCodeParams()) -- so that the FIRST request of that shape finds what a second one
        would: the generation session (KV cache, token / hidden-state buffers: 2.3 GB at batch 64 x 2048 new tokens), the captured and
        instantiated decode graphs with their first replays behind them, the acoustic decoder's workspace and the pinned staging buffers
        (profiles/r4u_ttfs_probe.log: first audio of a fresh engine after 280 ms, 32-51 ms from the second call on).  Runs the real path
        on a synthetic prompt up to its first streamed chunk and interrupts it (core.py:272-273); a request of another shape still
        benefits from the allocator's cached blocks and the decoder's buffers (`kw`: the extra keywords the requests will pass to
        `infer_code`, e.g. a `stop_at` tensor -- the session is keyed on their presence).  The reference has no counterpart: its first call pays
        torch's lazy initialisation the same way (core.py:137-163 loads weights only).  Returns the seconds it took."""
        import time
        assert self.has_loaded()
        t0 = time.perf_counter()
        params = params or Chat.InferCodeParams(show_tqdm=False)
        ids = torch.ones((batch, prompt_len, GPT.n_vq), dtype=torch.int64)
        attn = torch.ones((batch, prompt_len), dtype=torch.bool)
        was = self.context.get()
        self.context.set(False)
        # an unseeded warm-up (manual_seed=None, the default) draws from torch's global CPU generator like any request: put the
        # generator back afterwards, so that a later unseeded request reproduces the reference's draws for a seed set before load().
        # stream=False: the generator is polled once per chunk of GptEngine.POLL steps, so the interrupt below stops it after the
        # first chunk, not after `max_new_token` steps.
        rng_state = torch.get_rng_state()
        try:
            n = 0
            for out in self.infer_code(ids, attn, torch.ones((batch, prompt_len), dtype=torch.bool), params, stream=stream, **kw):
                n += 1
                if out is not None and len(out.hiddens) and max(int(h.shape[0]) for h in out.hiddens) > 0 and self.codec is not None:
                    if stream:
                        self._stream_piece(out.hiddens, 0, params.stream_speed)      # window decode + staging buffers
                    else:
                        self.decode_to_wavs(out.hiddens)
                self.context.set(True)          # one chunk is enough: the generator stops at its next poll
        finally:
            self.context.set(was)
            torch.set_rng_state(rng_state)
        torch.cuda.synchronize(self.device)
        return time.perf_counter() - t0

    def unload(self):               # core.py:165-174
        self.gpt = None
        self.codec = None
        self.dvae = None
        self.tokenizer = None
        self.speaker = None

    # -- speakers (core.py:176-180) ---------------------------------------------------------------------------
    def sample_random_speaker(self) -> str:
        if self.speaker is None:
            raise RuntimeError("speaker statistics not loaded: pass spk_stat= to Chat.load")
        return self.speaker.sample_random()

    def sample_audio_speaker(self, wav) -> str:
        """24 kHz waveform -> `spk_smp` string: DVAE encode to [4,T] codes, packed like `Speaker.encode_prompt`"""
        if self.dvae is None:
            raise RuntimeError("full DVAE not loaded (asset/DVAE.safetensors, or state_dicts['dvae'])")
        return Speaker.encode_prompt(self.dvae.sample_audio(wav))

    def interrupt(self):            # core.py:272-273
        self.context.set(True)

    def infer_code(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, text_mask: torch.Tensor,
                   params: InferCodeParams = InferCodeParams(), stream: bool = False, return_hidden: bool = True,
                   spk_emb_ids: Optional[int] = None, **shard_kw) -> Iterator[GenerationOutputs]:
        """`Chat._infer_code` from `gen_logits` on (core.py:580-658).  With `params.spk_emb` and `spk_emb_ids` (the
        tokenizer's id of `[spk_emb]`) the prompt embedding gets the speaker vector at those positions (:630-637)."""
        assert self.has_loaded()
        # core.py:558-561: a scalar temperature is replicated over the 4 codebooks, a list is used as is
        temperature = torch.tensor(params.temperature if isinstance(params.temperature, list) else [params.temperature] * GPT.n_vq)
        warpers, procs = gen_logits(GPT.n_audio - 1, params.top_P, params.top_K, params.repetition_penalty)
        emb = self.gpt.embed_prompt(input_ids, text_mask)
        if params.spk_emb is not None and spk_emb_ids is not None:
            apply_speaker(emb, params.spk_emb, input_ids, spk_emb_ids)
        return self.gpt.generate(
            emb, input_ids, temperature, GPT.n_audio - 1, attention_mask, params.max_new_token, params.min_new_token,
            (*procs, *warpers), False, False, return_hidden, stream, params.show_tqdm, params.ensure_non_empty,
            params.stream_batch, params.manual_seed, self.context, **shard_kw)

    def refine_text_ids(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, text_mask: torch.Tensor, eos_token: int,
                        params: RefineTextParams = RefineTextParams(), num_code: int = GPT.n_text, **kw) -> GenerationOutputs:
        """`Chat._refine_text` from `gen_logits` on (core.py:682-751): the same generator in text mode
        (`infer_text=True`: text embedding, 21178-way text head, tokens replicated over the 4 slots); `eos_token`
        is `tokenizer.eos_token` ([Ebreak]).  Returns `GenerationOutputs` whose `ids[b]` is the 1-D refined token row."""
        assert self.has_loaded()
        warpers, procs = gen_logits(num_code, params.top_P, params.top_K, params.repetition_penalty)   # core.py:682-687: len(tokenizer)
        emb = self.gpt.embed_prompt(input_ids, text_mask)
        return next(self.gpt.generate(
            emb, input_ids, torch.tensor([params.temperature]), eos_token, attention_mask, params.max_new_token,
            params.min_new_token, (*procs, *warpers), True, False, False, False, params.show_tqdm, params.ensure_non_empty,
            24, params.manual_seed, self.context, **kw))

    def decode_to_wavs(self, result_list: List[torch.Tensor], use_decoder: bool = True, pad_to: Optional[int] = None) -> np.ndarray:
        """`Chat._decode_to_wavs` (core.py:513-539) -> np.float32 [B, n]: per-row hidden states [T_b,768] through the
        decoder, or (use_decoder=False) per-row token ids [T_b,4] through the full DVAE's codebook; then Vocos.
        `pad_to`: decode as rows of a batch whose longest row has that many tokens (dist.infer_sharded)."""
        assert self.has_loaded(use_decoder)
        if len(result_list) == 0:
            return np.array([], dtype=np.float32)
        if use_decoder:
            return self.codec.to_host(self.codec.decode_to_wavs(result_list, pad_to=pad_to))
        return self.codec.to_host(self.codec.vocos_decode(self.dvae.decode_codes(result_list, pad_to=pad_to)))

    def decode_to_pcm16(self, result_list: List[torch.Tensor], use_decoder: bool = True, strip: bool = True,
                        product: str = "f64") -> List[np.ndarray]:
        """`_decode_to_wavs` followed by what the reference's callers do with every waveform -- the sample-level silence strip of
        core.py:262-265 and `float_to_int16` (tools/audio/np.py:7-11; examples/web/funcs.py:209, tools/audio/pcm.py:29), one peak per
        utterance -- with the conversion ON THE DEVICE: the batch crosses PCIe as int16 + one mask bit per sample instead of float32.
        Returns one int16 array per utterance, equal to `float_to_int16(wav[np.abs(wav) > 1e-5])` bit for bit (the strip only removes
        samples that are far below one count, so the peak -- hence the scale -- is that of the unstripped row)."""
        assert self.has_loaded(use_decoder)
        if len(result_list) == 0:
            return []
        wav = self.codec.decode_to_wavs(result_list) if use_decoder else self.codec.vocos_decode(self.dvae.decode_codes(result_list))
        pcm, keep = self.codec.float_to_int16(wav, per_row=True, product=product, keep_thr=1e-5 if strip else None)
        pcm_h = self.codec.to_host(pcm)
        if not strip:
            return [pcm_h[b] for b in range(pcm_h.shape[0])]
        keep_h = self.codec.to_host(keep)
        n = pcm_h.shape[1]
        return [pcm_h[b][np.unpackbits(keep_h[b])[:n].astype(bool)] for b in range(pcm_h.shape[0])]

    def infer_ids(self, input_ids, attention_mask, text_mask, params: InferCodeParams = InferCodeParams(), **kw) -> np.ndarray:
        """non-stream `Chat._infer` body for one batch (core.py:469-481, split_text=False, skip_refine_text=True),
        BEFORE the sample-level silence strip of core.py:258-270."""
        last = None
        for last in self.infer_code(input_ids, attention_mask, text_mask, params, stream=False, **kw):
            pass
        if last is None:
            return np.zeros((0,), np.float32)
        return self.decode_to_wavs(last.hiddens)

    def infer_ids_pipelined(self, batches, params: InferCodeParams = InferCodeParams(), **kw) -> Iterator[np.ndarray]:
        """`infer_ids` over a QUEUE of batches (an iterable of `(input_ids, attention_mask, text_mask)` or of
        `(input_ids, attention_mask, text_mask, kwargs)`), software-pipelined: the acoustic decode + host copy of batch i run on the
        codec engine's side stream while batch i+1 is being generated (`CodecEngine.decode_to_wavs_async`).  Yields one
        np.float32 [B_i, n_i] array per batch, in order, each identical to `infer_ids` of that batch; a batch's result is yielded
        once the NEXT batch's generation has been issued and finished, the last one at the end."""
        pending = []        # results not yet handed out, oldest first: a PendingWavs, or None for a batch that produced nothing
        for item in batches:
            a, extra = (item[:3], item[3]) if len(item) == 4 else (item, {})
            last = None
            for last in self.infer_code(*a, params, stream=False, **{**kw, **extra}):
                pass
            pending.append(None if last is None else self.codec.decode_to_wavs_async(last.hiddens))
            while len(pending) > 1:     # exactly ONE item per input batch, in order, also for batches that yielded no output
                p = pending.pop(0)
                yield np.zeros((0,), np.float32) if p is None else p.result()
        for p in pending:
            yield np.zeros((0,), np.float32) if p is None else p.result()

    def _stream_piece(self, hiddens, a: int, b: Optional[int], use_decoder: bool = True, pcm16: bool = False) -> np.ndarray:
        """samples [a, b) (b=None: to the end) of the decode of the current prefix (core.py:482-497), from a token window with
        halos instead of the whole prefix (`CodecEngine.decode_window`); `incremental_stream=False` restores the reference's
        full re-decode per yield"""
        Tn = max(int(r.size(0)) for r in hiddens)
        total = 256 * (2 * Tn - 1) if use_decoder else None
        if not use_decoder or not self.incremental_stream:
            wavs = self.decode_to_wavs(hiddens, use_decoder)
            piece = wavs[:, a: wavs.shape[1] if b is None else min(b, wavs.shape[1])]
            return np.stack([float_to_int16(r) for r in piece]) if (pcm16 and piece.shape[1]) else piece
        hi = total if b is None else min(b, total)
        if hi <= a:     # a window that starts past the end of this prefix (a later split batch, see `_infer`): nothing to decode
            return np.zeros((len(hiddens), 0), np.int16 if pcm16 else np.float32)
        win = self.codec.decode_window(hiddens, a, hi)
        if pcm16 and win.shape[1] > 0:     # every row by its own peak -- float_to_int16(chunk[b]), examples/web/funcs.py:203-206 -- on the device
            return self.codec.to_host(self.codec.float_to_int16(win, per_row=True)[0])
        return self.codec.to_host(win)

    def infer_ids_stream(self, input_ids, attention_mask, text_mask, params: InferCodeParams = InferCodeParams(), **kw):
        """stream=True body of `Chat._infer` for one batch (core.py:455-503): every `stream_batch` live steps the
        generator yields the cumulative result and the next `stream_speed` samples of the decode of that prefix are
        emitted; the first `pass_first_n_batches` yields are dropped (core.py:488-490); the tail is emitted with all-silent
        columns removed (core.py:500-503).  The reference decodes the WHOLE prefix at every yield (O(n^2), App. D-10);
        here only the token window those samples depend on is decoded (same samples, O(n) in total), on the caller's
        stream while the generator's own stream already runs the next chunk."""
        length = 0
        pass_batch_count = 0
        result = None
        for result in self.infer_code(input_ids, attention_mask, text_mask, params, stream=True, **kw):
            pass_batch_count += 1
            if pass_batch_count <= params.pass_first_n_batches:
                continue
            piece = self._stream_piece(result.hiddens, length, length + params.stream_speed)
            length += piece.shape[1]
            yield piece
        if result is not None:
            new_wavs = self._stream_piece(result.hiddens, length, None)
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

    # ---------------------------------------------------------------------------------------------------------
    # text level: the reference's public call (core.py:208-270) and its private helpers
    # ---------------------------------------------------------------------------------------------------------
    def _need_tokenizer(self):
        if self.tokenizer is None:
            raise RuntimeError("no tokenizer loaded: Chat.load(custom_path=<dir holding asset/tokenizer>) or tokenizer=<dir>")

    def _infer_code(self, text, stream: bool, device, return_hidden: bool, params: InferCodeParams) -> Iterator[GenerationOutputs]:
        """core.py:542-662: decorate -> tokenise (+ audio-code prompt `spk_smp`) -> embed -> speaker -> generate."""
        self._need_tokenizer()
        if not isinstance(text, list):
            text = [text]
        assert len(text), "text should not be empty"
        prompt = Speaker.decode_prompt(params.spk_smp) if params.spk_smp is not None else None
        ids, attn, tmask = self.tokenizer.encode(
            Speaker.decorate_code_prompts(text, params.prompt, params.txt_smp, params.spk_emb), GPT.n_vq, prompt=prompt)
        return self.infer_code(ids, attn, tmask, params, stream=stream, return_hidden=return_hidden,
                               spk_emb_ids=self.tokenizer.spk_emb_ids)

    def infer_sharded(self, text, params_infer_code: InferCodeParams = InferCodeParams(), lang=None, do_text_normalization: bool = True,
                      do_homophone_replacement: bool = True, policy: str = "snake", group=None, dst: int = 0, use_decoder: bool = True):
        """`Chat.infer(text, skip_refine_text=True, split_text=False, use_decoder=...)` (core.py:208-270) over the ranks of the `torch.distributed` process
        group: every rank calls this with the SAME texts and parameters; the batch is tokenised everywhere (host work, deterministic), dealt
        by prompt length, generated and decoded shard by shard (`dist.infer_sharded`: global row numbering for the CPU draws and the
        rows >= 625 quirk, decode padded to the global longest utterance) and rank `dst` returns the list of stripped waveforms in the
        caller's order -- what the single-process call returns; the other ranks return None.  The reference has no data-parallel mode."""
        from .dist import infer_sharded
        assert self.has_loaded(use_decoder=use_decoder)
        self._need_tokenizer()
        self.context.set(False)
        if not isinstance(text, list):
            text = [text]
        if len(text) == 0:
            return []
        text = [self.normalizer(t, do_text_normalization, do_homophone_replacement, lang) for t in text]
        params = params_infer_code
        prompt = Speaker.decode_prompt(params.spk_smp) if params.spk_smp is not None else None
        ids, attn, tmask = self.tokenizer.encode(
            Speaker.decorate_code_prompts(text, params.prompt, params.txt_smp, params.spk_emb), GPT.n_vq, prompt=prompt)
        wavs = infer_sharded(self, ids, attn, tmask, params, policy=policy, group=group, dst=dst, use_decoder=use_decoder,
                             spk_emb_ids=self.tokenizer.spk_emb_ids)
        if wavs is None:
            return None
        thr = np.float32(1e-5)
        return [wav[np.abs(wav) > thr] for wav in wavs]     # core.py:258-266: the sample-level strip

    def _refine_text(self, text, device, params: RefineTextParams) -> GenerationOutputs:
        """core.py:665-751"""
        self._need_tokenizer()
        if not isinstance(text, list):
            text = [text]
        ids, attn, tmask = self.tokenizer.encode(Speaker.decorate_text_prompts(text, params.prompt), GPT.n_vq)
        return self.refine_text_ids(ids, attn, tmask, self.tokenizer.eos_token, params, num_code=self.tokenizer.len)

    def infer(self, text, stream=False, lang=None, skip_refine_text=False, refine_text_only=False, use_decoder=True,
              do_text_normalization=True, do_homophone_replacement=True, split_text=True, max_split_batch=4,
              params_refine_text: RefineTextParams = RefineTextParams(), params_infer_code: InferCodeParams = InferCodeParams(),
              *, pcm16: bool = False):
        """core.py:208-270: `List[np.ndarray]` (one stripped waveform per text, or ONE concatenated waveform when
        `split_text`), a generator of `np.ndarray [B, n]` chunks when `stream`, the refined text when `refine_text_only`.
        `pcm16=True` (keyword-only, not in the reference): the same results as 16-bit PCM -- what the reference's callers get from
        `float_to_int16` (tools/audio/np.py:7-11) on each returned waveform / on each row of each streamed chunk, computed on the
        device so that half the bytes cross PCIe: `infer(t, pcm16=True)[i] == float_to_int16(infer(t)[i])` bit for bit."""
        self.context.set(False)
        if split_text and isinstance(text, str):
            if "\n" in text:
                text = text.split("\n")
            else:                                  # sentence ends: after a CJK full stop, or after ". "
                text = [t for t in re.split(r"(?<=\u3002)|(?<=\.\s)", text) if t]
            self.logger.info("split text into %d parts", len(text))
        if len(text) == 0:
            return []
        res_gen = self._infer(text, stream, lang, skip_refine_text, refine_text_only, use_decoder, do_text_normalization,
                              do_homophone_replacement, split_text, max_split_batch, params_refine_text, params_infer_code,
                              pcm16=pcm16 and not refine_text_only and not (split_text and not stream))
        if stream:
            return res_gen
        if refine_text_only:
            return next(res_gen)
        if pcm16 and not split_text:
            return [w for wavs in res_gen for w in wavs]          # already stripped and converted, utterance by utterance, on the device
        thr = np.float32(1e-5)
        stripped = [wav[np.abs(wav) > thr] for wavs in res_gen for wav in wavs]   # sample-level strip, also mid-utterance
        if pcm16:      # split_text: ONE concatenated waveform, hence one peak over all sentences -- converted on the host
            return [float_to_int16(np.concatenate(stripped))]
        return [np.concatenate(stripped)] if split_text else stripped

    def _infer(self, text, stream, lang, skip_refine_text, refine_text_only, use_decoder, do_text_normalization,
               do_homophone_replacement, split_text, max_split_batch, params_refine_text, params_infer_code, pcm16: bool = False):
        """core.py:395-503 (generator)."""
        assert self.has_loaded(use_decoder=use_decoder)
        if not isinstance(text, list):
            text = [text]
        text = [self.normalizer(t, do_text_normalization, do_homophone_replacement, lang) for t in text]
        if not skip_refine_text:
            refined = self._refine_text(text, self.device, params_refine_text)
            tokens = [row[row.less(self.tokenizer.break_0_ids)] for row in refined.ids]   # control tokens >= [break_0] dropped
            text = self.tokenizer.decode(tokens)
            refined.destroy()
            if refine_text_only:
                yield "\n".join(text) if (split_text and isinstance(text, list)) else text
                return
        if split_text and len(text) > 1 and params_infer_code.spk_smp is None:
            # core.py:435-453: the first sentence is synthesised alone and its audio becomes the speaker prompt of the rest
            refer_text = text[0]
            result = next(self._infer_code(refer_text, False, self.device, use_decoder, params_infer_code))
            wavs = self.decode_to_wavs(result.hiddens if use_decoder else result.ids, use_decoder)
            result.destroy()
            params_infer_code.spk_smp = self.sample_audio_speaker(wavs[0])
            params_infer_code.txt_smp = refer_text
        length = 0
        pass_batch_count = 0
        step = max_split_batch if split_text else len(text)
        for lo in range(0, len(text), step):
            batch = text[lo: lo + step]
            if split_text:
                self.logger.info("infer split %d~%d", lo, lo + len(batch))
            last = None
            for result in self._infer_code(batch, stream, self.device, use_decoder, params_infer_code):
                if not stream:
                    src = result.hiddens if use_decoder else result.ids
                    wavs = self.decode_to_pcm16(src, use_decoder) if pcm16 else self.decode_to_wavs(src, use_decoder)
                    result.destroy()
                    yield wavs
                    continue
                if last is not None:
                    last.destroy()
                last = result
                pass_batch_count += 1
                if pass_batch_count <= params_infer_code.pass_first_n_batches:
                    continue     # the reference decodes these yields and drops the audio (core.py:482-490)
                src = result.hiddens if use_decoder else result.ids
                piece = self._stream_piece(src, length, length + params_infer_code.stream_speed, use_decoder, pcm16)
                # core.py:491-496: `b = a + stream_speed`, clamped to the width of THIS decode, becomes the new `length` -- also when
                # that is BELOW `a`: `length` and `pass_batch_count` are not reset between split batches, so the first yields of a
                # later batch (a short prefix again) are empty and pull `length` back (tests/test_host_flow.py, stream_split_batches)
                # (the decode width of a T-token prefix is 256 (2 T - 1) samples on both decode paths; never negative for an empty yield)
                length = min(length + params_infer_code.stream_speed, max(0, 256 * (2 * max(int(r.size(0)) for r in src) - 1)))
                yield piece
            if stream and last is not None:
                new_wavs = self._stream_piece(last.hiddens if use_decoder else last.ids, length, None, use_decoder)
                last.destroy()
                keep_cols = np.sum(np.abs(new_wavs) > 1e-5, axis=0) > 0
                tail = new_wavs[:, keep_cols]
                # the last chunk is filtered by columns on the float samples first (core.py:500-503): converted on the host, row by row
                if pcm16:     # int16 like every other chunk of the stream, also when nothing survives the column filter
                    tail = np.stack([float_to_int16(r) for r in tail]) if tail.shape[1] else tail.astype(np.int16)
                yield tail
