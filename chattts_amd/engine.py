"""Host side of the hot path: weight repacking, per-call device state, the generate loop
(mirror of `GPT.generate`, /root/reference/ChatTTS/model/gpt.py:316-618) and the acoustic decoder
(mirror of `Chat._decode_to_wavs`, /root/reference/ChatTTS/core.py:513-539).

PyTorch is used for device memory, streams and host<->device copies only; every arithmetic op of
the path runs in libchattts_amd.so (hand-written gfx950 kernels) through the C ABI in
include/chattts_amd.h.  There is no eager/PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import logging
import math
from dataclasses import dataclass, field
from typing import Iterator, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._sync import wait_event, wait_stream
from .config import DVAE, GPT, VOCOS
from .rng import ExpDraws, penalty_table
from .weights import fold_weight_norm, gpt_layer_count

log = logging.getLogger("chattts_amd")


# ---------------------------------------------------------------------------------------------
# logits-processor descriptors (mirror of gen_logits, /root/reference/ChatTTS/model/processors.py:38-58)
# ---------------------------------------------------------------------------------------------
@dataclass
class RepetitionPenalty:          # CustomRepetitionPenaltyLogitsProcessorRepeat, processors.py:6-35
    penalty: float
    max_input_ids: int
    past_window: int = 16


@dataclass
class TopP:                       # transformers TopPLogitsWarper(top_p, min_tokens_to_keep=3)
    top_p: float
    min_tokens_to_keep: int = 3


@dataclass
class TopK:                       # transformers TopKLogitsWarper(top_k, min_tokens_to_keep=3)
    top_k: int
    min_tokens_to_keep: int = 3


def gen_logits(num_code: int, top_P=0.7, top_K=20, repetition_penalty=1.0):
    """Same signature / return shape as the reference's `gen_logits` (processors.py:38-58); returns
    descriptors the fused sampling kernel understands instead of Python callables."""
    warpers, procs = [], []
    if top_P is not None:
        warpers.append(TopP(top_P, 3))
    if top_K is not None:
        warpers.append(TopK(top_K, 3))
    if repetition_penalty is not None and repetition_penalty != 1:
        procs.append(RepetitionPenalty(repetition_penalty, num_code, 16))
    return warpers, procs


@dataclass
class SamplingPlan:
    top_p: Optional[float] = None
    top_k: Optional[int] = None
    penalty: Optional[float] = None


def plan_from_processors(processors: Sequence, infer_text: bool = False) -> SamplingPlan:
    """Accepts this module's descriptors AND the reference's own objects (duck-typed on the attribute
    names of transformers' warpers / the reference's penalty class).  Anything else cannot be fused
    into the kernel and is rejected loudly.  Order must be penalty -> top-p -> top-k (core.py:649)."""
    plan = SamplingPlan()
    stage = 0
    for p in processors:
        if hasattr(p, "penalty") and hasattr(p, "past_window"):
            if infer_text:
                # the reference's processor receives [B, n, 1] histories in text mode (gpt.py:477-485) and its
                # one_hot(...).sum(1) then broadcasts [B,V] against [B,1,V]: it only works because refine-text
                # defaults to repetition_penalty = 1.0, which creates no processor (processors.py:52)
                raise NotImplementedError("refine-text mode supports repetition_penalty = 1.0 only (reference default)")
            if stage > 0 or p.past_window != 16 or p.max_input_ids != GPT.n_audio - 1:
                raise NotImplementedError("repetition penalty must come first with past_window=16, max_input_ids=625")
            plan.penalty = float(p.penalty)
            stage = 1
        elif hasattr(p, "top_p"):
            if stage > 1 or getattr(p, "min_tokens_to_keep", 3) != 3:
                raise NotImplementedError("top-p must precede top-k and use min_tokens_to_keep=3")
            plan.top_p = float(p.top_p)
            stage = 2
        elif hasattr(p, "top_k"):
            if getattr(p, "min_tokens_to_keep", 3) != 3:
                raise NotImplementedError("top-k must use min_tokens_to_keep=3")
            plan.top_k = int(p.top_k)
            stage = 3
        else:
            raise NotImplementedError(f"logits processor {type(p).__name__} cannot be fused into the HIP sampling kernel")
    return plan


@dataclass(repr=False, eq=False)
class GenerationOutputs:          # gpt.py:276-285
    ids: List[torch.Tensor]
    attentions: list
    hiddens: List[torch.Tensor]

    def destroy(self):
        self.ids.clear()
        self.attentions.clear()
        self.hiddens.clear()


class RowList(list):
    """The per-utterance rows of a result (what the reference hands around as a plain list) that are VIEWS of one zero-padded
    [B, Tmax, ...] tensor, `padded` -- the batch `Chat._decode_to_wavs` would rebuild from them row by row (core.py:525-533).
    `CodecEngine.decode_to_wavs` takes it as is; any list operation (slicing, copying) yields a plain list and the generic path."""
    padded: Optional[torch.Tensor] = None


class Context:                    # gpt.py:103-111
    def __init__(self):
        self._interrupt = False

    def set(self, v: bool):
        self._interrupt = v

    def get(self) -> bool:
        return self._interrupt


class _EventGroup:
    """several CUDA events waited on as one"""

    def __init__(self, evs):
        self.evs = evs

    def synchronize(self):
        for e in self.evs:
            wait_event(e)


def cu_masked_stream(lib, dev, first: int, n: int, stride: int, complement: bool) -> "torch.cuda.Stream":
    """a HIP stream confined to CUs first, first + stride, ... (n of them) -- or to every OTHER CU (complement): ctts_stream_create_cu_mask"""
    h = C.c_void_p()
    with torch.cuda.device(dev):
        _lib.check(lib.ctts_stream_create_cu_mask(first, n, stride, 1 if complement else 0, C.byref(h)), "ctts_stream_create_cu_mask")
    return torch.cuda.ExternalStream(h.value, device=dev)


def codec_cu_spec():
    """CTTS_CODEC_CUS="n[:stride[:first]]" -> (first, n, stride) or None"""
    spec = os.environ.get("CTTS_CODEC_CUS", "")
    if not spec:
        return None
    parts = [int(v) for v in spec.split(":")]
    return (parts[2] if len(parts) > 2 else 0), parts[0], (parts[1] if len(parts) > 1 else 1)


def left_pad_starts(attention_mask: torch.Tensor) -> torch.Tensor:
    """kv_start[b] = number of leading zeros; raises unless the mask is left padding (tokenizer.py:73-110)."""
    m = attention_mask.to(torch.bool).cpu()
    T = m.shape[1]
    n = m.sum(1)
    start = T - n
    ar = torch.arange(T)[None, :]
    if not torch.equal(m, ar >= start[:, None]):
        raise NotImplementedError("only left-padded attention masks (what Tokenizer.encode produces) are supported")
    if int(n.min()) <= 0:
        raise ValueError("empty prompt row")
    return start.to(torch.int32)


def rope_row_perm() -> torch.Tensor:
    """Row order of q_proj / k_proj inside wqkv in perf mode: within every head the 16-row tiles hold
    dims [8t..8t+7] followed by [32+8t..32+8t+7], so that the two halves of a rotate-half pair land in
    the same 16-column MFMA tile of the QKV kernel and RoPE + KV append run in its epilogue."""
    idx = []
    for h in range(GPT.n_heads):
        for t in range(4):
            idx += [h * 64 + 8 * t + i for i in range(8)] + [h * 64 + 32 + 8 * t + i for i in range(8)]
    return torch.tensor(idx, dtype=torch.long)


def pack_frag(w: torch.Tensor) -> torch.Tensor:
    """[R, C] (R % 16 == 0, C % 32 == 0) -> the fragment-packed order of csrc/decode.hip,
    [R/16][C/32][lane = (c%32)/8 * 16 + r%16][c%8]: every (16-row tile, 32-column chunk) becomes one contiguous KiB (bf16)
    in exactly the lane order v_mfma_f32_16x16x32_bf16 reads its operand, so a wave's fragment load is ONE coalesced KiB
    instead of 64 scattered 16-byte pieces."""
    R, Cc = w.shape
    assert R % 16 == 0 and Cc % 32 == 0
    return w.reshape(R // 16, 16, Cc // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()


X3_LO_SCALE = 2048.0


def split_f16(w: torch.Tensor):
    """float32 -> (hi, lo') float16: the "f32x3" operand format of the GPT projections (csrc/common.hpp x3_split): hi = fp16(w),
    lo' = fp16((w - hi) * 2^11), w = hi + lo' / 2^11 to 22 significant bits; inputs saturated at the fp16 range."""
    w = w.to(torch.float32).clamp(-65504.0, 65504.0)
    hi = w.to(torch.float16)
    lo = ((w - hi.to(torch.float32)) * X3_LO_SCALE).to(torch.float16)
    return hi, lo


def pack_frag_x3(w: torch.Tensor) -> torch.Tensor:
    """float32 [R, C] -> the SPLIT-fp16 planes of csrc/decode32x.hip: [2][R/16][C/32][64][8] float16, plane 0 = hi = fp16(w), plane 1 =
    lo' = fp16((w - hi) * 2^11) (w = hi + lo' / 2^11 to 22 significant bits), each plane in pack_frag's fragment order.  Three fp16 MFMAs
    (hi*hi; lo'*hi + hi*lo' on a second accumulator, x 2^-11; f32 accumulation) then stand for one float32 product."""
    hi, lo = split_f16(w)
    return torch.stack([pack_frag(hi), pack_frag(lo)], 0).contiguous()


def unpack_frag_x3(p: torch.Tensor, R: int, Cc: int) -> torch.Tensor:
    """planes -> float32 hi + lo' / 2048 (tests)"""
    return unpack_frag(p[0].float(), R, Cc) + unpack_frag(p[1].float(), R, Cc) / X3_LO_SCALE


def pack_wo_heads(wo: torch.Tensor) -> torch.Tensor:
    """o_proj.weight [768 out, 768 in] (bf16) -> [12 heads][8 (k / 8)][768 out][8] : the slice of Wo one attention head multiplies,
    laid out so that a wave's 16-byte loads (one output column per lane) are one contiguous KiB (csrc/gpt.hip attention_k<OPJ>)"""
    assert tuple(wo.shape) == (GPT.hidden, GPT.hidden)
    return wo.view(GPT.hidden, GPT.n_heads, 8, 8).permute(1, 2, 0, 3).contiguous()


def pack_frag32(w: torch.Tensor) -> torch.Tensor:
    """float32 twin of pack_frag for the parity mode (csrc/decode32.hip): [R, C] (R % 16 == 0, C % 16 == 0) ->
    [R/16][C/16][lane = (c%16)/4 * 16 + r%16][c%4], one contiguous KiB per (16-row tile, 16-column chunk) in the lane order of
    four consecutive v_mfma_f32_16x16x4_f32 steps."""
    R, Cc = w.shape
    assert R % 16 == 0 and Cc % 16 == 0
    return w.reshape(R // 16, 16, Cc // 16, 4, 4).permute(0, 2, 3, 1, 4).contiguous()


def unpack_frag32(p: torch.Tensor, R: int, Cc: int) -> torch.Tensor:
    """inverse of pack_frag32 (tests / debugging)"""
    return p.reshape(R // 16, Cc // 16, 4, 16, 4).permute(0, 3, 1, 2, 4).reshape(R, Cc).contiguous()


def unpack_frag(p: torch.Tensor, R: int, Cc: int) -> torch.Tensor:
    """inverse of pack_frag (tests / debugging)"""
    return p.reshape(R // 16, Cc // 32, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(R, Cc).contiguous()


def rope_tables(max_pos: int, head_dim: int = GPT.head_dim, theta: float = GPT.rope_theta):
    """cos/sin [max_pos, head_dim/2] float32, evaluated with the same torch f32 ops HF's
    LlamaRotaryEmbedding uses (inv_freq = 1/theta^(arange(0,d,2)/d); freqs = pos * inv_freq)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).to(torch.float32) / head_dim))
    pos = torch.arange(max_pos, dtype=torch.float32)
    freqs = pos[:, None] * inv_freq[None, :]
    return freqs.cos().contiguous(), freqs.sin().contiguous()


# ---------------------------------------------------------------------------------------------
class GptEngine:
    """Device-resident, repacked GPT weights + the generate loop."""

    NQ_RING = 64   # unseeded sampling: ring of per-step Exp(1) draws uploaded ahead of the GPU
    POLL = 16      # decode steps enqueued between two looks at the device-side finish flags
    # Stated bound on what the split-fp16 projections ("f32x3") move a PRE-temperature logit by, against the exact f32 kernels
    # under the same token history, RELATIVE to the head's logit scale (rms over the vocabulary of |W_v| * rms(final norm gain): the
    # standard deviation a logit has for a unit-rms hidden state; 4.02 for the synthetic checkpoint).  Measured with tools/x3_logit_bound.py
    # on the bench workload (all 64 rows, every step, teacher-forced on the reference's stream; profiles/r6o_x3_logit_bound_fp16split.log):
    # max |dlogit| 1.42e-5 = 3.5e-6 of the scale over 13.4 M logits (rms 3.5e-7; the bf16 split of round 5: 2.1e-5 / 3.3e-6), stated here
    # with a 1.5x allowance; re-measure on a trained checkpoint.  The certificate compares 2 * REL_ERR_X3 * scale / min(temperature) with the smallest decision margin of the call
    # (ctts_gen_state.margin).  What the same measurement says about LONG calls: the margins of the workload's 85,752 draws have density
    # ~0.8 per tempered-logit unit near 0 (smallest 1.3e-5), so ~10 draws sit below the bound: a worst-case bound flags several 500-step
    # utterances per batch although not one of the 85,752 draws actually differs between the two arithmetics (empirical rate on random
    # inputs: 1.3e-6 per draw, every divergence flagged: profiles/r6o_x3_flip_rate_fp16split.log).
    REL_ERR_X3 = 5.3e-6

    def __init__(self, gpt_sd: dict, embed_sd: dict, device: torch.device, dtype: str = "bf16",
                 max_pos: int = GPT.max_pos, logger: logging.Logger = log, rms_eps: float = GPT.rms_eps,
                 rope_theta: float = GPT.rope_theta, certify: Optional[bool] = None, exact_fallback: bool = False):
        """`dtype` names the arithmetic, explicitly:
          "f32"   -- the parity mode on float32 arithmetic throughout (f32-input MFMA, csrc/decode32.hip / prefill32.hip);
          "f32x3" -- the parity mode with the Llama projections on SPLIT-fp16 operands (csrc/decode32x.hip, prefill32x.hip: 22 significant
                     bits per operand, exact products, f32 accumulation -- measured closer to a float64 evaluation of the model than the
                     f32 MFMA kernels, profiles/r6p_f64_distance.log; attention, heads and sampling stay float32).  Same token ids as "f32" whenever no
                     draw of the call was decided by less than the arithmetic's logit error: every call computes its smallest decision
                     margin on the device (`certify`, default on in this mode; `last_stats["min_margin"]`, `["certified"]`).  With
                     `exact_fallback=True` the utterances whose margin is below the stated bound are generated again on the exact kernels
                     (both weight copies are resident): the result is then the "f32" engine's by construction.  OFF by default: the bound
                     is a worst case, long utterances almost always hold a draw below it (see REL_ERR_X3), and regenerating them costs
                     more than running "f32" in the first place -- a caller that needs the guarantee on long calls loads "f32";
          "bf16"  -- the perf mode (bf16 weights / KV / activations).
        `rms_eps` / `rope_theta` / `max_pos`: the run-time fields of `asset/gpt/config.json` (weights.check_gpt_config;
        `LlamaModel.from_pretrained`, gpt.py:75); everything else in that file is the geometry the kernels are compiled for."""
        if dtype not in ("bf16", "f32", "f32x3"):
            raise ValueError("dtype must be 'bf16' (perf), 'f32' (parity, exact float32) or 'f32x3' (parity, split-fp16 projections)")
        self.lib = _lib.lib()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.EngineError("GptEngine needs a ROCm GPU device (there is no CPU path)")
        self.logger = logger
        self.dtype = dtype
        self.wdt = torch.bfloat16 if dtype == "bf16" else torch.float32
        self.code = _lib.BF16 if dtype == "bf16" else _lib.F32
        self.n_layers = gpt_layer_count(gpt_sd)
        self.max_pos = max_pos
        dev = self.device
        f = lambda t: t.to(torch.float32).contiguous().to(dev)
        wcast = lambda t: t.to(torch.float32).to(self.wdt).contiguous().to(dev)
        self.wqkv, self.wo, self.wgu, self.wd, self.ln1, self.ln2 = [], [], [], [], [], []
        # perf mode folds the RMSNorm gains into the following projection (W' = W * gain[None, :], rounded to bf16
        # once at load): the kernels then only need the per-row 1/rms, applied in the GEMM epilogue.
        fold = (lambda w, g: w.float() * g.float()[None, :]) if dtype == "bf16" else (lambda w, g: w)
        perm = rope_row_perm() if dtype == "bf16" else torch.arange(GPT.hidden)
        for i in range(self.n_layers):
            p = f"layers.{i}."
            ln1, ln2 = gpt_sd[p + "input_layernorm.weight"], gpt_sd[p + "post_attention_layernorm.weight"]
            self.wqkv.append(wcast(fold(torch.cat([gpt_sd[p + "self_attn.q_proj.weight"][perm], gpt_sd[p + "self_attn.k_proj.weight"][perm],
                                                   gpt_sd[p + "self_attn.v_proj.weight"]], 0), ln1)))
            self.wo.append(wcast(gpt_sd[p + "self_attn.o_proj.weight"]))
            self.wgu.append(wcast(fold(torch.cat([gpt_sd[p + "mlp.gate_proj.weight"], gpt_sd[p + "mlp.up_proj.weight"]], 0), ln2)))
            self.wd.append(wcast(gpt_sd[p + "mlp.down_proj.weight"]))
            self.ln1.append(f(gpt_sd[p + "input_layernorm.weight"]))
            self.ln2.append(f(gpt_sd[p + "post_attention_layernorm.weight"]))
        self.norm = f(gpt_sd["norm.weight"])
        self.emb_code = f(torch.stack([embed_sd[f"emb_code.{k}.weight"] for k in range(GPT.n_vq)], 0))
        self.emb_text = f(embed_sd["emb_text.weight"])
        self.heads = f(torch.cat([fold_weight_norm(embed_sd[f"head_code.{k}.parametrizations.weight.original0"].float(),
                                                   embed_sd[f"head_code.{k}.parametrizations.weight.original1"].float())
                                  for k in range(GPT.n_vq)], 0))
        self.head_text = f(fold_weight_norm(embed_sd["head_text.parametrizations.weight.original0"].float(),
                                            embed_sd["head_text.parametrizations.weight.original1"].float()))
        # logit scale of the two heads (see REL_ERR_X3): rms_v |W_v|_2 * rms(final norm gain)
        ng = float(self.norm.pow(2).mean().sqrt())
        self.logit_scale = {False: float(self.heads.pow(2).sum(1).mean().sqrt()) * ng, True: float(self.head_text.pow(2).sum(1).mean().sqrt()) * ng}
        cos, sin = rope_tables(max_pos, theta=rope_theta)
        self.rope_cos, self.rope_sin = cos.to(dev), sin.to(dev)
        self._arrs = [_lib.ptr_array(x) for x in (self.wqkv, self.wo, self.wgu, self.wd, self.ln1, self.ln2)]
        # a second copy of the four matrices in the fragment-packed order the DECODE kernels read (bf16: decode.hip, f32:
        # decode32.hip); the row-major copy stays for the prefill kernels (+0.38 GB / +0.75 GB of the 288 GB)
        # (CTTS_DEC_PACKED=0, the A/B switch back to the row-major decode kernels, skips building them)
        use_packed = os.environ.get("CTTS_DEC_PACKED", "1") != "0"
        self.certify = (dtype == "f32x3") if certify is None else bool(certify)
        self.exact_fallback = bool(exact_fallback) and dtype == "f32x3"
        pk = pack_frag if dtype == "bf16" else pack_frag32
        self.packed, self._pk_arrs = None, None
        rp = rope_row_perm().to(dev)
        qk_perm = torch.cat([rp, GPT.hidden + rp, 2 * GPT.hidden + torch.arange(GPT.hidden, device=dev)])
        # "f32x3" reads the packed f32 copies only on an exact call (the fallback of its certificate, generate(exact=True)): without
        # `exact_fallback` they are not built (0.75 GB, and the packing time at load) -- an exact call then runs the row-major f32
        # kernels, bit-identical results (test_packed_f32_decode_is_bit_identical_to_row_major)
        if use_packed and (dtype != "f32x3" or self.exact_fallback):
            self.packed = [[pk(t) for t in ws] for ws in (self.wqkv, self.wo, self.wgu, self.wd)]
            if dtype != "bf16":
                # the packed f32 QKV matrix carries the RoPE row permutation too (its decode kernel rotates in the epilogue); an
                # output column's dot product does not depend on where the column sits, so the parity arithmetic is untouched
                self.packed[0] = [pack_frag32(t[qk_perm]) for t in self.wqkv]
            self._pk_arrs = [_lib.ptr_array(x) for x in self.packed]
        # dtype "f32x3": decode step on split-fp16 operands (csrc/decode32x.hip): the four matrices once more as hi | lo' fp16 planes (the
        # float32 bytes again), the RMSNorm gains folded in before the split; the prompt pass splits the row-major f32 matrices in its
        # tile loader (csrc/prefill32x.hip)
        self.x3, self._x3_arrs = None, None
        if use_packed and dtype == "f32x3":
            gain = lambda w_, g_: w_.float() * g_.float()[None, :]
            self.x3 = [[pack_frag_x3(gain(t[qk_perm], g_)) for t, g_ in zip(self.wqkv, self.ln1)], [pack_frag_x3(t) for t in self.wo],
                       [pack_frag_x3(gain(t, g_)) for t, g_ in zip(self.wgu, self.ln2)], [pack_frag_x3(t) for t in self.wd]]
            self._x3_arrs = [_lib.ptr_array(x) for x in self.x3]
        # perf mode: o_proj once more, sliced per attention head -- the decode step folds o_proj + residual into the attention launch
        # a second, per-head copy of every o_proj (23.6 MB of bf16) only for the opt-in fused attention + o_proj launch (CTTS_ATT_OPROJ=1)
        self.wo_hd = [pack_wo_heads(t) for t in self.wo] if (use_packed and dtype == "bf16" and os.environ.get("CTTS_ATT_OPROJ") == "1") else None
        self._wo_hd_arr = None if self.wo_hd is None else _lib.ptr_array(self.wo_hd)

        def pad16(t):      # zero rows up to a multiple of 16: the padded columns of the last logits tile are never stored
            r = (-t.shape[0]) % 16
            return t if r == 0 else torch.cat([t, torch.zeros((r, t.shape[1]), dtype=t.dtype, device=t.device)], 0)
        self.heads_pk = pack_frag32(pad16(self.heads)) if use_packed else None
        self.head_text_pk = pack_frag32(pad16(self.head_text)) if use_packed else None
        w = _lib.GptWeights()
        w.n_layers, w.weight_dtype, w.kv_dtype, w.max_pos = self.n_layers, self.code, self.code, max_pos
        w.wqkv, w.wo, w.wgu, w.wd, w.ln1, w.ln2 = [C.cast(a, _lib.PP) for a in self._arrs]
        w.norm, w.emb_code, w.heads = self.norm.data_ptr(), self.emb_code.data_ptr(), self.heads.data_ptr()
        w.rope_cos, w.rope_sin = self.rope_cos.data_ptr(), self.rope_sin.data_ptr()
        w.rms_eps = float(rms_eps)
        w.emb_text, w.head_text, w.n_text = self.emb_text.data_ptr(), self.head_text.data_ptr(), GPT.n_text
        if self._pk_arrs is not None:
            w.wqkv_pk, w.wo_pk, w.wgu_pk, w.wd_pk = [C.cast(a, _lib.PP) for a in self._pk_arrs]
        w.heads_pk, w.head_text_pk = _lib.ptr(self.heads_pk), _lib.ptr(self.head_text_pk)
        if self._wo_hd_arr is not None:
            w.wo_hd = C.cast(self._wo_hd_arr, _lib.PP)
        if self._x3_arrs is not None:
            w.wqkv_x3, w.wo_x3, w.wgu_x3, w.wd_x3 = [C.cast(a, _lib.PP) for a in self._x3_arrs]
        self._w = w
        h = C.c_void_p()
        _lib.check(self.lib.ctts_gpt_create(C.byref(h), C.byref(w)), "ctts_gpt_create")
        self.handle = h
        # CTTS_GPT_COMPLEMENT=1 (with CTTS_CODEC_CUS): the generator's stream runs on every CU EXCEPT the acoustic decoder's side-stream CUs
        spec = codec_cu_spec() if os.environ.get("CTTS_GPT_COMPLEMENT") == "1" else None
        self.stream = cu_masked_stream(self.lib, dev, *spec, True) if spec else torch.cuda.Stream(device=dev)
        self._lane_res = [(self.handle, self.stream)]
        self._lane_res_alt = []
        self._lane_res_text = []
        self.default_lanes = 1
        self.rng = "host"         # default source of the multinomial's Exp(1) draws: "host" (the reference's CPU stream) | "device"
        self.last_stats = {}
        self._session = None      # buffers + instantiated graph of the last generate() geometry (see generate)
        self._session_alt = None  # ... and of the last exact re-run of certificate-flagged utterances (kept apart: it must not evict the main one)
        self._session_text = None  # ... and of the last refine-text call: the default `Chat.infer` alternates text and code calls (core.py:341-360)
        self._draws_cache = None  # (key, ExpDraws) of the last seeded call: the constant Exp(1) tensor

    def __del__(self):
        try:
            for h, _ in [*getattr(self, "_lane_res", []), *getattr(self, "_lane_res_alt", []), *getattr(self, "_lane_res_text", [])]:
                self.lib.ctts_gpt_destroy(h)
            self._lane_res, self._lane_res_alt, self._lane_res_text = [], [], []
            self.handle = None
        except Exception:
            pass

    def weight_bytes_per_step(self) -> int:
        """HBM bytes of weights one decode step streams (190,698,240 params)."""
        es = 2 if self.dtype == "bf16" else 4
        per_layer = (3 * 768 * 768 + 768 * 768 + 2 * 3072 * 768 + 768 * 3072) * es
        return per_layer * self.n_layers + self.heads.numel() * 4

    # -- a2: Embed.forward (embed.py:52-79): gathers only, done with torch indexing on the device
    def embed_prompt(self, input_ids: torch.Tensor, text_mask: torch.Tensor) -> torch.Tensor:
        ids = input_ids.to(self.device)
        tm = text_mask.to(self.device).bool()
        et = self.emb_text[ids[..., 0].clamp(0, GPT.n_text - 1)]
        cid = ids.clamp(0, GPT.n_audio - 1)
        ec = self.emb_code[0][cid[..., 0]]
        for k in range(1, GPT.n_vq):
            ec = ec + self.emb_code[k][cid[..., k]]
        return torch.where(tm[..., None], et, ec).contiguous()

    # -- a3: GPT.generate
    def _lane_resources(self, n: int, pool: str = ""):
        """(handle, stream) pairs for n concurrent lanes; lane 0 of the main pool is the engine's own handle/stream.  `pool`: "" (code
        mode), "_alt" (the exact re-run of flagged utterances), "_text" (refine-text) -- a handle owns ONE captured decode graph, so every
        session slot decodes on handles of its own and none of them replaces another's graph."""
        pool = getattr(self, "_lane_res" + pool)
        while len(pool) < n:
            h = C.c_void_p()
            _lib.check(self.lib.ctts_gpt_create(C.byref(h), C.byref(self._w)), "ctts_gpt_create")
            pool.append((h, torch.cuda.Stream(device=self.device)))
        return pool[:n]

    def generate(self, emb: torch.Tensor, inputs_ids: torch.Tensor, temperature: torch.Tensor, eos_token: int = GPT.n_audio - 1,
                 attention_mask: Optional[torch.Tensor] = None, max_new_token: int = 2048, min_new_token: int = 0,
                 logits_processors: Sequence = (), infer_text: bool = False, return_attn: bool = False,
                 return_hidden: bool = False, stream: bool = False, show_tqdm: bool = False, ensure_non_empty: bool = True,
                 stream_batch: int = 24, manual_seed: Optional[int] = None, context: Optional[Context] = None,
                 *, use_graph: bool = True, stop_at: Optional[torch.Tensor] = None, row_offset: int = 0,
                 total_rows: Optional[int] = None, profile_tag: Optional[int] = None,
                 profile_stride: int = 1, lanes: Optional[int] = None,
                 teacher_ids: Optional[torch.Tensor] = None, prefill_chunk: Optional[int] = None,
                 return_sampled: bool = False, rng: Optional[str] = None, rng_seed: Optional[int] = None,
                 rng_nonce=None, row_ids: Optional[torch.Tensor] = None, exact: bool = False) -> Iterator[GenerationOutputs]:
        """Drop-in for `GPT.generate` (gpt.py:316-337), code mode.  Extra keyword-only arguments:
        `use_graph` (hipGraph replay of the decode step), `stop_at` ([B] int32 forced output lengths,
        benchmark hook), `row_offset`/`total_rows` (this shard's position inside a data-parallel batch:
        keeps the CPU draw and the rows>=625 penalty quirk keyed on the global row index), `lanes`
        (the batch is cut into that many contiguous row groups, each decoding on its own HIP stream
        with its own captured graph; utterances never interact, so the result is identical, while the
        per-kernel launch / dependent-load latency of one lane overlaps the others' kernels), `teacher_ids`
        ([B, max_new_token, 4] int64: teacher forcing -- the token written at step i is teacher_ids[:, i] instead of
        the sampled one; evaluation hook used to bound the bf16 mode's drift on the reference's token stream),
        `prefill_chunk` (tokens: the prompt is prefilled in pieces of that many slots -- `ctts_gpt_prefill_chunk` -- which
        bounds the activation workspace of a long prompt such as an `spk_smp` audio-code prompt, core.py:435-453: the workspace is
        then sized for a piece, not for the whole prompt), `return_sampled` (evaluation hook, the companion of `teacher_ids`: after
        the call `self.last_sampled` holds, per utterance, the [T_b, 4] tokens the sampler itself drew at every step before teacher
        forcing replaced them -- the teacher-forced token agreement of a numeric mode), `rng` ("host" | "device", default
        `self.rng` = "host"): where the Exp(1) draws of the multinomial come from.  "host" is the parity contract -- the reference's
        CPU generator call, uploaded (rng.py).  "device" draws them inside the sampling kernel (Philox4x32-10, counter = token / global
        row / step; the reference on a GPU device draws from the device generator too, gpt.py:39): with `manual_seed=None` -- the
        reference's DEFAULT -- that removes the ~2 ms per-step host draw + upload; the key is `rng_seed` or, if None, one draw from
        torch's global CPU generator (so `torch.manual_seed` still makes a run repeatable).  Code mode only.  `rng_nonce` (device generator
        only; an int or one int per utterance): the fourth word of the generator's counter instead of its constant -- a slot pool gives
        every admission its own (serving.SlotPool), and passing a request's number here reproduces that request in isolation.
        `row_ids` ([B] ints, with `total_rows`): the GLOBAL utterance index of every row of this call when they are not the contiguous
        block `row_offset` describes (a length-balanced data-parallel shard, dist.infer_sharded; the exact re-run below): the Exp(1) draw
        of row b is the full-batch draw's row, and the rows >= 625 penalty quirk / device generator are keyed on it.  `exact` (dtype
        "f32x3" only): this call's decode steps on the exact f32 kernels.
        PARITY CERTIFICATE (`self.certify`, default in "f32x3"): after the call `last_stats["min_margin"]` is the smallest decision margin
        of any step of any utterance in tempered-logit units (include/chattts_amd.h, ctts_gen_state.margin), `last_stats["margin_bound"]` =
        2 * REL_ERR_X3 * logit_scale / min(temperature), `last_margins` the per-utterance minima.  Utterances below the bound: a warning, and
        (`self.exact_fallback`, non-streaming calls) they are generated again on the exact kernels and replace their rows of the result."""
        if return_attn:
            raise NotImplementedError("return_attn is not supported by the fused attention kernel")
        context = context or Context()
        plan = plan_from_processors(logits_processors, infer_text)
        lib, dev = self.lib, self.device
        B, T, nvq = inputs_ids.shape
        assert nvq == GPT.n_vq and emb.shape == (B, T, GPT.hidden)
        nrow = 1 if infer_text else nvq                   # sampling rows per utterance (gpt.py:459-464 vs :439-440)
        V = GPT.n_text if infer_text else GPT.n_audio
        max_new = int(max_new_token)
        if T + max_new > self.max_pos:
            raise ValueError(f"T + max_new_token = {T + max_new} exceeds max_position_embeddings {self.max_pos}")
        if attention_mask is None:
            attention_mask = torch.ones((B, T), dtype=torch.bool)
        kv_start_all = left_pad_starts(attention_mask)
        n_lanes = self.default_lanes if lanes is None else int(lanes)
        n_lanes = max(1, min(n_lanes, B))
        exact = bool(exact) and self.x3 is not None      # (the other modes have one arithmetic)
        certify = self.certify and teacher_ids is None
        rng_state0 = torch.get_rng_state() if (manual_seed is None and self.exact_fallback and not exact) else None
        rid = None
        if row_ids is not None:
            rid = torch.as_tensor(row_ids, dtype=torch.int64).reshape(-1).cpu()
            assert rid.numel() == B and total_rows is not None, "row_ids needs one global index per utterance and total_rows"
        if profile_tag is not None or not use_graph:
            n_lanes = 1
        from .dist import shard_bounds
        bounds = [shard_bounds(B, n_lanes, i) for i in range(n_lanes)]
        slot_sfx = "_alt" if exact else ("_text" if infer_text else "")
        res = self._lane_resources(n_lanes, slot_sfx)
        caller = torch.cuda.current_stream(dev)
        # seeded sampling re-seeds the CPU generator at every step (gpt.py:504-507): ONE constant tensor per (seed, batch
        # geometry).  Drawing it costs ~4 ms of host time per 256 rows (30 ms for the 2048 rows of an 8-GPU batch), so the
        # last one is kept across calls.
        rng_mode = rng or self.rng
        if rng_mode not in ("host", "device"):
            raise ValueError("rng must be 'host' or 'device'")
        device_rng = rng_mode == "device" and not infer_text     # refine-text keeps the host stream
        dkey = (total_rows if total_rows is not None else B * nrow, V, manual_seed, row_offset, B * nrow, None if rid is None else tuple(rid.tolist()))
        if device_rng:
            draws = None
            seed_val = int(rng_seed) if rng_seed is not None else (int(manual_seed) if manual_seed is not None
                                                                    else int(torch.randint(0, 2 ** 62, (1,)).item()))
        elif manual_seed is not None and self._draws_cache is not None and self._draws_cache[0] == dkey:
            draws = self._draws_cache[1]
        else:
            srows = None if rid is None else (rid[:, None] * nrow + torch.arange(nrow)[None, :]).reshape(-1)
            draws = ExpDraws(dkey[0], V, manual_seed, row_begin=row_offset, row_end=row_offset + B * nrow, rows=srows)
            self._draws_cache = (dkey, draws) if draws.constant else None
        const_q = device_rng or draws.constant      # no per-step host draw / upload
        ptab = penalty_table(plan.penalty)
        nq = 1 if const_q else self.NQ_RING
        emb_all = emb.to(torch.float32).contiguous().to(dev)
        ids_all = inputs_ids.to(dev)
        temp_d = temperature.to(torch.float32).reshape(-1).to(dev)
        assert temp_d.numel() == nrow, "temperature must have one entry per sampling row of an utterance"
        ptab_d = None if ptab is None else ptab.to(dev)
        wait_stream(caller)  # inputs above were produced on the caller's stream

        class Lane:
            pass

        # ---- session reuse: a call with the same geometry and sampling constants as the previous one keeps that call's
        # device buffers AND its instantiated decode graph (every pointer and by-value field the captured kernels hold is
        # then unchanged); only buffer contents are re-initialised, stream-ordered on the lane's stream.  The first
        # replays of a freshly instantiated graph cost ~17 ms at B = 64 (tools/ttfs_probe.py), which a serving loop or a
        # second batch of the same shape should not pay again.  Results are handed out as copies (see outputs()).
        top_p_thr = float(np.float32(1.0 - plan.top_p)) if plan.top_p is not None else 0.0
        key = (tuple(bounds), T, max_new, nrow, V, nq, self.dtype, top_p_thr, plan.top_p is not None, int(plan.top_k or 0),
               plan.top_k is not None, int(min_new_token), int(eos_token), int(row_offset), bool(infer_text), stop_at is not None,
               ptab is not None, teacher_ids is not None, bool(return_sampled),
               None if prefill_chunk is None else int(prefill_chunk), device_rng, manual_seed is None, device_rng and rng_nonce is not None,
               exact, certify, rid is not None)
        # three slots: code mode, the exact re-run, refine-text -- an exact re-run keeps the main call's session (and graph) alive, and
        # the text / code calls the default `Chat.infer` alternates between each find theirs from the previous request
        slot = "_session" + slot_sfx
        sess = getattr(self, slot)
        sess = sess if (sess is not None and sess["key"] == key) else None
        if sess is None:
            setattr(self, slot, None)     # drop the previous session's buffers before allocating new ones
            sess = dict(key=key, lanes=[], graph=False, temp=torch.empty((nrow,), dtype=torch.float32, device=dev),
                        ptab=None if ptab is None else torch.empty_like(ptab, device=dev), q_sig=None,
                        seed=torch.zeros((1,), dtype=torch.int64, device=dev))
            for (lo, hi), (handle, st) in zip(bounds, res):
                ln = Lane()
                ln.lo, ln.hi, ln.handle, ln.st = lo, hi, handle, st
                Bl = hi - lo
                with torch.cuda.stream(st):
                    ln.ids_buf = torch.empty((Bl, T + max_new, nvq), dtype=torch.int64, device=dev)  # gpt.py:372-379
                    ln.len_d = torch.empty((Bl,), dtype=torch.int32, device=dev)
                    # finish flags and end_idx live in ONE padded device block (16-byte granules: [Bp] uint8 | [Bp] int32) so that a
                    # poll is one shader copy of it into pinned host memory (ctts_copy_bytes) -- see snapshot()
                    Bp = (Bl + 15) // 16 * 16
                    ln.state_blk = torch.zeros((5 * Bp,), dtype=torch.uint8, device=dev)
                    ln.finish = ln.state_blk[:Bl]                                             # gpt.py:346
                    ln.order = torch.empty((Bl,), dtype=torch.int32, device=dev)              # visiting order of the device-side compaction
                    ln.n_active = torch.empty((1,), dtype=torch.int32, device=dev)
                    ln.end_idx = ln.state_blk[Bp:].view(torch.int32)[:Bl]                     # gpt.py:343
                    ln.hiddens = torch.empty((Bl, max_new, GPT.hidden), dtype=torch.float32, device=dev)
                    kv_shape = (self.n_layers, Bl, GPT.n_heads, T + max_new, GPT.head_dim)
                    ln.kcache = torch.empty(kv_shape, dtype=self.wdt, device=dev)
                    ln.vcache = torch.empty(kv_shape, dtype=self.wdt, device=dev)
                    # the prefill carves for the rows it pushes through the layers at once (the whole prompt, or one chunk of it);
                    # the decode steps for one row per utterance
                    tc_ws = T if (prefill_chunk is None or int(prefill_chunk) >= T) else max(1, int(prefill_chunk))
                    ln.ws_bytes = max(lib.ctts_gpt_workspace_bytes(Bl, tc_ws), lib.ctts_gpt_workspace_bytes(Bl, 1))
                    ln.workspace = torch.empty((ln.ws_bytes,), dtype=torch.uint8, device=dev)
                    ln.kv_start = torch.empty((Bl,), dtype=kv_start_all.dtype, device=dev)
                    ln.stop_d = None if stop_at is None else torch.empty((Bl,), dtype=torch.int32, device=dev)
                    ln.q_d = torch.empty((1,) if device_rng else (nq, Bl * nrow, V), dtype=torch.float32, device=dev)
                    ln.teacher = None if teacher_ids is None else torch.empty((Bl, max_new, nvq), dtype=torch.int64, device=dev)
                    ln.sampled = torch.zeros((Bl, max_new, nvq), dtype=torch.int64, device=dev) if return_sampled else None
                    ln.nonce = torch.zeros((Bl,), dtype=torch.int32, device=dev) if (device_rng and rng_nonce is not None) else None
                    ln.margin = torch.empty((Bl,), dtype=torch.float32, device=dev) if certify else None
                    ln.row_base = torch.empty((Bl,), dtype=torch.int32, device=dev) if rid is not None else None
                s = _lib.GenState()
                s.B, s.T, s.max_new = Bl, T, max_new
                s.ids_buf, s.len, s.kv_start = ln.ids_buf.data_ptr(), ln.len_d.data_ptr(), ln.kv_start.data_ptr()
                s.finish, s.end_idx, s.hiddens = ln.finish.data_ptr(), ln.end_idx.data_ptr(), ln.hiddens.data_ptr()
                s.kcache, s.vcache, s.q, s.nq = ln.kcache.data_ptr(), ln.vcache.data_ptr(), ln.q_d.data_ptr(), nq
                s.temperature = sess["temp"].data_ptr()
                s.pow_table = _lib.ptr(sess["ptab"])
                s.top_p_thr = top_p_thr
                s.use_top_p = int(plan.top_p is not None)
                s.top_k = int(plan.top_k or 0)
                s.use_top_k = int(plan.top_k is not None)
                s.min_new, s.eos = int(min_new_token), int(eos_token)
                s.row_offset = int(row_offset + lo * nrow)
                s.infer_text = int(infer_text)
                s.stop_at = _lib.ptr(ln.stop_d)
                s.teacher_ids = _lib.ptr(ln.teacher)
                s.sampled_ids = _lib.ptr(ln.sampled)
                s.rng_device, s.rng_per_step, s.rng_seed = int(device_rng), int(manual_seed is None), sess["seed"].data_ptr()
                s.rng_nonce = _lib.ptr(ln.nonce)
                s.margin, s.row_base, s.proj_exact = _lib.ptr(ln.margin), _lib.ptr(ln.row_base), int(exact)
                s.prefill_valid_rows = 0      # set per call below (the session is keyed on geometry, not on the mask)
                # compaction order: utterances by descending context = ascending left padding (contexts of a batch differ only by
                # the static valid prompt length), so the attention grid starts its longest units first.  CTTS_ORDER=0: ascending slot
                s.order = ln.order.data_ptr() if os.environ.get("CTTS_ORDER", "1") != "0" else None
                s.workspace, s.workspace_bytes = ln.workspace.data_ptr(), ln.ws_bytes
                # no row_map: the library compacts on the device -- the first kernel of every decode step ranks the utterances
                # whose finish flag is 0 and writes n_active (include/chattts_amd.h), so finished utterances leave the step
                # at once and the host only ever READS finish / end_idx
                s.row_map, s.n_active = None, ln.n_active.data_ptr()
                ln.s = s
                sess["lanes"].append(ln)
            setattr(self, slot, sess)
        L = sess["lanes"]
        q_sig = dkey if (draws is not None and draws.constant) else None
        for ln in L:
            lo, hi = ln.lo, ln.hi
            Bl = hi - lo
            ln.done, ln.end_snap = False, None
            ln.live_ub = Bl      # upper bound of the lane's live utterances: what the last collected finish poll saw (flags only go up)
            with torch.cuda.stream(ln.st):
                if sess.get("out_ev") is not None:
                    ln.st.wait_event(sess["out_ev"])   # the previous call's result copies have read these buffers
                ln.ids_buf.zero_()
                ln.ids_buf[:, :T] = ids_all[lo:hi]
                ln.len_d.fill_(T)
                ln.finish.zero_()
                ln.order.copy_(torch.argsort(kv_start_all[lo:hi].to(torch.int64), stable=True).to(torch.int32))
                ln.n_active.fill_(Bl)
                ln.end_idx.zero_()
                ln.kv_start.copy_(kv_start_all[lo:hi])
                # "f32x3": the prompt pass runs over the valid prompt tokens only (ctts_gen_state.prefill_valid_rows; whole-prompt prefill only)
                ln.s.prefill_valid_rows = int((T - kv_start_all[lo:hi].to(torch.int64)).sum()) if self.x3 is not None else 0
                if ln.stop_d is not None:
                    ln.stop_d.copy_(stop_at[lo:hi].to(torch.int32))
                if ln.margin is not None:
                    ln.margin.fill_(float("inf"))
                if ln.row_base is not None:
                    ln.row_base.copy_((rid[lo:hi] * nrow).to(torch.int32))
                if getattr(ln, "nonce", None) is not None:
                    nz = torch.as_tensor(rng_nonce, dtype=torch.int64).reshape(-1)
                    nz = nz.expand(B) if nz.numel() == 1 else nz
                    ln.nonce.copy_(nz[lo:hi].to(torch.int32))
                if ln.teacher is not None:
                    assert tuple(teacher_ids.shape) == (B, max_new, nvq)
                    ln.teacher.copy_(teacher_ids[lo:hi].to(torch.int64))
                ln.emb = emb_all[lo:hi].contiguous()
                if draws is not None and draws.constant and sess["q_sig"] != q_sig:
                    ln.q_d[0].copy_(draws.step(0)[lo * nrow: hi * nrow])
                if ln is L[0]:
                    if device_rng:
                        sess["seed"].fill_(seed_val)
                    sess["temp"].copy_(temp_d)
                    if ptab is not None:
                        sess["ptab"].copy_(ptab)
        sess["q_sig"] = q_sig
        if n_lanes > 1:   # temp / ptab were written on lane 0's stream
            ev0 = torch.cuda.Event()
            ev0.record(L[0].st)
            for ln in L[1:]:
                ln.st.wait_event(ev0)

        # ---- Exp(1) draws of the unseeded mode: the reference draws one [rows, V] tensor per step from torch's global
        # CPU generator (gpt.py:498-500).  That stream is serial (~2 ms per step at B=64, more than a decode step), so
        # a worker thread draws 32-step blocks ahead into two pinned buffers while the GPU consumes the previous block;
        # uploads are stream-ordered behind the launches that still read the ring half they overwrite (no host sync).
        half = nq // 2
        feeder = None
        if not const_q:
            from concurrent.futures import ThreadPoolExecutor
            feeder = dict(pool=ThreadPoolExecutor(max_workers=1), bufs=[torch.empty((half, B * nrow, V), dtype=torch.float32).pin_memory()
                                                                           for _ in range(2)],
                          evs=[None, None], fut=None, next_block=0, states=[])

            def draw_block(blk):
                buf = feeder["bufs"][blk % 2]
                if feeder["evs"][blk % 2] is not None:
                    e_ = feeder["evs"][blk % 2]                 # the H2D copy that last read this buffer is done
                    e_.synchronize() if isinstance(e_, _EventGroup) else wait_event(e_)
                feeder["states"].append((blk, torch.get_rng_state()))
                nthr = torch.get_num_threads()
                torch.set_num_threads(1)   # exponential_ is a serial stream; a 100+-thread OpenMP wake-up costs ~10 ms per call
                try:
                    n = min(half, max_new - blk * half)
                    for j in range(max(n, 0)):
                        draws.step_into(blk * half + j, buf[j])
                finally:
                    torch.set_num_threads(nthr)
                return blk

            feeder["fut"] = feeder["pool"].submit(draw_block, 0)
        uploaded = 0  # steps whose Exp(1) draws are on the device (unseeded mode only)

        def ensure_q(upto: int):
            nonlocal uploaded
            if feeder is None:
                return
            while uploaded < upto:
                blk = feeder["fut"].result()
                n = min(half, max_new - uploaded)
                buf = feeder["bufs"][blk % 2]
                slab = uploaded % nq
                evs = []
                for ln in L:
                    with torch.cuda.stream(ln.st):
                        ln.q_d[slab: slab + n].copy_(buf[:n, ln.lo * nrow: ln.hi * nrow], non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(ln.st)
                        evs.append(ev)
                feeder["evs"][blk % 2] = evs[-1] if len(evs) == 1 else _EventGroup(evs)
                uploaded += n
                if uploaded < max_new:
                    feeder["fut"] = feeder["pool"].submit(draw_block, blk + 1)
                else:
                    feeder["fut"] = None

        def finish_rng(steps_reference: int):
            """Leave torch's global CPU generator exactly where the reference would: it draws once per executed step
            (the loop of gpt.py:394 stops at the step where the last row hits EOS), while the feeder ran ahead."""
            if feeder is None:
                return
            if feeder["fut"] is not None:
                feeder["fut"].result()
            feeder["pool"].shutdown(wait=True)
            blk = min(steps_reference // half, len(feeder["states"]) - 1)
            torch.set_rng_state(feeder["states"][blk][1])
            tmp = torch.empty((draws.total_rows, V), dtype=torch.float32)
            for _ in range(steps_reference - blk * half):
                tmp.exponential_(1)

        def outputs() -> GenerationOutputs:
            """Snapshot of the result so far.  Uses each lane's `end_snap` (host copy of end_idx taken by the last
            poll) and makes the caller's stream wait for the poll-time event only, so a chunk that was already
            enqueued behind the poll keeps running while the consumer (e.g. the DVAE/Vocos decode of a streamed
            prefix) works on the caller's stream -- the two overlap on the GPU."""
            ids, hid = [], []
            # copies, on the caller's stream behind the poll-time event: the session's buffers are reused by the next
            # call of the same geometry, results must not alias them
            with torch.cuda.stream(caller):
                if len(L) == 1 and max(L[0].end_snap, default=0) > 0:
                    # one lane (the default): ONE strided copy per result instead of one clone per row (128 launches for a batch of 64,
                    # ~1.5 ms of host time between the last decode step and the acoustic decoder); rows are views of the copies, the
                    # hidden states come zero padded -- the batch the decoder needs (RowList.padded)
                    ln = L[0]
                    e, nb = ln.end_snap, ln.hi - ln.lo
                    Tm = max(e)
                    caller.wait_event(ln.ev)
                    ids_all = (ln.ids_buf[:nb, T: T + Tm, 0] if infer_text else ln.ids_buf[:nb, T: T + Tm]).clone()   # gpt.py:297-301
                    ids = [ids_all[b, : e[b]] for b in range(nb)]
                    hid = RowList()
                    if return_hidden:                                                                                 # gpt.py:303-307
                        e_d = torch.tensor(e, dtype=torch.int32).to(dev, non_blocking=True)
                        keep = torch.arange(Tm, device=dev, dtype=torch.int32)[None, :] < e_d[:, None]
                        hid.padded = torch.where(keep[..., None], ln.hiddens[:nb, :Tm], ln.hiddens.new_zeros(()))
                        hid.extend(hid.padded[b, : e[b]] for b in range(nb))
                    sess["out_ev"] = torch.cuda.Event()
                    sess["out_ev"].record(caller)
                    return GenerationOutputs(ids=ids, attentions=[], hiddens=hid)
                for ln in L:
                    e = ln.end_snap
                    caller.wait_event(ln.ev)
                    if infer_text:
                        ids += [ln.ids_buf[b, T: T + e[b], 0].clone() for b in range(ln.hi - ln.lo)]   # gpt.py:300-301
                    else:
                        ids += [ln.ids_buf[b, T: T + e[b]].clone() for b in range(ln.hi - ln.lo)]      # gpt.py:297-299
                    if return_hidden:
                        hid += [ln.hiddens[b, : e[b]].clone() for b in range(ln.hi - ln.lo)]           # gpt.py:303-307
                sess["out_ev"] = torch.cuda.Event()
                sess["out_ev"].record(caller)
            return GenerationOutputs(ids=ids, attentions=[], hiddens=hid)

        sync_poll = os.environ.get("CTTS_SYNC_POLL") == "1"   # profiling fallback: plain blocking D2H polls, no pinned buffers

        def snapshot(ln):
            """stream-ordered, ASYNCHRONOUS D2H of the lane's finish flags and end_idx into pinned host buffers, plus the event
            that marks the copy (and everything enqueued before it) done.  Two buffer sets: up to two chunks are in flight."""
            k = ln.snap_i = (getattr(ln, "snap_i", 0) + 1) % 2
            Bl_, Bp_ = ln.hi - ln.lo, ln.state_blk.numel() // 5
            if not hasattr(ln, "snap_buf"):
                ln.snap_buf = []
                for _ in range(2):
                    blk = torch.empty((5 * Bp_,), dtype=torch.uint8).pin_memory()
                    ln.snap_buf.append((blk[:Bl_], blk[Bp_:].view(torch.int32)[:Bl_], torch.cuda.Event(), blk))
            fin_h, end_h, ev, blk = ln.snap_buf[k]
            with torch.cuda.stream(ln.st):
                if sync_poll:
                    fin_h.copy_(ln.finish.cpu())
                    end_h.copy_(ln.end_idx.cpu())
                else:
                    # a SHADER copy into the pinned block (one packet of the stream; no copy-engine hand-off behind the step's kernels)
                    _lib.check(lib.ctts_copy_bytes(blk.data_ptr(), ln.state_blk.data_ptr(), blk.numel(), ln.st.cuda_stream), "ctts_copy_bytes")
                ev.record(ln.st)
            return k

        def collect(ln, k):
            """wait for snapshot k; the reference keeps stepping finished rows until the last one is done (gpt.py:512-518,592)
            but cuts their output at end_idx -- here they left the step on the device the moment they finished"""
            fin_h, end_h, ev, _ = ln.snap_buf[k]
            wait_event(ev)
            ln.ev = ev
            ln.end_snap = end_h.tolist()
            ln.done = bool(fin_h.all())
            ln.live_ub = (ln.hi - ln.lo) - int(fin_h.sum())
            return fin_h.clone()

        def poll(ln):
            return collect(ln, snapshot(ln))

        def enqueue(n):
            """n decode steps on every unfinished lane; unseeded mode: in pieces that stay inside one block of the Exp(1) ring,
            so a block is uploaded only after every step that still reads the half it overwrites has been enqueued"""
            nonlocal steps_enq
            while n > 0:
                k = n if feeder is None else min(n, half - (steps_enq % half))
                ensure_q(steps_enq + k)
                for ln in L:
                    if ln.done:
                        continue
                    if graph_ok:
                        # opt-in: the step captured for a BOUND on the live rows (16-row buckets below the batch; include/chattts_amd.h
                        # ctts_gpt_graph_build_rows): finished utterances then cost no workgroups at all.  The bound is what the last
                        # COLLECTED poll saw -- one or two chunks old, and finish flags only go up -- so it holds for every step enqueued
                        # here; the kernels read the exact live count themselves, the bits do not depend on the bound.
                        Bl_ = ln.hi - ln.lo
                        rows = (max(ln.live_ub, 1) + 15) // 16 * 16 if rows_graphs else Bl_
                        if rows < Bl_:
                            if rows not in sess.setdefault("rows_built", [set() for _ in L])[L.index(ln)]:
                                _lib.check(lib.ctts_gpt_graph_build_rows(ln.handle, C.byref(ln.s), rows, ln.st.cuda_stream), "ctts_gpt_graph_build_rows")
                                sess["rows_built"][L.index(ln)].add(rows)
                            _lib.check(lib.ctts_gpt_graph_launch_rows(ln.handle, k, rows, ln.st.cuda_stream), "ctts_gpt_graph_launch_rows")
                        else:
                            _lib.check(lib.ctts_gpt_graph_launch(ln.handle, k, ln.st.cuda_stream), "ctts_gpt_graph_launch")
                    else:
                        for _ in range(k):
                            _lib.check(lib.ctts_gpt_decode_step(ln.handle, C.byref(ln.s), ln.st.cuda_stream), "ctts_gpt_decode_step")
                steps_enq += k
                n -= k

        # ---- step 0: prefill ----
        ensure_q(1)
        for ln in L:
            if prefill_chunk is None or int(prefill_chunk) >= T:
                _lib.check(lib.ctts_gpt_prefill(ln.handle, C.byref(ln.s), ln.emb.data_ptr(), ln.st.cuda_stream), "ctts_gpt_prefill")
            else:
                tc = max(1, int(prefill_chunk))
                with torch.cuda.stream(ln.st):
                    for t0 in range(0, T, tc):
                        n = min(tc, T - t0)
                        piece = ln.emb[:, t0: t0 + n].contiguous()
                        _lib.check(lib.ctts_gpt_prefill_chunk(ln.handle, C.byref(ln.s), piece.data_ptr(), t0, n, int(t0 + n == T),
                                                              ln.st.cuda_stream), "ctts_gpt_prefill_chunk")
                        ln.keep_piece = piece   # stream-ordered use: stays alive until the next piece replaces it / the poll syncs
        steps_done = 1
        steps_enq = 1     # steps enqueued so far (>= steps_done: chunks run ahead of the host's finish polls)
        graph_ok = False
        for ln in L:
            ln.ev = torch.cuda.Event()
        fin0 = torch.cat([poll(ln) for ln in L])  # syncs: the step-0 rule needs it (gpt.py:527)
        if bool(fin0.any()):
            self.logger.warning("unexpected end at index %s", str(fin0.nonzero().flatten().tolist()))
            finish_rng(1)   # the reference drew exactly one tensor (step 0) before it got here
            if ensure_non_empty and manual_seed is None:
                self.logger.warning("regenerate in order to ensure non-empty")
                yield from self.generate(emb, inputs_ids, temperature, eos_token, attention_mask, max_new_token,
                                         min_new_token, logits_processors, infer_text, return_attn, return_hidden,
                                         stream, show_tqdm, ensure_non_empty, stream_batch, manual_seed, context,
                                         use_graph=use_graph, stop_at=stop_at, row_offset=row_offset, total_rows=total_rows,
                                         profile_tag=profile_tag, profile_stride=profile_stride, lanes=lanes,
                                         teacher_ids=teacher_ids, prefill_chunk=prefill_chunk, return_sampled=return_sampled,
                                         rng=rng, rng_seed=None, rng_nonce=rng_nonce, row_ids=row_ids, exact=exact)
            return  # gpt.py:570: the seeded case yields nothing

        graph_ok = use_graph and max_new > 1
        # OPT-IN (CTTS_GRAPH_ROWS=1): measured on the C3 bench it changes nothing -- 1399-1406 vs 1403-1413 audio-s/s, parity mode 1011-1031
        # vs 1021-1037 (profiles/r5o_ab_graph_rows.log): like the persistent attention grid, it removes workgroups that were not what the
        # step was waiting for.  Default: every chunk on the batch-sized graph.
        rows_graphs = os.environ.get("CTTS_GRAPH_ROWS", "0") == "1"
        if graph_ok and not sess["graph"]:
            for ln in L:
                _lib.check(lib.ctts_gpt_graph_build(ln.handle, C.byref(ln.s), ln.st.cuda_stream), "ctts_gpt_graph_build")
            sess["graph"] = True
            sess["rows_built"] = [set() for _ in L]     # (graph_build dropped whatever bounded graphs the handle had)
        if profile_tag is not None:
            _lib.check(lib.ctts_gpt_profile_begin(L[0].handle, int(profile_tag), 4096, int(profile_stride)), "profile_begin")

        chunk = stream_batch if stream else self.POLL
        all_done = all(ln.done for ln in L)
        interrupted = False
        # the reference yields when (i+1) % stream_batch == 0: keep chunk ends on those steps
        next_n = lambda done: min(chunk - (done % chunk), max_new - done)
        # Run-ahead: chunk k+1 is enqueued BEFORE the host looks at chunk k's finish flags (asynchronous snapshots), so the
        # GPU never idles on the host between chunks; a streamed chunk's consumer (DVAE/Vocos of the prefix) overlaps the
        # generation of the next one.  If everything turns out finished, the surplus chunk is ~100 empty launches per step.
        import time as _time
        t_dec0 = _time.perf_counter()
        inflight = []   # [(steps, [snapshot index per lane])], oldest first

        def launch_chunk():
            n = next_n(steps_enq)
            enqueue(n)
            inflight.append((n, [None if ln.done else snapshot(ln) for ln in L]))

        try:
            if steps_enq < max_new and not all_done:
                launch_chunk()
            while inflight:
                if len(inflight) < 2 and steps_enq < max_new and not context.get():
                    launch_chunk()
                n, snaps = inflight.pop(0)
                for ln, k in zip(L, snaps):
                    if k is not None and not ln.done:
                        collect(ln, k)
                steps_done += n
                all_done = all(ln.done for ln in L)
                if all_done:
                    inflight.clear()   # a surplus chunk computes nothing (no live rows); its launches drain in `finally`
                if stream:
                    emit = False
                    if not all_done and steps_done % stream_batch == 0:
                        emit = True                                      # gpt.py:579-589
                    elif all_done:
                        i_star = max(max(ln.end_snap) for ln in L)       # step at which the last row hit EOS
                        emit = i_star > 0 and i_star % stream_batch == 0  # the reference's duplicate yield (stream_iter quirk)
                    if emit:
                        yield outputs()
                # the reference tests the interrupt flag after the step's stream yield (gpt.py:579-592): a chunk that ends on a
                # yield boundary is still handed out.  Granularity here is one chunk (POLL / stream_batch steps), not one step.
                if context.get():
                    interrupted = True
                    # run-ahead: a chunk may already be enqueued behind the one just collected.  Its tokens WILL be in the buffers
                    # the final outputs() reads, so it is collected and counted too -- tokens, `steps` and the rewound position of
                    # the global CPU generator (unseeded mode) then describe the same number of steps.
                    while inflight:
                        n2, snaps2 = inflight.pop(0)
                        for ln, k in zip(L, snaps2):
                            if k is not None and not ln.done:
                                collect(ln, k)
                        steps_done += n2
                    all_done = all(ln.done for ln in L)
                    break
            if profile_tag is not None:
                n_s, tot = C.c_int32(0), C.c_double(0.0)
                buf = (C.c_float * 4096)()
                _lib.check(lib.ctts_gpt_profile_samples(L[0].handle, buf, 4096, C.byref(n_s)), "profile_samples")
                self.last_stats["profile_samples_ms"] = np.ctypeslib.as_array(buf)[: int(n_s.value)].copy()
                _lib.check(lib.ctts_gpt_profile_end(L[0].handle, C.byref(n_s), C.byref(tot)), "profile_end")
                self.last_stats["profile"] = (int(n_s.value), float(tot.value))
        finally:
            if feeder is not None:
                feeder["pool"].shutdown(wait=True)   # idempotent; the generator-state rewind happens in finish_rng
            for ln in L:
                wait_stream(ln.st)   # nothing of this call is still running when its buffers are handed to the next one
            self.last_stats["decode_ms"] = 1e3 * (_time.perf_counter() - t_dec0)   # host wall of the decode loop (streaming: incl. consumer time)
        if not all_done:
            if interrupted:
                self.logger.warning("generation is interrupted")
            else:
                self.logger.warning(f"incomplete result. hit max_new_token: {max_new_token}")   # gpt.py:601-607
        self.last_stats.update(steps=steps_done, B=B, T=T, lanes=n_lanes)
        for ln in L:
            if not ln.done or ln.end_snap is None:
                poll(ln)
        # steps the reference loop would have executed: up to and including the step at which the last row hit EOS
        steps_ref = (max(max(ln.end_snap) for ln in L) + 1) if all_done else min(steps_done, max_new)
        finish_rng(min(steps_ref, max_new))
        if return_sampled:
            self.last_sampled = [ln.sampled[b, : ln.end_snap[b]].clone() for ln in L for b in range(ln.hi - ln.lo)]
        final = outputs()
        if certify:
            # ---- parity certificate: the smallest decision margin of every utterance (tempered-logit units) against twice the stated
            # logit error of the split-fp16 projections.  The lanes are idle here (wait_stream above).
            marg = torch.cat([ln.margin for ln in L]).cpu().numpy()
            x3_run = self.x3 is not None and not exact
            bound = 2.0 * self.REL_ERR_X3 * self.logit_scale[bool(infer_text)] / max(float(temperature.min()), 1e-6) if x3_run else 0.0
            unsafe = np.nonzero(marg < bound)[0].tolist()
            self.last_margins = marg
            self.last_stats.update(min_margin=float(marg.min()) if marg.size else float("inf"), margin_bound=bound,
                                   certified=not unsafe, uncertified_rows=unsafe, exact_rerun_rows=[])
            if unsafe:
                (self.logger.warning if self.exact_fallback else self.logger.info)(
                                    "parity certificate: %d of %d utterances had a draw decided by less than %.2e tempered-logit units "
                                    "(smallest margin %.2e)%s", len(unsafe), B, bound, float(marg.min()),
                                    "; generating them again on the exact f32 kernels" if (self.exact_fallback and not stream and not interrupted)
                                    else "; the split-fp16 result stands as it is")
                if self.exact_fallback and not stream and not interrupted:
                    final = self._rerun_exact(final, unsafe, emb, inputs_ids, temperature, eos_token, attention_mask, max_new_token, min_new_token,
                                              logits_processors, infer_text, return_hidden, stream_batch, manual_seed, context, use_graph, stop_at,
                                              row_offset // nrow, (total_rows if total_rows is not None else B * nrow), prefill_chunk, rng,
                                              seed_val if device_rng else None, rng_nonce, rid, rng_state0)
        yield final

    def _rerun_exact(self, final, rows, emb, inputs_ids, temperature, eos_token, attention_mask, max_new_token, min_new_token,
                     logits_processors, infer_text, return_hidden, stream_batch, manual_seed, context, use_graph, stop_at, first_row,
                     total_rows, prefill_chunk, rng, rng_seed, rng_nonce, rid, rng_state0):
        """The exact fallback of the certificate: utterances `rows` of the call once more, as a sub-batch with the same padded prompt
        geometry, on the f32 MFMA decode kernels, with the draws of their own global rows -- what the "f32" engine gives for them (utterances
        never interact) -- spliced over their rows of `final`."""
        stats = dict(self.last_stats)
        sel = torch.tensor(rows, dtype=torch.long)
        gid = rid[sel] if rid is not None else sel + int(first_row)
        nz = None
        if rng_nonce is not None:
            nz = torch.as_tensor(rng_nonce, dtype=torch.int64).reshape(-1)
            nz = nz if nz.numel() == 1 else nz[sel]
        state_after = None
        if rng_state0 is not None:     # unseeded host draws: the same per-step tensors the main call consumed
            state_after = torch.get_rng_state()
            torch.set_rng_state(rng_state0)
        sub = None
        try:
            for sub in self.generate(emb[sel.to(emb.device)], inputs_ids[sel.to(inputs_ids.device)], temperature, eos_token,
                                     None if attention_mask is None else attention_mask[sel.to(attention_mask.device)], max_new_token, min_new_token,
                                     logits_processors, infer_text, False, return_hidden, False, False, False, stream_batch, manual_seed, context,
                                     use_graph=use_graph, stop_at=None if stop_at is None else stop_at[sel.to(stop_at.device)], total_rows=total_rows,
                                     lanes=1, prefill_chunk=prefill_chunk, rng=rng, rng_seed=rng_seed, rng_nonce=nz, row_ids=gid, exact=True):
                pass
        finally:
            # where the global generator is left: the reference draws once per step of its loop, which ends when the LAST utterance
            # finishes -- the run (main or re-run) that executed more steps decides; a re-run of every utterance IS the call
            if state_after is not None:
                keep_rerun = sub is not None and (len(rows) == len(final.ids) or self.last_stats.get("steps", 0) > stats.get("steps", 0))
                if not keep_rerun:
                    torch.set_rng_state(state_after)
        rerun_stats = self.last_stats
        self.last_stats = stats
        if sub is None:     # (an EOS at step 0 of the exact run: the seeded reference yields nothing, gpt.py:570)
            self.logger.warning("parity certificate: the exact re-run ended at step 0; the split-fp16 rows stand")
            return final
        same_len = all(int(sub.ids[j].shape[0]) == int(final.ids[b].shape[0]) for j, b in enumerate(rows))
        padded = getattr(final.hiddens, "padded", None)
        if return_hidden and not (same_len and padded is not None):
            final.hiddens = RowList(list(final.hiddens))      # plain rows: the decoder re-pads (core.py:525-533)
            padded = None
        for j, b in enumerate(rows):
            final.ids[b] = sub.ids[j]
            if return_hidden:
                if padded is not None:
                    padded[b, : sub.hiddens[j].shape[0]].copy_(sub.hiddens[j])     # (final.hiddens[b] is a view of it)
                else:
                    final.hiddens[b] = sub.hiddens[j]
        self.last_stats.update(exact_rerun_rows=list(rows), exact_rerun_steps=rerun_stats.get("steps"),
                               exact_rerun_min_margin=rerun_stats.get("min_margin"), certified=True)
        return final


def split_bf16(w: torch.Tensor) -> torch.Tensor:
    """[N, K] float32 -> [N, Kp/32, 2, 32] bfloat16, Kp = K rounded up to 32 and zero padded: for every row and every
    32-wide k block the hi values then the lo values, w = hi + lo up to 2^-17 |w| (the weight operand of the bf16x3
    GEMM tiles).  One k-step of one row is ONE 128-byte line (hi 64 B + lo 64 B), so a tile step never fetches a
    half-used cache line."""
    N, K = w.shape
    Kp = (K + 31) // 32 * 32
    wp = torch.zeros((N, Kp), dtype=torch.float32)
    wp[:, :K] = w
    hi = wp.to(torch.bfloat16)
    lo = (wp - hi.to(torch.float32)).to(torch.bfloat16)
    return torch.stack([hi.view(N, Kp // 32, 32), lo.view(N, Kp // 32, 32)], 2).contiguous()


def pack_x3p(w: torch.Tensor) -> torch.Tensor:
    """[R, K] float32 (R % 32 == 0, K % 16 == 0) -> the PRE-SPLIT planes of csrc/codec_gemm.hip in MFMA fragment order,
    [R/32][K/16][hi | lo][lane = (k%16)/8*32 + r%32][k%8] bfloat16: w = hi + lo up to 2^-17 |w|, and every (32-row tile,
    16-wide k block, plane) is one contiguous KiB -- what `v_mfma_f32_32x32x16_bf16` takes as an operand, so a tile step is
    staged by plain LDS-DMA copies."""
    R, K = w.shape
    assert R % 32 == 0 and K % 16 == 0
    w = w.to(torch.float32)
    hi = w.to(torch.bfloat16)
    lo = (w - hi.to(torch.float32)).to(torch.bfloat16)
    x = torch.stack([hi, lo], 0).reshape(2, R // 32, 32, K // 16, 2, 8)
    return x.permute(1, 3, 0, 4, 2, 5).contiguous()


def unpack_x3p(p: torch.Tensor, R: int, K: int) -> torch.Tensor:
    """inverse of pack_x3p: -> float32 [R, K] = hi + lo (tests)"""
    x = p.reshape(R // 32, K // 16, 2, 2, 32, 8).permute(2, 0, 4, 1, 3, 5).reshape(2, R, K).to(torch.float32)
    return x[0] + x[1]


def pack_h1p(w: torch.Tensor) -> torch.Tensor:
    """[R, K] float32 (R % 32 == 0, K % 16 == 0) -> ONE fp16 plane in MFMA fragment order (csrc/codec_gemm.hip: gemm_h1p_k),
    [R/32][K/16][lane = (k%16)/8*32 + r%32][k%8] float16, values saturated to the finite half range like the kernels' own
    activation rounding."""
    R, K = w.shape
    assert R % 32 == 0 and K % 16 == 0
    h = w.to(torch.float32).clamp(-65504.0, 65504.0).to(torch.float16)
    return h.reshape(R // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()


def unpack_h1p(p: torch.Tensor, R: int, K: int) -> torch.Tensor:
    """inverse of pack_h1p: -> float32 [R, K] (tests)"""
    return p.reshape(R // 32, K // 16, 2, 32, 8).permute(0, 3, 1, 2, 4).reshape(R, K).to(torch.float32)


# ---------------------------------------------------------------------------------------------
class PendingWavs:
    """result of `CodecEngine.decode_to_wavs_async`: the float32 waveforms of one batch, in flight"""

    def __init__(self, codec, view, done, keep):
        self._codec, self._view, self._done, self._keep, self._out = codec, view, done, keep, None

    def done(self) -> bool:
        return self._out is not None or self._done.query()

    def result(self) -> np.ndarray:
        if self._out is None:
            wait_event(self._done)
            self._out = self._codec._copy_out(self._view) if self._view.numel() else np.zeros(tuple(self._view.shape), np.float32)
            self._keep = None
        return self._out


class CodecEngine:
    """DVAE decoder + Vocos on the device (channels-last)."""

    def __init__(self, decoder_sd: dict, vocos_sd: dict, device: torch.device, gemm: str = "bf16x3"):
        """gemm="bf16x3": dense layers run on split-bf16 MFMA tiles (x = hi + lo, 3 products; f32-class accuracy,
        measured wav RMS error vs the reference ~1e-6 against the 1e-4 bar); gemm="f32": f32-input MFMA tiles;
        gemm="f16" (the perf mode's decoder): as "bf16x3", but the ConvNeXt point-wise pairs of batches from 1024 frames take
        one fp16 MFMA per product with f32 accumulation -- waveform within 1e-5 RMS of "bf16x3" (stated and tested bound; the
        north-star bar is 1e-4), about half the decode time."""
        if gemm not in ("bf16x3", "f32", "f16"):
            raise ValueError("gemm must be 'bf16x3', 'f16' or 'f32'")
        self.gemm = gemm
        self.lib = _lib.lib()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.EngineError("CodecEngine needs a ROCm GPU device (there is no CPU path)")
        dev = self.device
        f = lambda t: t.to(torch.float32).contiguous().to(dev)
        dww = lambda t: f(t[:, 0, :].t())                # [C,1,7] -> [7,C]

        def dense(t):
            """[N, K] float32 dense weight -> device tensor in the layout the selected GEMM tiles read"""
            t = t.to(torch.float32).reshape(t.shape[0], -1).contiguous()
            if gemm == "f32":
                return t.to(dev)
            return split_bf16(t).to(dev)

        convw = lambda t: dense(t.permute(0, 2, 1))      # [Cout,Cin,k] -> [Cout,k*Cin]
        self.keep = []
        k = self.keep.append
        w = _lib.CodecWeights()

        def P(t):
            k(t)
            return t.data_ptr()

        def PA(ts):
            k(ts)
            arr = _lib.ptr_array(ts)
            k(arr)
            return C.cast(arr, _lib.PP)

        d = decoder_sd
        w.conv_in0_w, w.conv_in0_b = P(convw(d["decoder.conv_in.0.weight"])), P(f(d["decoder.conv_in.0.bias"]))
        w.conv_in2_w, w.conv_in2_b = P(convw(d["decoder.conv_in.2.weight"])), P(f(d["decoder.conv_in.2.bias"]))
        nb = 0
        while f"decoder.decoder_block.{nb}.weight" in d:
            nb += 1
        w.n_dvae_blocks = nb
        blk = lambda i, n: d[f"decoder.decoder_block.{i}.{n}"]
        w.d_dw_w = PA([dww(blk(i, "dwconv.weight")) for i in range(nb)])
        w.d_dw_b = PA([f(blk(i, "dwconv.bias")) for i in range(nb)])
        w.d_ln_w = PA([f(blk(i, "norm.weight")) for i in range(nb)])
        w.d_ln_b = PA([f(blk(i, "norm.bias")) for i in range(nb)])
        w.d_pw1_w = PA([dense(blk(i, "pwconv1.weight")) for i in range(nb)])
        w.d_pw1_b = PA([f(blk(i, "pwconv1.bias")) for i in range(nb)])
        w.d_pw2_w = PA([dense(blk(i, "pwconv2.weight")) for i in range(nb)])
        w.d_pw2_b = PA([f(blk(i, "pwconv2.bias")) for i in range(nb)])
        w.d_gamma = PA([f(blk(i, "weight")) for i in range(nb)])
        w.conv_out_w = P(dense(d["decoder.conv_out.weight"][:, :, 0]))
        w.out_conv_w = P(convw(d["out_conv.weight"]))
        w.coef = P(f(d["coef"].reshape(-1)))
        v = vocos_sd
        w.v_embed_w, w.v_embed_b = P(convw(v["backbone.embed.weight"])), P(f(v["backbone.embed.bias"]))
        w.v_norm_w, w.v_norm_b = P(f(v["backbone.norm.weight"])), P(f(v["backbone.norm.bias"]))
        nv = 0
        while f"backbone.convnext.{nv}.gamma" in v:
            nv += 1
        w.n_vocos_blocks = nv
        vb = lambda i, n: v[f"backbone.convnext.{i}.{n}"]
        w.v_dw_w = PA([dww(vb(i, "dwconv.weight")) for i in range(nv)])
        w.v_dw_b = PA([f(vb(i, "dwconv.bias")) for i in range(nv)])
        w.v_ln_w = PA([f(vb(i, "norm.weight")) for i in range(nv)])
        w.v_ln_b = PA([f(vb(i, "norm.bias")) for i in range(nv)])
        w.v_pw1_w = PA([dense(vb(i, "pwconv1.weight")) for i in range(nv)])
        w.v_pw1_b = PA([f(vb(i, "pwconv1.bias")) for i in range(nv)])
        w.v_pw2_w = PA([dense(vb(i, "pwconv2.weight")) for i in range(nv)])
        w.v_pw2_b = PA([f(vb(i, "pwconv2.bias")) for i in range(nv)])
        w.v_gamma = PA([f(vb(i, "gamma")) for i in range(nv)])
        w.v_final_w, w.v_final_b = P(f(v["backbone.final_layer_norm.weight"])), P(f(v["backbone.final_layer_norm.bias"]))
        w.head_w, w.head_b = P(dense(v["head.out.weight"])), P(f(v["head.out.bias"]))
        w.window = P(f(v["head.istft.window"]))
        kk = torch.arange(VOCOS.n_fft // 2, dtype=torch.float64) * (2.0 * math.pi / VOCOS.n_fft)
        w.twiddle = P(f(torch.stack([kk.cos(), kk.sin()], 1)))
        w.gemm_mode = {"f32": 0, "bf16x3": 1, "f16": 2}[gemm]
        if gemm == "f16":
            h1p = lambda t: pack_h1p(t.to(torch.float32).reshape(t.shape[0], -1)).to(dev)
            w.d_pw1_x3p = PA([h1p(blk(i, "pwconv1.weight")) for i in range(nb)])
            w.d_pw2_x3p = PA([h1p(blk(i, "pwconv2.weight")) for i in range(nb)])
            w.v_pw1_x3p = PA([h1p(vb(i, "pwconv1.weight")) for i in range(nv)])
            w.v_pw2_x3p = PA([h1p(vb(i, "pwconv2.weight")) for i in range(nv)])
        if gemm == "bf16x3":
            # the ConvNeXt point-wise layers once more as pre-split fragment-order planes: from 12288 frames they run on the
            # LDS-DMA staged kernel of csrc/codec_gemm.hip (40 of the 45 GEMM launches of a decode)
            x3p = lambda t: pack_x3p(t.to(torch.float32).reshape(t.shape[0], -1)).to(dev)
            w.d_pw1_x3p = PA([x3p(blk(i, "pwconv1.weight")) for i in range(nb)])
            w.d_pw2_x3p = PA([x3p(blk(i, "pwconv2.weight")) for i in range(nb)])
            w.v_pw1_x3p = PA([x3p(vb(i, "pwconv1.weight")) for i in range(nv)])
            w.v_pw2_x3p = PA([x3p(vb(i, "pwconv2.weight")) for i in range(nv)])
        self._w = w
        h = C.c_void_p()
        _lib.check(self.lib.ctts_codec_create(C.byref(h), C.byref(w)), "ctts_codec_create")
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.ctts_codec_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def _ws(self, B, F):
        """the codec workspace: ONE buffer per engine that only ever grows.  A streamed utterance asks for a different window size
        at every yield; going through the caching allocator each time meant a device allocation per yield whenever no cached block
        fitted (tens of milliseconds each on this pool's hosts, profiles/r3e_c5_sched_probe.log).  Kernels of consecutive calls
        are stream-ordered on the caller's stream, so sharing the buffer needs no extra synchronisation as long as ONE stream
        drives the engine at a time (the documented contract of a handle, include/chattts_amd.h)."""
        n = self.lib.ctts_codec_workspace_bytes(B, F)
        cur = torch.cuda.current_stream(self.device)
        buf = getattr(self, "_ws_buf", None)
        if buf is None or buf.numel() < n or getattr(self, "_ws_stream", None) != cur.cuda_stream:
            if buf is not None:
                buf.record_stream(cur)
            with torch.cuda.stream(cur):
                self._ws_buf = buf = torch.empty((max(n, 0 if buf is None else buf.numel()),), dtype=torch.uint8, device=self.device)
            self._ws_stream = cur.cuda_stream
        return buf, buf.numel()

    def dvae_decode(self, hid: torch.Tensor) -> torch.Tensor:
        """hid [B,T,768] f32 (zero padded rows) -> mel [B,2T,100]   (dvae.py:276-297)."""
        B, T, H = hid.shape
        assert H == GPT.hidden
        hid = hid.to(torch.float32).contiguous().to(self.device)
        mel = torch.empty((B, 2 * T, DVAE.n_mels), dtype=torch.float32, device=self.device)
        ws, n = self._ws(B, 2 * T)
        st = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self.lib.ctts_dvae_decode(self.handle, hid.data_ptr(), mel.data_ptr(), B, T, ws.data_ptr(), n, st), "ctts_dvae_decode")
        return mel

    def vocos_decode(self, mel: torch.Tensor) -> torch.Tensor:
        """mel [B,F,100] -> wav [B,256(F-1)]   (vocos.Vocos.decode, core.py:505-510)."""
        B, F, M = mel.shape
        assert M == VOCOS.n_mels and F >= 2
        mel = mel.to(torch.float32).contiguous().to(self.device)
        wav = torch.empty((B, VOCOS.hop * (F - 1)), dtype=torch.float32, device=self.device)
        ws, n = self._ws(B, F)
        st = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self.lib.ctts_vocos_decode(self.handle, mel.data_ptr(), wav.data_ptr(), B, F, ws.data_ptr(), n, st), "ctts_vocos_decode")
        return wav

    def float_to_int16(self, wav: torch.Tensor, per_row: bool = True, product: str = "f64", keep_thr: Optional[float] = None):
        """`float_to_int16` of the reference's back end (tools/audio/np.py:7-11) ON THE DEVICE, before the waveform crosses PCIe: wav
        [B, n] float32 (any row stride) -> (pcm [B, n] int16, keep) with am = 32767 * 32768 // (ceil(peak) * 32768) and truncation toward
        zero.  per_row: a peak per utterance (one call per utterance, examples/web/funcs.py:206-209) -- False: ONE peak over the block
        (the function applied to a [B, n] array, examples/cmd/stream.py:44).  product "f64": the reference's numba runtime (exact product),
        "f32": plain NumPy's reading of the same line.  keep_thr: also return the packed mask |x| > keep_thr ([B, ceil(n/8)] uint8,
        np.packbits order) -- the selector of Chat.infer's silence strip (core.py:262-265) -- else None.  Bit-exact against the reference
        function (tests/test_backend.py)."""
        assert wav.dim() == 2 and wav.dtype == torch.float32 and wav.is_cuda and (wav.shape[1] == 0 or wav.stride(1) == 1)
        B, n = int(wav.shape[0]), int(wav.shape[1])
        pcm = torch.empty((B, n), dtype=torch.int16, device=wav.device)
        keep = torch.empty((B, (n + 7) // 8), dtype=torch.uint8, device=wav.device) if keep_thr is not None else None
        if B == 0 or n == 0:
            return pcm, keep
        peak = torch.empty((B,), dtype=torch.int32, device=wav.device)
        st = torch.cuda.current_stream(wav.device).cuda_stream
        _lib.check(self.lib.ctts_float_to_int16(wav.data_ptr(), pcm.data_ptr(), _lib.ptr(keep), B, n, int(wav.stride(0)) if B > 1 else n,
                                                1 if per_row else 0, {"f64": 0, "f32": 1}[product], float(keep_thr or 0.0), peak.data_ptr(), st),
                   "ctts_float_to_int16")
        return pcm, keep

    # receptive field of one output sample, in mel frames either side: ISTFT 4 overlapping frames; Vocos embed k7 + 8 ConvNeXt
    # blocks k7 = 27; DVAE conv_in k3 + k3, 12 ConvNeXt blocks k7 dilation 2, out_conv k3 = 75 (dvae.py:145-161, config.py:83-121)
    HALO_FRAMES = 27 + 75

    def decode_window(self, result_list: List[torch.Tensor], s_lo: int, s_hi: int) -> torch.Tensor:
        """Samples [s_lo, s_hi) of `decode_to_wavs(result_list)` WITHOUT decoding the whole batch: the acoustic decoder is a
        stack of short symmetric convolutions, so those samples depend only on the tokens within HALO_FRAMES mel frames
        (+ the 4 overlapping ISTFT frames) of them.  Only that token window (zero padded like core.py:525-533) goes
        through DVAE + Vocos; at the true ends of the sequence the window is the sequence's own edge, so the result equals
        the full decode up to summation order.  This is what makes streaming O(n): the reference re-decodes the entire
        prefix at every yield (core.py:482-497) to hand out the next `stream_speed` samples of it."""
        Tn = max(int(r.size(0)) for r in result_list)
        total = VOCOS.hop * (2 * Tn - 1)
        s_lo, s_hi = max(0, int(s_lo)), min(total, int(s_hi))
        if s_hi <= s_lo:
            return torch.empty((len(result_list), 0), dtype=torch.float32, device=self.device)
        hop, nfft = VOCOS.hop, VOCOS.n_fft
        f_a = (s_lo + nfft // 2) // hop - (nfft // hop - 1)          # first / last ISTFT frame that overlaps the samples
        f_b = min((s_hi - 1 + nfft // 2) // hop, 2 * Tn - 1)
        t_lo = max(0, (f_a - self.HALO_FRAMES) // 2)
        t_hi = min(Tn, (f_b + self.HALO_FRAMES) // 2 + 1)
        batch = torch.zeros((len(result_list), t_hi - t_lo, GPT.hidden), dtype=torch.float32, device=self.device)
        for i, r in enumerate(result_list):
            n = min(int(r.size(0)), t_hi) - t_lo
            if n > 0:
                batch[i, :n] = r[t_lo: t_lo + n]
        wav = self.vocos_decode(self.dvae_decode(batch))
        off = 2 * hop * t_lo                                          # sample index of the window's first sample
        return wav[:, s_lo - off: s_hi - off]

    def to_host(self, t: torch.Tensor) -> np.ndarray:
        """device tensor -> numpy, the `.cpu().numpy()` that ends the reference path (core.py:508-510), through a cached PINNED
        staging buffer: a pageable `.cpu()` of the 67 MB of waveforms of a 64-utterance batch runs at a fraction of the PCIe rate.
        The returned array is a copy (the staging buffer is reused by the next call)."""
        t = t.contiguous()
        n = t.numel()
        if n == 0:
            return np.zeros(tuple(t.shape), dtype={torch.int16: np.int16, torch.uint8: np.uint8}.get(t.dtype, np.float32))
        pins = self.__dict__.setdefault("_pinned", {})     # one grow-only staging buffer per element type (float32 waveforms, int16 PCM, uint8 masks)
        buf = pins.get(t.dtype)
        if buf is None or buf.numel() < n:
            buf = pins[t.dtype] = torch.empty(((n + 15) // 16 * 16,), dtype=t.dtype).pin_memory()
        view = buf[:n].view(t.shape)
        nbytes = n * t.element_size()
        if nbytes >= (16 << 20) and nbytes % 64 == 0 and t.data_ptr() % 16 == 0 and os.environ.get("CTTS_D2H_PIPE", "1") != "0":
            # a batch's worth of waveforms: the copy over PCIe (1.2 ms for 67 MB) and the copy out of the staging buffer (2 ms on 4
            # threads) in 4 pieces, piece i leaving the staging buffer while piece i + 1 crosses the bus (tools/pass_timeline.py)
            return self._to_host_piped(t, view, nbytes)
        if nbytes % 16 == 0 and t.data_ptr() % 16 == 0 and os.environ.get("CTTS_D2H_SHADER", "1") != "0":
            # shader copy (plain stores over PCIe into the pinned buffer): the same rate as hipMemcpyAsync (67 MB in 1.2 ms,
            # profiles/r3j_d2h_probe.log) and just the next packet of the stream -- no copy-engine hand-off behind the decode kernels
            _lib.check(self.lib.ctts_copy_bytes(view.data_ptr(), t.data_ptr(), nbytes, torch.cuda.current_stream(self.device).cuda_stream),
                       "ctts_copy_bytes")
        else:
            view.copy_(t, non_blocking=True)
        wait_stream(torch.cuda.current_stream(self.device))
        # The result must not alias the staging buffer: copy it out with plain memcpy (numpy), NOT `view.clone()`: torch's intra-op
        # pool has one thread per hardware thread (128 here) and this pool's hosts run under a 16-CPU cgroup quota, so waking it up
        # intermittently costs 60-90 ms of CFS throttling (profiles/r3l_hostcopy_probe.log: clone max 92 ms vs numpy 0.1 ms for one
        # streamed chunk) -- that was what turned the streaming config C5 from 266 into 600 ms per batch.  Large results are cut into
        # 4 slices copied by 4 plain threads (memcpy releases the GIL).
        return self._copy_out(view)

    # -- software pipelining across batches ------------------------------------------------------------------------------------
    def decode_to_wavs_async(self, result_list: List[torch.Tensor]) -> "PendingWavs":
        """`decode_to_wavs` + the `.cpu().numpy()` of core.py:508-510, ASYNCHRONOUSLY on the engine's own side stream: the call
        returns at once (everything is enqueued), `PendingWavs.result()` hands out the numpy array.  A caller that works through a
        queue of batches starts the NEXT batch's generation before asking for the result, so the acoustic decode (MFMA-bound, big
        kernels) overlaps the next batch's decode steps (latency-bound, most SIMDs idle) on the same GPU -- the batch-level form of
        what BASELINE config 4 does per streamed chunk.  The input tensors must not be modified until `result()` returned
        (`GptEngine.generate` hands out copies, so its outputs qualify).  Results are bit-identical to the synchronous calls."""
        dev = self.device
        if not hasattr(self, "_side"):
            # CTTS_CODEC_CUS="n[:stride[:first]]": the side stream is confined to n compute units (ctts_stream_create_cu_mask), so the big
            # MFMA kernels of the decode of batch i do not take the CUs the latency-bound decode steps of batch i + 1 run on
            spec = codec_cu_spec()
            if spec:
                self._side = cu_masked_stream(self.lib, dev, *spec, False)
            else:
                self._side = torch.cuda.Stream(device=dev)
            self._side_pins = [None, None]     # two staging buffers: a result may still be in its buffer when the next decode is enqueued
            self._side_n = 0
        caller = torch.cuda.current_stream(dev)
        ready = torch.cuda.Event()
        ready.record(caller)                   # the hidden states were produced on the caller's stream
        k = self._side_n = (self._side_n + 1) % 2
        with torch.cuda.stream(self._side):
            self._side.wait_event(ready)
            wav = self.decode_to_wavs(result_list)
            n = wav.numel()
            pin = self._side_pins[k]
            if pin is None or pin.numel() < n:
                pin = self._side_pins[k] = torch.empty(((n + 3) // 4 * 4,), dtype=torch.float32).pin_memory()
            view = pin[:n].view(wav.shape)
            if n and (n * 4) % 16 == 0:
                _lib.check(self.lib.ctts_copy_bytes(view.data_ptr(), wav.data_ptr(), n * 4, self._side.cuda_stream), "ctts_copy_bytes")
            elif n:
                view.copy_(wav, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self._side)
            for r in result_list:
                r.record_stream(self._side)
            wav.record_stream(self._side)
        return PendingWavs(self, view, done, wav)

    def _to_host_piped(self, t: torch.Tensor, view: torch.Tensor, nbytes: int, pieces: int = 4) -> np.ndarray:
        st = torch.cuda.current_stream(self.device)
        per = (nbytes // pieces + 63) // 64 * 64
        cuts = [(o, min(per, nbytes - o)) for o in range(0, nbytes, per)]
        evs = []
        for off, ln in cuts:
            _lib.check(self.lib.ctts_copy_bytes(view.data_ptr() + off, t.data_ptr() + off, ln, st.cuda_stream), "ctts_copy_bytes")
            ev = torch.cuda.Event()
            ev.record(st)
            evs.append(ev)
        src = view.numpy()
        out = np.empty(src.shape, dtype=src.dtype)
        fs, fo = src.reshape(-1).view(np.uint8), out.reshape(-1).view(np.uint8)
        pool = getattr(self, "_copy_pool", None)
        if pool is None:
            from concurrent.futures import ThreadPoolExecutor
            pool = self._copy_pool = ThreadPoolExecutor(max_workers=4)
        for (off, ln), ev in zip(cuts, evs):
            wait_event(ev)
            q = (ln // 4 + 63) // 64 * 64
            list(pool.map(lambda a: np.copyto(fo[off + a: off + min(a + q, ln)], fs[off + a: off + min(a + q, ln)]), range(0, ln, q)))
        return out

    def _copy_out(self, view: torch.Tensor) -> np.ndarray:
        """pinned staging view -> fresh numpy array by plain memcpy (see to_host)"""
        src = view.numpy()
        out = np.empty(src.shape, dtype=src.dtype)
        flat_s, flat_o = src.reshape(-1), out.reshape(-1)
        if flat_s.nbytes < (8 << 20):
            np.copyto(flat_o, flat_s)
        else:
            pool = getattr(self, "_copy_pool", None)
            if pool is None:
                from concurrent.futures import ThreadPoolExecutor
                pool = self._copy_pool = ThreadPoolExecutor(max_workers=4)
            step = (flat_s.size + 3) // 4
            list(pool.map(lambda i: np.copyto(flat_o[i * step: (i + 1) * step], flat_s[i * step: (i + 1) * step]), range(4)))
        return out

    def decode_to_wavs(self, result_list: List[torch.Tensor], pad_to: Optional[int] = None) -> torch.Tensor:
        """`Chat._decode_to_wavs` (core.py:513-539): zero-pad the per-row [T_b,768] hidden lists to the
        longest row, DVAE decode, Vocos decode -> [B, 256(2Tmax-1)] float32 on the device.  `pad_to`: pad to that many tokens instead
        (>= the longest row): a data-parallel shard decodes its rows as part of the GLOBAL batch (dist.infer_sharded) -- the reference
        decodes a shorter row's tail from zero hidden states and returns it."""
        if len(result_list) == 0:
            return torch.empty((0,), dtype=torch.float32)
        longest = max(int(r.size(0)) for r in result_list)
        if pad_to is not None and int(pad_to) < longest:
            raise ValueError("pad_to is shorter than the longest row")
        pad = getattr(result_list, "padded", None)
        if (pad is not None and pad.dim() == 3 and pad.shape[0] == len(result_list) and pad.shape[2] == GPT.hidden and pad.device == self.device
                and pad.dtype == torch.float32 and pad.shape[1] == longest and (pad_to is None or int(pad_to) == longest)):
            return self.vocos_decode(self.dvae_decode(pad))      # GptEngine.generate's rows: views of exactly this batch (RowList)
        Tmax = longest if pad_to is None else int(pad_to)
        batch = torch.zeros((len(result_list), Tmax, GPT.hidden), dtype=torch.float32, device=self.device)
        for i, r in enumerate(result_list):
            batch[i, : r.size(0)] = r
        return self.vocos_decode(self.dvae_decode(batch))
