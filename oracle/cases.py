"""Seeded input recipes for the golden cases (shared by oracle/make_goldens.py and tests/).
numpy RandomState only, so the inputs are bit-reproducible everywhere.  TEST INFRASTRUCTURE."""
from __future__ import annotations

import numpy as np

from chattts_amd import synth

f32 = np.float32
V = 626

# ---- sampling-kernel unit cases (rows = B*4) -------------------------------------------------
SAMPLING_CASES = {
    # default InferCodeParams (core.py:195-206): top_P .7, top_K 20, rep 1.05, temp .3
    "default_h0": dict(rows=32, hist=0, top_P=0.7, top_K=20, rep=1.05, temp=0.3, seed=42, mask_eos=False, scale=4.0, inseed=1),
    "default_h5": dict(rows=32, hist=5, top_P=0.7, top_K=20, rep=1.05, temp=0.3, seed=42, mask_eos=False, scale=4.0, inseed=2),
    "default_h16": dict(rows=32, hist=16, top_P=0.7, top_K=20, rep=1.05, temp=0.3, seed=7, mask_eos=True, scale=4.0, inseed=3),
    "default_h40": dict(rows=64, hist=40, top_P=0.7, top_K=20, rep=1.05, temp=0.3, seed=42, mask_eos=False, scale=4.0, inseed=4),
    # tests/#511.py:31-37 "greedy": top_K=1, top_P=0.005 (min_tokens_to_keep=3 still keeps 3)
    "greedy511": dict(rows=32, hist=8, top_P=0.005, top_K=1, rep=1.05, temp=0.3, seed=42, mask_eos=False, scale=4.0, inseed=5),
    # flat distribution: top-p keeps many, top-k cuts to 20
    "flat": dict(rows=32, hist=3, top_P=0.7, top_K=20, rep=1.05, temp=1.0, seed=11, mask_eos=False, scale=0.5, inseed=6),
    # top-p only / top-k only / neither / no penalty
    "p_only": dict(rows=16, hist=4, top_P=0.9, top_K=None, rep=1.2, temp=0.7, seed=3, mask_eos=False, scale=2.0, inseed=7),
    "k_only_ties": dict(rows=16, hist=4, top_P=None, top_K=5, rep=None, temp=1.0, seed=3, mask_eos=False, scale=2.0, inseed=8, quant=0.5),
    "none": dict(rows=16, hist=4, top_P=None, top_K=None, rep=None, temp=0.5, seed=5, mask_eos=True, scale=3.0, inseed=9),
    # processors.py:24-27 quirk: rows >= 625 get no penalty (B >= 157)
    "rows640": dict(rows=640, hist=12, top_P=0.7, top_K=20, rep=1.5, temp=0.3, seed=42, mask_eos=False, scale=4.0, inseed=10),
}


def _random_sampling_cases(n: int = 36, seed: int = 77):
    """seeded random kernel cases: 4..96 rows, histories of 0..40 tokens, both warpers present / absent, top_K from 1 to above the
    vocabulary (above 64 the kernel leaves its counting fast path), penalties below / above 1, flat to peaky logits, logits quantised to
    0.5 / 2.0 so that ties sit on the top-k and top-p boundaries"""
    rs = np.random.RandomState(seed)
    out = {}
    for i in range(n):
        out[f"rnd{i:02d}"] = dict(
            rows=4 * int(rs.randint(1, 25)), hist=int(rs.choice([0, 1, 3, 8, 16, 17, 40])),
            top_P=[None, 0.05, 0.3, 0.7, 0.9, 0.999][int(rs.randint(6))], top_K=[None, 1, 2, 3, 4, 20, 64, 65, 100, 626, 700][int(rs.randint(11))],
            rep=[None, 0.9, 1.05, 1.5, 2.0][int(rs.randint(5))], temp=float(rs.choice([0.05, 0.3, 1.0, 1.7])), seed=int(rs.randint(1, 10 ** 6)),
            mask_eos=bool(rs.rand() < 0.3), scale=float(rs.choice([0.1, 1.0, 4.0, 8.0])), inseed=1000 + i,
            **({"quant": float(rs.choice([0.5, 2.0]))} if rs.rand() < 0.4 else {}))
    for c in out.values():
        # exactly equal logits that straddle the TOP-P cut are the one place where the reference's result is not a function of the values:
        # TopPLogitsWarper sorts with torch.sort(stable=False), whose order among ties is an implementation detail of the torch build (for
        # 626 all-equal logits and top_P = 0.05, torch 2.10 keeps tokens {0, 197..227}) -- DESIGN.md "Known deviations".  Ties stay in the
        # top-k-only cases, where every token tied with the k-th value survives (order-free).
        if c["top_P"] is not None:
            c.pop("quant", None)
    return out


SAMPLING_CASES.update(_random_sampling_cases())


def sampling_inputs(c):
    rs = np.random.RandomState(c["inseed"])
    logits = (rs.standard_normal((c["rows"], V)) * c["scale"]).astype(f32)
    if c.get("quant"):
        logits = (np.round(logits / c["quant"]) * c["quant"]).astype(f32)
    # histories drawn from the likely tokens so penalties actually hit the head of the distribution
    top = np.argsort(-logits, axis=1)[:, :8]
    pick = rs.randint(0, 8, size=(c["rows"], c["hist"]))
    hist = np.take_along_axis(top, pick, axis=1).astype(np.int64) if c["hist"] else np.zeros((c["rows"], 0), np.int64)
    temp = np.full(c["rows"], c["temp"], dtype=f32)
    return logits, hist, temp


# ---- end-to-end generation cases -------------------------------------------------------------
GEN_CASES = {
    # BASELINE config C1: one 16-token sentence, near-greedy (tests/#511.py:31-37 parameters)
    "c1": dict(B=1, t_min=16, t_max=16, pseed=0, temperature=[0.3] * 4, top_P=0.005, top_K=1, rep=1.05,
               max_new=48, min_new=0, manual_seed=42, keep_hidden_rows=[0], keep_logit_steps=[0, 1, 47]),
    # mixed-length left-padded batch, default sampling
    "b8": dict(B=8, t_min=12, t_max=28, pseed=1, temperature=[0.3] * 4, top_P=0.7, top_K=20, rep=1.05,
               max_new=64, min_new=8, manual_seed=42, keep_hidden_rows=[0, 5], keep_logit_steps=[0, 1, 30, 63]),
    # long run: 320 autoregressive steps (context up to ~350 keys: several attention blocks per wave); any argmax flip
    # anywhere would diverge the whole suffix
    "long": dict(B=2, t_min=20, t_max=30, pseed=3, temperature=[0.3] * 4, top_P=0.7, top_K=20, rep=1.05,
                 max_new=320, min_new=320, manual_seed=1234, keep_hidden_rows=[], keep_logit_steps=[]),
    # per-codebook temperatures, no seed => torch global CPU generator advances per step
    "unseeded": dict(B=2, t_min=10, t_max=14, pseed=2, temperature=[0.3, 0.5, 0.7, 1.0], top_P=0.7, top_K=20, rep=1.05,
                     max_new=24, min_new=0, manual_seed=None, global_seed=7, keep_hidden_rows=[1], keep_logit_steps=[0, 23]),
}


# ---- BASELINE-size cases (tests/golden/generate_big.npz): the widths / lengths BASELINE.json's configs name --------
BIG_CASES = {
    # C3 at full width: 64 utterances, the bench's own mixed-length left-padded prompts (synth seed 0, 16..48 tokens),
    # default sampling; rows hit EOS naturally at different steps after min_new (compaction across 64 rows, the
    # M = 64 projection tiles, 4-row groups up to global sampling row 255)
    "c3w": dict(B=64, t_min=16, t_max=48, pseed=0, temperature=[0.3] * 4, top_P=0.7, top_K=20, rep=1.05,
                max_new=40, min_new=6, manual_seed=42, keep_hidden_rows=[0, 37, 63], keep_logit_steps=[0, 39]),
    # C2: batch 1, 512 speech tokens (context up to 544 keys), nothing may flip over 512 autoregressive steps
    "c2": dict(B=1, t_min=32, t_max=32, pseed=7, temperature=[0.3] * 4, top_P=0.7, top_K=20, rep=1.05,
               max_new=512, min_new=512, manual_seed=42, keep_hidden_rows=[0], keep_logit_steps=[0, 511]),
}


# ---- the sampling-parameter space and batch widths beyond one GPU's share (tests/golden/generate_params.npz) ----------------------
# Everything `Chat.InferCodeParams` lets a caller turn (core.py:195-206 -> gen_logits, processors.py:36-58): no top-k / no top-p
# warper at all, top_K below min_tokens_to_keep, a wide nucleus at temperature 1, strong / no repetition penalty -- and a batch of
# 160 utterances = 640 sampling rows: the reference's own run of a batch wider than 64 rows per GPU, whose rows >= 625 get NO
# repetition penalty (processors.py:24-27); the engine must reproduce that through its GLOBAL row numbering, on one GPU and from a
# shard (SURVEY 8e caveats 1-2).
PARAM_CASES = {
    "wide160": dict(B=160, t_min=8, t_max=16, pseed=11, temperature=[0.3] * 4, top_P=0.7, top_K=20, rep=1.3,
                    max_new=24, min_new=4, manual_seed=42, keep_hidden_rows=[0, 156, 159], keep_logit_steps=[]),
    "hot": dict(B=5, t_min=10, t_max=18, pseed=12, temperature=[1.0, 0.9, 0.8, 0.7], top_P=0.95, top_K=50, rep=1.2,
                max_new=48, min_new=0, manual_seed=3, keep_hidden_rows=[4], keep_logit_steps=[0, 47]),
    "k2": dict(B=3, t_min=9, t_max=13, pseed=13, temperature=[0.5] * 4, top_P=0.3, top_K=2, rep=1.0,
               max_new=32, min_new=2, manual_seed=11, keep_hidden_rows=[1], keep_logit_steps=[0]),
    "ponly": dict(B=2, t_min=12, t_max=12, pseed=14, temperature=[0.3] * 4, top_P=0.5, top_K=None, rep=1.0,
                  max_new=32, min_new=0, manual_seed=5, keep_hidden_rows=[0], keep_logit_steps=[0]),
    "konly": dict(B=4, t_min=6, t_max=20, pseed=15, temperature=[0.6] * 4, top_P=None, top_K=8, rep=1.3,
                  max_new=32, min_new=0, manual_seed=6, keep_hidden_rows=[3], keep_logit_steps=[0]),
    "nowarp": dict(B=2, t_min=8, t_max=10, pseed=16, temperature=[0.2] * 4, top_P=None, top_K=None, rep=1.05,
                   max_new=24, min_new=0, manual_seed=8, keep_hidden_rows=[1], keep_logit_steps=[0]),
}


# ---- `InferCodeParams.max_new_token`'s default, 2048 (core.py:197), in full (tests/golden/generate_max.npz) ----------------------------
# batch 2 (one row left-padded), EOS masked to the last step: 2048 autoregressive steps, contexts up to 2088 keys -- the longest
# generation the reference's defaults allow; ~10 min of reference CPU time (its DynamicCache re-concatenates the KV every step)
MAX_CASES = {
    "max2048": dict(B=2, t_min=24, t_max=40, pseed=31, temperature=[0.3] * 4, top_P=0.7, top_K=20, rep=1.05,
                    max_new=2048, min_new=2048, manual_seed=42, keep_hidden_rows=[], keep_logit_steps=[]),
}


# ---- GPT.generate(stream=True): the yield schedule of gpt.py:579-589 (tests/golden/generate_stream.npz) -----------------------
# a yield whenever the count of steps with any unfinished row is a multiple of `stream_batch`, every row cut at its own end_idx,
# plus the final yield (a duplicate when the last step was a multiple).  GEN_CASES entries + the stream_batch to run them with.
GEN_STREAM_CASES = {
    "b8_s24": ("b8", 24),          # the default stream_batch; rows finish between yields
    "b8_s5": ("b8", 5),            # 12 yields + final; not a divisor of max_new = 64
    "unseeded_s7": ("unseeded", 7),   # global-generator draws, stream_batch not aligned with anything
    "c1_s16": ("c1", 16),          # 48 steps = 3 x 16: the final yield repeats the third
}


# ---- "unexpected end at index" (gpt.py:527-570, tests/golden/generate_regen.npz) ----------------------------------------------
# a row draws EOS at the very first step (min_new_token = 0, temperature 1, no warpers; the seed was searched for on the reference):
#   unseeded: the reference throws the attempt away and calls itself again -- the global generator has moved on by the one [rows, 626]
#             draw of the failed step 0, so the second attempt sees different draws;
#   seeded  : the reference logs the warning and RETURNS WITHOUT YIELDING anything (gpt.py:570).
REGEN_CASES = {
    "regen57": dict(B=6, t_min=8, t_max=14, pseed=21, temperature=[1.0] * 4, top_P=None, top_K=None, rep=1.05,
                    max_new=16, min_new=0, manual_seed=None, global_seed=57),
    "regen195": dict(B=6, t_min=8, t_max=14, pseed=21, temperature=[1.0] * 4, top_P=None, top_K=None, rep=1.05,
                     max_new=16, min_new=0, manual_seed=None, global_seed=195),
    "seeded57": dict(B=6, t_min=8, t_max=14, pseed=21, temperature=[1.0] * 4, top_P=None, top_K=None, rep=1.05,
                     max_new=16, min_new=0, manual_seed=57),
}


# ---- a seeded random sweep over everything GPT.generate's caller can turn (tests/golden/generate_sweep.npz: ids only) ------------------
def sweep_cases(n: int = 40, seed: int = 2025):
    """deterministic list of GEN_CASES-shaped dicts: batch widths around the 16-row tile edges, one-token prompts, max_new_token = 1,
    min_new_token above max_new_token, top_K in {None, 1, 2, 3, ..., above the vocabulary}, top_P from 0.1 to 0.99 or absent, temperatures
    from 0.05 to 1.5 per codebook, repetition penalties below / at / above 1, seeded and unseeded"""
    rs = np.random.RandomState(seed)
    out = {}
    for i in range(n):
        B = int(rs.choice([1, 2, 3, 5, 7, 9, 16, 17, 33]))
        t_min = int(rs.randint(1, 13))
        t_max = t_min + int(rs.randint(0, 21))
        max_new = int(rs.choice([1, 2, 3, 8, 17, 24, 40]))
        seeded = bool(rs.rand() < 0.75)
        c = dict(B=B, t_min=t_min, t_max=t_max, pseed=100 + i,
                 temperature=[float(rs.choice([0.05, 0.3, 0.7, 1.0, 1.5])) for _ in range(4)],
                 top_P=[None, 0.1, 0.5, 0.7, 0.9, 0.99][int(rs.randint(6))],
                 top_K=[None, 1, 2, 3, 5, 20, 100, 700][int(rs.randint(8))],
                 rep=float(rs.choice([1.0, 1.05, 1.2, 2.0, 0.9])),
                 max_new=max_new, min_new=int(rs.randint(0, max_new + 3)),
                 manual_seed=int(rs.randint(1, 10 ** 6)) if seeded else None, global_seed=int(rs.randint(1, 10 ** 6)),
                 keep_hidden_rows=[], keep_logit_steps=[])
        out[f"s{i:02d}"] = c
    return out


def text_sweep_cases(n: int = 12, seed: int = 2026):
    """the same for refine-text mode (infer_text=True: one sampling row per utterance over the 21178-way text head, repetition penalty 1 --
    the only value the reference's text mode can run with)"""
    rs = np.random.RandomState(seed)
    out = {}
    for i in range(n):
        t_min = int(rs.randint(1, 10))
        max_new = int(rs.choice([1, 2, 5, 12, 20]))
        seeded = bool(rs.rand() < 0.75)
        out[f"t{i:02d}"] = dict(B=int(rs.choice([1, 2, 3, 6, 17])), t_min=t_min, t_max=t_min + int(rs.randint(0, 12)), pseed=300 + i,
                                temperature=[float(rs.choice([0.1, 0.7, 1.0, 1.3]))],
                                top_P=[None, 0.2, 0.7, 0.95][int(rs.randint(4))], top_K=[None, 1, 2, 20, 500, 30000][int(rs.randint(6))],
                                rep=1.0, max_new=max_new, min_new=int(rs.randint(0, max_new + 2)),
                                manual_seed=int(rs.randint(1, 10 ** 6)) if seeded else None, global_seed=int(rs.randint(1, 10 ** 6)),
                                keep_hidden_rows=[], keep_logit_steps=[])
    return out


# refine-text mode (core.py:665-751 defaults: temperature 0.7, top_P 0.7, top_K 20, repetition_penalty 1.0)
TEXT_EOS = 21000  # stands in for tokenizer.eos_token ([Ebreak]); any id works with synthetic weights
TEXT_CASES = {
    "text3": dict(B=3, t_min=9, t_max=15, pseed=4, temperature=[0.7], top_P=0.7, top_K=20, rep=1.0, max_new=24, min_new=2,
                  manual_seed=12345, keep_hidden_rows=[2], keep_logit_steps=[0, 5]),
    "text1_greedyish": dict(B=1, t_min=12, t_max=12, pseed=5, temperature=[0.3], top_P=0.1, top_K=3, rep=1.0, max_new=16, min_new=0,
                            manual_seed=7, keep_hidden_rows=[0], keep_logit_steps=[0]),
}


def gen_inputs(c):
    return synth.make_prompts(c["B"], c["t_min"], c["t_max"], seed=c["pseed"])


# ---- acoustic decoder cases ------------------------------------------------------------------
CODEC_CASES = {
    "c24": dict(B=2, T=24, seed=0),
    "c1x5": dict(B=1, T=5, seed=1),   # shorter than the receptive field: edge handling
    # round 5: the short end -- one token (2 mel frames, a 256-sample waveform: every convolution is all padding), 2 / 3 / 4 tokens, lengths
    # around the dilated kernels' reach, odd batches with a zero-padded second row
    "s1x1": dict(B=1, T=1, seed=11), "s1x2": dict(B=1, T=2, seed=12), "s2x3": dict(B=2, T=3, seed=13), "s3x4": dict(B=3, T=4, seed=14),
    "s1x7": dict(B=1, T=7, seed=15), "s4x9": dict(B=4, T=9, seed=16), "s2x13": dict(B=2, T=13, seed=17), "s3x33": dict(B=3, T=33, seed=18),
}


# ragged rows through the reference's own `Chat._decode_to_wavs` (core.py:513-539: zero padding to the longest row, [T, 768] -> [768, T],
# decoder, vocos) -- codec.npz `ragged.wav`
RAGGED_LENS = (40, 17, 33, 1)


def ragged_rows():
    rs = np.random.RandomState(4)
    return [rs.standard_normal((n, 768)).astype(f32) for n in RAGGED_LENS]


def codec_inputs(c):
    rs = np.random.RandomState(100 + c["seed"])
    hid = rs.standard_normal((c["B"], c["T"], 768)).astype(f32)
    if c["B"] > 1:
        hid[1, c["T"] * 2 // 3:] = 0  # a shorter row zero-padded like core.py:525-533
    return hid


# ---- acoustic decoder at BASELINE sizes (tests/golden/codec_big.npz) -----------------------------------------------------------
# Inputs are the REFERENCE's own GPT hidden states: contiguous runs of `c2.hid0` (tests/golden/generate_big.npz: the 512 final-norm
# rows of the reference's 512-step C2 generation), not random normals -- so nothing but the golden outputs has to be stored.
#   c2size  : BASELINE configs[1] -- 1 x 512 tokens = 1024 frames: the size from which the point-wise ConvNeXt pairs take the LDS-DMA
#             kernels (gemm_x3p_k / gemm_h1p_k, ctts_codec.x3p_min_rows)
#   r16x400 : 16 ragged rows (128..400 tokens, one of them 400) zero-padded to 400 like core.py:525-533 = 12800 frames: the size from
#             which the dense layers take the 256 x 256 split-bf16 tile and the depthwise conv the sliding-window kernel (>= 12288)
CODEC_BIG_CASES = {
    "c2size": dict(B=1, T=512, seed=None),
    "r16x400": dict(B=16, T=400, seed=3),
}
CODEC_BIG_MEL_STRIDE = 4     # the golden keeps mel frames f = (b + 4 k) ...: every 4th frame, phase = row index (all rows, all phases)
CODEC_BIG_WAV_STRIDE = 16    # ... and every 16th waveform sample, phase = row index


def codec_big_inputs(c, hid0: np.ndarray):
    """hid0: generate_big.npz['c2.hid0'] [512, 768] f32.  Returns (hid [B, T, 768] f32, lens [B])."""
    assert hid0.shape == (512, 768)
    if c["seed"] is None:
        return hid0[None, : c["T"]].astype(f32).copy(), np.array([c["T"]], np.int64)
    rs = np.random.RandomState(200 + c["seed"])
    lens = rs.randint(128, c["T"] + 1, size=c["B"]).astype(np.int64)
    lens[rs.randint(c["B"])] = c["T"]
    hid = np.zeros((c["B"], c["T"], 768), f32)
    for b in range(c["B"]):
        off = rs.randint(0, 512 - lens[b] + 1)
        hid[b, : lens[b]] = hid0[off: off + lens[b]]
    return hid, lens


def codec_big_subsample(mel: np.ndarray, wav: np.ndarray):
    """what the golden stores of a case's outputs (mel [B,100,F], wav [B,N]): strided samples whose phase walks with the row, plus
    float64 block sums that cover every element (mel: 32-frame blocks, wav: 2048-sample blocks)"""
    B, _, F = mel.shape
    N = wav.shape[1]
    ms, ws = CODEC_BIG_MEL_STRIDE, CODEC_BIG_WAV_STRIDE
    mel_s = np.stack([mel[b][:, (b % ms)::ms][:, : F // ms] for b in range(B)])
    wav_s = np.stack([wav[b][(b % ws)::ws][: N // ws] for b in range(B)])
    nbm, nbw = F // 32, N // 2048
    mel_blk = mel[:, :, : nbm * 32].astype(np.float64).reshape(B, 100, nbm, 32).sum(-1)
    wav_blk = wav[:, : nbw * 2048].astype(np.float64).reshape(B, nbw, 2048).sum(-1)
    wav_sq = (wav[:, : nbw * 2048].astype(np.float64) ** 2).reshape(B, nbw, 2048).sum(-1)
    return dict(mel_s=mel_s, wav_s=wav_s, mel_blk=mel_blk, wav_blk=wav_blk, wav_sq=wav_sq,
                mel_peak=np.array([np.abs(mel).max()], np.float64), wav_rms=np.array([np.sqrt(np.mean(wav.astype(np.float64) ** 2))]))


# ---- host / device back end (tests/golden/backend.npz): float_to_int16 and ChatStreamer ------------------------------------------
def pcm_inputs():
    """waveforms for the float_to_int16 goldens: name -> float32 array (1-D = one utterance, 2-D = a [B, n] block).  Covers a peak below 1
    (scale 32767), above 1 (ceil 2: scale 16383) and above 2, exact +-peak samples, values around the 1e-5 strip threshold, zeros
    inside, a length that is not a multiple of 8, and products that sit within an ulp of an integer."""
    rs = np.random.RandomState(77)
    out = {}
    a = (rs.standard_normal(10007) * 0.11).astype(f32)
    a[100], a[200], a[300:310] = 0.73, -0.73, 0.0
    out["utt_quiet"] = a
    b = (rs.standard_normal(8192) * 0.35).astype(f32)
    b[5], b[6] = 1.5, -1.25
    out["utt_peak2"] = b
    c = (rs.standard_normal((4, 6001)) * np.array([[0.2], [0.02], [1e-5], [0.0]])).astype(f32)
    c[0, 17] = 0.999999
    out["block4"] = c
    d = (np.arange(1, 4097, dtype=np.float64) / 32767.0).astype(f32)          # x * 32767 within rounding of the integers 1..4096
    d = np.concatenate([d, -d, np.nextafter(d, f32(0)), np.nextafter(d, f32(2))]).astype(f32)
    out["integers"] = d
    e = (rs.standard_normal(3000) * 0.9).astype(f32)
    e[0] = 2.5
    out["utt_peak3"] = e
    return out


def stream_chunks(name: str):
    """chunk sequences like Chat.infer(stream=True) yields them ([B, n] float32, utterances that have finished are silent from then on)
    for the ChatStreamer goldens"""
    if name.startswith("rnd"):   # seeded random: 1..4 utterances ending at random times (silent afterwards), 1..12 chunks of 100..30000
        rs = np.random.RandomState(900 + int(name[3:]))   # samples, some chunks silent for everybody, some utterances silent throughout
        B, n_chunks = int(rs.randint(1, 5)), int(rs.randint(1, 13))
        widths = [int(rs.choice([100, 700, 3000, 12000, 12001, 24000, 30000])) for _ in range(n_chunks)]
        total = sum(widths)
        ends = [0 if rs.rand() < 0.15 else int(rs.randint(1, total + 1)) for _ in range(B)]
        chunks, pos = [], 0
        for n in widths:
            ch = (rs.standard_normal((B, n)) * float(rs.choice([0.05, 0.3, 1.2]))).astype(f32)
            for b, e in enumerate(ends):
                ch[b, max(0, min(n, e - pos)):] = 0.0
            if rs.rand() < 0.15:
                ch[:] = 0.0
            chunks.append(ch)
            pos += n
        return chunks
    rs = np.random.RandomState({"three": 5, "one": 6, "late": 7}[name])
    if name == "one":        # a single utterance, chunks shorter than a block
        return [(rs.standard_normal((1, n)) * 0.2).astype(f32) for n in (3000, 3000, 3000, 12000, 500, 7000)]
    if name == "three":      # three utterances of different lengths: 0 ends first, then 2, then 1
        ends = [30000, 90000, 55000]
        chunks, pos = [], 0
        for n in [12000] * 8 + [7000]:
            ch = (rs.standard_normal((3, n)) * 0.15).astype(f32)
            for b, e in enumerate(ends):
                k = max(0, min(n, e - pos))
                ch[b, k:] = 0.0
            chunks.append(ch)
            pos += n
        return chunks
    # "late": utterance 0 silent from the start, a silent chunk in the middle, a tail that stays under the block size
    chunks = []
    for i, n in enumerate((9000, 9000, 4000, 9000, 2500, 2500)):
        ch = (rs.standard_normal((2, n)) * 0.3).astype(f32)
        ch[0] = 0.0
        if i == 2:
            ch[:] = 0.0
        chunks.append(ch)
    return chunks


STREAM_CASES = ("three", "one", "late") + tuple(f"rnd{i}" for i in range(16))
