"""TEST INFRASTRUCTURE (oracle/): deterministic CPU stand-ins for the two engine seams of `Chat`, used to run the HOST
control flow of `Chat.infer` / `Chat._infer` -- text splitting, the refer-sentence speaker prompt, split batches, the streaming window
arithmetic (`stream_speed`, `pass_first_n_batches`, the `length` carried across split batches), the silence strip -- without a GPU:

  * oracle/make_host_goldens.py plugs them into the REFERENCE's own `Chat` class (imported from /root/reference, core.py:208-270 and
    :386-503 run unmodified) and records everything it returns / yields -> tests/golden/host_flow.json + .npz;
  * tests/test_host_flow.py plugs the same stand-ins into `chattts_amd.core.Chat` and requires the same bytes and the same calls.

Nothing here computes audio: a "hidden state" is a [T, 1] tensor of the utterance's number, a "waveform" a ramp that depends on the
utterance number and the sample index, silent (exact zeros) past the utterance's end and at every 97th sample (the reference strips
|x| <= 1e-5 samples ANYWHERE, core.py:258-266, not only at the tail).
"""
from __future__ import annotations

import zlib
from typing import List

import numpy as np
import torch


def text_len(t: str) -> int:
    """tokens the fake engine 'generates' for a text: 30..89, from a checksum of the text (stable across processes)"""
    return 30 + zlib.crc32(t.encode("utf-8")) % 60


def text_tag(t: str) -> float:
    return float(1 + zlib.crc32(t.encode("utf-8")) % 7)


class FakeOutputs:
    """what `GPT.GenerationOutputs` (gpt.py:276-285) offers its consumer"""

    def __init__(self, ids, hiddens, log):
        self.ids, self.hiddens, self.attentions = ids, hiddens, []
        self._log = log

    def destroy(self):
        self._log.append(["destroy"])


class Seams:
    """`log` collects every seam call in order: what text batch reached `_infer_code` with which flags and which speaker prompt."""

    def __init__(self, stream_batch: int = 24):
        self.log: List[list] = []
        self.stream_batch = stream_batch

    # -- seam 1: Chat._infer_code(text, stream, device, return_hidden / use_decoder, params) (core.py:542-662) ------------------
    def infer_code(self, text, stream, device, use_decoder, params):
        if not isinstance(text, list):
            text = [text]
        self.log.append(["infer_code", list(text), bool(stream), bool(use_decoder), params.spk_smp, params.txt_smp])
        lens = [text_len(t) for t in text]
        tags = [text_tag(t) for t in text]

        def outputs(upto):
            n = [min(l, upto) for l in lens]
            hid = [torch.full((k, 1), tag, dtype=torch.float32) for k, tag in zip(n, tags)]
            ids = [torch.full((k, 4), int(tag), dtype=torch.int64) for k, tag in zip(n, tags)]
            return FakeOutputs(ids, hid, self.log)

        if stream:      # the yield schedule of gpt.py:579-589: every stream_batch steps while any row is alive, then the final state
            step = self.stream_batch
            while step < max(lens):
                yield outputs(step)
                step += self.stream_batch
        yield outputs(max(lens))

    # -- seam 2: Chat._decode_to_wavs(result_list, use_decoder) (core.py:513-539) -> np.float32 [B, 256 (2 Tmax - 1)] -----------
    def decode_to_wavs(self, result_list, use_decoder=True):
        self.log.append(["decode", [int(r.shape[0]) for r in result_list], bool(use_decoder)])
        Tm = max(int(r.shape[0]) for r in result_list)
        n = 256 * (2 * Tm - 1)
        wav = np.zeros((len(result_list), n), np.float32)
        idx = np.arange(n)
        for b, r in enumerate(result_list):
            tag = float(r[0, 0]) if r.shape[0] else 0.0
            live = 256 * (2 * int(r.shape[0]) - 1)
            row = (np.float32(0.01) * np.float32(tag) + (idx % 1000).astype(np.float32) * np.float32(1e-4)).astype(np.float32)
            row[idx % 97 == 0] = 0
            row[live:] = 0
            wav[b] = row
        return wav

    def sample_audio_speaker(self, wav):
        s = f"SPK<{len(wav)}:{float(np.abs(wav).sum(dtype=np.float64)):.3f}>"
        self.log.append(["sample_audio_speaker", s])
        return s

    # -- refine-text seam: Chat._refine_text(text, device, params) (core.py:665-751) ---------------------------------------------
    def refine_text(self, text, device, params):
        if not isinstance(text, list):
            text = [text]
        self.log.append(["refine_text", list(text), params.prompt])
        # token ids: the characters' code points, with one control token (>= break_0) spliced in that the caller must drop
        ids = [torch.tensor([ord(ch) for ch in t] + [FakeTokenizer.break_0_ids + 3], dtype=torch.int64) for t in text]
        return FakeOutputs(ids, [], self.log)


class FakeTokenizer:
    break_0_ids = 0x20000      # above every code point used in the scenarios

    @staticmethod
    def decode(tokens):
        return ["".join(chr(int(i)) for i in row) + "~" for row in tokens]


class FakeNormalizer:
    def __init__(self, log):
        self.log = log

    def __call__(self, t, do_text_normalization=True, do_homophone_replacement=True, lang=None):
        self.log.append(["normalize", t, bool(do_text_normalization), bool(do_homophone_replacement), lang])
        return t.strip()


# ---- scenarios: (name, infer kwargs; the params objects are built per side from these plain fields) -------------------------------------
SCENARIOS = {
    "list3_nosplit": dict(text=["alpha one", "beta two is longer", "c"], split_text=False, skip_refine_text=True),
    "lines_split_refer": dict(text="first line\nsecond line here\nthird\nfourth and last\nfifth", split_text=True, skip_refine_text=True,
                              max_split_batch=2),
    "sentences_regex": dict(text="One sentence. Another one. 中文句子。第二句。tail without stop", split_text=True,
                            skip_refine_text=True, max_split_batch=4),
    "split_with_spk_smp": dict(text=["a b", "c d", "e f"], split_text=True, skip_refine_text=True, max_split_batch=2, spk_smp="GIVEN", txt_smp="given text"),
    "single_string_nosplit": dict(text="just one", split_text=False, skip_refine_text=True),
    "refine_then_code": dict(text=["refine me", "and me too"], split_text=False, skip_refine_text=False),
    "refine_only_split": dict(text=["only text", "second"], split_text=True, skip_refine_text=False, refine_text_only=True),
    "refine_only_nosplit": dict(text=["only text", "second"], split_text=False, skip_refine_text=False, refine_text_only=True),
    "use_decoder_false": dict(text=["codes path", "x"], split_text=False, skip_refine_text=True, use_decoder=False),
    "empty_list": dict(text=[], split_text=False, skip_refine_text=True),
    "empty_string_split": dict(text="", split_text=True, skip_refine_text=True),
    "norm_flags": dict(text=["  padded  "], split_text=False, skip_refine_text=True, do_text_normalization=False, do_homophone_replacement=False, lang="en"),
    "stream_default": dict(text=["stream a", "stream bb is long"], stream=True, split_text=False, skip_refine_text=True),
    "stream_fast": dict(text=["stream a", "stream bb is long", "zz"], stream=True, split_text=False, skip_refine_text=True, stream_speed=3000,
                        pass_first_n_batches=0),
    "stream_huge_window": dict(text=["stream a"], stream=True, split_text=False, skip_refine_text=True, stream_speed=10 ** 6, pass_first_n_batches=1),
    "stream_split_batches": dict(text="s one\ns two two\ns three\ns four is the longest of them\ns five", stream=True, split_text=True,
                                 skip_refine_text=True, max_split_batch=2, stream_speed=6000, pass_first_n_batches=1),
    "stream_short_never_passes": dict(text=["k"], stream=True, split_text=False, skip_refine_text=True, pass_first_n_batches=9),
}

def _random_scenarios(n: int = 40, seed: int = 7):
    """seeded random calls of Chat.infer: 0..7 sentences as a list / newline string / regex-split string, every flag, window sizes from
    a few hundred samples to larger than any waveform, 0..4 passed batches, split batches of 1..4"""
    rs = np.random.RandomState(seed)
    words = ["alpha", "beta", "gamma", "delta", "epsilon", "zeta", "eta", "theta", "iota", "kappa", "lambda", "mu", "中文", "句子", "x"]
    out = {}
    for i in range(n):
        k = int(rs.randint(0, 8))
        sents = [" ".join(str(rs.choice(words)) for _ in range(int(rs.randint(1, 6)))) + f" {i}.{j}" for j in range(k)]
        form = int(rs.randint(3))
        split = bool(rs.rand() < 0.5)
        if form == 0 or not split:
            text = sents if (k != 1 or rs.rand() < 0.5) else sents[0]
        elif form == 1:
            text = "\n".join(sents)
        else:
            text = "".join(t + (". " if rs.rand() < 0.5 else "。") for t in sents)
        sc = dict(text=text, split_text=split, skip_refine_text=bool(rs.rand() < 0.7), stream=bool(rs.rand() < 0.5),
                  use_decoder=bool(rs.rand() < 0.8), max_split_batch=int(rs.randint(1, 5)))
        if not sc["skip_refine_text"] and rs.rand() < 0.3:
            sc["refine_text_only"] = True
        if sc["stream"]:
            sc["stream_speed"] = int(rs.choice([500, 3000, 12000, 24000, 10 ** 6]))
            sc["pass_first_n_batches"] = int(rs.randint(0, 5))
        if rs.rand() < 0.3:
            sc["spk_smp"], sc["txt_smp"] = "GIVEN", "given text"
        out[f"rand{i:02d}"] = sc
    return out


SCENARIOS.update(_random_scenarios())

INFER_KEYS = ("stream", "lang", "skip_refine_text", "refine_text_only", "use_decoder", "do_text_normalization", "do_homophone_replacement",
              "split_text", "max_split_batch")
CODE_PARAM_KEYS = ("spk_smp", "txt_smp", "stream_speed", "pass_first_n_batches")


def call_infer(chat, sc: dict, code_params, refine_params):
    """chat.infer(...) with the scenario's arguments; returns a JSON-able description + the arrays, generators drained"""
    kw = {k: sc[k] for k in INFER_KEYS if k in sc}
    for k in CODE_PARAM_KEYS:
        if k in sc:
            setattr(code_params, k, sc[k])
    res = chat.infer(sc["text"], params_refine_text=refine_params, params_infer_code=code_params, **kw)
    arrays = []
    if isinstance(res, str):
        desc = {"kind": "str", "value": res}
    elif isinstance(res, list):
        if all(isinstance(r, str) for r in res) and len(res):
            desc = {"kind": "list_str", "value": list(res)}
        else:
            desc = {"kind": "list", "n": len(res)}
            arrays = [np.asarray(r) for r in res]
    else:       # generator (stream=True)
        arrays = [np.asarray(r) for r in res]
        desc = {"kind": "stream", "n": len(arrays)}
    desc["shapes"] = [list(a.shape) for a in arrays]
    desc["dtypes"] = [str(a.dtype) for a in arrays]
    desc["spk_smp_after"] = code_params.spk_smp
    desc["txt_smp_after"] = code_params.txt_smp
    return desc, arrays


# ---- the argument side of engine seam #1: what `Chat._infer_code` / `Chat._refine_text` hand to `GPT.generate` ----------------------
# (oracle/make_seam_goldens.py runs the reference's two methods unmodified -- real reference Tokenizer on the synthetic vocabulary of
# tests/golden/tokenizer, real reference Speaker and Embed -- with a recorder in place of `GPT.generate`; tests/test_seam_args.py does the
# same with chattts_amd.core.Chat and a recorder in place of the engine.)  "$SPK" / "$SMP" are replaced by the speaker / audio-prompt strings
# of tests/golden/frontend.json (`speaker.sample_str`, `speaker.prompt_str`).
GENERATE_ARGS = ("emb", "inputs_ids", "temperature", "eos_token", "attention_mask", "max_new_token", "min_new_token", "logits_processors",
                 "infer_text", "return_attn", "return_hidden", "stream", "show_tqdm", "ensure_non_empty", "stream_batch", "manual_seed", "context")

SEAM_CODE_SCENARIOS = {
    "defaults_one_text": dict(text=["what is your favorite english food?"], stream=False, return_hidden=True, params={}),
    "string_not_list": dict(text="hello world", stream=False, return_hidden=False, params=dict(prompt="[speed_2]")),
    "speaker_three_texts": dict(text=["chat tts test string [laugh] like that", "你好", "the time of day"], stream=True, return_hidden=True,
                                params=dict(prompt="[speed_3]", spk_emb="$SPK", temperature=[0.1, 0.2, 0.4, 0.8], top_P=None, top_K=5,
                                            repetition_penalty=1.0, max_new_token=100, min_new_token=5, manual_seed=7, stream_batch=12,
                                            ensure_non_empty=False, show_tqdm=False)),
    "audio_prompt": dict(text=["hello world", "四川美食确实以辣闻名"], stream=False, return_hidden=True,
                         params=dict(spk_smp="$SMP", txt_smp="sample text", top_P=0.9, top_K=None, repetition_penalty=1.3, temperature=0.0003)),
    "speaker_and_audio_prompt": dict(text=["hot water"], stream=False, return_hidden=True,
                                     params=dict(spk_emb="$SPK", spk_smp="$SMP", txt_smp="the way", max_new_token=17)),
}

SEAM_REFINE_SCENARIOS = {
    "defaults": dict(text=["what is your favorite food", "你好"], params={}),
    "custom": dict(text="hello world", params=dict(prompt="[oral_2][laugh_0][break_6]", top_P=0.5, top_K=10, temperature=0.5, repetition_penalty=1.2,
                                                   max_new_token=77, min_new_token=3, manual_seed=11, show_tqdm=False, ensure_non_empty=False)),
}


def describe_processors(procs) -> list:
    """class-agnostic description of a logits-processor chain: the reference's objects (transformers' warpers, processors.py:8-35) and
    this package's descriptors expose the same numbers under these attribute names"""
    out = []
    for p in procs:
        if hasattr(p, "top_p"):
            out.append(["top_p", float(p.top_p), int(p.min_tokens_to_keep)])
        elif hasattr(p, "top_k"):
            # transformers' TopKLogitsWarper stores max(top_k, min_tokens_to_keep) as `top_k`; the descriptor keeps both numbers
            out.append(["top_k", max(int(p.top_k), int(getattr(p, "min_tokens_to_keep", 0)))])
        elif hasattr(p, "penalty"):
            out.append(["penalty", float(p.penalty), int(p.max_input_ids), int(p.past_window)])
        else:
            out.append(["?", type(p).__name__])
    return out


def describe_generate_call(real_generate, args, kwargs) -> dict:
    """positional + keyword arguments of one generate call, bound to the signature of the REAL function the recorder stands in for
    (its defaults fill what the caller left out) -> {name: JSON-able value} over the reference's 17 parameters; tensors by shape / dtype /
    values, the embedding by checksum"""
    import hashlib
    import inspect
    ba = inspect.signature(real_generate).bind(None, *args, **kwargs)
    ba.apply_defaults()
    out = {}
    for k in GENERATE_ARGS:
        v = ba.arguments[k]
        if k == "context":
            out[k] = "context" if v is not None else None
        elif k == "logits_processors":
            out[k] = describe_processors(v)
        elif k == "emb":
            a = v.detach().cpu().numpy()
            out[k] = {"shape": list(a.shape), "dtype": str(a.dtype), "sha256": hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest(),
                      "abs_sum": float(np.abs(a).sum(dtype=np.float64))}
        elif isinstance(v, torch.Tensor):
            out[k] = {"shape": list(v.shape), "dtype": str(v.dtype).replace("torch.", ""), "values": v.detach().cpu().tolist()}
        else:
            out[k] = v
    return out
