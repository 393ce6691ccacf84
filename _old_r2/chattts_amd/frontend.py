"""Host front end of the hot path: what `Chat._infer_code` / `Chat._refine_text` do to a
string before the engine sees tensors -- text normalisation, prompt decoration, tokenisation with left padding and
an optional audio-code prompt, and the speaker vector that replaces the embedding at `[spk_emb]` positions.

All of it is host string / small-tensor work (nothing here is on the GPU roofline); it exists so that
`chattts_amd.core.Chat.infer(text, ...)` is the same call as the reference's.  Reference anchors:
  * `Normalizer`          /root/reference/ChatTTS/norm.py:67-253
  * `Tokenizer`           /root/reference/ChatTTS/model/tokenizer.py:17-138
  * `Speaker`             /root/reference/ChatTTS/model/speaker.py:10-154
  * base16384 strings     third-party `pybase16384` (un-pinned in requirements.txt, not installed here): the public
                          base16384 format is restated in `b14_encode` / `b14_decode` and pinned by decoding the
                          reference's own `Config.spk_stat` string (tests/golden/spk_stat.txt -> 2 x 768 float16).
"""
from __future__ import annotations

import json
import logging
import lzma
import re
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F

# ---------------------------------------------------------------------------------------------------------------------
# base16384: 7 bytes <-> 4 code points of 14 bits each, big-endian bit order, offset U+4E00; a trailing U+3D0r
# marks an input whose length was r (mod 7).
# ---------------------------------------------------------------------------------------------------------------------
_B14_BASE = 0x4E00
_B14_TAIL = 0x3D00


def b14_encode(data: bytes) -> str:
    n = len(data)
    if n == 0:
        return ""
    r = n % 7
    bits = np.unpackbits(np.frombuffer(data, dtype=np.uint8))
    nchar = (n // 7) * 4 + (8 * r + 13) // 14
    pad = nchar * 14 - bits.size
    if pad:
        bits = np.concatenate([bits, np.zeros(pad, dtype=np.uint8)])
    vals = (bits.reshape(nchar, 14).astype(np.int64) << np.arange(13, -1, -1)).sum(1) + _B14_BASE
    s = "".join(map(chr, vals.tolist()))
    return s + chr(_B14_TAIL + r) if r else s


def b14_decode(s: str) -> bytes:
    r = 0
    if s and (ord(s[-1]) & 0xFF00) == _B14_TAIL:
        r = ord(s[-1]) & 0xFF
        s = s[:-1]
    if not s:
        return b""
    v = np.fromiter(map(ord, s), dtype=np.int64, count=len(s)) - _B14_BASE
    if ((v < 0) | (v >= 1 << 14)).any() or r > 6:
        raise ValueError("not a base16384 string")
    bits = ((v[:, None] >> np.arange(13, -1, -1)) & 1).astype(np.uint8).reshape(-1)
    nbytes = ((len(s) - (8 * r + 13) // 14) // 4) * 7 + r
    return np.packbits(bits[: nbytes * 8]).tobytes()


_LZMA_KW = dict(format=lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA2, "preset": 9 | lzma.PRESET_EXTREME}])


# ---------------------------------------------------------------------------------------------------------------------
# Speaker (speaker.py:10-154)
# ---------------------------------------------------------------------------------------------------------------------
@torch.inference_mode()
def apply_speaker(emb: torch.Tensor, spk_emb: Union[str, torch.Tensor], input_ids: torch.Tensor, spk_emb_ids: int,
                  inplace: bool = True) -> torch.Tensor:
    """speaker.py:21-51 (`Speaker.apply`; it uses no speaker state).  `emb` [B,T,H] may live on the GPU; the vector
    is normalised with the same torch call on its own device/dtype as the reference (float16 on the host when it comes
    from a string), then moved, so the substituted values are bit-identical.  Only slot 0 of `input_ids` is looked at."""
    vec = torch.from_numpy(Speaker.decode_vector(spk_emb)) if isinstance(spk_emb, str) else spk_emb
    unit = F.normalize(vec, p=2.0, dim=0, eps=1e-12).to(emb.device)
    at_spk = input_ids[..., 0].to(emb.device).eq(spk_emb_ids).unsqueeze(-1)          # [B,T,1]
    out = torch.where(at_spk, unit.to(emb.dtype).view(1, 1, -1), emb)
    if inplace:
        emb.copy_(out)
        return emb
    return out


class Speaker:
    """`spk_cfg` is the reference's `Config.spk_stat` string: float16 [2*dim] = (std, mean) (speaker.py:11-16)."""

    def __init__(self, dim: int, spk_cfg: str, device=torch.device("cpu")):
        stat = torch.from_numpy(np.frombuffer(b14_decode(spk_cfg), dtype=np.float16).copy()).to(device)
        if stat.numel() != 2 * dim:
            raise ValueError(f"spk_stat holds {stat.numel()} values, expected {2 * dim}")
        self.std, self.mean = stat.chunk(2)
        self.dim = dim

    # -- strings <-> tensors ----------------------------------------------------------------------------------
    @staticmethod
    def encode_vector(spk_emb: torch.Tensor) -> str:            # speaker.py:131-142 (`_encode`)
        raw = spk_emb.to(dtype=torch.float16, device="cpu").numpy().tobytes()
        return b14_encode(lzma.compress(raw, **_LZMA_KW))

    @staticmethod
    def decode_vector(spk_emb: str) -> np.ndarray:              # speaker.py:144-154 (`_decode`)
        return np.frombuffer(lzma.decompress(b14_decode(spk_emb), **_LZMA_KW), dtype=np.float16).copy()

    @staticmethod
    def encode_prompt(prompt: torch.Tensor) -> str:             # speaker.py:87-102: u16 shape header + LZMA2 body
        arr = prompt.cpu().numpy().astype(np.uint16)
        if arr.ndim != 2:
            raise AssertionError("prompt must be a 2D tensor")
        head = np.array(arr.shape, dtype="<u2").tobytes()
        return b14_encode(head + lzma.compress(arr.astype("<u2").tobytes(), **_LZMA_KW))

    @staticmethod
    def decode_prompt(prompt: str) -> torch.Tensor:             # speaker.py:104-119 -> int32 [num_vq, T]
        dec = b14_decode(prompt)
        shp = np.frombuffer(dec[:4], dtype="<u2")
        body = np.frombuffer(lzma.decompress(dec[4:], **_LZMA_KW), dtype="<u2")
        return torch.from_numpy(body.astype(np.int32)).view(int(shp[0]), int(shp[1]))

    def sample_random_tensor(self) -> torch.Tensor:             # speaker.py:121-129: consumes torch's global RNG
        return torch.randn(self.dim, device=self.std.device, dtype=self.std.dtype).mul_(self.std).add_(self.mean)

    def sample_random(self) -> str:                             # speaker.py:18-19
        return self.encode_vector(self.sample_random_tensor())

    # -- the one tensor op: put the unit-norm speaker vector at every `[spk_emb]` position -----------------------
    def apply(self, emb: torch.Tensor, spk_emb: Union[str, torch.Tensor], input_ids: torch.Tensor, spk_emb_ids: int,
              device: Optional[torch.device] = None, inplace: bool = True) -> torch.Tensor:
        return apply_speaker(emb, spk_emb, input_ids, spk_emb_ids, inplace)

    # -- prompt decoration (pure string work) ---------------------------------------------------------------------
    @staticmethod
    def decorate_code_prompts(text: Sequence[str], prompt: str, txt_smp: Optional[str], spk_emb: Optional[str]) -> List[str]:
        """speaker.py:53-80: strip user-typed control tags, prepend the style prompt, wrap in [Stts]...[Ptts]."""
        tag = "[spk_emb]" if spk_emb is not None else "[empty_spk]"
        smp = txt_smp or ""
        out = []
        for t in text:
            for banned in ("[Stts]", "[spk_emb]", "[empty_spk]"):
                t = t.replace(banned, "")
            t = t.strip()
            if prompt:
                t = prompt + t
            out.append(f"[Stts]{tag}{smp}{t}[Ptts]")
        return out

    @staticmethod
    def decorate_text_prompts(text: Sequence[str], prompt: str) -> List[str]:   # speaker.py:82-85
        return [f"[Sbreak]{t}[Pbreak]{prompt}" for t in text]


# ---------------------------------------------------------------------------------------------------------------------
# Tokenizer (tokenizer.py:17-138)
# ---------------------------------------------------------------------------------------------------------------------
class Tokenizer:
    """Wraps the `BertTokenizerFast` saved under `asset/tokenizer` (config.py:10).  `encode` returns the three
    tensors the engine consumes: `input_ids [B,T,num_vq] int64`, `attention_mask [B,T] int64`, `text_mask [B,T] bool`
    -- rows are LEFT padded with id 0 / mask 0, an audio-code prompt `[num_vq, P]` is appended to every row."""

    def __init__(self, tokenizer_path_or_obj):
        if isinstance(tokenizer_path_or_obj, (str, bytes)) or hasattr(tokenizer_path_or_obj, "__fspath__"):
            from transformers import BertTokenizerFast
            tok = BertTokenizerFast.from_pretrained(tokenizer_path_or_obj)
        else:
            tok = tokenizer_path_or_obj
        self._tokenizer = tok
        self.len = len(tok)
        self.spk_emb_ids = tok.convert_tokens_to_ids("[spk_emb]")
        self.break_0_ids = tok.convert_tokens_to_ids("[break_0]")
        self.eos_token = tok.convert_tokens_to_ids("[Ebreak]")

    @torch.inference_mode()
    def encode(self, text: Sequence[str], num_vq: int, prompt: Optional[torch.Tensor] = None,
               device="cpu") -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        P = 0
        if prompt is not None:
            assert prompt.size(0) == num_vq, "prompt dim 0 must equal to num_vq"
            P = int(prompt.size(1))
        rows = []
        for t in text:   # one string at a time, no special tokens added (tokenizer.py:54-60)
            enc = self._tokenizer(t, add_special_tokens=False, return_attention_mask=True)
            rows.append((torch.tensor(enc["input_ids"], dtype=torch.int64), torch.tensor(enc["attention_mask"], dtype=torch.int64)))
        T = max(r[0].numel() for r in rows) + P
        B = len(rows)
        ids = torch.zeros((B, T), dtype=torch.int64, device=device)
        attn = torch.zeros((B, T), dtype=torch.int64, device=device)
        for b, (i, m) in enumerate(rows):
            n = i.numel()
            ids[b, T - P - n: T - P] = i
            attn[b, T - P - n: T - P] = m
        if P:
            attn[:, T - P:] = 1
        text_mask = attn.bool()
        ids4 = ids.unsqueeze(-1).expand(-1, -1, num_vq).clone()
        if P:
            text_mask[:, T - P:] = False                      # code positions use the 4 code embeddings (embed.py:52-79)
            ids4[:, T - P:, :] = prompt.t().to(device=device, dtype=torch.int64).unsqueeze(0)
        return ids4, attn, text_mask

    @torch.inference_mode()
    def decode(self, sequences, skip_special_tokens: bool = False, clean_up_tokenization_spaces=None, **kw):
        return self._tokenizer.batch_decode(sequences, skip_special_tokens=skip_special_tokens,
                                            clean_up_tokenization_spaces=clean_up_tokenization_spaces, **kw)


# ---------------------------------------------------------------------------------------------------------------------
# Normalizer (norm.py:67-253)
# ---------------------------------------------------------------------------------------------------------------------
def _pairs(src: str, dst: str) -> Dict[int, str]:
    assert len(src) == len(dst)
    return {ord(a): b for a, b in zip(src, dst)}


# norm.py:95-121: punctuation the model was not trained on -> comma / full stop (only applied when a text holds
# characters outside the accepted set)
_SIMPLIFY = _pairs("：；！（）【】『』「」《》－:;!()><-",
                   "，，。，，，，，，，，，，，,,.,,,,,")
# norm.py:122-156: ASCII punctuation -> full-width forms for Chinese text ('[', ']' and '_' stay: control tags)
_HALF2FULL = _pairs("!\"'#$%&(),-*+./:;<=>?@\\^`{|}~",
                    "！“‘＃＄％＆（），－＊＋。／：；＜＝＞？＠＼＾｀｛｜｝～")

_TAG = re.compile(r"\[[\w_]+\]")                                   # norm.py:91 control tags such as [uv_break]
_REJECT = re.compile(r"[^\u4e00-\u9fffA-Za-z，。、,\. ]")           # norm.py:90 everything the model cannot read
_ZH_CHAR = re.compile(r"[\u4e00-\u9fff]")
_EN_WORD = re.compile(r"\b[A-Za-z]+\b")


def split_tags(text: str) -> Tuple[List[str], List[str]]:
    """norm.py:37-56: cut at '[' ... ']' -> (plain pieces, tags).  A piece is emitted before every tag (possibly
    empty) and once more for a non-empty remainder."""
    texts, tags, cur, tag = [], [], "", ""
    for c in text:
        if c == "[":
            texts.append(cur)
            cur, tag = "", c
        elif tag:
            tag += c
        else:
            cur += c
        if c == "]":
            tags.append(tag)
            tag = ""
    if cur:
        texts.append(cur)
    return texts, tags


def combine_tags(texts: List[str], tags: List[str]) -> str:      # norm.py:59-66
    tags = list(tags)
    return "".join(t + (tags.pop(0) if tags else "") for t in texts)


class Normalizer:
    def __init__(self, map_file_path: Optional[str] = None, logger=logging.getLogger(__name__)):
        self.logger = logger
        self.normalizers: Dict[str, Callable[[str], str]] = {}
        self.homophones: Dict[int, int] = {}
        if map_file_path is not None:
            with open(map_file_path, "r", encoding="utf-8") as f:
                self.homophones = {ord(k): ord(v) for k, v in json.load(f).items()}    # norm.py:222-229

    def __call__(self, text: str, do_text_normalization=True, do_homophone_replacement=True, lang: Optional[str] = None) -> str:
        if do_text_normalization:
            _lang = self.detect_language(text) if lang is None else lang
            if _lang in self.normalizers:
                texts, tags = split_tags(text)
                texts = [self.normalizers[_lang](t) for t in texts]
                text = combine_tags(texts, tags) if tags else texts[0]
            if _lang == "zh":
                text = text.translate(_HALF2FULL)
        invalid = set(_REJECT.findall(_TAG.sub("", text)))       # norm.py:231-234
        if invalid:
            self.logger.warning(f"found invalid characters: {invalid}")
            text = text.translate(_SIMPLIFY)
        if do_homophone_replacement and self.homophones:
            # norm.py:23-34 works on UTF-16 code units; the map's keys are BMP characters, so per-character
            # translation is the same thing
            hits = [(c, chr(self.homophones[ord(c)])) for c in text if ord(c) in self.homophones]
            if hits:
                text = text.translate(self.homophones)
                self.logger.info("replace homophones: " + ", ".join(f"{a}->{b}" for a, b in hits))
        if invalid:
            texts, tags = split_tags(text)
            texts = [_REJECT.sub("", t) for t in texts]
            text = combine_tags(texts, tags) if tags else texts[0]
        return text

    def register(self, name: str, normalizer: Callable[[str], str]) -> bool:   # norm.py:198-212
        if name in self.normalizers:
            self.logger.warning(f"name {name} has been registered")
            return False
        try:
            if not isinstance(normalizer("test string 测试字符串"), str):
                self.logger.warning("normalizer must have caller type (str) -> str")
                return False
        except Exception as e:  # noqa: BLE001 -- the reference swallows and reports any failure of the probe call
            self.logger.warning(e)
            return False
        self.normalizers[name] = normalizer
        return True

    def unregister(self, name: str):
        self.normalizers.pop(name, None)

    @staticmethod
    def detect_language(sentence: str) -> str:                    # norm.py:243-253
        return "zh" if len(_ZH_CHAR.findall(sentence)) > len(_EN_WORD.findall(sentence)) else "en"
