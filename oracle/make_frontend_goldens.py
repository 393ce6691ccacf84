"""BUILD-CONTAINER ONLY.  Golden vectors for the host front end (chattts_amd/frontend.py), produced by the REAL
reference classes imported from /root/reference:

  tests/golden/tokenizer/        a small synthetic BertTokenizerFast (the real `asset/tokenizer` is not reachable
                                 offline) holding every control token the reference's prompts use
  tests/golden/frontend.json     Normalizer outputs, decorated prompts, `Tokenizer.encode` tensors (with / without an
                                 audio-code prompt), `Tokenizer.decode` strings, `Speaker.apply` output checksum inputs
  tests/golden/spk_stat.txt      `Config.spk_stat` (config.py:132): the known-answer string for the base16384 codec
  tests/golden/homophones_small.json   8-entry synthetic homophone map used by both sides

`pybase16384` is not installed, so the reference `Speaker` is imported with that module stubbed by OUR codec
(frontend.b14_encode / b14_decode); this checks the Speaker logic around the codec, while the codec itself is pinned
by spk_stat decoding to exactly 2 x 768 finite float16 values (positive std half).  `numba.jit` is stubbed by the
identity decorator for norm.py.

    python -m oracle.make_frontend_goldens
"""
from __future__ import annotations

import importlib
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"

CONTROL = ["[Stts]", "[Ptts]", "[spk_emb]", "[empty_spk]", "[Sbreak]", "[Pbreak]", "[Ebreak]", "[uv_break]", "[v_break]",
           "[lbreak]", "[llbreak]", "[laugh]", "[undefine]"] + [f"[speed_{i}]" for i in range(10)] + \
          [f"[oral_{i}]" for i in range(10)] + [f"[laugh_{i}]" for i in range(3)] + [f"[break_{i}]" for i in range(8)]
WORDS = ("what is your favorite english food like the a of and to in it you that he was for on are with as i his they be at "
         "one have this from or had by hot but some we can out other were all there when up use how said an each she which do "
         "their time if will way about many then them would write so these her long make thing see him two has look more day "
         "could go come did my sound no most number who over know water than call first people may down side been now find "
         "chat tts test string hello world ##s ##ing ##ed ##ly").split()
ZH = list("四川美食确实以辣闻名但也有不辣的选择比如甜水面赖汤圆蛋烘糕叶儿粑等这些小吃口味温和甜而不腻也很受欢迎测试字符串你好")
PUNCT = list("?.,!'，。、？！")


def build_tokenizer(out_dir: str):
    from transformers import BertTokenizerFast
    os.makedirs(out_dir, exist_ok=True)
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + PUNCT + sorted(set(WORDS)) + sorted(set(ZH)) + list("abcdefghijklmnopqrstuvwxyz") \
        + ["##" + c for c in "abcdefghijklmnopqrstuvwxyz"]
    seen, uniq = set(), []
    for v in vocab:
        if v not in seen:
            seen.add(v)
            uniq.append(v)
    tok = BertTokenizerFast(vocab={v: i for i, v in enumerate(uniq)}, do_lower_case=True)   # transformers 5.x signature
    tok.add_special_tokens({"additional_special_tokens": CONTROL})
    tok.save_pretrained(out_dir)
    return out_dir


def ref_frontend():
    for name, path in [("ChatTTS", f"{REF}/ChatTTS"), ("ChatTTS.model", f"{REF}/ChatTTS/model")]:
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [path]
            sys.modules[name] = m
    sys.path.insert(0, ROOT)
    from chattts_amd import frontend as F
    b14 = types.ModuleType("pybase16384")
    b14.encode_to_string = F.b14_encode
    b14.decode_from_string = F.b14_decode
    sys.modules["pybase16384"] = b14
    nb = types.ModuleType("numba")
    nb.jit = lambda *a, **k: (lambda fn: fn)
    sys.modules.setdefault("numba", nb)
    tok = importlib.import_module("ChatTTS.model.tokenizer")
    spk = importlib.import_module("ChatTTS.model.speaker")
    norm = importlib.import_module("ChatTTS.norm")
    cfg = importlib.import_module("ChatTTS.config.config")
    return tok, spk, norm, cfg


NORM_CASES = [
    ("What is [uv_break]your favorite english food?[laugh][lbreak]", True, True, None),
    ("四川美食确实以辣闻名,但也有不辣的选择.比如甜水面!", True, True, None),
    ("四川美食(确实)以辣闻名;但也有不辣的选择:比如甜水面", True, False, None),
    ("hello world: this (is) a test-string; ok!", True, True, None),
    ("hello <world> 100% [laugh] done", True, True, "en"),
    ("chat T T S is a text to speech model [uv_break] designed for dialogue 测试", False, True, None),
    ("你好 hello 你好 world 测试 字符串 tts", True, True, "zh"),
    ("[laugh]", True, True, None),
    ("plain text without anything", True, True, None),
    ("甜[uv_break]而不腻#也很受欢迎[lbreak]", True, True, None),
]
CODE_TEXTS = [
    ["What is [uv_break]your favorite english food?[laugh][lbreak]"],
    ["[Stts] hello [spk_emb]world [empty_spk] ", "the time of day", "chat tts test string [laugh] like that"],
    ["四川美食确实以辣闻名", "hello world", "你好"],
]


def norm_sweep_cases(n: int = 120, seed: int = 31):
    """seeded random strings over everything norm.py looks at: ASCII / CJK letters (incl. the keys of the homophone map), digits, both
    punctuation families and the characters its tables rewrite, control tags (whole, unclosed, nested), brackets, whitespace runs,
    characters outside every accepted range; random (do_text_normalization, do_homophone_replacement, lang)"""
    rs = np.random.RandomState(seed)
    pools = [list("abcdefghijklmnopqrstuvwxyzABCDEFXYZ"), list("0123456789"), list("四川美食辣面圆糕甜腻你好选择比如水汤蛋"),
             list(",.!?;:'\"-_()<>{}#%&*+=/~`@$^|\\"), list("，。！？；：、（）《》【】“”‘’—…·"), [" ", "  ", "\t", "\n"],
             ["[uv_break]", "[laugh]", "[lbreak]", "[speed_5]", "[oral_2]", "[break_6]", "[", "]", "[[x]", "[unclosed"],
             ["é", "ß", "ø", "Ω", "я", "あ", "한", "😀", "\u200b", "１２", "ＡＢ"]]
    weights = np.array([6, 2, 6, 3, 3, 2, 2, 1], np.float64)
    weights /= weights.sum()
    out = []
    for _ in range(n):
        k = int(rs.randint(1, 30))
        t = "".join(str(rs.choice(pools[int(rs.choice(len(pools), p=weights))])) for _ in range(k))
        out.append((t, bool(rs.rand() < 0.8), bool(rs.rand() < 0.7), [None, None, "en", "zh"][int(rs.randint(4))]))
    return out


def main():
    tok_dir = build_tokenizer(os.path.join(GOLD, "tokenizer"))
    tokm, spkm, normm, cfgm = ref_frontend()
    cfg = cfgm.Config()
    with open(os.path.join(GOLD, "spk_stat.txt"), "w", encoding="utf-8") as f:
        f.write(cfg.spk_stat)
    hmap = dict(zip("辣面圆糕甜腻你好", "拉免元高田泥尼号"))
    hpath = os.path.join(GOLD, "homophones_small.json")
    with open(hpath, "w", encoding="utf-8") as f:
        json.dump(hmap, f, ensure_ascii=False)

    out = {}
    nz = normm.Normalizer(hpath)
    nz.register("en", lambda s: s.replace("100%", "one hundred percent"))
    out["norm"] = [{"in": t, "tn": tn, "hp": hp, "lang": lang, "out": nz(t, tn, hp, lang)} for t, tn, hp, lang in NORM_CASES]

    out["norm_sweep"] = [{"in": t, "tn": tn, "hp": hp, "lang": lang, "out": nz(t, tn, hp, lang)} for t, tn, hp, lang in norm_sweep_cases()]

    T = tokm.Tokenizer(tok_dir)
    if not hasattr(T._tokenizer, "encode_plus"):
        # harness-side shim: transformers 5.x dropped `encode_plus` (the reference, written against >=4.41, calls it at
        # tokenizer.py:56); `__call__` on one string is the same operation
        type(T._tokenizer).encode_plus = lambda self, *a, **k: self(*a, **k)
    S = spkm.Speaker
    out["tokenizer_meta"] = {"len": T.len, "spk_emb_ids": T.spk_emb_ids, "break_0_ids": T.break_0_ids, "eos_token": T.eos_token}
    enc = []
    g = torch.Generator().manual_seed(5)
    for i, texts in enumerate(CODE_TEXTS):
        for spk_emb, txt_smp, with_prompt in [(None, None, False), ("x", None, False), ("x", "sample text", True)]:
            deco = S.decorate_code_prompts(list(texts), "[speed_5]", txt_smp, spk_emb)
            prompt = torch.randint(0, 626, (4, 7 + i), generator=g, dtype=torch.int32) if with_prompt else None
            ids, attn, tmask = T.encode(deco, 4, prompt=prompt)
            enc.append({"texts": texts, "spk_emb": spk_emb, "txt_smp": txt_smp, "decorated": deco,
                        "prompt": None if prompt is None else prompt.tolist(), "ids": ids.tolist(), "attn": attn.tolist(),
                        "tmask": tmask.to(torch.int64).tolist(), "decoded": T.decode(ids[..., 0])})
    out["encode"] = enc
    # seeded random batches: 1..6 texts of 0..12 vocabulary words / CJK characters / control tokens / unknown words, random prompt string,
    # optional speaker embedding and audio-code prompt (1..40 frames)
    rs = np.random.RandomState(17)
    vocab_like = WORDS + ZH + CONTROL + PUNCT + ["zzzqq", "Ünknown", "12", " "]
    sweep = []
    for i in range(30):
        texts = ["".join(str(rs.choice(vocab_like)) + ("" if rs.rand() < 0.3 else " ") for _ in range(int(rs.randint(0, 13))))
                 for _ in range(int(rs.randint(1, 7)))]
        spk_emb = None if rs.rand() < 0.5 else "x"
        with_prompt = bool(rs.rand() < 0.4)
        txt_smp = "sample text" if (with_prompt and rs.rand() < 0.7) else None
        prm = ["", "[speed_5]", "[oral_2][laugh_0][break_6]"][int(rs.randint(3))]
        deco = S.decorate_code_prompts(list(texts), prm, txt_smp, spk_emb)
        prompt = torch.from_numpy(rs.randint(0, 626, (4, int(rs.randint(1, 41)))).astype(np.int32)) if with_prompt else None
        ids, attn, tmask = T.encode(deco, 4, prompt=prompt)
        sweep.append({"texts": texts, "spk_emb": spk_emb, "txt_smp": txt_smp, "prompt_str": prm, "decorated": deco,
                      "prompt": None if prompt is None else prompt.tolist(), "ids": ids.tolist(), "attn": attn.tolist(),
                      "tmask": tmask.to(torch.int64).tolist(), "decoded": T.decode(ids[..., 0])})
    out["encode_sweep"] = sweep
    deco = S.decorate_text_prompts(["what is your favorite food", "你好"], "[oral_2][laugh_0][break_6]")
    ids, attn, tmask = T.encode(deco, 4)
    out["refine"] = {"decorated": deco, "ids": ids.tolist(), "attn": attn.tolist(), "tmask": tmask.to(torch.int64).tolist()}

    # Speaker: statistics, sampling (torch global RNG), string round trips, apply
    sp = S(768, cfg.spk_stat)
    torch.manual_seed(77)
    s_str = sp.sample_random()
    vec = S._decode(s_str)
    pr = torch.randint(0, 626, (4, 33), generator=g)
    p_str = S.encode_prompt(pr)
    emb = torch.randn(3, 9, 768, generator=g)
    iid = torch.randint(0, 100, (3, 9, 4), generator=g)
    iid[0, 2, 0] = T.spk_emb_ids
    iid[2, 5, 0] = T.spk_emb_ids
    iid[1, 1, 1] = T.spk_emb_ids     # slot 1 does not count (speaker.py:46 looks at slot 0 only)
    ap = sp.apply(emb.clone(), s_str, iid, T.spk_emb_ids, torch.device("cpu"))
    out["speaker"] = {"std_head": sp.std[:4].tolist(), "mean_head": sp.mean[:4].tolist(), "seed": 77, "sample_str": s_str,
                      "sample_vec_head": vec[:6].astype(np.float32).tolist(), "prompt": pr.tolist(), "prompt_str": p_str,
                      "apply_ids": iid.tolist(), "apply_row": ap[0, 2, :8].tolist(), "apply_sum": float(ap.double().sum()),
                      "apply_untouched": float((ap[1] - emb[1]).abs().max())}
    with open(os.path.join(GOLD, "frontend.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=0)
    print("wrote frontend goldens:", len(out["norm"]), "norm cases,", len(enc), "encode cases")


if __name__ == "__main__":
    main()
