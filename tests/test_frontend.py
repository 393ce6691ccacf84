"""Host front end (chattts_amd/frontend.py) against outputs of the reference's own Normalizer / Tokenizer / Speaker
(tests/golden/frontend.json, generated in the build container by oracle/make_frontend_goldens.py) and against the
reference's `Config.spk_stat` string as the known-answer test of the base16384 codec.  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from chattts_amd import frontend as F

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(GOLD, "frontend.json"), encoding="utf-8") as f:
        return json.load(f)


@pytest.fixture(scope="module")
def tok():
    return F.Tokenizer(os.path.join(GOLD, "tokenizer"))


@pytest.fixture(scope="module")
def spk_stat():
    with open(os.path.join(GOLD, "spk_stat.txt"), encoding="utf-8") as f:
        return f.read()


def test_b14_known_answer_spk_stat(spk_stat, gold):
    """config.py:132 decodes to exactly 2 x 768 float16 (std, mean) -- speaker.py:11-16"""
    raw = F.b14_decode(spk_stat)
    assert len(raw) == 2 * 768 * 2 and len(raw) % 7 == 6 and ord(spk_stat[-1]) == 0x3D06
    a = np.frombuffer(raw, dtype=np.float16).astype(np.float32)
    assert np.isfinite(a).all() and (a[:768] > 1.0).all() and a[:768].max() < 20 and np.abs(a[768:]).max() < 10
    assert F.b14_encode(raw) == spk_stat                      # the encoder reproduces the reference's own string
    sp = F.Speaker(768, spk_stat)
    assert sp.std[:4].tolist() == gold["speaker"]["std_head"] and sp.mean[:4].tolist() == gold["speaker"]["mean_head"]
    with pytest.raises(ValueError):
        F.Speaker(512, spk_stat)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 6, 7, 8, 13, 14, 15, 700, 3072])
def test_b14_round_trip_every_tail_length(n):
    rng = np.random.default_rng(n)
    data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    s = F.b14_encode(data)
    assert len(s) == (n // 7) * 4 + (8 * (n % 7) + 13) // 14 + (1 if n % 7 else 0)
    assert all(0x4E00 <= ord(c) < 0x4E00 + 16384 for c in s[: len(s) - (1 if n % 7 else 0)])
    assert F.b14_decode(s) == data


def test_b14_rejects_foreign_text():
    with pytest.raises(ValueError):
        F.b14_decode("hello")


def test_normalizer_matches_reference(gold):
    nz = F.Normalizer(os.path.join(GOLD, "homophones_small.json"))
    assert nz.register("en", lambda s: s.replace("100%", "one hundred percent"))
    assert not nz.register("en", lambda s: s)                   # already registered (norm.py:199-201)
    assert not nz.register("xx", lambda s: 5)                   # not str -> str
    for c in gold["norm"]:
        assert nz(c["in"], c["tn"], c["hp"], c["lang"]) == c["out"], c["in"]
    nz.unregister("en")
    assert "one hundred" not in nz("hello 100% done", True, True, "en")
    assert F.Normalizer()("你好", True, True) == "你好"        # no map file: homophone step is a no-op


def test_normalizer_random_sweep_matches_reference(gold):
    """frontend.json `norm_sweep`: 120 seeded random strings (ASCII / CJK letters incl. the homophone keys, digits, both punctuation
    families, control tags whole / unclosed / nested, whitespace runs, characters outside every accepted range) with random flags through
    the reference's Normalizer (oracle/make_frontend_goldens.norm_sweep_cases)"""
    nz = F.Normalizer(os.path.join(GOLD, "homophones_small.json"))
    nz.register("en", lambda s: s.replace("100%", "one hundred percent"))
    bad = [(c["in"], c["tn"], c["hp"], c["lang"], c["out"], nz(c["in"], c["tn"], c["hp"], c["lang"])) for c in gold["norm_sweep"]
           if nz(c["in"], c["tn"], c["hp"], c["lang"]) != c["out"]]
    assert not bad, bad[:5]


def test_split_and_combine_tags():
    t, g = F.split_tags("a[x]b[y]")
    assert (t, g) == (["a", "b"], ["[x]", "[y]"])
    assert F.combine_tags(t, g) == "a[x]b[y]"
    t, g = F.split_tags("[x]tail")
    assert (t, g) == (["", "tail"], ["[x]"]) and F.combine_tags(t, g) == "[x]tail"


def test_decorate_prompts_match_reference(gold):
    for e in gold["encode"]:
        assert F.Speaker.decorate_code_prompts(list(e["texts"]), "[speed_5]", e["txt_smp"], e["spk_emb"]) == e["decorated"]
    assert F.Speaker.decorate_text_prompts(["what is your favorite food", "你好"], "[oral_2][laugh_0][break_6]") == gold["refine"]["decorated"]
    assert F.Speaker.decorate_code_prompts(["x"], "", None, None) == ["[Stts][empty_spk]x[Ptts]"]


def test_tokenizer_encode_matches_reference(gold, tok):
    m = gold["tokenizer_meta"]
    assert (tok.len, tok.spk_emb_ids, tok.break_0_ids, tok.eos_token) == (m["len"], m["spk_emb_ids"], m["break_0_ids"], m["eos_token"])
    for e in gold["encode"]:
        prompt = None if e["prompt"] is None else torch.tensor(e["prompt"], dtype=torch.int32)
        ids, attn, tmask = tok.encode(e["decorated"], 4, prompt=prompt)
        assert ids.dtype == torch.int64 and attn.dtype == torch.int64 and tmask.dtype == torch.bool
        assert ids.tolist() == e["ids"] and attn.tolist() == e["attn"] and tmask.to(torch.int64).tolist() == e["tmask"]
        assert tok.decode(ids[..., 0]) == e["decoded"]
    r = gold["refine"]
    ids, attn, tmask = tok.encode(r["decorated"], 4)
    assert ids.tolist() == r["ids"] and attn.tolist() == r["attn"] and tmask.to(torch.int64).tolist() == r["tmask"]


def test_tokenizer_encode_random_sweep_matches_reference(gold, tok):
    """frontend.json `encode_sweep`: 30 seeded random batches (1..6 texts of vocabulary words / CJK characters / control tokens / unknown
    words, empty texts, random prompt strings, optional speaker embedding and audio-code prompt of 1..40 frames) decorated and encoded by
    the reference's Speaker / Tokenizer: decoration, ids, left padding, masks and the decoded strings"""
    for e in gold["encode_sweep"]:
        assert F.Speaker.decorate_code_prompts(list(e["texts"]), e["prompt_str"], e["txt_smp"], e["spk_emb"]) == e["decorated"]
        prompt = None if e["prompt"] is None else torch.tensor(e["prompt"], dtype=torch.int32)
        ids, attn, tmask = tok.encode(e["decorated"], 4, prompt=prompt)
        assert ids.tolist() == e["ids"] and attn.tolist() == e["attn"] and tmask.to(torch.int64).tolist() == e["tmask"], e["texts"]
        assert tok.decode(ids[..., 0]) == e["decoded"]


def test_reference_test_655_round_trip(tok):
    """/root/reference/tests/#655.py:56-92: decorate -> encode -> decode gives the pinned string"""
    text = ["What is [uv_break]your favorite english food?[laugh][lbreak]"]
    ids, _, _ = tok.encode(F.Speaker.decorate_code_prompts(text, "[speed_5]", None, "some speaker"), 4)
    assert tok.decode(ids[..., 0])[0] == "[Stts] [spk_emb] [speed_5] what is [uv_break] your favorite english food? [laugh] [lbreak] [Ptts]"


def test_speaker_strings_and_apply_match_reference(gold, spk_stat):
    g = gold["speaker"]
    sp = F.Speaker(768, spk_stat)
    torch.manual_seed(g["seed"])
    s = sp.sample_random()
    assert s == g["sample_str"]                                  # randn * std + mean -> f16 -> LZMA2 -> base16384
    vec = F.Speaker.decode_vector(s)
    assert vec.dtype == np.float16 and vec.shape == (768,) and vec[:6].astype(np.float32).tolist() == g["sample_vec_head"]
    pr = torch.tensor(g["prompt"])
    assert F.Speaker.encode_prompt(pr) == g["prompt_str"]
    back = F.Speaker.decode_prompt(g["prompt_str"])
    assert back.dtype == torch.int32 and torch.equal(back.long(), pr)
    gen = torch.Generator().manual_seed(5)                       # replay make_frontend_goldens' draws up to `emb`
    for i in range(3):
        torch.randint(0, 626, (4, 7 + i), generator=gen, dtype=torch.int32)
    torch.randint(0, 626, (4, 33), generator=gen)
    emb = torch.randn(3, 9, 768, generator=gen)
    iid = torch.tensor(g["apply_ids"])
    spk_id = int(iid[0, 2, 0])
    out = sp.apply(emb.clone(), s, iid, spk_id, torch.device("cpu"))
    assert out[0, 2, :8].tolist() == g["apply_row"] and float(out.double().sum()) == g["apply_sum"]
    assert torch.equal(out[1], emb[1]) and g["apply_untouched"] == 0.0   # slot 1 carrying the id does not count
    assert torch.equal(out[0, 2], out[2, 5]) and abs(float(out[0, 2].norm()) - 1.0) < 1e-3
    keep = F.apply_speaker(emb, torch.from_numpy(vec), iid, spk_id, inplace=False)
    assert torch.equal(keep, out) and not torch.equal(emb, out)


def test_chat_text_level_needs_tokenizer():
    from chattts_amd.core import Chat
    c = Chat()
    with pytest.raises(RuntimeError):
        c._need_tokenizer()
    with pytest.raises(RuntimeError):
        c.sample_random_speaker()
    assert Chat.InferCodeParams().prompt == "[speed_5]" and Chat.RefineTextParams().temperature == 0.7
    assert c.infer([], split_text=False) == []


def test_audio_back_end():
    """float_to_int16 / WAV container (tools/audio/np.py:7-12, pcm.py:8-33): values restated here as literals computed
    by the reference's formula: am = 32767*32768 // (ceil(max|x|)*32768), int16(x*am) truncating toward zero"""
    import io
    import wave
    from chattts_amd import audio as A
    x = np.array([0.0, 0.5, -0.5, 0.999, -1.0, 0.25001], np.float32)
    assert A.float_to_int16(x).tolist() == [0, 16383, -16383, 32734, -32767, 8192]
    y = np.array([1.5, -0.75], np.float32)              # peak above 1 -> ceil = 2 -> scale 16383
    assert A.float_to_int16(y).tolist() == [24574, -12287]
    assert A.float_to_int16(np.zeros(4, np.float32)).tolist() == [0, 0, 0, 0]
    b = A.pcm_to_wav_bytes(x)
    with wave.open(io.BytesIO(b), "rb") as wf:
        assert (wf.getnchannels(), wf.getsampwidth(), wf.getframerate(), wf.getnframes()) == (1, 2, 24000, 6)
        assert np.frombuffer(wf.readframes(6), dtype="<i2").tolist() == A.float_to_int16(x).tolist()


def test_chat_load_rejects_remote_sources_before_touching_the_gpu():
    """the reference's `load` returns False on a failed download (core.py:149-151); this engine has no network path"""
    from chattts_amd.core import Chat
    c = Chat()
    assert c.load(source="huggingface") is False and not c.has_loaded()
    # sentence splitting of `infer` (core.py:225-238) happens before anything is loaded: an empty text yields []
    assert c.infer("", split_text=True) == []


def test_chat_load_signature_matches_reference():
    """the reference's positional parameters and defaults of Chat.load (core.py:137-148), extras keyword-only"""
    import inspect
    from chattts_amd.core import Chat
    ps = list(inspect.signature(Chat.load).parameters.values())[1:]
    pos = [(p.name, p.default) for p in ps if p.kind == p.POSITIONAL_OR_KEYWORD]
    assert pos == [("source", "local"), ("force_redownload", False), ("compile", False), ("custom_path", None), ("device", None),
                   ("coef", None), ("use_flash_attn", False), ("use_vllm", False), ("experimental", False)]
    assert all(p.kind == p.KEYWORD_ONLY for p in ps[len(pos):])
