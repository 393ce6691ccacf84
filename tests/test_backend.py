"""Back end of the path against the reference's own code run on the same inputs (tests/golden/backend.npz,
oracle/make_backend_goldens.py): `float_to_int16` (tools/audio/np.py:7-11) on the host and -- `-m gpu` -- on the device, `ChatStreamer`
(examples/cmd/stream.py:9-145) block for block, `Chat.infer(..., pcm16=True)`.  Bit-exact bars: integer / byte work."""
import hashlib
import os

import numpy as np
import pytest
import torch

from chattts_amd import audio
from chattts_amd.streamer import ChatStreamer
from oracle import cases

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "backend.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.mark.parametrize("product", ["f64", "f32"])
def test_host_float_to_int16_equals_the_reference_function(gold, product):
    """every sample of every case: the reference function's own output, for the arithmetic of its numba runtime (float64 product) and
    for plain NumPy's reading of the same line (float32 product); the two do differ (4608 samples of the `integers` case)"""
    for name, x in cases.pcm_inputs().items():
        got = audio.float_to_int16(x, product)
        assert got.dtype == np.int16 and got.shape == x.shape
        assert np.array_equal(got, gold[f"pcm.{name}.{product}"]), (name, int((got != gold[f"pcm.{name}.{product}"]).sum()))
    assert (gold["pcm.integers.f32"] != gold["pcm.integers.f64"]).sum() > 1000
    assert not audio.float_to_int16(np.zeros(100, np.float32)).any()          # the reference divides by zero here
    assert audio.pcm_scale(0.3) == 32767 and audio.pcm_scale(1.5) == 16383 and audio.pcm_scale(2.5) == 10922


def _digest(blocks, as_float):
    raw = b"".join(np.ascontiguousarray(b, dtype="<f4").tobytes() for b in blocks) if as_float else b"".join(blocks)
    return hashlib.sha256(raw).hexdigest()


@pytest.mark.parametrize("name", cases.STREAM_CASES)
def test_chat_streamer_equals_the_reference_class(gold, name):
    """the reference's ChatStreamer fed the same chunk sequence: the same PCM16 byte blocks in the same order with the same boundaries
    (three utterances ending at different times; one utterance in sub-block chunks; a silent first utterance, a silent chunk, a
    leftover under the block size; 16 seeded random sequences of 1..4 utterances, 1..12 chunks of 100..30000 samples), for both product
    arithmetics, and the float pieces of output_format=None"""
    if f"stream.{name}.f64.bytes.raises" in gold.files:
        # seeded random sequences on which the reference class itself crashes (`is_keep_next` unbound, stream.py:124, when the first
        # chunk is silent for every utterance): the port must not -- what it yields there is its own behaviour, not pinned
        for fmt in ("PCM16_byte", "PCM16", None):
            list(ChatStreamer().generate(iter(cases.stream_chunks(name)), output_format=fmt))
        return
    for product in ("f64", "f32"):
        blocks = list(ChatStreamer(product=product).generate(iter(cases.stream_chunks(name)), output_format="PCM16_byte"))
        key = f"stream.{name}.{product}.bytes"
        assert all(isinstance(b, bytes) for b in blocks)
        assert [len(b) for b in blocks] == gold[key + ".lens"].tolist()
        if key in gold.files:
            assert np.array_equal(np.frombuffer(b"".join(blocks), np.uint8), gold[key])
        assert _digest(blocks, False) == str(gold[key + ".sha256"])
    pieces = list(ChatStreamer().generate(iter(cases.stream_chunks(name)), output_format=None))
    assert [len(p) for p in pieces] == gold[f"stream.{name}.f64.float.lens"].tolist()
    assert _digest(pieces, True) == str(gold[f"stream.{name}.f64.float.sha256"])
    ints = list(ChatStreamer().generate(iter(cases.stream_chunks(name)), output_format="PCM16"))
    assert all(p.dtype == np.int16 for p in ints)
    assert _digest([p.astype("<i2").tobytes() for p in ints], False) == str(gold[f"stream.{name}.f64.bytes.sha256"])


def test_chat_streamer_edge_inputs():
    """no chunks, only silent chunks, an empty chunk: nothing is yielded (the reference raises on the first and the last)"""
    s = ChatStreamer()
    assert list(s.generate(iter([]), "PCM16_byte")) == []
    assert list(s.generate(iter([np.zeros((2, 500), np.float32)] * 3), "PCM16_byte")) == []
    assert list(s.generate(iter([np.zeros((2, 0), np.float32)]), "PCM16_byte")) == []


def test_wav_bytes_round_trip():
    import io
    import wave
    x = cases.pcm_inputs()["utt_quiet"]
    with wave.open(io.BytesIO(audio.pcm_to_wav_bytes(x)), "rb") as wf:
        assert (wf.getnchannels(), wf.getsampwidth(), wf.getframerate(), wf.getnframes()) == (1, 2, 24000, x.size)
        assert np.array_equal(np.frombuffer(wf.readframes(x.size), "<i2"), audio.float_to_int16(x))


# ---- device ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("product", ["f64", "f32"])
def test_device_float_to_int16_equals_the_reference_function(gold, weights, product):
    """ctts_float_to_int16 (csrc/codec.hip absmax_rows_k + pcm16_k) through CodecEngine.float_to_int16: np.array_equal with the reference
    function's output on every case -- one peak over a block (`per_row=False`, the function applied to a 2-D array) and a peak per row
    (one call per utterance); the keep mask equals |x| > 1e-5; a strided view (a window of a wider buffer) converts like its copy."""
    from chattts_amd import engine as E
    dev = torch.device("cuda:0")
    codec = E.CodecEngine(weights["decoder"], weights["vocos"], dev)
    for name, x in cases.pcm_inputs().items():
        x2 = x if x.ndim == 2 else x[None]
        pcm, keep = codec.float_to_int16(torch.from_numpy(x2).to(dev), per_row=False, product=product, keep_thr=1e-5)
        want = gold[f"pcm.{name}.{product}"].reshape(x2.shape)
        assert pcm.dtype == torch.int16 and np.array_equal(pcm.cpu().numpy(), want), (name, int((pcm.cpu().numpy() != want).sum()))
        bits = np.unpackbits(keep.cpu().numpy(), axis=1)[:, : x2.shape[1]].astype(bool)
        assert np.array_equal(bits, np.abs(x2) > np.float32(1e-5))
        rows, _ = codec.float_to_int16(torch.from_numpy(x2).to(dev), per_row=True, product=product)
        for b in range(x2.shape[0]):
            assert np.array_equal(rows[b].cpu().numpy(), audio.float_to_int16(x2[b], product)), (name, b)
    wide = torch.from_numpy(cases.pcm_inputs()["block4"]).to(dev)
    view = wide[:, 1001:5000]
    a, _ = codec.float_to_int16(view, per_row=True, product=product)
    b, _ = codec.float_to_int16(view.contiguous(), per_row=True, product=product)
    assert not view.is_contiguous() and torch.equal(a, b)
    z, _ = codec.float_to_int16(torch.zeros((2, 77), device=dev), per_row=True)
    assert not bool(z.any())


@pytest.mark.gpu
def test_chat_infer_pcm16_equals_float_to_int16_of_infer(weights):
    """`Chat.infer(..., pcm16=True)` (the conversion + the silence strip on the device, int16 + mask bits over PCIe) against the float path
    followed by the reference's host function: non-stream per utterance, stream per row of every chunk -- np.array_equal."""
    from chattts_amd.core import Chat
    dev = torch.device("cuda:0")
    gold_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    with open(os.path.join(gold_dir, "spk_stat.txt"), encoding="utf-8") as f:
        spk_stat = f.read()
    chat = Chat()
    assert chat.load(state_dicts=weights, device=dev, dtype="f32", tokenizer=os.path.join(gold_dir, "tokenizer"), spk_stat=spk_stat)
    torch.manual_seed(11)
    spk = chat.sample_random_speaker()
    texts = ["What is [uv_break]your favorite english food?[laugh][lbreak]", "hello world"]

    def params(n):
        return Chat.InferCodeParams(spk_emb=spk, max_new_token=n, min_new_token=n - 7, manual_seed=7, show_tqdm=False, stream_batch=8,
                                    stream_speed=3000, pass_first_n_batches=1)
    f32 = chat.infer(list(texts), skip_refine_text=True, split_text=False, params_infer_code=params(40))
    pcm = chat.infer(list(texts), skip_refine_text=True, split_text=False, params_infer_code=params(40), pcm16=True)
    assert len(pcm) == len(f32) == 2
    for a, b in zip(pcm, f32):
        assert a.dtype == np.int16 and a.shape == b.shape and np.array_equal(a, audio.float_to_int16(b))
    chunks_f = list(chat.infer(list(texts), stream=True, skip_refine_text=True, split_text=False, params_infer_code=params(40)))
    chunks_p = list(chat.infer(list(texts), stream=True, skip_refine_text=True, split_text=False, params_infer_code=params(40), pcm16=True))
    assert len(chunks_f) == len(chunks_p) >= 3
    for cf, cp in zip(chunks_f, chunks_p):
        assert cp.shape == cf.shape
        if cf.shape[1]:
            assert cp.dtype == np.int16 and all(np.array_equal(cp[b], audio.float_to_int16(cf[b])) for b in range(cf.shape[0]))
    # ... and the streamer over the float chunks is the reference's byte stream for ONE listener (its conversion is per collected block)
    blocks = list(ChatStreamer(base_block_size=2000).generate(iter(chunks_f), "PCM16_byte"))
    assert blocks and all(isinstance(b, bytes) and len(b) % 2 == 0 for b in blocks)


@pytest.mark.gpu
def test_chat_warm_hands_the_first_request_a_ready_session(weights):
    """`Chat.warm(B, T, params)` / `Chat.load(..., warm=...)`: the first real request of that geometry reuses the session the warm-up
    built (same buffers, graph already instantiated) and yields exactly what an un-warmed engine yields; the interrupt flag the warm-up
    used is restored."""
    from chattts_amd import synth
    from chattts_amd.core import Chat, InferCodeParams
    dev = torch.device("cuda:0")
    ids, mask, tmask = synth.make_prompts(4, 7, 12, seed=3)
    a = (torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(tmask))
    p = InferCodeParams(max_new_token=40, min_new_token=40, manual_seed=5, show_tqdm=False, stream_batch=8, stream_speed=3000, pass_first_n_batches=1)
    cold = Chat()
    assert cold.load(state_dicts=weights, device=dev, dtype="bf16")
    want = list(cold.infer_ids_stream(*a, p))
    warm = Chat()
    assert warm.load(state_dicts=weights, device=dev, dtype="bf16", warm=(4, ids.shape[1], p))
    sess = warm.gpt._session
    assert sess is not None and sess["graph"] and not warm.context.get()
    got = list(warm.infer_ids_stream(*a, p))
    assert warm.gpt._session is sess                         # not rebuilt: the request found its geometry
    assert len(got) == len(want) >= 3 and all(np.array_equal(x, y) for x, y in zip(got, want))
    assert warm.warm(4, ids.shape[1], p) < 5.0               # idempotent, and cheap the second time


# ---- the OpenAI-compatible endpoint (reference: examples/api/openai_api.py:149-294) ------------------------------------
class _FakeChat:
    """records the call, returns deterministic int16 audio: what `Chat.infer(..., pcm16=True)` hands the endpoint"""

    class InferCodeParams:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    def __init__(self):
        self.calls = []

    def has_loaded(self):
        return True

    def infer(self, text, stream=False, **kw):
        self.calls.append((list(text), stream, kw))
        n = 1000 + 10 * len(text[0])
        full = (np.arange(n) % 1000 - 500).astype(np.int16)
        if not stream:
            return [full]
        return (c[None, :] for c in (full[:300], full[300:300], full[300:]))     # an empty chunk in the middle, like a dropped yield


def test_openai_endpoint_shapes_the_reference_responses():
    """POST /v1/audio/speech (openai_api.py:149-288): parameter whitelist + validation, the reference's InferCodeParams, WAV file for a
    plain request, open-ended RIFF header + raw PCM16 chunks for a streamed one, "pcm" = the bare samples, mp3 / ogg refused with a message
    that names PyAV where it is absent; GET /health."""
    import importlib.util
    import io
    import wave
    from starlette.testclient import TestClient
    from chattts_amd import server
    chat = _FakeChat()
    app = server.create_app(chat, voices={"default": "SPK-D", "alloy": "SPK-A"})
    with TestClient(app) as c:
        assert c.get("/health").json()["model_loaded"] is True
        r = c.post("/v1/audio/speech", json={"model": "whatever", "input": "hello there", "voice": "alloy", "response_format": "wav", "bogus": 1})
        assert r.status_code == 200 and r.headers["content-type"] == "audio/wav" and "output.wav" in r.headers["content-disposition"]
        with wave.open(io.BytesIO(r.content), "rb") as wf:
            assert (wf.getnchannels(), wf.getsampwidth(), wf.getframerate()) == (1, 2, 24000)
            pcm = np.frombuffer(wf.readframes(wf.getnframes()), dtype="<i2")
        want = (np.arange(1000 + 10 * len("hello there")) % 1000 - 500).astype(np.int16)
        assert np.array_equal(pcm, want)
        text, stream, kw = chat.calls[-1]
        p = kw["params_infer_code"]
        assert text == ["hello there"] and stream is False and kw["skip_refine_text"] is True and kw["pcm16"] is True
        assert (p.prompt, p.top_P, p.top_K, p.temperature, p.repetition_penalty, p.manual_seed, p.spk_emb) == ("[speed_5]", 0.5, 10, 0.1, 1.1, 42, "SPK-A")
        assert (p.stream_batch, p.stream_speed, p.pass_first_n_batches, p.max_new_token) == (24, 12000, 2, 2048)
        r = c.post("/v1/audio/speech", json={"input": "hello there", "voice": "nobody", "response_format": "pcm"})
        assert r.status_code == 200 and np.array_equal(np.frombuffer(r.content, dtype="<i2"), want)
        assert chat.calls[-1][2]["params_infer_code"].spk_emb == "SPK-D"                   # unknown voice -> default
        r = c.post("/v1/audio/speech", json={"input": "hello there", "response_format": "wav", "stream": True})
        assert r.status_code == 200 and r.content[:44] == server.wav_stream_header() and r.content[4:8] == b"\xff\xff\xff\xff"
        assert np.array_equal(np.frombuffer(r.content[44:], dtype="<i2"), want) and chat.calls[-1][1] is True
        assert c.post("/v1/audio/speech", json={"input": "x", "response_format": "flac"}).status_code == 400
        if importlib.util.find_spec("av") is None:
            r = c.post("/v1/audio/speech", json={"input": "x", "response_format": "mp3"})
            assert r.status_code == 400 and "PyAV" in r.text
        assert c.post("/v1/audio/speech", json={"input": "x" * 2049, "response_format": "wav"}).status_code == 422
        assert c.post("/v1/audio/speech", json={"input": "x", "speed": 3.0, "response_format": "wav"}).status_code == 422
        assert c.post("/v1/audio/speech", json={"response_format": "wav"}).status_code == 422


@pytest.mark.gpu
def test_openai_endpoint_streams_what_chat_infer_streams(weights):
    """the endpoint on the real engine: a plain request's WAV samples == `float_to_int16` of `Chat.infer`'s waveform, a streamed request's
    body == the open-ended header + `float_to_int16` of every chunk `Chat.infer(stream=True)` yields (openai_api.py:259-288), byte for byte."""
    import io
    import wave
    from starlette.testclient import TestClient
    from chattts_amd import server
    from chattts_amd.core import Chat
    dev = torch.device("cuda:0")
    gold_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    with open(os.path.join(gold_dir, "spk_stat.txt"), encoding="utf-8") as f:
        spk_stat = f.read()
    chat = Chat()
    assert chat.load(state_dicts=weights, device=dev, dtype="bf16", tokenizer=os.path.join(gold_dir, "tokenizer"), spk_stat=spk_stat)
    torch.manual_seed(11)
    spk = chat.sample_random_speaker()
    text = "What is [uv_break]your favorite english food?[lbreak]"
    app = server.create_app(chat, voices={"default": spk})

    def direct(stream):
        p = Chat.InferCodeParams(prompt="[speed_5]", top_P=0.5, top_K=10, temperature=0.1, repetition_penalty=1.1, max_new_token=144, min_new_token=0,
                                 show_tqdm=False, ensure_non_empty=True, manual_seed=42, spk_emb=spk, stream_batch=24, stream_speed=12000,
                                 pass_first_n_batches=2)
        return chat.infer([text], stream=stream, skip_refine_text=True, params_infer_code=p)
    orig = chat.InferCodeParams
    # (random weights do not emit [Ebreak] on cue: cap the length the endpoint's fixed max_new_token = 2048 would otherwise run to)
    chat.InferCodeParams = lambda **kw: orig(**{**kw, "max_new_token": 144})
    try:
        want = direct(False)
        chunks = list(direct(True))
        with TestClient(app) as c:
            r = c.post("/v1/audio/speech", json={"input": text, "response_format": "wav"})
            assert r.status_code == 200
            with wave.open(io.BytesIO(r.content), "rb") as wf:
                got = np.frombuffer(wf.readframes(wf.getnframes()), dtype="<i2")
            assert np.array_equal(got, audio.float_to_int16(want[0]))
            r = c.post("/v1/audio/speech", json={"input": text, "response_format": "wav", "stream": True})
            body = server.wav_stream_header() + b"".join(audio.float_to_int16(ch).astype("<i2").tobytes() for ch in chunks if ch.size)
            assert r.status_code == 200 and r.content == body and len(chunks) >= 1 and sum(ch.size for ch in chunks) > 0
    finally:
        chat.InferCodeParams = orig
