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
