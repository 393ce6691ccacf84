"""The HOST control flow of `Chat.infer` / `Chat._infer` against the reference's own (CPU only, no engine).

tests/golden/host_flow.{json,npz} were produced by oracle/make_host_goldens.py: the REFERENCE's `Chat.infer` + `Chat._infer` (core.py:208-270,
:386-503), imported from /root/reference and run unmodified, with the two engine seams (`_infer_code`, `_decode_to_wavs`) and the
refine-text / speaker-prompt helpers replaced by the deterministic stand-ins of oracle/host_fakes.py.  Here the same stand-ins are plugged into
`chattts_amd.core.Chat`; every scenario must return / yield the same bytes in the same shapes AND reach the seams with the same calls in the
same order (which text batch, which flags, which speaker prompt -- incl. the refer-sentence prompt of `split_text` and the streaming `length`
that the reference carries across split batches)."""
import json
import os

import numpy as np
import pytest

from chattts_amd.core import Chat
from oracle import host_fakes as HF

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def flow():
    with open(os.path.join(GOLDEN, "host_flow.json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(GOLDEN, "host_flow.npz"))


def make_chat(seams):
    chat = Chat()
    chat.device = "cpu"
    chat.normalizer = HF.FakeNormalizer(seams.log)
    chat.tokenizer = HF.FakeTokenizer()
    chat.has_loaded = lambda use_decoder=True: True
    chat._infer_code = seams.infer_code
    chat.decode_to_wavs = seams.decode_to_wavs          # the reference's `_decode_to_wavs`
    chat._refine_text = seams.refine_text
    chat.sample_audio_speaker = seams.sample_audio_speaker
    chat.incremental_stream = False     # the reference's full re-decode per yield (the windowed decode is GPU-tested to equal its slices)
    return chat


def _normal(log):
    """seam calls that the two implementations may legitimately order differently are not compared: `destroy` is bookkeeping"""
    return [e for e in log if e[0] != "destroy"]


@pytest.mark.parametrize("name", list(HF.SCENARIOS))
def test_infer_control_flow_equals_the_reference(flow, name):
    meta, arrs = flow
    want = meta[name]
    seams = HF.Seams()
    chat = make_chat(seams)
    desc, got = HF.call_infer(chat, HF.SCENARIOS[name], Chat.InferCodeParams(), Chat.RefineTextParams())
    assert desc["kind"] == want["kind"]
    if "value" in want:
        assert desc["value"] == want["value"]
    assert desc["shapes"] == want["shapes"], (desc["shapes"], want["shapes"])
    assert desc["dtypes"] == want["dtypes"]
    assert desc["spk_smp_after"] == want["spk_smp_after"] and desc["txt_smp_after"] == want["txt_smp_after"]
    for i, a in enumerate(got):
        assert np.array_equal(a, arrs[f"{name}.{i}"]), (name, i)
    mine, ref = json.loads(json.dumps(_normal(seams.log))), _normal(want["log"])
    if want["kind"] == "stream":
        # the reference decodes the first `pass_first_n_batches` yields and drops the audio (core.py:482-490); this Chat does not decode
        # them at all -- its decode calls must be a subsequence of the reference's, everything else (which batches reached the engine,
        # with which flags and speaker prompt) the same list
        # (and the tail chunk decodes the final state again where the reference reuses its last `wavs`: consecutive repeats collapse)
        dec = [e for e in mine if e[0] == "decode"]
        dec = [e for i, e in enumerate(dec) if i == 0 or e != dec[i - 1]]
        it = iter(e for e in ref if e[0] == "decode")
        assert all(any(e == r for r in it) for e in dec), (dec, [e for e in ref if e[0] == "decode"])
        mine, ref = [e for e in mine if e[0] != "decode"], [e for e in ref if e[0] != "decode"]
    assert mine == ref
    # every GenerationOutputs the engine handed out is released exactly once, as in the reference
    assert sum(e[0] == "destroy" for e in seams.log) == sum(e[0] == "destroy" for e in want["log"])
