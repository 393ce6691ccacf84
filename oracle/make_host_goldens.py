"""Generates tests/golden/host_flow.{json,npz}: the REFERENCE's own `Chat.infer` / `Chat._infer` (core.py:208-270, :386-503), imported
from /root/reference and run unmodified, with the two engine seams replaced by the deterministic CPU stand-ins of oracle/host_fakes.py.

Run in the build container only:   python -m oracle.make_host_goldens

Import recipe: `ChatTTS.core` imports `vocos` (not installed, not vendored) and the whole `ChatTTS.model` package (pybase16384, torchaudio,
vector_quantize_pytorch, numba missing) at module level -- none of which the control flow under test touches.  The package objects are
stubbed as in oracle/ref_harness.py, the model submodules that import cleanly are imported individually and their classes put on the stub,
`vocos` is an empty module.  `Chat.__init__` is NOT run (it builds the normalizer from files and needs nothing we test): the instance is made
with `__new__` and given exactly the attributes `infer` / `_infer` read.
"""
from __future__ import annotations

import importlib
import json
import logging
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import host_fakes as HF, ref_harness  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def ref_chat_class():
    ref_harness.ref_modules()                     # ChatTTS / ChatTTS.model stubs + gpt, embed, processors, dvae imported
    nb = types.ModuleType("numba")
    nb.jit = lambda *a, **k: (lambda f: f)
    sys.modules.setdefault("numba", nb)
    if "pybase16384" in sys.modules and not hasattr(sys.modules["pybase16384"], "encode_to_string"):
        sys.modules["pybase16384"].encode_to_string = None
        sys.modules["pybase16384"].decode_from_string = None
    for n in ("vocos", "vocos.pretrained"):
        if n not in sys.modules:
            sys.modules[n] = types.ModuleType(n)
    sys.modules["vocos"].Vocos = None
    sys.modules["vocos.pretrained"].instantiate_class = None
    model = sys.modules["ChatTTS.model"]
    tok = importlib.import_module("ChatTTS.model.tokenizer")
    spk = importlib.import_module("ChatTTS.model.speaker")
    m = ref_harness.ref_modules()
    model.DVAE, model.Embed, model.GPT, model.gen_logits = m["dvae"].DVAE, m["embed"].Embed, m["gpt"].GPT, m["processors"].gen_logits
    model.Tokenizer, model.Speaker = tok.Tokenizer, spk.Speaker
    core = importlib.import_module("ChatTTS.core")
    return core.Chat


class _Ctx:
    def __init__(self):
        self.v = False

    def set(self, v):
        self.v = v

    def get(self):
        return self.v


def make_ref_chat(Chat, seams: HF.Seams):
    chat = Chat.__new__(Chat)
    chat.logger = logging.getLogger("ref_host_flow")
    chat.device = "cpu"
    chat.context = _Ctx()
    chat.normalizer = HF.FakeNormalizer(seams.log)
    chat.tokenizer = HF.FakeTokenizer()
    chat.has_loaded = lambda use_decoder=False: True
    chat._infer_code = seams.infer_code
    chat._decode_to_wavs = seams.decode_to_wavs
    chat._refine_text = seams.refine_text
    chat.sample_audio_speaker = seams.sample_audio_speaker
    return chat


def main():
    assert ref_harness.available(), "/root/reference is required"
    Chat = ref_chat_class()
    meta, arrays = {}, {}
    for name, sc in HF.SCENARIOS.items():
        seams = HF.Seams()
        chat = make_ref_chat(Chat, seams)
        desc, arrs = HF.call_infer(chat, sc, Chat.InferCodeParams(), Chat.RefineTextParams())
        desc["log"] = seams.log
        meta[name] = desc
        for i, a in enumerate(arrs):
            arrays[f"{name}.{i}"] = a
        print(name, desc["kind"], desc["shapes"][:6], "calls", len(seams.log))
    with open(os.path.join(OUT, "host_flow.json"), "w") as f:
        json.dump(meta, f, indent=1, ensure_ascii=False)
    np.savez_compressed(os.path.join(OUT, "host_flow.npz"), **arrays)


if __name__ == "__main__":
    main()
