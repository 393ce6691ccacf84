"""What `Chat._infer_code` / `Chat._refine_text` hand to the engine, against what the reference's hand to `GPT.generate` (CPU only).

tests/golden/seam_args.json (oracle/make_seam_goldens.py): the reference's two methods (core.py:542-662, :665-751), imported from
/root/reference and run unmodified on its own Tokenizer (synthetic vocabulary), Speaker and Embed (synthetic weights), with a recorder in
place of `GPT.generate`.  Here `chattts_amd.core.Chat` runs the same scenarios with a recorder in place of the engine (its `embed_prompt` is
the numpy oracle's, itself pinned bit for bit on the reference's Embed): all 17 arguments of the generate call must be the same -- the
decorated, tokenised, left-padded ids and masks, the prompt embedding WITH the speaker vector substituted (sha256 of the float32 bytes),
per-codebook temperatures, eos, the processor chain and its numbers, every flag."""
import json
import os

import numpy as np
import pytest
import torch

from chattts_amd import engine as E, frontend as F
from chattts_amd.core import Chat
from oracle import generate_np, host_fakes as HF

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(GOLD, "seam_args.json"), encoding="utf-8") as f:
        seam = json.load(f)
    with open(os.path.join(GOLD, "frontend.json"), encoding="utf-8") as f:
        front = json.load(f)
    return seam, front


class RecorderEngine:
    """stands in for GptEngine: `embed_prompt` on the CPU (numpy oracle), `generate` records its arguments"""

    def __init__(self, esd):
        self.esd, self.calls = esd, []

    def embed_prompt(self, input_ids, text_mask):
        return torch.from_numpy(generate_np.embed_prompt(self.esd, input_ids.numpy(), text_mask.numpy()))

    def generate(self, *args, **kwargs):
        self.calls.append(HF.describe_generate_call(E.GptEngine.generate, args, kwargs))
        B = args[1].shape[0]
        yield HF.FakeOutputs([torch.arange(5 + b, dtype=torch.int64) for b in range(B)], [], [])


@pytest.fixture()
def chat(weights):
    c = Chat()
    c.tokenizer = F.Tokenizer(os.path.join(GOLD, "tokenizer"))
    with open(os.path.join(GOLD, "spk_stat.txt"), encoding="utf-8") as f:
        c.speaker = F.Speaker(768, f.read())
    c.gpt = RecorderEngine({k: v.numpy() for k, v in weights["embed"].items()})
    c.has_loaded = lambda use_decoder=True: True
    c.device = torch.device("cpu")
    return c


def fill(params, fields, front):
    for k, v in fields.items():
        v = {"$SPK": front["speaker"]["sample_str"], "$SMP": front["speaker"]["prompt_str"]}.get(v, v) if isinstance(v, str) else v
        setattr(params, k, v)
    return params


def same(mine: dict, ref: dict):
    assert list(mine) == list(ref) == list(HF.GENERATE_ARGS)
    for k in HF.GENERATE_ARGS:
        a, b = mine[k], ref[k]
        if k == "temperature":      # the reference builds float32 from Python floats; so must we
            assert a["dtype"] == b["dtype"] == "float32" and a["shape"] == b["shape"]
            assert np.array_equal(np.array(a["values"], np.float32), np.array(b["values"], np.float32))
        elif k in ("inputs_ids", "attention_mask"):
            assert a["shape"] == b["shape"] and np.array_equal(np.array(a["values"]).astype(np.int64), np.array(b["values"]).astype(np.int64)), k
        else:
            assert a == b, (k, a, b)


@pytest.mark.parametrize("name", list(HF.SEAM_CODE_SCENARIOS))
def test_infer_code_hands_the_engine_what_the_reference_hands_gpt_generate(gold, chat, name):
    seam, front = gold
    sc = HF.SEAM_CODE_SCENARIOS[name]
    res = chat._infer_code(sc["text"], sc["stream"], chat.device, sc["return_hidden"], fill(Chat.InferCodeParams(), sc["params"], front))
    assert len(list(res)) == seam["code"][name]["yields"]
    want = seam["code"][name]["calls"]
    assert len(chat.gpt.calls) == len(want) == 1
    same(json.loads(json.dumps(chat.gpt.calls[0])), want[0])


@pytest.mark.parametrize("name", list(HF.SEAM_REFINE_SCENARIOS))
def test_refine_text_hands_the_engine_what_the_reference_hands_gpt_generate(gold, chat, name):
    seam, front = gold
    sc = HF.SEAM_REFINE_SCENARIOS[name]
    res = chat._refine_text(sc["text"], chat.device, fill(Chat.RefineTextParams(), sc["params"], front))
    assert [int(r.shape[0]) for r in res.ids] == seam["refine"][name]["ids_lens"]
    want = seam["refine"][name]["calls"]
    assert len(chat.gpt.calls) == len(want) == 1
    same(json.loads(json.dumps(chat.gpt.calls[0])), want[0])
