"""Generates tests/golden/seam_args.json: every argument the REFERENCE's `Chat._infer_code` (core.py:542-662) and `Chat._refine_text`
(core.py:665-751) hand to `GPT.generate`, for the scenarios of oracle/host_fakes.py.  Both methods are imported from /root/reference and
run unmodified on the reference's own Tokenizer (synthetic vocabulary, tests/golden/tokenizer), Speaker and Embed (synthetic weights);
`GPT.generate` is a recorder.   Build container only:   python -m oracle.make_seam_goldens
"""
from __future__ import annotations

import json
import logging
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from chattts_amd import weights as W  # noqa: E402
from oracle import host_fakes as HF, make_frontend_goldens as MF, make_host_goldens as MH, ref_harness  # noqa: E402

GOLD = MF.GOLD


class RecorderGPT:
    is_vllm = False
    device_gpt = torch.device("cpu")

    def __init__(self, real_generate):
        self.calls = []
        self.real = real_generate

    def generate(self, *args, **kwargs):
        self.calls.append(HF.describe_generate_call(self.real, args, kwargs))
        ids = kwargs.get("inputs_ids", args[1] if len(args) > 1 else None)
        B = ids.shape[0]
        yield HF.FakeOutputs([torch.arange(5 + b, dtype=torch.int64) for b in range(B)], [], [])


def fill(params, fields: dict, gold: dict):
    for k, v in fields.items():
        if v == "$SPK":
            v = gold["speaker"]["sample_str"]
        elif v == "$SMP":
            v = gold["speaker"]["prompt_str"]
        setattr(params, k, v)
    return params


def main():
    assert ref_harness.available()
    with open(os.path.join(GOLD, "frontend.json"), encoding="utf-8") as f:
        gold = json.load(f)
    tokm, spkm, normm, cfgm = MF.ref_frontend()
    Chat = MH.ref_chat_class()
    m = ref_harness.ref_modules()
    cfg = cfgm.Config()
    sds = W.synthetic_all()
    embed = m["embed"].Embed(cfg.embed.hidden_size, cfg.embed.num_audio_tokens, cfg.embed.num_text_tokens, cfg.embed.num_vq)
    embed.load_state_dict(sds["embed"])
    embed.eval()
    T = tokm.Tokenizer(os.path.join(GOLD, "tokenizer"))
    if not hasattr(T._tokenizer, "encode_plus"):     # transformers 5.x dropped it (see make_frontend_goldens.py)
        type(T._tokenizer).encode_plus = lambda self, *a, **k: self(*a, **k)

    def make():
        chat = Chat.__new__(Chat)
        chat.logger = logging.getLogger("ref_seam")
        chat.config = cfg
        chat.device = chat.device_gpt = torch.device("cpu")
        chat.context = MH._Ctx()
        chat.tokenizer, chat.speaker, chat.embed = T, spkm.Speaker(cfg.gpt.hidden_size, cfg.spk_stat), embed
        chat.gpt = RecorderGPT(m["gpt"].GPT.generate)
        return chat

    out = {"code": {}, "refine": {}}
    with torch.no_grad():
        for name, sc in HF.SEAM_CODE_SCENARIOS.items():
            chat = make()
            res = chat._infer_code(sc["text"], sc["stream"], chat.device, sc["return_hidden"], fill(Chat.InferCodeParams(), sc["params"], gold))
            n = len(list(res))
            out["code"][name] = {"calls": chat.gpt.calls, "yields": n}
            print(name, chat.gpt.calls[0]["emb"]["shape"], chat.gpt.calls[0]["logits_processors"], "yields", n)
        for name, sc in HF.SEAM_REFINE_SCENARIOS.items():
            chat = make()
            res = chat._refine_text(sc["text"], chat.device, fill(Chat.RefineTextParams(), sc["params"], gold))
            out["refine"][name] = {"calls": chat.gpt.calls, "ids_lens": [int(r.shape[0]) for r in res.ids]}
            print(name, chat.gpt.calls[0]["emb"]["shape"], chat.gpt.calls[0]["logits_processors"])
    with open(os.path.join(GOLD, "seam_args.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=1)


if __name__ == "__main__":
    main()
