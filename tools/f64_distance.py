"""Distance of the two parity arithmetics to EXACT arithmetic: the final-norm hidden state of every generated position, `"f32"` (f32 MFMA)
and `"f32x3"` (split-fp16 projections), against the same model evaluated in float64 (HF LlamaModel.double() on the CPU, one teacher-forced
forward over prompt + generated tokens).  If the two engines sit at a similar distance from the float64 truth, their difference from
EACH OTHER is float32 rounding noise of the same class, not a systematic loss of the split.  GPU box:  python tools/f64_distance.py"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chattts_amd import engine as E, synth  # noqa: E402
from chattts_amd import weights as W  # noqa: E402
from oracle import torch_port  # noqa: E402  (measurement tool, not the product path)


def main():
    dev = torch.device("cuda:0")
    sds = W.synthetic_all()
    B, steps = 16, 96
    ids, mask, tmask = synth.make_prompts(B, 10, 30, seed=5)
    T = ids.shape[1]
    ids_t, mask_t = torch.from_numpy(ids), torch.from_numpy(mask)
    warpers, procs = E.gen_logits(625, 0.7, 20, 1.05)
    temp = torch.tensor([0.3] * 4)
    res, teacher = {}, None
    for dt in ("f32", "f32x3"):
        eng = E.GptEngine(sds["gpt"], sds["embed"], dev, dtype=dt, certify=False)
        emb = eng.embed_prompt(ids_t, torch.from_numpy(tmask))
        kw = {} if teacher is None else dict(teacher_ids=teacher)
        out = list(eng.generate(emb, ids_t, temp, 625, mask_t, steps, steps, (*procs, *warpers), return_hidden=True, manual_seed=3, **kw))[-1]
        if teacher is None:
            teacher = torch.stack([t.cpu() for t in out.ids], 0)          # [B, steps, 4]: the f32 engine's own stream
            emb_prompt = emb.double().cpu()
        res[dt] = torch.stack([h.cpu() for h in out.hiddens], 0).double().numpy()     # [B, steps, 768]
        del eng
        torch.cuda.empty_cache()
    # float64 truth: one forward over [prompt | generated tokens]
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    llama = torch_port.build_llama(sds["gpt"]).double()
    emb_code = [sds["embed"][f"emb_code.{k}.weight"].double() for k in range(4)]
    gen_emb = sum(emb_code[k][teacher[:, :-1, k]] for k in range(4))             # inputs of steps 1 .. steps-1
    x = torch.cat([emb_prompt, gen_emb], 1)
    am = torch.cat([mask_t.bool(), torch.ones((B, steps - 1), dtype=torch.bool)], 1)
    pos = (am.long().cumsum(-1) - 1).masked_fill(am == 0, 1)
    with torch.inference_mode():
        h = llama(inputs_embeds=x, attention_mask=am, position_ids=pos, use_cache=False).last_hidden_state.numpy()
    truth = h[:, T - 1:, :]                                                       # position T-1+i produced token i
    out = {}
    for name, a, b in (("f32_vs_float64", res["f32"], truth), ("f32x3_vs_float64", res["f32x3"], truth), ("f32x3_vs_f32", res["f32x3"], res["f32"])):
        d = a - b
        rel = np.abs(d).max(-1) / np.abs(b).max(-1)
        out[name] = {"max_rel_hidden_err": float(rel.max()), "rms_rel_hidden_err": float(np.sqrt((d ** 2).mean() / (b ** 2).mean())),
                     "first_quarter_rms": float(np.sqrt((d[:, : steps // 4] ** 2).mean() / (b[:, : steps // 4] ** 2).mean())),
                     "last_quarter_rms": float(np.sqrt((d[:, -steps // 4:] ** 2).mean() / (b[:, -steps // 4:] ** 2).mean()))}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
