"""Generates tests/golden/*.npz by running the reference itself (oracle/ref_harness.py) on CPU.

Run in the build container only:   python -m oracle.make_goldens
Inputs are NOT stored: every test re-creates them from the seeded recipes in `cases.py`
(chattts_amd.synth / numpy RandomState) and the weight recipe in chattts_amd.weights, whose sha256
fingerprints ARE stored, so a drifting recipe fails loudly instead of silently mis-comparing.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from chattts_amd import weights as W  # noqa: E402
from oracle import cases, ref_harness  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def golden_sampling():
    m = ref_harness.ref_modules()
    out = {}
    for name, c in cases.SAMPLING_CASES.items():
        logits, hist, temp = cases.sampling_inputs(c)
        rows, V = logits.shape
        warpers, procs = m["processors"].gen_logits(num_code=V - 1, top_P=c["top_P"], top_K=c["top_K"],
                                                    repetition_penalty=c["rep"])
        x = torch.from_numpy(logits).clone()
        x /= torch.from_numpy(temp).view(-1, 1)  # gpt.py:487
        for p in (*procs, *warpers):  # core.py:649 order
            x = p(torch.from_numpy(hist), x)
        if c["mask_eos"]:
            x[:, V - 1] = -torch.inf  # gpt.py:494-495
        scores = torch.nn.functional.softmax(x, dim=-1)
        g = torch.Generator()
        idx = torch.multinomial(scores, 1, generator=g.manual_seed(c["seed"])).view(-1)  # gpt.py:501-508
        out[name + ".idx"] = idx.numpy()
        out[name + ".kept"] = np.packbits(torch.isfinite(x).numpy(), axis=1)
    np.savez_compressed(os.path.join(OUT, "sampling.npz"), **out)
    print("sampling.npz", {k: v.shape for k, v in out.items()})


def golden_generate(sds, which=None, fname="generate.npz"):
    embed, gpt = ref_harness.build_gpt(sds)
    out = {}
    for name, c in (which or cases.GEN_CASES).items():
        t0 = time.time()
        ids, mask, tmask = cases.gen_inputs(c)
        if c["manual_seed"] is None:
            torch.manual_seed(c["global_seed"])
        res, emb, cap = ref_harness.run_generate(
            embed, gpt, ids, mask, tmask, temperature=c["temperature"], top_P=c["top_P"], top_K=c["top_K"],
            repetition_penalty=c["rep"], max_new_token=c["max_new"], min_new_token=c["min_new"],
            manual_seed=c["manual_seed"], capture_logits=True)
        B = ids.shape[0]
        out[name + ".emb_sum"] = np.array([np.abs(emb).sum(dtype=np.float64)])
        out[name + ".emb_row0"] = emb[0]
        out[name + ".lens"] = np.array([r.shape[0] for r in res.ids], dtype=np.int64)
        out[name + ".ids"] = np.concatenate([r.numpy() for r in res.ids], 0)
        for b in c["keep_hidden_rows"]:
            out[name + f".hid{b}"] = res.hiddens[b].numpy()
        for s in c["keep_logit_steps"]:
            if s < len(cap):
                out[name + f".tlogits{s}"] = cap[s]  # logits / temperature at step s, [B*4, 626]
        print(name, "steps", len(cap), "lens", out[name + ".lens"].tolist(), f"{time.time() - t0:.1f}s")
    np.savez_compressed(os.path.join(OUT, fname), **out)


def golden_stream(sds):
    """every yield of the reference's GPT.generate(stream=True): per yield the length of every row's ids / hidden rows, and the ids
    of the yield themselves as one concatenated array (rows in order)"""
    embed, gpt = ref_harness.build_gpt(sds)
    out = {}
    for name, (base, sb) in cases.GEN_STREAM_CASES.items():
        c = cases.GEN_CASES[base]
        ids, mask, tmask = cases.gen_inputs(c)
        if c["manual_seed"] is None:
            torch.manual_seed(c["global_seed"])
        ys = []
        ref_harness.run_generate(
            embed, gpt, ids, mask, tmask, temperature=c["temperature"], top_P=c["top_P"], top_K=c["top_K"],
            repetition_penalty=c["rep"], max_new_token=c["max_new"], min_new_token=c["min_new"],
            manual_seed=c["manual_seed"], stream=True, stream_batch=sb, yields=ys)
        out[name + ".lens"] = np.array([[r.shape[0] for r in y[0]] for y in ys], np.int64)         # [yields, B]
        out[name + ".hid_lens"] = np.array([y[1] for y in ys], np.int64)
        out[name + ".ids"] = np.concatenate([r for y in ys for r in y[0]], 0)
        print(name, "yields", len(ys), out[name + ".lens"].tolist())
    np.savez_compressed(os.path.join(OUT, "generate_stream.npz"), **out)


def golden_regen(sds):
    """the step-0 EOS path: how many times the reference ran step 0 (`attempts`), what it finally returned (nothing, when seeded),
    and where it left torch's global generator (three uniforms drawn right after)"""
    embed, gpt = ref_harness.build_gpt(sds)
    out = {}
    for name, c in cases.REGEN_CASES.items():
        ids, mask, tmask = cases.gen_inputs(c)
        torch.manual_seed(c.get("global_seed", 999))
        res, emb, cap = ref_harness.run_generate(
            embed, gpt, ids, mask, tmask, temperature=c["temperature"], top_P=c["top_P"], top_K=c["top_K"],
            repetition_penalty=c["rep"], max_new_token=c["max_new"], min_new_token=c["min_new"],
            manual_seed=c["manual_seed"], capture_logits=True)
        out[name + ".rand_after"] = torch.rand(3).numpy()
        out[name + ".yielded"] = np.array([res is not None])
        if res is not None:
            out[name + ".lens"] = np.array([r.shape[0] for r in res.ids], dtype=np.int64)
            out[name + ".ids"] = np.concatenate([r.numpy() for r in res.ids], 0)
            out[name + ".attempts"] = np.array([len(cap) - int(out[name + ".lens"].max()) + 1])
        else:
            out[name + ".attempts"] = np.array([len(cap)])
        print(name, "yielded", res is not None, "spy calls", len(cap), "attempts", int(out[name + ".attempts"][0]),
              out.get(name + ".lens", np.array([])).tolist())
    np.savez_compressed(os.path.join(OUT, "generate_regen.npz"), **out)


def golden_sweep(sds):
    """cases.sweep_cases() through the reference: ids only (+ whether anything was yielded, + the global generator's position after)"""
    embed, gpt = ref_harness.build_gpt(sds)
    out = {}
    t0 = time.time()
    for name, c in {**cases.sweep_cases(), **cases.text_sweep_cases()}.items():
        text = name.startswith("t")
        ids, mask, tmask = cases.gen_inputs(c)
        torch.manual_seed(c["global_seed"])
        res, emb, cap = ref_harness.run_generate(
            embed, gpt, ids, mask, tmask, temperature=c["temperature"], top_P=c["top_P"], top_K=c["top_K"],
            repetition_penalty=c["rep"], max_new_token=c["max_new"], min_new_token=c["min_new"], manual_seed=c["manual_seed"],
            **(dict(infer_text=True, eos_token=cases.TEXT_EOS) if text else {}))
        out[name + ".rand_after"] = torch.rand(2).numpy()
        out[name + ".yielded"] = np.array([res is not None])
        if res is not None:
            out[name + ".lens"] = np.array([r.shape[0] for r in res.ids], dtype=np.int16)
            out[name + ".ids"] = np.concatenate([r.numpy().reshape(-1) if text else r.numpy() for r in res.ids], 0).astype(np.int16)
        print(name, {k: c[k] for k in ("B", "t_min", "t_max", "top_P", "top_K", "rep", "max_new", "min_new", "manual_seed")},
              "yielded", res is not None, "" if res is None else out[name + ".lens"].tolist()[:8])
    print(f"sweep: {time.time() - t0:.1f}s")
    np.savez_compressed(os.path.join(OUT, "generate_sweep.npz"), **out)


def golden_text(sds):
    embed, gpt = ref_harness.build_gpt(sds)
    out = {}
    for name, c in cases.TEXT_CASES.items():
        ids, mask, tmask = cases.gen_inputs(c)
        res, emb, cap = ref_harness.run_generate(
            embed, gpt, ids, mask, tmask, temperature=c["temperature"], top_P=c["top_P"], top_K=c["top_K"],
            repetition_penalty=c["rep"], max_new_token=c["max_new"], min_new_token=c["min_new"],
            manual_seed=c["manual_seed"], capture_logits=True, infer_text=True, eos_token=cases.TEXT_EOS)
        out[name + ".lens"] = np.array([r.shape[0] for r in res.ids], dtype=np.int64)
        out[name + ".ids"] = np.concatenate([r.numpy().reshape(-1) for r in res.ids], 0)
        for b in c["keep_hidden_rows"]:
            out[name + f".hid{b}"] = res.hiddens[b].numpy()
        for s in c["keep_logit_steps"]:
            if s < len(cap):
                out[name + f".tlogits{s}"] = cap[s].astype(np.float16)  # [B, 21178] logits / temperature (f16: size)
        print(name, "steps", len(cap), "lens", out[name + ".lens"].tolist())
    np.savez_compressed(os.path.join(OUT, "text.npz"), **out)


def golden_codec(sds):
    dec = ref_harness.build_decoder(sds)
    out = {}
    for name, c in cases.CODEC_CASES.items():
        hid = cases.codec_inputs(c)  # [B, T, 768]
        with torch.inference_mode():
            mel = dec(torch.from_numpy(hid).permute(0, 2, 1).contiguous())  # core.py:519-535 layout (B,768,T)
            wav = ref_harness.torch_vocos_decode(sds["vocos"], mel)
        out[name + ".mel"] = mel.numpy()  # [B, 100, 2T]
        out[name + ".wav"] = wav.numpy()  # [B, 256(2T-1)]
        print(name, mel.shape, wav.shape, "wav rms", float(wav.pow(2).mean().sqrt()))
    # the reference's own Chat._decode_to_wavs (core.py:513-539, unmodified) on ragged rows; its `self.decoder` is the reference DVAE, its
    # `self.vocos` the torch restatement (the package is absent)
    from oracle import make_host_goldens as MH

    class _V:
        def decode(self, spec):
            return ref_harness.torch_vocos_decode(sds["vocos"], spec)

    Chat = MH.ref_chat_class()
    chat = Chat.__new__(Chat)
    chat.decoder, chat.vocos, chat.device = dec, _V(), torch.device("cpu")
    rows = [torch.from_numpy(r) for r in cases.ragged_rows()]
    wav = chat._decode_to_wavs(rows, True)
    assert rows == []          # core.py:534 `del_all(result_list)`: the reference EMPTIES the caller's list
    out["ragged.wav"] = np.asarray(wav)
    print("ragged", out["ragged.wav"].shape, out["ragged.wav"].dtype)
    np.savez_compressed(os.path.join(OUT, "codec.npz"), **out)


def golden_codec_big(sds):
    """BASELINE-size acoustic goldens: mel from the reference's own DVAE class (ChatTTS/model/dvae.py:276-297), waveform
    from oracle/torch_port.vocos_decode, inputs = the reference GPT's hidden states (cases.codec_big_inputs)."""
    dec = ref_harness.build_decoder(sds)
    hid0 = np.load(os.path.join(OUT, "generate_big.npz"))["c2.hid0"]
    out = {}
    for name, c in cases.CODEC_BIG_CASES.items():
        t0 = time.time()
        hid, lens = cases.codec_big_inputs(c, hid0)
        with torch.inference_mode():
            mel = dec(torch.from_numpy(hid).permute(0, 2, 1).contiguous())  # core.py:519-535 layout (B,768,T)
            wav = ref_harness.torch_vocos_decode(sds["vocos"], mel)
        sub = cases.codec_big_subsample(mel.numpy(), wav.numpy())
        for k, v in sub.items():
            out[f"{name}.{k}"] = v
        out[name + ".lens"] = lens
        out[name + ".hid_sha256"] = np.array(W.fingerprint({"h": torch.from_numpy(hid)}))
        print(name, tuple(mel.shape), tuple(wav.shape), "mel peak", float(sub["mel_peak"][0]), "wav rms", float(sub["wav_rms"][0]), f"{time.time() - t0:.1f}s")
    np.savez_compressed(os.path.join(OUT, "codec_big.npz"), **out)


def main():
    assert ref_harness.available(), "/root/reference is required to generate goldens"
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    sds = W.synthetic_all()
    fp = {k: W.fingerprint(v) for k, v in sds.items()}
    fpath = os.path.join(OUT, "weights_fingerprint.txt")
    keep = {}
    if os.path.exists(fpath):     # lines other generators own (make_dvae_goldens.py: "dvae") stay
        with open(fpath) as f:
            keep = dict(line.split(None, 1) for line in f if line.strip())
    keep.update({k: fp[k] + "\n" for k in fp})
    keep["torch"] = torch.__version__ + "\n"
    with open(fpath, "w") as f:
        for k in sorted(keep):
            f.write(f"{k} {keep[k]}")
    which = sys.argv[1:] or ["sampling", "generate", "codec", "text"]
    if "sampling" in which:
        golden_sampling()
    if "generate" in which:
        golden_generate(sds)
    if "big" in which:   # BASELINE-size cases (C3 at B = 64, C2 at 512 steps): separate file, ~2 min of reference CPU time
        golden_generate(sds, cases.BIG_CASES, "generate_big.npz")
    if "params" in which:   # the sampling-parameter space + a 160-utterance batch (rows >= 625): ~1 min of reference CPU time
        golden_generate(sds, cases.PARAM_CASES, "generate_params.npz")
    if "max" in which:      # 2048 steps: ~10 min
        golden_generate(sds, cases.MAX_CASES, "generate_max.npz")
    if "sweep" in which:    # 40 seeded random configurations, ids only
        golden_sweep(sds)
    if "regen" in which:    # the "unexpected end at index" / regenerate path
        golden_regen(sds)
    if "stream" in which:   # the yield schedule of GPT.generate(stream=True)
        golden_stream(sds)
    if "codec" in which:
        golden_codec(sds)
    if "text" in which:
        golden_text(sds)
    if "codec_big" in which:   # needs generate_big.npz ("big") to exist: its c2.hid0 is the input
        golden_codec_big(sds)


if __name__ == "__main__":
    main()
