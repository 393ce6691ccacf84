"""Real-checkpoint readiness: what to run the day the assets and the third-party packages are reachable.

    python tools/validate_assets.py <dir holding asset/ | the asset dir itself> [--json] [--write-goldens]

1. ASSETS.  Reads only the safetensors HEADERS (no tensor is loaded) of the files the hot path repacks from -- `gpt/model.safetensors`,
   `Embed.safetensors`, `Decoder.safetensors`, `Vocos.safetensors` (+ `DVAE.safetensors` when present) -- and diffs key names / shapes /
   dtypes against SURVEY App. B as `chattts_amd.weights.expected_schema` states it (reference: config.py:4-11 paths, gpt.py:75-78,
   embed.py:18-35, dvae.py:145-161,226,239; the Vocos key set is the one item of App. B that could only be INFERRED offline -- this is
   where it gets checked against the real file); `gpt/config.json` goes through `check_gpt_config` (the geometry the kernels are built for).
2. THIRD-PARTY PINS.  SURVEY 8c rows a17 / f2 are "parity unpinned" because `vocos`, `vector_quantize_pytorch` and `torchaudio` are not
   installed in the build container.  For each one that imports HERE, the restatement the oracle uses is compared with the package itself
   on seeded inputs (the asset's weights when given, the synthetic recipe otherwise) and, with --write-goldens, package-generated goldens
   are written to tests/golden/pkg_*.npz, which tests/test_oracle_vs_golden.py picks up -- that flips the rows to "pinned".
   A package that does not import is reported as `absent` (not an error).
Exit code: 0 = every present asset file matches; 1 = a layout mismatch (the diff is printed); 2 = usage.
Test infrastructure / tooling: imports oracle/, never imported by chattts_amd/."""
from __future__ import annotations

import argparse
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from chattts_amd import weights as W  # noqa: E402

ST_DTYPES_FLOAT = {"F64", "F32", "F16", "BF16"}


def read_header(path: str) -> dict:
    """safetensors: u64 little-endian header length, then that many bytes of JSON {name: {dtype, shape, data_offsets}}"""
    with open(path, "rb") as fh:
        (n,) = struct.unpack("<Q", fh.read(8))
        hdr = json.loads(fh.read(n).decode("utf-8"))
    hdr.pop("__metadata__", None)
    return hdr


def asset_root(root: str) -> str:
    if os.path.isdir(os.path.join(root, "asset")) and not os.path.exists(os.path.join(root, W.ASSET_FILES["embed"])):
        return os.path.join(root, "asset")
    return root


def diff_file(name: str, path: str) -> dict:
    hdr = read_header(path)
    keys = {}
    for k, meta in hdr.items():
        kk = k[len("model."):] if (name == "gpt" and k.startswith("model.")) else k
        keys[kk] = (tuple(meta["shape"]), meta["dtype"])
    keys.pop("embed_tokens.weight", None)      # gpt.py:78
    rep = {"file": path, "tensors": len(keys), "ok": True}
    if name == "dvae":      # the full DVAE's key set depends on the quantiser package's module tree: listed, not judged
        rep["note"] = "optional file: key names listed by prefix only"
        rep["prefixes"] = sorted({k.split(".")[0] for k in keys})
        return rep
    n_layers = 0
    while name == "gpt" and f"layers.{n_layers}.input_layernorm.weight" in keys:
        n_layers += 1
    want = W.expected_schema(name, n_layers) if name == "gpt" else W.expected_schema(name)
    ign = W.IGNORED_PREFIXES.get(name, ())
    rep["missing"] = sorted(k for k in want if k not in keys)
    rep["unexpected"] = sorted(k for k in keys if k not in want and not (ign and k.startswith(ign)) and ".rotary_emb." not in k)
    rep["ignored"] = sorted(k for k in keys if k not in want and ((ign and k.startswith(ign)) or ".rotary_emb." in k))
    rep["wrong_shape"] = [f"{k}: file {keys[k][0]} != expected {tuple(want[k])}" for k in want if k in keys and keys[k][0] != tuple(want[k])]
    rep["not_float"] = [f"{k}: {keys[k][1]}" for k in want if k in keys and keys[k][1] not in ST_DTYPES_FLOAT]
    rep["dtypes"] = sorted({v[1] for v in keys.values()})
    if name == "gpt":
        rep["layers"] = n_layers
    rep["ok"] = not (rep["missing"] or rep["unexpected"] or rep["wrong_shape"] or rep["not_float"]) and (name != "gpt" or n_layers > 0)
    return rep


def check_assets(root: str) -> dict:
    root = asset_root(root)
    out = {"root": root, "files": {}}
    for name, rel in {**W.ASSET_FILES, **W.OPTIONAL_ASSET_FILES}.items():
        path = os.path.join(root, rel)
        if not os.path.exists(path):
            out["files"][name] = {"file": path, "ok": name in W.OPTIONAL_ASSET_FILES, "absent": True}
            continue
        try:
            out["files"][name] = diff_file(name, path)
        except (OSError, ValueError, KeyError, struct.error) as e:
            out["files"][name] = {"file": path, "ok": False, "error": f"unreadable safetensors header: {e}"}
    try:
        n_layers = out["files"]["gpt"].get("layers")
        out["gpt_config"] = {"ok": True, "runtime_fields": W.check_gpt_config(W.load_gpt_config(root), n_layers)}
    except W.AssetError as e:
        out["gpt_config"] = {"ok": False, "error": str(e)}
    out["ok"] = all(f["ok"] for f in out["files"].values()) and out["gpt_config"]["ok"]
    return out


# ---- third-party pins ---------------------------------------------------------------------------------------------------------------
def pin_vocos(vocos_sd, write_to=None) -> dict:
    """`vocos.Vocos.decode` (core.py:505-510; init arguments config.py:83-121) against oracle/torch_port.vocos_decode (the restatement
    the a17 tests compare the kernels with)."""
    try:
        import vocos  # noqa: F401
        from vocos.heads import ISTFTHead
        from vocos.models import VocosBackbone
    except ImportError as e:
        return {"status": "absent", "detail": str(e)}
    import torch
    from oracle import torch_port
    try:
        backbone = VocosBackbone(input_channels=100, dim=512, intermediate_dim=1536, num_layers=8)
        head = ISTFTHead(dim=512, n_fft=1024, hop_length=256, padding="center")
        missing = []
        for mod, pref in ((backbone, "backbone."), (head, "head.")):
            sd = {k[len(pref):]: v.float() for k, v in vocos_sd.items() if k.startswith(pref)}
            r = mod.load_state_dict(sd, strict=False)
            missing += [pref + k for k in r.missing_keys] + ["(unexpected) " + pref + k for k in r.unexpected_keys]
        g = torch.Generator().manual_seed(7)
        mel = torch.randn((2, 100, 48), generator=g)
        with torch.inference_mode():
            want = head(backbone(mel)).numpy()
            got = torch_port.vocos_decode({k: v.float() for k, v in vocos_sd.items()}, mel).numpy()
        d = got.astype(np.float64) - want.astype(np.float64)
        rep = {"status": "pinned" if (not missing and float(np.abs(d).max()) < 1e-4) else "MISMATCH", "package": getattr(vocos, "__version__", "?"),
               "state_dict_problems": missing, "max_abs_diff": float(np.abs(d).max()), "rms_diff": float(np.sqrt(np.mean(d ** 2))),
               "rms_signal": float(np.sqrt(np.mean(want.astype(np.float64) ** 2)))}
        if write_to:
            np.savez_compressed(os.path.join(write_to, "pkg_vocos.npz"), mel=mel.numpy(), wav=want)
            rep["golden"] = os.path.join(write_to, "pkg_vocos.npz")
        return rep
    except Exception as e:   # a package whose API differs from the one restated: say so, do not die
        return {"status": "attempt_failed", "detail": f"{type(e).__name__}: {e}"}


def pin_gfsq(dvae_sd, write_to=None) -> dict:
    """`vector_quantize_pytorch.GroupedResidualFSQ` (dvae.py:69-128; config.py:24-28: dim 1024, levels (5,5,5,5), G 2, R 2) against
    oracle/dvae_np.gfsq_encode / gfsq_embed."""
    try:
        from vector_quantize_pytorch import GroupedResidualFSQ
    except ImportError as e:
        return {"status": "absent", "detail": str(e)}
    import torch
    from oracle import dvae_np
    try:
        q = GroupedResidualFSQ(dim=1024, levels=[5, 5, 5, 5], num_quantizers=2, groups=2)
        sd = {k[len("vq_layer.quantizer."):]: v.float() for k, v in dvae_sd.items() if k.startswith("vq_layer.quantizer.")}
        r = q.load_state_dict(sd, strict=False)
        q.eval()
        g = torch.Generator().manual_seed(9)
        x = torch.randn((2, 40, 1024), generator=g)
        with torch.inference_mode():
            _, ind = q(x)                                     # [G, B, T, R]
            want = ind.permute(1, 2, 0, 3).flatten(2).transpose(1, 2).numpy()      # dvae.py:99-106 -> [B, G*R, T]
        nsd = {k: v.float().numpy() for k, v in dvae_sd.items()}
        res = {}
        for bf in (True, False):
            got = dvae_np.gfsq_encode(nsd, x.numpy(), bound_first=bf)
            res[bf] = float((got != want).mean())
        best = min(res, key=res.get)
        rep = {"status": "pinned" if res[best] == 0.0 and not r.missing_keys else "MISMATCH", "differing_code_fraction": res[best],
               "bound_first": best, "state_dict_problems": list(r.missing_keys) + ["(unexpected) " + k for k in r.unexpected_keys]}
        if write_to:
            np.savez_compressed(os.path.join(write_to, "pkg_gfsq.npz"), x=x.numpy(), codes=want, bound_first=np.array(best))
            rep["golden"] = os.path.join(write_to, "pkg_gfsq.npz")
        return rep
    except Exception as e:
        return {"status": "attempt_failed", "detail": f"{type(e).__name__}: {e}"}


def pin_mel(write_to=None) -> dict:
    """`torchaudio.transforms.MelSpectrogram(24000, n_fft 1024, hop 256, n_mels 100, center, power 1)` + log(clip(., 1e-5))
    (dvae.py:175-206) against oracle/dvae_np.mel_features."""
    try:
        import torchaudio
    except ImportError as e:
        return {"status": "absent", "detail": str(e)}
    import torch
    from oracle import dvae_np
    try:
        ms = torchaudio.transforms.MelSpectrogram(sample_rate=24000, n_fft=1024, hop_length=256, n_mels=100, center=True, power=1)
        g = torch.Generator().manual_seed(11)
        wav = torch.randn((1, 24000), generator=g) * 0.1
        with torch.inference_mode():
            want = torch.log(torch.clip(ms(wav), min=1e-5)).numpy()
        got = dvae_np.mel_features(wav.numpy(), dvae_np.hann_periodic(1024), dvae_np.melscale_fbanks())
        got = np.asarray(got)
        if got.shape != want.shape and got.shape == want.transpose(0, 2, 1).shape:
            got = got.transpose(0, 2, 1)
        d = got.astype(np.float64) - want.astype(np.float64)
        rep = {"status": "pinned" if float(np.abs(d).max()) < 1e-3 else "MISMATCH", "package": torchaudio.__version__, "max_abs_diff": float(np.abs(d).max())}
        if write_to:
            np.savez_compressed(os.path.join(write_to, "pkg_mel.npz"), wav=wav.numpy(), mel=want)
            rep["golden"] = os.path.join(write_to, "pkg_mel.npz")
        return rep
    except Exception as e:
        return {"status": "attempt_failed", "detail": f"{type(e).__name__}: {e}"}


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("root", help="directory that holds asset/ (the reference's download path) or the asset directory itself")
    ap.add_argument("--json", action="store_true", help="one JSON object instead of the readable report")
    ap.add_argument("--write-goldens", action="store_true", help="write package-generated goldens to tests/golden/pkg_*.npz")
    ap.add_argument("--no-pins", action="store_true", help="asset layout only")
    args = ap.parse_args(argv)
    if not os.path.isdir(args.root):
        print(f"{args.root}: not a directory", file=sys.stderr)
        return 2
    rep = check_assets(args.root)
    if not args.no_pins:
        gold = os.path.join(ROOT, "tests", "golden") if args.write_goldens else None
        sds = None
        if rep["ok"]:
            try:
                sds = W.load_assets(args.root)
            except W.AssetError:
                sds = None
        # goldens always come from the synthetic recipe (portable: tests re-create those weights); a loadable asset is pinned too
        rep["pins"] = {"weights": "synthetic recipe" + (" (goldens) + the asset's" if sds else ""),
                       "vocos": pin_vocos(W.synthetic_vocos(), gold), "vector_quantize_pytorch": pin_gfsq(W.synthetic_dvae(), gold),
                       "torchaudio": pin_mel(gold)}
        if sds:
            rep["pins"]["vocos_asset"] = pin_vocos(sds["vocos"])
            if "dvae" in sds:
                rep["pins"]["vector_quantize_pytorch_asset"] = pin_gfsq(sds["dvae"])
    if args.json:
        print(json.dumps(rep))
    else:
        print(f"asset root: {rep['root']}")
        for name, f in rep["files"].items():
            state = "absent (optional)" if f.get("absent") and f["ok"] else "ABSENT" if f.get("absent") else "ok" if f["ok"] else "MISMATCH"
            print(f"  {name:8s} {state:18s} {f['file']}" + (f"  [{f.get('tensors')} tensors, {','.join(f.get('dtypes', []))}]" if "tensors" in f else ""))
            for title in ("error", "missing", "unexpected", "wrong_shape", "not_float"):
                items = f.get(title)
                if items:
                    items = [items] if isinstance(items, str) else items
                    print(f"      {title} ({len(items)}): " + "; ".join(items[:12]) + (" ..." if len(items) > 12 else ""))
            if f.get("ignored"):
                print(f"      present but not read by the hot path ({len(f['ignored'])}): " + "; ".join(f["ignored"][:6]) + (" ..." if len(f["ignored"]) > 6 else ""))
        gc = rep["gpt_config"]
        print("  gpt/config.json " + ("ok, run-time fields " + json.dumps(gc["runtime_fields"]) if gc["ok"] else "MISMATCH\n      " + gc["error"].replace("\n", "\n      ")))
        for pkg, p in rep.get("pins", {}).items():
            if pkg != "weights":
                print(f"  pin {pkg:24s} {p['status']}" + "".join(f"  {k}={v}" for k, v in p.items() if k not in ("status",) and not isinstance(v, list)))
        print("RESULT: " + ("assets match the layout the engine repacks from" if rep["ok"] else "LAYOUT MISMATCH (see above)"))
    return 0 if rep["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
