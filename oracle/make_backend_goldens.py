"""BUILD-CONTAINER ONLY.  Golden vectors for the back end of the path, produced by the REAL reference code imported from
/root/reference:

  float_to_int16   /root/reference/tools/audio/np.py:7-11 -- `numba.jit` stubbed by the identity decorator (numba is not installed), so the
                   function body runs as plain NumPy.  Two goldens per input: the float32 input as is (NumPy >= 2: python-int weak scalar,
                   float32 product -- key `.f32`), and the same input widened to float64 first (key `.f64`): that is the arithmetic of the
                   reference's RUNTIME, where numba types `float32[:] * int64` as float64 (same peak, same integer scale, exact product).
  ChatStreamer     /root/reference/examples/cmd/stream.py:9-145 fed the chunk sequences of oracle/cases.stream_chunks, for
                   output_format "PCM16_byte" and None (float pieces): sha256 of the yielded blocks concatenated + their lengths (the small
                   cases' bytes also in full);
                   its float_to_int16 is the function above in the `.f64` arithmetic (the runtime's) -- and once more in `.f32`.

    python -m oracle.make_backend_goldens      ->  tests/golden/backend.npz
TEST INFRASTRUCTURE."""
from __future__ import annotations

import contextlib
import hashlib
import importlib.util
import io
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import cases  # noqa: E402


def ref_float_to_int16():
    if "numba" not in sys.modules:
        nb = types.ModuleType("numba")
        nb.jit = lambda *a, **k: (lambda f: f)
        sys.modules["numba"] = nb
    spec = importlib.util.spec_from_file_location("_ref_audio_np", os.path.join(REF, "tools", "audio", "np.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.float_to_int16


def ref_streamer(f2i):
    """the reference's ChatStreamer with `tools.audio.float_to_int16` bound to `f2i` (tools/audio/__init__.py imports PyAV, absent here)"""
    pkg, sub = types.ModuleType("tools"), types.ModuleType("tools.audio")
    sub.float_to_int16 = f2i
    pkg.audio = sub
    saved = {k: sys.modules.get(k) for k in ("tools", "tools.audio")}
    sys.modules["tools"], sys.modules["tools.audio"] = pkg, sub
    try:
        spec = importlib.util.spec_from_file_location("_ref_stream", os.path.join(REF, "examples", "cmd", "stream.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod.ChatStreamer


def main():
    f2i = ref_float_to_int16()
    out = {}
    for name, x in cases.pcm_inputs().items():
        out[f"pcm.{name}.f32"] = f2i(x)
        out[f"pcm.{name}.f64"] = f2i(x.astype(np.float64))
        assert out[f"pcm.{name}.f32"].dtype == np.int16 and out[f"pcm.{name}.f64"].dtype == np.int16
        print(name, x.shape, "f32 != f64 on", int((out[f"pcm.{name}.f32"] != out[f"pcm.{name}.f64"]).sum()), "samples")
    for prod, fn in (("f64", lambda a: f2i(a.astype(np.float64))), ("f32", f2i)):
        Streamer = ref_streamer(fn)
        for name in cases.STREAM_CASES:
            for fmt in ("PCM16_byte", None):
                key = f"stream.{name}.{prod}.{'bytes' if fmt else 'float'}"
                try:
                    with contextlib.redirect_stdout(io.StringIO()):        # the class prints its progress
                        blocks = list(Streamer().generate(iter(cases.stream_chunks(name)), output_format=fmt))
                except (UnboundLocalError, ValueError, IndexError) as exc:
                    # the reference class crashes on some sequences (e.g. `is_keep_next` unbound when the first chunk is silent for everybody,
                    # stream.py:124): recorded as such -- the port's behaviour there is its own (tests/test_backend.py)
                    out[key + ".raises"] = np.array(type(exc).__name__)
                    print(key, "reference raises", type(exc).__name__)
                    continue
                if not fmt and prod == "f32":
                    continue          # no conversion in this format: one golden is enough
                raw = b"".join(blocks) if fmt else b"".join(np.ascontiguousarray(b, dtype="<f4").tobytes() for b in blocks)
                out[key + ".sha256"] = np.array(hashlib.sha256(raw).hexdigest())     # the blocks, concatenated, byte for byte
                out[key + ".lens"] = np.array([len(b) for b in blocks], np.int64)    # ... and where each one ends
                if fmt and name in ("one", "late"):
                    out[key] = np.frombuffer(raw, dtype=np.uint8)                    # the small cases also in full (a failing test can show where)
                print(key, len(blocks), "blocks", out[key + ".lens"].tolist()[:12])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "backend.npz"), **out)
    print("backend.npz", os.path.getsize(os.path.join(ROOT, "tests", "golden", "backend.npz")), "bytes")


if __name__ == "__main__":
    main()
