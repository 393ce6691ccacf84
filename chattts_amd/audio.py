"""Host back end of the path: float32 waveform -> 16-bit PCM / WAV bytes, what the reference's
examples do with `Chat.infer`'s output (/root/reference/tools/audio/np.py:7-12, pcm.py:8-33; examples/cmd/run.py).
The device twin of `float_to_int16` is `CodecEngine.float_to_int16` (csrc/codec.hip pcm16_k, C ABI ctts_float_to_int16); this host
form serves the odd pieces (a stream's last chunk after its column filter, split_text concatenations) and the tests.
mp3 / ogg go through PyAV in the reference (tools/audio/av.py), which is not part of this engine."""
from __future__ import annotations

import math
import wave
from io import BytesIO

import numpy as np


def pcm_scale(peak: float) -> int:
    """np.py:9-10: am = 32767 * 32768 // (int(ceil(peak)) * 32768); 0 for a silent clip (the reference divides by zero there)"""
    c = int(math.ceil(float(peak))) * 32768
    return 32767 * 32768 // c if c else 0


def float_to_int16(audio: np.ndarray, product: str = "f64") -> np.ndarray:
    """np.py:7-11: scale by 32767 / ceil(max|x|) in integer arithmetic, truncate toward zero; ONE peak over the whole array, whatever its
    rank.  `product`: "f64" = the reference as it RUNS (the function is numba-jitted, and numba types float32[:] * int64 as float64: the
    product is exact before the truncation); "f32" = what plain NumPy >= 2 makes of the same source line (python int = weak scalar: the
    float32 product is rounded first).  The two differ by one count on roughly one sample in 10^4.  A silent clip returns zeros."""
    audio = np.asarray(audio)
    am = pcm_scale(np.abs(audio).max()) if audio.size else 0
    if am == 0:
        return np.zeros(audio.shape, dtype=np.int16)
    if product == "f32":
        return np.multiply(audio.astype(np.float32, copy=False), np.float32(am)).astype(np.int16)
    return np.multiply(audio.astype(np.float64), float(am)).astype(np.int16)


def pcm_to_wav_bytes(wav: np.ndarray, sample_rate: int = 24000) -> bytes:
    """pcm.py:8-33: mono, 16-bit little-endian RIFF/WAVE."""
    buf = BytesIO()
    with wave.open(buf, "wb") as wf:
        wf.setnchannels(1)
        wf.setsampwidth(2)
        wf.setframerate(sample_rate)
        wf.writeframes(float_to_int16(np.asarray(wav, dtype=np.float32).reshape(-1)).astype("<i2").tobytes())
    return buf.getvalue()
