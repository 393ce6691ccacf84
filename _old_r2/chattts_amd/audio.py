"""Host back end of the path: float32 waveform -> 16-bit PCM / WAV bytes, what the reference's
examples do with `Chat.infer`'s output (/root/reference/tools/audio/np.py:7-12, pcm.py:8-33; examples/cmd/run.py).
mp3 / ogg go through PyAV in the reference (tools/audio/av.py), which is not part of this engine."""
from __future__ import annotations

import math
import wave
from io import BytesIO

import numpy as np


def float_to_int16(audio: np.ndarray) -> np.ndarray:
    """np.py:7-12: scale by 32767 / ceil(max|x|) (integer arithmetic as in the reference: 32767 * 32768 // (ceil * 32768)),
    truncate toward zero.  A silent clip (max 0) divides by zero there; here it returns zeros."""
    peak = int(math.ceil(float(np.abs(audio).max()))) * 32768 if audio.size else 0
    if peak == 0:
        return np.zeros(audio.shape, dtype=np.int16)
    return np.multiply(audio, 32767 * 32768 // peak).astype(np.int16)


def pcm_to_wav_bytes(wav: np.ndarray, sample_rate: int = 24000) -> bytes:
    """pcm.py:8-33: mono, 16-bit little-endian RIFF/WAVE."""
    buf = BytesIO()
    with wave.open(buf, "wb") as wf:
        wf.setnchannels(1)
        wf.setsampwidth(2)
        wf.setframerate(sample_rate)
        wf.writeframes(float_to_int16(np.asarray(wav, dtype=np.float32).reshape(-1)).tobytes())
    return buf.getvalue()
