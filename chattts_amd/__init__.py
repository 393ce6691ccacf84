"""chattts_amd -- MI355X-native (gfx950) engine for the ChatTTS hot path:
GPT speech-token generation -> DVAE decoder -> Vocos, as hand-written HIP kernels behind a C ABI
(include/chattts_amd.h, csrc/libchattts_amd.so).  See DESIGN.md / INTEGRATION.md.

    from chattts_amd.core import Chat, InferCodeParams      # Chat-level seam (token ids in, waveforms out)
    from chattts_amd.engine import GptEngine, CodecEngine   # GPT.generate / DVAE+Vocos drop-ins

Nothing here computes on the CPU: the engines raise `EngineError` without the HIP library or a GPU.
"""
__version__ = "0.1.0"
