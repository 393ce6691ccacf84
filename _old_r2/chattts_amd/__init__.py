"""chattts_amd -- MI355X-native (gfx950) engine for the ChatTTS hot path:
GPT speech-token generation -> DVAE decoder -> Vocos, as hand-written HIP kernels behind a C ABI
(include/chattts_amd.h, csrc/libchattts_amd.so).  See DESIGN.md / INTEGRATION.md.

    from chattts_amd.core import Chat                       # Chat.infer(text, ...) with the reference's arguments, or token-level calls
    from chattts_amd.engine import GptEngine, CodecEngine   # GPT.generate / DVAE-decoder + Vocos drop-ins
    from chattts_amd.dvae import DvaeEngine                 # full DVAE: audio -> codes, codes -> mel
    from chattts_amd.frontend import Normalizer, Tokenizer, Speaker   # host front end (strings, prompts, speaker vectors)
    from chattts_amd.serving import SlotPool                # continuous batching over utterance slots

Nothing here computes on the CPU: the engines raise `EngineError` without the HIP library or a GPU.
"""
__version__ = "0.1.0"
