"""Shape constants of the hot path (GPT speech-token generator -> DVAE decoder -> Vocos).

Values restate /root/reference/ChatTTS/config/config.py:14-20 (Decoder), :50-63 (GPT),
:66-71 (Embed), :75-121 (Vocos feature/backbone/head init args).  They are the contract
the HIP kernels are compiled/launched against; nothing here is tunable at run time.
"""
from dataclasses import dataclass


@dataclass(frozen=True)
class GptDims:
    hidden: int = 768          # config.py:52
    inter: int = 3072          # config.py:53
    n_heads: int = 12          # config.py:54
    head_dim: int = 64
    n_layers: int = 20         # config.py:55
    max_pos: int = 4096        # config.py:57
    n_audio: int = 626         # config.py:61  (625 FSQ codes + EOS id 625)
    n_text: int = 21178        # config.py:62
    n_vq: int = 4              # config.py:63
    rms_eps: float = 1e-6      # HF LlamaConfig default used by the reference asset
    rope_theta: float = 10000.0


@dataclass(frozen=True)
class DvaeDims:
    idim: int = 384            # config.py:15
    odim: int = 384            # config.py:16
    hidden: int = 512          # config.py:17
    n_layers: int = 12         # config.py:18
    bn_dim: int = 128          # config.py:19
    kernel: int = 7            # dvae.py:139
    dilation: int = 2          # dvae.py:140
    n_mels: int = 100          # dvae.py:239
    ln_eps: float = 1e-6       # dvae.py:35


@dataclass(frozen=True)
class VocosDims:
    n_mels: int = 100          # config.py:89
    dim: int = 512             # config.py:90
    inter: int = 1536          # config.py:91
    n_layers: int = 8          # config.py:92
    n_fft: int = 1024          # config.py:105
    hop: int = 256             # config.py:106
    ln_eps: float = 1e-6       # vocos.models.VocosBackbone (un-vendored dep)


GPT = GptDims()
DVAE = DvaeDims()
VOCOS = VocosDims()

EOS_CODE = 625                 # core.py:580  (num_audio_tokens - 1)
SAMPLES_PER_TOKEN = 512        # 2 mel frames x hop 256
SAMPLE_RATE = 24000
