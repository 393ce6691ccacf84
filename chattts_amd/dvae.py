"""`DvaeEngine`: the full DVAE (`asset/DVAE.safetensors`) on the device -- SURVEY.md 8f-2.

  * `sample_audio(wav)`   == `DVAE.sample_audio` (/root/reference/ChatTTS/model/dvae.py:299-303): 24 kHz waveform ->
                             audio codes [4, T] (what `Chat.sample_audio_speaker` packs into the `spk_smp` string)
  * `decode_codes(ids)`   == `DVAE.forward(batch_ids)` (dvae.py:276-297) as used by
                             `Chat._decode_to_wavs(result.ids, use_decoder=False)` (core.py:513-539): zero-padded code
                             rows -> mel [B, 2Tmax, 100]
Both are thin wrappers over `ctts_dvae_encode` / `ctts_dvae_decode_codes` (include/chattts_amd.h); there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Union

import numpy as np
import torch

from . import _lib


class DvaeEngine:
    LEVELS, G, R = (5, 5, 5, 5), 2, 2          # config.py:23-28

    def __init__(self, dvae_sd: dict, device: torch.device, bound_first: bool = True):
        self.lib = _lib.lib()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.EngineError("DvaeEngine needs a ROCm GPU device (there is no CPU path)")
        dev = self.device
        sd = dvae_sd
        f = lambda t: t.to(torch.float32).contiguous().to(dev)
        self.keep = []
        k = self.keep.append

        def P(t):
            k(t)
            return t.data_ptr()

        def PA(ts):
            k(ts)
            arr = _lib.ptr_array(ts)
            k(arr)
            return C.cast(arr, _lib.PP)

        convw = lambda t: f(t.permute(0, 2, 1).reshape(t.shape[0], -1))      # [Cout,Cin,k] -> [Cout][k*Cin]
        dww = lambda t: f(t[:, 0, :].t())                                     # [C,1,7] -> [7,C]

        def trunk(tw: _lib.TrunkWeights, p: str):
            w0, w2, wo = sd[p + "conv_in.0.weight"], sd[p + "conv_in.2.weight"], sd[p + "conv_out.weight"]
            tw.idim, tw.bn_dim, tw.hidden, tw.odim = int(w0.shape[1]), int(w0.shape[0]), int(w2.shape[0]), int(wo.shape[0])
            nb = 0
            while f"{p}decoder_block.{nb}.weight" in sd:
                nb += 1
            tw.n_blocks = nb
            tw.conv_in0_w, tw.conv_in0_b = P(convw(w0)), P(f(sd[p + "conv_in.0.bias"]))
            tw.conv_in2_w, tw.conv_in2_b = P(convw(w2)), P(f(sd[p + "conv_in.2.bias"]))
            blk = lambda i, n: sd[f"{p}decoder_block.{i}.{n}"]
            tw.dw_w = PA([dww(blk(i, "dwconv.weight")) for i in range(nb)])
            tw.dw_b = PA([f(blk(i, "dwconv.bias")) for i in range(nb)])
            tw.ln_w = PA([f(blk(i, "norm.weight")) for i in range(nb)])
            tw.ln_b = PA([f(blk(i, "norm.bias")) for i in range(nb)])
            tw.pw1_w = PA([f(blk(i, "pwconv1.weight")) for i in range(nb)])
            tw.pw1_b = PA([f(blk(i, "pwconv1.bias")) for i in range(nb)])
            tw.pw2_w = PA([f(blk(i, "pwconv2.weight")) for i in range(nb)])
            tw.pw2_b = PA([f(blk(i, "pwconv2.bias")) for i in range(nb)])
            tw.gamma = PA([f(blk(i, "weight")) for i in range(nb)])
            tw.conv_out_w = P(f(wo[:, :, 0]))

        w = _lib.DvaeWeights()
        trunk(w.encoder, "encoder.")
        trunk(w.decoder, "decoder.")
        w.ds0_w, w.ds0_b = P(convw(sd["downsample_conv.0.weight"])), P(f(sd["downsample_conv.0.bias"]))
        # Conv1d(512,512,4,stride 2,pad 1) over frames == Conv1d(1024,512,3,pad 1) over frame PAIRS:
        # out[t] = W0 x[2t-1] + W1 x[2t] + W2 x[2t+1] + W3 x[2t+2]  ->  taps {[0,W0], [W1,W2], [W3,0]} on (x[2u], x[2u+1])
        w4 = sd["downsample_conv.2.weight"].to(torch.float32)             # [512, 512, 4]
        z = torch.zeros_like(w4[:, :, 0])
        pair = torch.stack([torch.cat([z, w4[:, :, 0]], 1), torch.cat([w4[:, :, 1], w4[:, :, 2]], 1), torch.cat([w4[:, :, 3], z], 1)], 1)
        w.ds1_w, w.ds1_b = P(f(pair.reshape(512, -1))), P(f(sd["downsample_conv.2.bias"]))
        w.out_conv_w = P(convw(sd["out_conv.weight"]))
        w.coef = P(f(sd["coef"].reshape(-1)))
        q = lambda g, n: sd[f"vq_layer.quantizer.rvqs.{g}.{n}"]
        w.q_in_w = P(f(torch.stack([q(g, "project_in.weight") for g in range(self.G)])))
        w.q_in_b = P(f(torch.stack([q(g, "project_in.bias") for g in range(self.G)])))
        w.q_out_w = P(f(torch.stack([q(g, "project_out.weight") for g in range(self.G)])))
        w.q_out_b = P(f(torch.stack([q(g, "project_out.bias") for g in range(self.G)])))
        for i, lv in enumerate(self.LEVELS):
            w.levels[i] = lv
        w.G, w.R, w.D, w.bound_first = self.G, self.R, 512, int(bound_first)
        w.mel_window = P(f(sd["preprocessor_mel.mel_spec.spectrogram.window"]))
        fb = sd["preprocessor_mel.mel_spec.mel_scale.fb"].to(torch.float32)   # [513, 100]
        fbt = torch.zeros((100, 516), dtype=torch.float32)
        fbt[:, :513] = fb.t()
        w.mel_fb = P(f(fbt))
        kk = torch.arange(512, dtype=torch.float64) * (2.0 * math.pi / 1024)
        w.twiddle = P(f(torch.stack([kk.cos(), kk.sin()], 1)))
        self._w = w
        h = C.c_void_p()
        _lib.check(self.lib.ctts_dvae_create(C.byref(h), C.byref(w)), "ctts_dvae_create")
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.ctts_dvae_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def sample_audio(self, wav: Union[np.ndarray, torch.Tensor]) -> torch.Tensor:
        """wav [n] float32 -> codes [4, T] int32 on the host (dvae.py:299-303)."""
        if isinstance(wav, np.ndarray):
            wav = torch.from_numpy(wav)
        wav = wav.reshape(-1).to(torch.float32).contiguous().to(self.device)
        n = int(wav.numel())
        T = int(self.lib.ctts_dvae_code_frames(n))
        codes = torch.empty((max(T, 0), self.G * self.R), dtype=torch.int32, device=self.device)
        nb = self.lib.ctts_dvae_encode_workspace_bytes(n)
        ws = torch.empty((nb,), dtype=torch.uint8, device=self.device)
        st = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self.lib.ctts_dvae_encode(self.handle, wav.data_ptr(), n, codes.data_ptr(), ws.data_ptr(), nb, st), "ctts_dvae_encode")
        return codes.t().contiguous().cpu()

    def decode_codes(self, ids: Union[torch.Tensor, List[torch.Tensor]], pad_to: Optional[int] = None) -> torch.Tensor:
        """[B,T,4] int64, or a list of [T_b,4] rows (zero padded to the longest: core.py:519-534) -> mel [B,2T,100].
        `pad_to` (rows): pad to that many tokens instead -- the rows of a batch whose longest row is elsewhere (dist.infer_sharded)."""
        if isinstance(ids, (list, tuple)):
            Tmax = max(int(r.size(0)) for r in ids)
            if pad_to is not None:
                assert int(pad_to) >= Tmax, "pad_to is shorter than the longest row"
                Tmax = int(pad_to)
            batch = torch.zeros((len(ids), Tmax, self.G * self.R), dtype=torch.int64, device=self.device)
            for i, r in enumerate(ids):
                batch[i, : r.size(0)] = r.to(self.device)
            ids = batch
        ids = ids.to(torch.int64).contiguous().to(self.device)
        B, T, _ = ids.shape
        mel = torch.empty((B, 2 * T, 100), dtype=torch.float32, device=self.device)
        nb = self.lib.ctts_dvae_decode_workspace_bytes(B, T)
        ws = torch.empty((nb,), dtype=torch.uint8, device=self.device)
        st = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self.lib.ctts_dvae_decode_codes(self.handle, ids.data_ptr(), mel.data_ptr(), B, T, ws.data_ptr(), nb, st),
                   "ctts_dvae_decode_codes")
        return mel
