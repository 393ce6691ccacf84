"""ctypes binding of libchattts_amd.so (include/chattts_amd.h).  There is NO fallback: if the HIP
library is missing or fails to load, importing this module's `lib()` raises."""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- MUST precede the dlopen below: torch bundles its own libamdhip64; loading ours first
#                  would bind libchattts_amd.so to a second HIP runtime that cannot see torch's device context

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CTTS_LIB") or os.path.join(HERE, "csrc", "libchattts_amd.so")   # CTTS_LIB: A/B builds of the same ABI

F32, BF16 = 0, 1
P = C.c_void_p
PP = C.POINTER(C.c_void_p)


class GptWeights(C.Structure):
    _fields_ = [
        ("n_layers", C.c_int32), ("weight_dtype", C.c_int32), ("kv_dtype", C.c_int32), ("max_pos", C.c_int32),
        ("wqkv", PP), ("wo", PP), ("wgu", PP), ("wd", PP), ("ln1", PP), ("ln2", PP),
        ("norm", P), ("emb_code", P), ("heads", P), ("rope_cos", P), ("rope_sin", P),
        ("rms_eps", C.c_float),
        ("emb_text", P), ("head_text", P), ("n_text", C.c_int32),
        ("wqkv_pk", PP), ("wo_pk", PP), ("wgu_pk", PP), ("wd_pk", PP),
        ("heads_pk", P), ("head_text_pk", P),
        ("wo_hd", PP),
        ("wqkv_x3", PP), ("wo_x3", PP), ("wgu_x3", PP), ("wd_x3", PP),
    ]


class GenState(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("T", C.c_int32), ("max_new", C.c_int32),
        ("ids_buf", P), ("len", P), ("kv_start", P), ("finish", P), ("end_idx", P), ("hiddens", P),
        ("kcache", P), ("vcache", P), ("q", P), ("nq", C.c_int32),
        ("temperature", P), ("pow_table", P),
        ("top_p_thr", C.c_float), ("use_top_p", C.c_int32), ("top_k", C.c_int32), ("use_top_k", C.c_int32),
        ("min_new", C.c_int32), ("eos", C.c_int32), ("row_offset", C.c_int32),
        ("stop_at", P), ("workspace", P), ("workspace_bytes", C.c_size_t), ("row_map", P), ("n_active", P),
        ("cap", C.c_int32), ("hid_cap", C.c_int32), ("kv_batch", C.c_int32), ("q_batch", C.c_int32), ("prompt_len", P),
        ("infer_text", C.c_int32), ("teacher_ids", P), ("sampled_ids", P), ("order", P),
        ("rng_device", C.c_int32), ("rng_per_step", C.c_int32), ("rng_seed", P), ("rng_nonce", P),
        ("margin", P), ("row_base", P), ("proj_exact", C.c_int32), ("prefill_valid_rows", C.c_int32),
    ]


class CodecWeights(C.Structure):
    _fields_ = [
        ("conv_in0_w", P), ("conv_in0_b", P), ("conv_in2_w", P), ("conv_in2_b", P),
        ("n_dvae_blocks", C.c_int32),
        ("d_dw_w", PP), ("d_dw_b", PP), ("d_ln_w", PP), ("d_ln_b", PP), ("d_pw1_w", PP), ("d_pw1_b", PP),
        ("d_pw2_w", PP), ("d_pw2_b", PP), ("d_gamma", PP),
        ("conv_out_w", P), ("out_conv_w", P), ("coef", P),
        ("v_embed_w", P), ("v_embed_b", P), ("v_norm_w", P), ("v_norm_b", P),
        ("n_vocos_blocks", C.c_int32),
        ("v_dw_w", PP), ("v_dw_b", PP), ("v_ln_w", PP), ("v_ln_b", PP), ("v_pw1_w", PP), ("v_pw1_b", PP),
        ("v_pw2_w", PP), ("v_pw2_b", PP), ("v_gamma", PP),
        ("v_final_w", P), ("v_final_b", P), ("head_w", P), ("head_b", P), ("window", P), ("twiddle", P),
        ("gemm_mode", C.c_int32),
        ("d_pw1_x3p", PP), ("d_pw2_x3p", PP), ("v_pw1_x3p", PP), ("v_pw2_x3p", PP),
    ]


class TrunkWeights(C.Structure):
    _fields_ = [
        ("idim", C.c_int32), ("odim", C.c_int32), ("hidden", C.c_int32), ("bn_dim", C.c_int32), ("n_blocks", C.c_int32),
        ("conv_in0_w", P), ("conv_in0_b", P), ("conv_in2_w", P), ("conv_in2_b", P),
        ("dw_w", PP), ("dw_b", PP), ("ln_w", PP), ("ln_b", PP), ("pw1_w", PP), ("pw1_b", PP), ("pw2_w", PP), ("pw2_b", PP),
        ("gamma", PP), ("conv_out_w", P),
    ]


class DvaeWeights(C.Structure):
    _fields_ = [
        ("encoder", TrunkWeights), ("decoder", TrunkWeights),
        ("ds0_w", P), ("ds0_b", P), ("ds1_w", P), ("ds1_b", P), ("out_conv_w", P), ("coef", P),
        ("q_in_w", P), ("q_in_b", P), ("q_out_w", P), ("q_out_b", P),
        ("levels", C.c_int32 * 4), ("G", C.c_int32), ("R", C.c_int32), ("D", C.c_int32), ("bound_first", C.c_int32),
        ("mel_window", P), ("mel_fb", P), ("twiddle", P),
    ]


I32, F, SZ = C.c_int32, C.c_float, C.c_size_t

# name -> (restype, argtypes): every symbol include/chattts_amd.h declares
SIGNATURES = {
    "ctts_last_error": (C.c_char_p, []),
    "ctts_version": (C.c_int, []),
    "ctts_gpt_create": (C.c_int, [PP, C.POINTER(GptWeights)]),
    "ctts_gpt_destroy": (None, [P]),
    "ctts_gpt_workspace_bytes": (SZ, [I32, I32]),
    "ctts_gpt_prefill": (C.c_int, [P, C.POINTER(GenState), P, P]),
    "ctts_gpt_prefill_chunk": (C.c_int, [P, C.POINTER(GenState), P, I32, I32, I32, P]),
    "ctts_gpt_decode_step": (C.c_int, [P, C.POINTER(GenState), P]),
    "ctts_gpt_graph_build": (C.c_int, [P, C.POINTER(GenState), P]),
    "ctts_gpt_graph_launch": (C.c_int, [P, I32, P]),
    "ctts_gpt_graph_build_rows": (C.c_int, [P, C.POINTER(GenState), I32, P]),
    "ctts_gpt_graph_launch_rows": (C.c_int, [P, I32, I32, P]),
    "ctts_gpt_graph_destroy": (None, [P]),
    "ctts_gpt_profile_begin": (C.c_int, [P, I32, I32, I32]),
    "ctts_gpt_profile_samples": (C.c_int, [P, C.POINTER(C.c_float), I32, C.POINTER(I32)]),
    "ctts_gpt_profile_end": (C.c_int, [P, C.POINTER(I32), C.POINTER(C.c_double)]),
    "ctts_codec_create": (C.c_int, [PP, C.POINTER(CodecWeights)]),
    "ctts_codec_destroy": (None, [P]),
    "ctts_codec_workspace_bytes": (SZ, [I32, I32]),
    "ctts_copy_bytes": (C.c_int, [P, P, SZ, P]),
    "ctts_float_to_int16": (C.c_int, [P, P, P, I32, C.c_int64, C.c_int64, I32, I32, F, P, P]),
    "ctts_dvae_decode": (C.c_int, [P, P, P, I32, I32, P, SZ, P]),
    "ctts_vocos_decode": (C.c_int, [P, P, P, I32, I32, P, SZ, P]),
    "ctts_dvae_create": (C.c_int, [PP, C.POINTER(DvaeWeights)]),
    "ctts_dvae_destroy": (None, [P]),
    "ctts_dvae_code_frames": (I32, [I32]),
    "ctts_dvae_encode_workspace_bytes": (SZ, [I32]),
    "ctts_dvae_decode_workspace_bytes": (SZ, [I32, I32]),
    "ctts_dvae_encode": (C.c_int, [P, P, I32, P, P, SZ, P]),
    "ctts_dvae_decode_codes": (C.c_int, [P, P, P, I32, I32, P, SZ, P]),
    "ctts_k_gemm": (C.c_int, [I32, P, P, P, I32, I32, I32, I32, I32, I32, I32, P, F, P, I32, P, P, I32, I32, I32, I32, I32, P]),
    "ctts_k_gemm_x3p": (C.c_int, [P, P, I32, I32, I32, I32, P, P, P, P, P, P]),
    "ctts_k_gemm_h1p": (C.c_int, [P, P, I32, I32, I32, I32, P, P, P, P, P, P]),
    "ctts_stream_create_cu_mask": (C.c_int, [I32, I32, I32, I32, P]),
    "ctts_stream_destroy": (C.c_int, [P]),
    "ctts_k_mlp_fused": (C.c_int, [P, P, P, I32, I32, P, P, P, P, I32, P]),
    "ctts_k_gemm_fast": (C.c_int, [P, I32, P, I32, I32, I32, P, F, I32, P, I32, P, I32, P, P]),
    "ctts_k_qkv_rope": (C.c_int, [P, P, I32, P, F, P, P, P, I32, P, P, I32, P, P, I32, P]),
    "ctts_k_gemm_dec": (C.c_int, [P, P, I32, I32, I32, P, P, F, I32, P, I32, P, I32, P, I32, P]),
    "ctts_k_gemm_dec32x": (C.c_int, [P, C.c_int64, P, C.c_int64, I32, I32, I32, P, P, I32, F, I32, P, I32, P, I32, P, C.c_int64, I32, P, I32, P, P, P]),
    "ctts_k_gemm_pre_x3": (C.c_int, [P, I32, P, P, I32, I32, I32, I32, I32, P, P, P, I32, P]),
    "ctts_k_dec32_last_variant": (C.c_char_p, []),
    "ctts_k_gemm_dec32": (C.c_int, [P, P, I32, I32, I32, P, P, I32, P, F, I32, P, I32, P, I32, P, I32, I32, I32, P]),
    "ctts_k_rows_prep": (C.c_int, [P, P, P, I32, P]),
    "ctts_k_rope_append": (C.c_int, [P, P, P, I32, I32, P, P, I32, P, P, I32, P]),
    "ctts_k_attention": (C.c_int, [P, P, P, I32, I32, P, I32, P, P, I32, P]),
    "ctts_k_attention_prefill": (C.c_int, [P, P, P, I32, P, I32, I32, P, I32, P]),
    "ctts_k_attention_dec": (C.c_int, [P, P, P, I32, P, P, P, I32, P, P, I32, P]),
    "ctts_k_attention_dec2": (C.c_int, [P, P, P, I32, I32, P, P, P, I32, I32, P]),
    "ctts_k_attention_cfg": (C.c_int, [I32, I32, I32]),
    "ctts_k_attention_heads_per_wg": (C.c_int, [I32]),
    "ctts_rccl_unique_id": (C.c_int, [P]),
    "ctts_rccl_comm_create": (C.c_int, [P, I32, P, I32]),
    "ctts_rccl_comm_destroy": (None, [P]),
    "ctts_broadcast_weights": (C.c_int, [P, P, I32, P, I32, P]),
    "ctts_k_attention_oproj": (C.c_int, [P, P, P, I32, P, P, P, I32, P, P, P, P, P, P]),
    "ctts_k_device_guard_probe": (C.c_int, [P, P, P, P, P]),
    "ctts_k_embed_codes": (C.c_int, [P, P, I32, P, P, I32, P]),
    "ctts_k_final_norm": (C.c_int, [P, I32, P, F, P, P, I32, P, I32, I32, P]),
    "ctts_k_sample": (C.c_int, [C.POINTER(GenState), P, P]),
    "ctts_k_sample_text": (C.c_int, [C.POINTER(GenState), P, I32, P]),
    "ctts_k_exp_draws": (C.c_int, [C.c_uint64, I32, I32, I32, I32, P, P]),
    "ctts_k_dwconv_ln": (C.c_int, [P, P, P, P, P, F, I32, P, I32, I32, P]),
    "ctts_k_layernorm": (C.c_int, [P, P, P, F, P, I32, P]),
    "ctts_k_istft": (C.c_int, [P, P, P, P, P, I32, I32, P]),
}

_lib = None


class EngineError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Loads the HIP library (once).  Raises if it is absent -- the product has no CPU path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EngineError(
                f"{LIB_PATH} not found: build it with `python -m chattts_amd.build` "
                "(hipcc --offload-arch=gfx950).  chattts_amd has no CPU/eager fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().ctts_last_error()
        raise EngineError(f"{what}: {msg.decode() if msg else 'unknown error'}")


def ptr(t) -> int:
    """Device/host pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr
