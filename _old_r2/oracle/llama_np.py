"""numpy float32 restatement of the HF LlamaModel forward the reference calls at
/root/reference/ChatTTS/model/gpt.py:419-427 (third-party `transformers`; the in-tree twin of the
arithmetic is /root/reference/examples/onnx/modeling_llama.py -- cited per function).

TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


class LlamaWeights:
    """Per-layer views of the HF state dict (SURVEY.md App. B key names)."""

    def __init__(self, sd: dict, n_heads: int = 12, eps: float = 1e-6, theta: float = 10000.0):
        g = lambda k: np.ascontiguousarray(np.asarray(sd[k], dtype=f32))
        self.n_layers = 0
        while f"layers.{self.n_layers}.input_layernorm.weight" in sd:
            self.n_layers += 1
        self.layers = []
        for i in range(self.n_layers):
            p = f"layers.{i}."
            self.layers.append(
                dict(
                    ln1=g(p + "input_layernorm.weight"),
                    wq=g(p + "self_attn.q_proj.weight"),
                    wk=g(p + "self_attn.k_proj.weight"),
                    wv=g(p + "self_attn.v_proj.weight"),
                    wo=g(p + "self_attn.o_proj.weight"),
                    ln2=g(p + "post_attention_layernorm.weight"),
                    wg=g(p + "mlp.gate_proj.weight"),
                    wu=g(p + "mlp.up_proj.weight"),
                    wd=g(p + "mlp.down_proj.weight"),
                )
            )
        self.norm = g("norm.weight")
        self.hidden = self.norm.shape[0]
        self.n_heads = n_heads
        self.head_dim = self.hidden // n_heads
        self.eps = f32(eps)
        # modeling_llama.py:131-137 -- inv_freq = 1 / theta^(2i/d)
        self.inv_freq = (1.0 / (theta ** (np.arange(0, self.head_dim, 2, dtype=np.float64) / self.head_dim))).astype(f32)


def rmsnorm(x: np.ndarray, w: np.ndarray, eps) -> np.ndarray:
    """modeling_llama.py:111-116: w * (x * rsqrt(mean(x^2) + eps)), all float32."""
    var = np.mean(x * x, axis=-1, keepdims=True, dtype=f32)
    return w * (x * (f32(1.0) / np.sqrt(var + eps)))


def rope_tables(pos: np.ndarray, inv_freq: np.ndarray):
    """modeling_llama.py:141-158: freqs = pos * inv_freq; emb = cat(freqs, freqs); cos/sin in f32."""
    freqs = pos.astype(f32)[..., None] * inv_freq  # [..., d/2]
    emb = np.concatenate([freqs, freqs], axis=-1)
    return np.cos(emb).astype(f32), np.sin(emb).astype(f32)


def apply_rope(x: np.ndarray, cos: np.ndarray, sin: np.ndarray) -> np.ndarray:
    """modeling_llama.py:239-256: x*cos + rotate_half(x)*sin, rotate_half = cat(-x2, x1).
    x: [B, nh, q, d]; cos/sin: [B, q, d]."""
    d = x.shape[-1]
    rot = np.concatenate([-x[..., d // 2:], x[..., : d // 2]], axis=-1)
    return x * cos[:, None] + rot * sin[:, None]


def silu(x: np.ndarray) -> np.ndarray:
    return x / (f32(1.0) + np.exp(-x))


class KVCache:
    """Dense per-layer K/V [B, nh, Cmax, d] float32 (the reference grows a DynamicCache by torch.cat)."""

    def __init__(self, n_layers, B, n_heads, cmax, d):
        self.k = np.zeros((n_layers, B, n_heads, cmax, d), dtype=f32)
        self.v = np.zeros((n_layers, B, n_heads, cmax, d), dtype=f32)
        self.len = 0


def forward(w: LlamaWeights, x: np.ndarray, cache: KVCache, kv_start: np.ndarray) -> np.ndarray:
    """One LlamaModel.forward over q new positions (modeling_llama.py:519-573 loop over layers,
    :375-505 attention, :259-295 MLP), left-padded rows.

    x        [B, q, H] input embeddings for slots cache.len .. cache.len+q-1
    kv_start [B] number of left-pad slots of each row (attention_mask == 0 there); keys in
             [kv_start[b], slot] are visible to query `slot` (causal + padding mask,
             gpt.py:357-366 builds the mask, HF adds it before the f32 softmax).
    Positions follow gpt.py:234-241: cumsum(mask)-1, pad slots get 1.
    Returns the final-norm hidden states [B, q, H].
    """
    B, q, H = x.shape
    nh, d = w.n_heads, w.head_dim
    c0 = cache.len
    c1 = c0 + q
    slots = np.arange(c0, c1)
    pos = slots[None, :] - kv_start[:, None]
    pos = np.where(pos < 0, 1, pos)
    cos, sin = rope_tables(pos, w.inv_freq)  # [B, q, d]
    keys = np.arange(c1)
    # mask[b, i, j]: key j visible to query slot c0+i
    vis = (keys[None, None, :] <= slots[None, :, None]) & (keys[None, None, :] >= kv_start[:, None, None])
    # a pad query row sees nothing; give it itself so softmax stays finite (its output is never read)
    selfvis = keys[None, None, :] == slots[None, :, None]
    vis = vis | (selfvis & ~vis.any(-1, keepdims=True))
    bias = np.where(vis, f32(0), f32(-np.inf)).astype(f32)[:, None]  # [B,1,q,c1]
    scale = f32(1.0 / np.sqrt(d))

    h = x.astype(f32)
    for li, L in enumerate(w.layers):
        xn = rmsnorm(h, L["ln1"], w.eps)
        qh = (xn @ L["wq"].T).reshape(B, q, nh, d).transpose(0, 2, 1, 3)
        kh = (xn @ L["wk"].T).reshape(B, q, nh, d).transpose(0, 2, 1, 3)
        vh = (xn @ L["wv"].T).reshape(B, q, nh, d).transpose(0, 2, 1, 3)
        qh = apply_rope(qh, cos, sin)
        kh = apply_rope(kh, cos, sin)
        cache.k[li, :, :, c0:c1] = kh
        cache.v[li, :, :, c0:c1] = vh
        K = cache.k[li, :, :, :c1]
        V = cache.v[li, :, :, :c1]
        s = np.matmul(qh, K.transpose(0, 1, 3, 2)) * scale + bias  # [B,nh,q,c1]
        s = s - s.max(-1, keepdims=True)
        p = np.exp(s)
        p = p / p.sum(-1, keepdims=True, dtype=f32)
        o = np.matmul(p, V).transpose(0, 2, 1, 3).reshape(B, q, H)
        h = h + o @ L["wo"].T
        xn = rmsnorm(h, L["ln2"], w.eps)
        a = silu(xn @ L["wg"].T) * (xn @ L["wu"].T)
        h = h + a @ L["wd"].T
    cache.len = c1
    return rmsnorm(h, w.norm, w.eps)
