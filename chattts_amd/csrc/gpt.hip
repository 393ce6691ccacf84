// GPT-step kernels for gfx950 other than the projections: code-embedding sum, RoPE + KV append,
// decode/prefill attention, final RMSNorm + hidden capture, and the fused sampling chain.
// Reference call sites are cited per kernel; all of them live in the per-step body of
// /root/reference/ChatTTS/model/gpt.py:396-577 or in the HF LlamaModel forward it calls.
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"
#include "decode_dev.hpp"

#ifndef CTTS_QF_DELAY
#define CTTS_QF_DELAY 8   // s_sleep units before a fused-launch attention unit requests its first keys (lets the QKV tiles' requests go first; A/B: 0 | 8 | 24)
#endif
#ifndef CTTS_QF_PRE
#define CTTS_QF_PRE 1   // blocks of old keys per wave a fused-launch attention unit requests before it waits for its QKV tiles (A/B: 0 | 1 | 2; measured 2: 1344, 1: 1422, separate launches: 1406 audio-s/s, profiles/r4f_ab_qkv_att_pre.log)
#endif
#define HID 768
#define NHEAD 12
#define HDIM 64
#define NVQ 4
#define NAUDIO 626

__device__ __forceinline__ void row_to_b_slot(const GptRowMap& rm, int m, int& b, int& slot) {
  if (rm.q_per_b == 1) { b = rm.row_map ? rm.row_map[m] : m; slot = rm.len[b] - 1; }
  else { b = m / rm.q_per_b; slot = rm.slot0 + m - b * rm.q_per_b; if (rm.row_map) b = rm.row_map[b]; }  // prefill (chunk), slot pool
}
// decode launches keep the captured grid (B rows); rows beyond the compact active count do nothing
__device__ __forceinline__ bool row_absent(const int32_t* n_active, int m) { return n_active != nullptr && m >= *n_active; }

// ------------------------------------------------------------------------------------------------
// G1  x[b] = sum_k emb_code[k][ids_buf[b, len[b]-1, k]]     (gpt.py:403-415; k-ordered f32 adds
//     like torch.stack(code_emb, 3).sum(3))
// ------------------------------------------------------------------------------------------------
// emit one residual-stream row: f32, optional bf16 copy, optional 48 partial sums of squares
// (thread t owns columns 4t..4t+3; a partial covers 16 columns = 4 consecutive threads)
__device__ __forceinline__ void emit_row(float4 s, int t, float* __restrict__ x_row, uint16_t* __restrict__ xb_row,
                                         float* __restrict__ ssq_row, float* __restrict__ xp32_at = nullptr, size_t lo_plane = 0) {
  if (x_row) *reinterpret_cast<float4*>(x_row + t * 4) = s;
  if (xp32_at) *reinterpret_cast<float4*>(xp32_at) = s;   // columns 4t..4t+3 are one lane's 16 bytes of the packed f32 order
  if (xb_row) {
    ushort4 o;
    o.x = f32_to_bf16(s.x); o.y = f32_to_bf16(s.y); o.z = f32_to_bf16(s.z); o.w = f32_to_bf16(s.w);
    *reinterpret_cast<ushort4*>(xb_row + t * 4) = o;   // columns 4t..4t+3 share one 8-column group in either layout
    if (lo_plane) {   // "f32x3" parity mode (decode32x.hip): BOTH planes in the split-fp16 format (common.hpp x3_split), lo_plane elements apart
      ushort4 h, l;
      x3_split(s.x, h.x, l.x); x3_split(s.y, h.y, l.y); x3_split(s.z, h.z, l.z); x3_split(s.w, h.w, l.w);
      *reinterpret_cast<ushort4*>(xb_row + t * 4) = h;
      *reinterpret_cast<ushort4*>(xb_row + lo_plane + t * 4) = l;
    }
  }
  if (ssq_row) {
    float q = (s.x * s.x + s.y * s.y) + (s.z * s.z + s.w * s.w);
    q += __shfl_xor(q, 1, 64);
    q += __shfl_xor(q, 2, 64);
    if ((t & 3) == 0) ssq_row[t >> 2] = q;
  }
}

// The first kernel of a decode step also writes the step's row descriptors (kernels.hpp RowDesc): it walks
// row_map -> len -> kv_start -> finish once, the 20 QKV epilogues and 20 attention launches behind it start from desc[m].
// ... and, for the QKV tiles of the fused QKV + attention launch, the RoPE factors of the row's position as one 256-byte row
// (cos[32] | sin[32]): their epilogue then needs no row -> position -> table chain
__device__ __forceinline__ void write_rope_cs(const StepPrep& sp, int m, int b, int slot, int t) {
  if (sp.rope_cs == nullptr || t >= 64) return;
  const int ks = sp.kv_start[b];
  const int pos = slot - ks < 0 ? 1 : slot - ks;   // as write_desc
  sp.rope_cs[(size_t)m * 64 + t] = t < 32 ? sp.cos_t[pos * 32 + t] : sp.sin_t[pos * 32 + t - 32];
}
__device__ __forceinline__ void write_desc(const StepPrep& sp, int m, int b, int slot) {
  RowDesc d;
  const int ks = sp.kv_start[b];
  d.b = (sp.finish != nullptr && sp.finish[b]) ? -1 : b;
  d.slot = slot;
  d.pos = slot - ks < 0 ? 1 : slot - ks;   // pad slots get position 1 (gpt.py:234-241)
  d.jlo = ks > slot ? slot : ks;
  sp.desc[m] = d;
}
// Device-side compaction: which utterance is compact row m, and how many rows are live.  Every wave of the workgroup
// evaluates this itself (the finish bytes are one load for B <= 64; no barrier).  Returns -1 when row m does not exist.
// `order` (optional): the utterances are visited in that order instead of ascending slot (host: descending context, so that the
// attention grid's first workgroups are its longest units); the return value is the utterance SLOT either way.
__device__ __forceinline__ int nth_unfinished(const uint8_t* __restrict__ finish, const int32_t* __restrict__ order, int B, int m,
                                              int& total) {
  const int lane = threadIdx.x & 63;
  int cnt = 0, found = -1;
  for (int base = 0; base < B; base += 64) {
    const int idx = base + lane;
    const int slot = idx < B ? (order ? order[idx] : idx) : 0;
    const bool alive = idx < B && finish[slot] == 0;
    const unsigned long long mask = __ballot(alive);
    const int c = __popcll(mask);
    if (found < 0 && m < cnt + c) {
      const int r = m - cnt;   // the r-th set bit of mask
      const bool mine = alive && __popcll(mask & ((1ull << lane) - 1ull)) == r;
      const unsigned long long pick = __ballot(mine);
      found = __shfl(slot, (int)__ffsll((long long)pick) - 1, 64);
    }
    cnt += c;
  }
  total = cnt;
  return found;
}

// The step's first kernel also zeroes the arrival words of the fused QKV + attention launches of the 20 layers behind it (plain stores:
// a kernel boundary lies between them and the first atomic).
__device__ __forceinline__ void step_zero(const StepPrep& sp) {
  if (sp.zero_p == nullptr) return;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < sp.zero_n; i += gridDim.x * blockDim.x) sp.zero_p[(size_t)i * sp.zero_stride] = 0;
}

__device__ __forceinline__ uint16_t* xb_row_ptr(uint16_t* xb, int m, int t, int packed) {
  // emit_row adds 4t itself: hand it a base such that base + 4t is where columns 4t..4t+3 of row m live
  if (!xb) return nullptr;
  return packed ? xb + pk_off(m, 4 * t, HID / 32) - 4 * t : xb + (size_t)m * HID;
}

__global__ __launch_bounds__(192) void embed_codes_k(const float* __restrict__ emb, const int64_t* __restrict__ ids_buf,
                                                     int tcap, const int32_t* __restrict__ len, float* __restrict__ x,
                                                     uint16_t* __restrict__ xb, float* __restrict__ ssq,
                                                     const int32_t* __restrict__ row_map, const int32_t* __restrict__ n_active,
                                                     StepPrep sp) {
  CTTS_PROBE_RETURN();
  step_zero(sp);
  const int m = blockIdx.x, t = threadIdx.x;
  int b;
  if (sp.row_map_out != nullptr) {   // device-side compaction: this step's row order comes from the finish flags
    int total;
    b = nth_unfinished(sp.finish, sp.order, gridDim.x, m, total);
    if (m == 0 && t == 0) *sp.n_active_out = total;
    if (b < 0) {   // row m does not exist this step: say so in its descriptor (the attention kernel reads nothing else)
      if (sp.desc != nullptr && t == 0) sp.desc[m] = RowDesc{-1, 0, 0, 0};
      return;
    }
    if (t == 0) sp.row_map_out[m] = b;
  } else {
    if (row_absent(n_active, m)) return;
    b = row_map ? row_map[m] : m;
  }
  const int slot = len[b] - 1;
  if (sp.desc != nullptr && t == 0) write_desc(sp, m, b, slot);
  write_rope_cs(sp, m, b, slot, t);
  const int64_t* tok = ids_buf + ((size_t)b * tcap + slot) * NVQ;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < NVQ; ++k) {
    int id = (int)tok[k];
    id = min(max(id, 0), NAUDIO - 1);
    const float4 v = *reinterpret_cast<const float4*>(emb + ((size_t)k * NAUDIO + id) * HID + t * 4);
    if (k == 0) s = v; else { s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
  }
  emit_row(s, t, x + (size_t)m * HID, xb_row_ptr(xb, m, t, sp.xb_packed), ssq ? ssq + (size_t)m * SSQ_PARTS : nullptr,
           sp.xp32 ? sp.xp32 + pk32_off(m, 4 * t, HID / 16) : nullptr, sp.xb_lo_plane);
}

static StepPrep prep_or_none(const StepPrep* p) {
  StepPrep sp{nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1};
  if (p) sp = *p;
  return sp;
}

hipError_t launch_embed_codes(const float* emb_code, const int64_t* ids_buf, int tcap, const int32_t* len, float* x, uint16_t* xb,
                              float* ssq, int B, const int32_t* row_map, const int32_t* n_active, hipStream_t st, const StepPrep* prep) {
  CTTS_LAUNCH(embed_codes_k, dim3(B), dim3(192), st, emb_code, ids_buf, tcap, len, x, xb, ssq, row_map, n_active, prep_or_none(prep));
  return hipGetLastError();
}

__global__ __launch_bounds__(192) void rows_prep_k(const float* __restrict__ x, uint16_t* __restrict__ xb, float* __restrict__ ssq) {
  const int m = blockIdx.x, t = threadIdx.x;
  const float4 s = *reinterpret_cast<const float4*>(x + (size_t)m * HID + t * 4);
  emit_row(s, t, nullptr, xb + (size_t)m * HID, ssq + (size_t)m * SSQ_PARTS);
}

__global__ __launch_bounds__(192) void prefill_prep32_k(const float* __restrict__ x, float* __restrict__ xp32, RowDesc* __restrict__ desc,
                                                        int q_per_b, int slot0, const int32_t* __restrict__ kv_start,
                                                        const int32_t* __restrict__ row_map) {
  const int m = blockIdx.x, t = threadIdx.x;
  const float4 v = *reinterpret_cast<const float4*>(x + (size_t)m * HID + t * 4);
  *reinterpret_cast<float4*>(xp32 + pk32_off(m, 4 * t, HID / 16)) = v;   // columns 4t..4t+3 are one lane's 16 bytes of the packed order
  if (t == 0) {
    const int bq = m / q_per_b, slot = slot0 + m - bq * q_per_b;
    const int b = row_map ? row_map[bq] : bq;
    const int ks = kv_start[b];
    desc[m] = RowDesc{b, slot, slot - ks < 0 ? 1 : slot - ks, ks > slot ? slot : ks};   // as write_desc / rope_append_k (pad slots: position 1)
  }
}
hipError_t launch_prefill_prep32(const float* x, float* xp32, RowDesc* desc, int q_per_b, int slot0, const int32_t* kv_start,
                                 const int32_t* row_map, int M, hipStream_t st) {
  CTTS_LAUNCH(prefill_prep32_k, dim3(M), dim3(192), st, x, xp32, desc, q_per_b, slot0, kv_start, row_map);
  return hipGetLastError();
}

// Prompt pass over the VALID prompt tokens only (round 6, "f32x3" mode): the caller's batch is left-padded to its longest prompt
// (tokenizer.py:73-110) and the reference pushes every pad row through every layer (their outputs are never consumed: attention masks
// them as keys, gpt.py:234-241).  One workgroup per utterance: its first compact row = the number of valid tokens of the utterances before
// it; rows [first, first + T - kv_start[b]) take the embeddings of its valid slots, their descriptors (utterance, slot, RoPE position,
// first visible key) drive rope_append_k and attention_k, last_row[b] is the row the final norm / heads / sampler read.
__global__ __launch_bounds__(192) void prefill_compact_k(const float* __restrict__ emb, float* __restrict__ x, RowDesc* __restrict__ desc,
                                                         int32_t* __restrict__ last_row, int T, const int32_t* __restrict__ kv_start) {
  __shared__ int part[3];
  const int b = blockIdx.x, t = threadIdx.x;
  int acc = 0;
  for (int i = t; i < b; i += 192) acc += T - kv_start[i];
  acc = (int)wave_sum((float)acc);     // (exact: < 2^24 rows)
  if ((t & 63) == 0) part[t >> 6] = acc;
  __syncthreads();
  const int first = part[0] + part[1] + part[2];
  const int ks = kv_start[b], n = T - ks;
  for (int j = 0; j < n; ++j) {
    const float4 v = *reinterpret_cast<const float4*>(emb + ((size_t)b * T + ks + j) * HID + t * 4);
    *reinterpret_cast<float4*>(x + (size_t)(first + j) * HID + t * 4) = v;
    if (t == 0) desc[first + j] = RowDesc{b, ks + j, j, ks};
  }
  if (t == 0) last_row[b] = first + n - 1;
}
hipError_t launch_prefill_compact(const float* emb, float* x, RowDesc* desc, int32_t* last_row, int B, int T, const int32_t* kv_start,
                                  hipStream_t st) {
  CTTS_LAUNCH(prefill_compact_k, dim3(B), dim3(192), st, emb, x, desc, last_row, T, kv_start);
  return hipGetLastError();
}

hipError_t launch_rows_prep(const float* x32, uint16_t* xb, float* ssq, int M, hipStream_t st) {
  CTTS_LAUNCH(rows_prep_k, dim3(M), dim3(192), st, x32, xb, ssq);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// G4  RoPE (rotate-half, HF modeling_llama apply_rotary_pos_emb; in-tree twin
//     /root/reference/examples/onnx/modeling_llama.py:239-256) on q (in place) and k, then append k,v
//     to the cache at `slot`.  Position = slot - kv_start[b] (gpt.py:234-241; pad slots get 1).
//     cos/sin come from a host table built with the reference's own f32 ops.
// ------------------------------------------------------------------------------------------------
template <typename KT> __device__ __forceinline__ KT to_kt(float v);
template <> __device__ __forceinline__ float to_kt<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t to_kt<bf16_t>(float v) { return f32_to_bf16(v); }

template <typename KT>
__global__ __launch_bounds__(384) void rope_append_k(float* __restrict__ qkv, KT* __restrict__ kc, KT* __restrict__ vc, int cmax,
                                                     const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                     GptRowMap rm) {
  const int m = blockIdx.x, t = threadIdx.x;  // t: head = t / 32, pair d = t % 32
  if (rm.q_per_b == 1 && row_absent(rm.n_active, m)) return;
  int b, slot, pos;
  if (rm.q_per_b != 1 && rm.desc != nullptr) {   // compact prompt rows (prefill_compact_k): the row's descriptor says where it belongs
    const RowDesc d = rm.desc[m];
    b = d.b; slot = d.slot; pos = d.pos;
  } else {
    row_to_b_slot(rm, m, b, slot);
    pos = slot - rm.kv_start[b];
    if (pos < 0) pos = 1;
  }
  const int h = t >> 5, d = t & 31;
  const float c = cos_t[pos * 32 + d], s = sin_t[pos * 32 + d];
  float* row = qkv + (size_t)m * (3 * HID);
  // q
  {
    float* q = row + h * HDIM;
    const float x1 = q[d], x2 = q[d + 32];
    q[d] = rope_lo(x1, x2, c, s);
    q[d + 32] = rope_hi(x1, x2, c, s);
  }
  const size_t base = (((size_t)b * NHEAD + h) * cmax + slot) * HDIM;
  {
    const float* k = row + HID + h * HDIM;
    const float x1 = k[d], x2 = k[d + 32];
    kc[base + d] = to_kt<KT>(rope_lo(x1, x2, c, s));
    kc[base + d + 32] = to_kt<KT>(rope_hi(x1, x2, c, s));
  }
  {
    const float* v = row + 2 * HID + h * HDIM;
    vc[base + d] = to_kt<KT>(v[d]);
    vc[base + d + 32] = to_kt<KT>(v[d + 32]);
  }
}

hipError_t launch_rope_append(float* qkv, void* kcache, void* vcache, int kv_wt, int cmax, const float* cos_tab,
                              const float* sin_tab, GptRowMap rm, int M, hipStream_t st) {
  if (kv_wt == WT_BF16)
    CTTS_LAUNCH((rope_append_k<bf16_t>), dim3(M), dim3(384), st, qkv, (bf16_t*)kcache, (bf16_t*)vcache, cmax, cos_tab,
                       sin_tab, rm);
  else
    CTTS_LAUNCH((rope_append_k<float>), dim3(M), dim3(384), st, qkv, (float*)kcache, (float*)vcache, cmax, cos_tab,
                       sin_tab, rm);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// G5  attention for one query row and one head: softmax_f32(q.K^T / 8 + mask) V with the causal +
//     left-pad mask of the reference (keys in [kv_start[b], slot]).  KV is streamed from HBM with
//     fully coalesced 16-byte lane loads: LPK lanes share one key row (128 B bf16 / 256 B f32), a wave
//     load instruction covers 64/LPK consecutive keys = 1 KiB contiguous, 8 K-loads + 8 V-loads are in
//     flight per wave per block; the dot product is finished with log2(LPK) xor-shuffles, softmax is
//     online (running max / sum per wave) and the NW waves of a workgroup split the key blocks and
//     merge through LDS.
// ------------------------------------------------------------------------------------------------
template <typename KT> struct KTraits;
template <> struct KTraits<float>  { static constexpr int DPL = 4; };  // dims per lane (16 B)
template <> struct KTraits<bf16_t> { static constexpr int DPL = 8; };

template <typename KT, int DPL> __device__ __forceinline__ void unpack16(const u128& r, float* f);
template <> __device__ __forceinline__ void unpack16<float, 4>(const u128& r, float* f) {
  f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y); f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t, 8>(const u128& r, float* f) {
  f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
  f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
  f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
  f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
}

// packed attention output: the A operand of o_proj in fragment order (bf16: decode.hip, f32: decode32.hip)
template <typename OT> __device__ __forceinline__ size_t pko_off(int m, int c);
template <> __device__ __forceinline__ size_t pko_off<bf16_t>(int m, int c) { return pk_off(m, c, HID / 32); }
template <> __device__ __forceinline__ size_t pko_off<float>(int m, int c) { return pk32_off(m, c, HID / 16); }
// split-bf16 parity mode (decode32x.hip): the attention output as hi / lo bf16 planes in decode.hip's fragment order
struct x3p_t { uint16_t v; };
template <> __device__ __forceinline__ size_t pko_off<x3p_t>(int m, int c) { return pk_off(m, c, HID / 32); }
template <typename OT> __device__ __forceinline__ void store_out(OT* p, float v);
template <> __device__ __forceinline__ void store_out<x3p_t>(x3p_t* p, float v) { p->v = f32_to_bf16(v); }   // (never the packed path's store)
// packed store of one output element: row m, column c
template <typename OT> __device__ __forceinline__ void store_pko(OT* out, int m, int c, float v, size_t);
template <> __device__ __forceinline__ void store_pko<x3p_t>(x3p_t* out, int m, int c, float v, size_t plane) {
  uint16_t* p = reinterpret_cast<uint16_t*>(out) + pk_off(m, c, HID / 32);
  uint16_t h, l;
  x3_split(v, h, l);
  p[0] = h;
  p[plane] = l;
}
template <> __device__ __forceinline__ void store_out<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void store_out<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }
template <> __device__ __forceinline__ void store_pko<float>(float* out, int m, int c, float v, size_t) { out[pko_off<float>(m, c)] = v; }
template <> __device__ __forceinline__ void store_pko<bf16_t>(bf16_t* out, int m, int c, float v, size_t) { out[pko_off<bf16_t>(m, c)] = f32_to_bf16(v); }

// Lane-group reductions of the attention kernels without the LDS crossbar (round 5).  `__shfl_xor` lowers to ds_bpermute_b32 -- an LDS
// round trip per step, 15 of them in a dependent chain per 32-key block, which a persistent workgroup with ONE wave per SIMD cannot
// hide.  The DPP forms are plain VALU operands.  Bit-identical to the xor butterfly: the partner of step o = 1, 2 is the same lane
// (quad_perm); for o = 4 (and 8) all lanes of a quad (half row) already hold the same value, so the mirrored lane's value IS lane ^ o's;
// IEEE addition commutes.
#ifndef CTTS_ATT_DPP
#define CTTS_ATT_DPP 1   // A/B builds: python -m chattts_amd.build --variant nodpp -DCTTS_ATT_DPP=0
#endif
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int LPK>
__device__ __forceinline__ float group_sum(float d) {   // sum over the LPK (8 | 16) lanes that share a key, every lane gets it
#if CTTS_ATT_DPP
  d += dpp_f<0xB1>(d);    // quad_perm [1,0,3,2] = lane ^ 1
  d += dpp_f<0x4E>(d);    // quad_perm [2,3,0,1] = lane ^ 2
  d += dpp_f<0x141>(d);   // row_half_mirror: the other quad of the 8 lanes
  if (LPK == 16) d += dpp_f<0x140>(d);   // row_mirror: the other half of the row
#else
#pragma unroll
  for (int o = 1; o < LPK; o <<= 1) d += __shfl_xor(d, o, 64);
#endif
  return d;
}
// x + its partners at lane ^ o for o = LPK, 2 LPK, ... 32 (the merge of a wave's key groups), in the xor butterfly's order and with its bits:
// o = 8 is a rotation by 8 inside the 16-lane row (DPP row_ror:8), o = 16 / 32 swap rows / halves between two copies of the register
// (v_permlane16_swap_b32 / v_permlane32_swap_b32, gfx950) -- no LDS round trip.  9 values x 3 steps at the end of every unit.
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
template <int LPK>
__device__ __forceinline__ float groups_sum(float x, const int lane) {
#if CTTS_ATT_DPP
  if (LPK == 8) x += dpp_f<0x128>(x);   // row_ror:8 = lane ^ 8 within the row
  {
    const unsigned xi = __float_as_uint(x);
    const u32x2_t r = __builtin_amdgcn_permlane16_swap(xi, xi, false, false);   // .x = [r0 r0 r2 r2], .y = [r1 r1 r3 r3]
    x += __uint_as_float((lane & 16) ? r.x : r.y);
  }
  {
    const unsigned xi = __float_as_uint(x);
    const u32x2_t r = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);   // .x = [lo lo], .y = [hi hi]
    x += __uint_as_float((lane & 32) ? r.x : r.y);
  }
  return x;
#else
#pragma unroll
  for (int o = LPK; o < 64; o <<= 1) x += __shfl_xor(x, o, 64);
  return x;
#endif
}
template <int LPK>
__device__ __forceinline__ float groups_max(float v) {  // max over the key groups of the wave (v is uniform within a group); exact in any order
#if CTTS_ATT_DPP
  if (LPK == 8) v = fmaxf(v, dpp_f<0x140>(v));                                                              // row_mirror: the row's other group
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x142, 0xa, 0xf, false)));  // row_bcast15 -> rows 1, 3
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x143, 0xc, 0xf, false)));  // row_bcast31 -> rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
#else
#pragma unroll
  for (int o = LPK; o < 64; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
#endif
}

// SPLIT (decode, perf mode): "remainder splitting".  A (utterance, head) unit streams ctx * 256 bytes of KV and a CU ingests
// only ~25 GB/s of that, so the kernel lasts as long as the CU with the most units: U = 12 * n_active units on C CUs cost
// ceil(U / C) units of time although the average CU holds U / C (540 units on 256 CUs: 3 instead of 2.1; below 256 units whole
// CUs idle).  Here the first floor(U / C) * C units stay whole (one workgroup each, merged through LDS as before) and each
// of the R = U mod C remainder units is cut into S = min(8, C / R) key ranges handled by S workgroups -- R * S <= C small
// workgroups, about one per CU -- that meet through memory: every piece writes its partial (o[64], m, l) write-through, draws
// a ticket from the unit's counter, and the LAST arriver reads the S partials (L1-bypassing loads), merges them and writes the
// output (MI355X guide, Guideline 16 hand-off in its counter form: sc1 payload -> vmcnt(0) -> relaxed agent atomic; placement
// independent).  grid = 12 * rows + C workgroups.  MEASURED (profiles/r2e_*): correct (tests/test_gpu_kernels.py), but on the C3
// bench the launch gets 0.5 us SLOWER (9.3 -> 9.8 us) -- the pieces' hand-off latency is not hidden behind the whole units --
// and the split geometry depends on the live-row count, which costs bf16 mode its batch invariance.  Kept behind
// CTTS_ATT_SPLIT=1 (default off) as a recorded negative result.
//
// NBUF (round 3): KV blocks in flight per wave.  2 = the block being consumed + the next one; 3 = one more.  A wave's share of a
// 500-key context is 4 blocks, and with fewer units than CUs (the last quarter of a C3 batch, every step of C2) nothing else on
// the CU covers the round trip between them (profiles/r3y_attn_phase_probe.log: 240 units at 430 keys spend 3.0 us behind the
// first block).  The order in which blocks are consumed is untouched, so the result is the same bit for bit in both modes.
//
// OPJ (round 4; decode, perf mode): o_proj + residual folded into this launch -- see the block behind the merge.
//
// QF (round 4; decode, perf mode): the unit runs INSIDE the fused QKV + attention launch (qkv_attention_k below).  Its q row and the
// newest key / value are written by QKV tiles of the same launch, so it (1) requests its first two blocks of OLD keys (everything
// below the step's slot: written by earlier launches), (2) waits for its head's arrival word, (3) reads q and the newest K / V row
// with sc1 loads and goes on as usual; the newest key is consumed last, by the last wave, as a block of its own.
template <typename KT, int NW, typename OT, bool PKO = false, bool SPLIT = false, bool PF = false, int NBUF = 2, bool OPJ = false, bool QF = false, int HPW = 1>
__device__ __forceinline__ void attention_body(const float* __restrict__ qkv, const KT* __restrict__ kc, const KT* __restrict__ vc, int cmax,
                                               OT* __restrict__ out, const GptRowMap& rm, const int h_in, const int m_in, const int wg_linear) {
  constexpr int DPL = KTraits<KT>::DPL;
  constexpr int LPK = HDIM / DPL;   // lanes per key: 8 (bf16) / 16 (f32)
  constexpr int KPI = 64 / LPK;     // keys per load instruction: 8 / 4
  constexpr int NI = 4;             // load instructions per block and operand (K and V): 8 loads per block in flight,
                                    // and the NEXT block's 8 are issued before the current block is consumed
  constexpr int KB = KPI * NI;      // keys per wave-block: 32 / 16
#ifndef CTTS_KV_NT
#define CTTS_KV_NT 1              // A/B builds: python -m chattts_amd.build --variant kvplain -DCTTS_KV_NT=0 (profiles/r3ap_kv_nt_ab.log)
#endif
  constexpr bool KV_NT = NW > 1 && CTTS_KV_NT;    // decode: every KV byte is read once per step by exactly one workgroup -> non-temporal
  constexpr int NU = QF ? 2 : HPW;   // units per workgroup: the fused launch's workgroups are 8 waves = heads 2p and 2p + 1 of one utterance;
                                     // HPW > 1 (attention_hpw_k): HPW consecutive heads of one utterance, NW waves each (same length: uniform barriers)
  __shared__ float sm_m_[NU][NW], sm_l_[NU][NW], sm_acc_[NU][NW][HDIM];
  __shared__ __attribute__((aligned(16))) bf16_t sm_on[OPJ ? HDIM : 8];   // OPJ: the unit's normalised output, bf16
  __shared__ __attribute__((aligned(16))) float sm_p[OPJ ? HID : 4];      // OPJ: its 768-wide o_proj partial
  __shared__ int sm_last;
#ifdef CTTS_PF_BUILD
  if (PF && (threadIdx.x >> 6) == NW) {   // fifth wave: this layer's gate/up weights towards this XCD's L2 (common.hpp), then gone
    prefetch_weight_tiles(rm.pf, threadIdx.x & 63, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    return;
  }
#endif

  const int unit = (QF || HPW > 1) ? (int)(threadIdx.x / (64 * NW)) : 0;   // which of the workgroup's units this wave belongs to
  int h = h_in + unit, m = m_in;
  const int tid = (QF || HPW > 1) ? (int)(threadIdx.x % (64 * NW)) : (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;   // thread / wave index WITHIN the unit
  float (&sm_m)[NW] = sm_m_[unit];
  float (&sm_l)[NW] = sm_l_[unit];
  float (&sm_acc)[NW][HDIM] = sm_acc_[unit];
  const int kg = lane / LPK, dl = lane % LPK;
  long long* dbg = (PKO && rm.dbg) ? rm.dbg + (size_t)wg_linear * 8 : nullptr;
#define ASTAMP(i) do { if (dbg && tid == 0) dbg[i] = wall_clock64(); } while (0)
  ASTAMP(0);
  int n_piece = 1, piece = 0, su = 0;   // SPLIT: pieces of this unit, this workgroup's piece, index among the split units
  if (SPLIT) {
    const int nact = rm.n_active ? *rm.n_active : (int)(gridDim.x - rm.sp_cus) / NHEAD;
    const int U = nact * NHEAD, C = rm.sp_cus;
    const int k = U / C, R = U - k * C;
    const int full = R == 0 ? U : k * C;                 // units [0, full) stay whole
    const int S = R == 0 ? 1 : min(ATT_SPLIT_MAX, C / R);
    const int w = blockIdx.x;
    int unit = w;
    if (w >= full) {
      const int idx = w - full;
      su = idx / S;
      unit = full + su;
      piece = idx - su * S;
      n_piece = S;
      if (unit >= U) return;
    }
    m = unit / NHEAD;
    h = unit - m * NHEAD;
  } else if (rm.q_per_b == 1 && !rm.desc_covers_all && row_absent(rm.n_active, m)) return;
  int b, slot, jlo;
  if (rm.desc != nullptr) {   // decode: one 16-byte load instead of the row_map -> len -> kv_start / finish chain
    const RowDesc d = rm.desc[m];
    if (d.b < 0) return;      // finished since the last compaction: nothing downstream is ever read
    b = d.b; slot = d.slot; jlo = d.jlo;
  } else {
    row_to_b_slot(rm, m, b, slot);
    if (rm.finish != nullptr && rm.finish[b]) return;
    jlo = rm.kv_start[b];
    if (jlo > slot) jlo = slot;  // pad query row: sees only itself (its output is never consumed)
  }

  ASTAMP(1);   // row descriptor known
  if (dbg && tid == 0) { dbg[6] = slot + 1 - jlo; }   // visible keys of this unit
  // q is REQUESTED after the first KV block (below) and stays unscaled: the 1/sqrt(64) = 2^-3 is applied to the finished dot product,
  // which is the same number bit for bit (a power-of-two scale commutes with every rounding of the fmaf chain and of the shuffle
  // adds).  Reading q first put one L2 miss (0.7 us median, and several us for a workgroup whose load queued behind a co-resident
  // workgroup's 64 KB of KV requests) in front of the whole KV stream (profiles/r3x_attn_phase_probe.log).
  float q[DPL];
  const float* qp = qkv + (size_t)m * (3 * HID) + h * HDIM + dl * DPL;
  const KT* kbase = kc + ((size_t)b * NHEAD + h) * cmax * HDIM + dl * DPL;
  const KT* vbase = vc + ((size_t)b * NHEAD + h) * cmax * HDIM + dl * DPL;

  float mrun = -INFINITY, lrun = 0.f, acc[DPL];
#pragma unroll
  for (int i = 0; i < DPL; ++i) acc[i] = 0.f;

  // this workgroup's key range [rbeg, rend): all visible keys, or one of n_piece KPI-aligned pieces of them
  int rbeg = jlo, rend = QF ? slot : slot + 1;   // QF: the newest key (this step's) is handled behind the loop
  if (SPLIT && n_piece > 1) {
    const int pper = ((slot + 1 - jlo + n_piece - 1) / n_piece + KPI - 1) / KPI * KPI;
    rbeg = jlo + piece * pper;
    rend = min(rbeg + pper, slot + 1);
  }
  // each wave owns one contiguous, KPI-aligned share of those keys (balanced: a context of n
  // keys costs every wave ceil(n / NW / KB) blocks instead of giving wave 0 the remainder blocks)
  const int nkeys = max(rend - rbeg, 0);
  const int per = ((nkeys + NW - 1) / NW + KPI - 1) / KPI * KPI;
  const int jbeg = rbeg + wave * per;
  const int jend = min(jbeg + per, rend);  // exclusive

  u128 kA[NI], vA[NI], kB[NI], vB[NI], kC[NBUF > 2 ? NI : 1], vC[NBUF > 2 ? NI : 1];
  // Loads are UNCONDITIONAL with the key index clamped into the wave's range: a per-lane `if (j < jend) load`
  // makes hipcc branch around every load and wait vmcnt(0) in between (one memory round trip per load).
  // Clamped lanes re-read the last key (an L1 hit) and are masked where they are consumed.
  const int jlast = max(jend - 1, jlo);
  // QF: buffer loads with 32-bit byte offsets (one VGPR per address instead of two; 16 loads are in flight across the wait, and two
  // 8-wave workgroups per CU need <= 128 VGPRs); aux 2 = the same non-temporal hint
  const auto krs = __builtin_amdgcn_make_buffer_rsrc((void*)kc, 0, QF ? rm.qf_kv_bytes : 0, 0x00020000);
  const auto vrs = __builtin_amdgcn_make_buffer_rsrc((void*)vc, 0, QF ? rm.qf_kv_bytes : 0, 0x00020000);
  const unsigned ubase = (((unsigned)b * NHEAD + (unsigned)h) * (unsigned)cmax) * (HDIM * (unsigned)sizeof(KT)) + (unsigned)dl * 16u;
  auto load_blk = [&](u128* kr, u128* vr, int j0) {
    if constexpr (QF) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(krs, ubase + (unsigned)min(j0 + i * KPI + kg, jlast) * (HDIM * (unsigned)sizeof(KT)), 0, 2);
        kr[i].x = t.x; kr[i].y = t.y; kr[i].z = t.z; kr[i].w = t.w;
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(vrs, ubase + (unsigned)min(j0 + i * KPI + kg, jlast) * (HDIM * (unsigned)sizeof(KT)), 0, 2);
        vr[i].x = t.x; vr[i].y = t.y; vr[i].z = t.z; vr[i].w = t.w;
      }
    } else {
#pragma unroll
    for (int i = 0; i < NI; ++i) kr[i] = KV_NT ? load16_nt(kbase + (size_t)min(j0 + i * KPI + kg, jlast) * HDIM)
                                               : load16(kbase + (size_t)min(j0 + i * KPI + kg, jlast) * HDIM);
#pragma unroll
    for (int i = 0; i < NI; ++i) vr[i] = KV_NT ? load16_nt(vbase + (size_t)min(j0 + i * KPI + kg, jlast) * HDIM)
                                               : load16(vbase + (size_t)min(j0 + i * KPI + kg, jlast) * HDIM);
    }
  };
  auto use_blk_e = [&](const u128* kr, const u128* vr, int j0, const int jend_) {
    float s[NI];
    bool ok[NI];
    float bmax = -INFINITY;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      ok[i] = (j0 + i * KPI + kg) < jend_;
      float kf[DPL];
      float d = 0.f;
      unpack16<KT, DPL>(kr[i], kf);
#pragma unroll
      for (int e = 0; e < DPL; ++e) d = fmaf(q[e], kf[e], d);
      d = group_sum<LPK>(d);
      d *= 0.125f;   // 1/sqrt(64), exact
      s[i] = ok[i] ? d : -INFINITY;
      bmax = fmaxf(bmax, s[i]);
    }
    bmax = groups_max<LPK>(bmax);
    // bmax is finite: key j0 (i = 0, kg = 0) is always < jend for a block that is consumed
    const float mnew = fmaxf(mrun, bmax);
    const float alpha = expf(mrun - mnew);  // exp(-inf) = 0 on the first block
    lrun *= alpha;
#pragma unroll
    for (int e = 0; e < DPL; ++e) acc[e] *= alpha;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float p = expf(s[i] - mnew);  // masked lanes: s = -inf -> p = 0 (their V is a finite, clamped re-read)
      lrun += p;
      float vf[DPL];
      unpack16<KT, DPL>(vr[i], vf);
#pragma unroll
      for (int e = 0; e < DPL; ++e) acc[e] = fmaf(p, vf[e], acc[e]);
    }
    mrun = mnew;
  };
  auto use_blk = [&](const u128* kr, const u128* vr, int j0) { use_blk_e(kr, vr, j0, jend); };

  int j = jbeg;
  if constexpr (QF) {
#if CTTS_QF_DELAY > 0
    __builtin_amdgcn_s_sleep(CTTS_QF_DELAY);   // A/B: let the QKV tiles' requests go first
#endif
#if CTTS_QF_PRE >= 1
    load_blk(kA, vA, j);        // old keys on their way before the wait
#endif
#if CTTS_QF_PRE == 2
    load_blk(kB, vB, j + KB);
#endif
    if (wave == 0) {
      // arrivals of this head and launch: its 12 QKV tiles (4 q + 4 k + 4 v column tiles; one arrival per tile, decode_dev.hpp).  The
      // polls queue behind this wave's own KV requests (vmcnt is in order): nothing is lost, the keys are needed anyway.
      // BOUNDED spin: a dependency that never arrives must not hang the GPU.
      const int32_t* fw = rm.qf_flag + h * HO_STRIDE;
      const long long t0 = wall_clock64();
      while (__hip_atomic_load(fw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 12) {
        if (wall_clock64() - t0 > 2000000ll) __builtin_trap();   // 20 ms: abort the launch (the host sees a HIP error), never compute on stale q
        __builtin_amdgcn_s_sleep(4);
      }
    }
    __syncthreads();   // both units: the pair is done waiting when both heads have arrived
#if CTTS_QF_PRE == 0
    load_blk(kA, vA, j);
#endif
    const auto qrs = __builtin_amdgcn_make_buffer_rsrc((void*)qkv, 0, rm.qf_rows * 3 * HID * 4, 0x00020000);
    constexpr int NV = DPL / 4;
    u32x4_t qv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) qv[i] = __builtin_amdgcn_raw_buffer_load_b128(qrs, (m * (3 * HID) + h * HDIM + dl * DPL + 4 * i) * 4, 0, 16);
    u128 kN[1], vN[1];
    if (wave == NW - 1) {   // the newest key: every lane group reads the same row (one 128-byte line), sc1
      const u32x4_t tk = __builtin_amdgcn_raw_buffer_load_b128(krs, ubase + (unsigned)slot * (HDIM * (unsigned)sizeof(KT)), 0, 16);
      const u32x4_t tv = __builtin_amdgcn_raw_buffer_load_b128(vrs, ubase + (unsigned)slot * (HDIM * (unsigned)sizeof(KT)), 0, 16);
      kN[0].x = tk.x; kN[0].y = tk.y; kN[0].z = tk.z; kN[0].w = tk.w;
      vN[0].x = tv.x; vN[0].y = tv.y; vN[0].z = tv.z; vN[0].w = tv.w;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      q[4 * i] = __uint_as_float(qv[i].x); q[4 * i + 1] = __uint_as_float(qv[i].y);
      q[4 * i + 2] = __uint_as_float(qv[i].z); q[4 * i + 3] = __uint_as_float(qv[i].w);
    }
    // the newest key FIRST (its row arrives together with q; consumed before the old blocks, its registers are free again)
    if (wave == NW - 1) {   // the newest key: every lane group holds the same row, lane group 0 accounts for it
      float kf[DPL], vf[DPL];
      float d = 0.f;
      unpack16<KT, DPL>(kN[0], kf);
#pragma unroll
      for (int e = 0; e < DPL; ++e) d = fmaf(q[e], kf[e], d);
#pragma unroll
      for (int o = 1; o < LPK; o <<= 1) d += __shfl_xor(d, o, 64);
      d *= 0.125f;
      const float mnew = fmaxf(mrun, d);
      const float alpha = expf(mrun - mnew);   // exp(-inf) = 0 when this wave had no old keys
      const float p = kg == 0 ? expf(d - mnew) : 0.f;
      lrun = lrun * alpha + p;
      unpack16<KT, DPL>(vN[0], vf);
#pragma unroll
      for (int e = 0; e < DPL; ++e) acc[e] = fmaf(p, vf[e], acc[e] * alpha);
      mrun = mnew;
    }
  } else {
  load_blk(kA, vA, j);
  {   // q: 16-byte vector loads, issued behind the first block's requests
    constexpr int NV = DPL / 4;
    float4 qv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) qv[i] = reinterpret_cast<const float4*>(qp)[i];
#pragma unroll
    for (int i = 0; i < NV; ++i) { q[4 * i] = qv[i].x; q[4 * i + 1] = qv[i].y; q[4 * i + 2] = qv[i].z; q[4 * i + 3] = qv[i].w; }
  }
  }
  ASTAMP(2);   // first block and q requested
  bool first_blk = true;
#if CTTS_QF_PRE == 2
  if constexpr (QF) {   // blocks j and j + KB are already here (or on their way): consume them, restoring the loop's invariant
    if (j < jend) {
      __builtin_amdgcn_sched_barrier(0);
      use_blk(kA, vA, j);
      j += KB;
      if (j < jend) {
        load_blk(kA, vA, j + KB);
        __builtin_amdgcn_sched_barrier(0);
        use_blk(kB, vB, j);
        j += KB;
      }
    }
  }
#endif
  if (NBUF > 2) {   // ring of three: blocks j + KB and j + 2 KB are on their way while block j is consumed
    load_blk(kB, vB, j + KB);
    while (j < jend) {
      load_blk(kC, vC, j + 2 * KB);
      __builtin_amdgcn_sched_barrier(0);
      use_blk(kA, vA, j);
      if (dbg && first_blk) { asm volatile("" :: "v"(lrun)); ASTAMP(3); first_blk = false; }
      j += KB;
      if (!(j < jend)) break;
      load_blk(kA, vA, j + 2 * KB);
      __builtin_amdgcn_sched_barrier(0);
      use_blk(kB, vB, j);
      j += KB;
      if (!(j < jend)) break;
      load_blk(kB, vB, j + 2 * KB);
      __builtin_amdgcn_sched_barrier(0);
      use_blk(kC, vC, j);
      j += KB;
    }
  } else
  while (j < jend) {
    load_blk(kB, vB, j + KB);   // prefetch (clamped, so harmless past the end)
    __builtin_amdgcn_sched_barrier(0);  // keep the 8 prefetch loads ahead of the consumer (hipcc sinks them otherwise)
    use_blk(kA, vA, j);
    if (dbg && first_blk) { asm volatile("" :: "v"(lrun)); ASTAMP(3); first_blk = false; }   // first block consumed (= landed)
    j += KB;
    if (!(j < jend)) break;
    load_blk(kA, vA, j + KB);
    __builtin_amdgcn_sched_barrier(0);
    use_blk(kB, vB, j);
    j += KB;
  }

  if (dbg) { asm volatile("" :: "v"(lrun)); ASTAMP(4); }   // this wave's keys done
  // OPJ: this head's slice of o_proj -- Wo[:, 64 h .. 64 h + 63] as [k / 8][column][8] bf16, 96 KB shared by every utterance (L2) --
  // is requested NOW, behind the wave's last KV block and ahead of the merge: thread t owns columns t, t + 256, t + 512
  u128 wv[OPJ ? 3 : 1][OPJ ? 8 : 1];
  if constexpr (OPJ) {
    const u128* wo = reinterpret_cast<const u128*>(rm.wo_h) + (size_t)h * 8 * HID + tid;
#pragma unroll
    for (int jc = 0; jc < 3; ++jc)
#pragma unroll
      for (int g8 = 0; g8 < 8; ++g8) wv[jc][g8] = load16(wo + (size_t)g8 * HID + 256 * jc);
    __builtin_amdgcn_sched_barrier(0);
  }
  // merge the key groups of this wave (same running max in every lane)
  lrun = groups_sum<LPK>(lrun, lane);
#pragma unroll
  for (int e = 0; e < DPL; ++e) acc[e] = groups_sum<LPK>(acc[e], lane);
  if (NW == 1) {
    if (kg == 0) {
      const float inv = 1.0f / lrun;
      OT* op = PKO ? out + pko_off<OT>(m, h * HDIM + dl * DPL) : out + (size_t)m * HID + h * HDIM + dl * DPL;
#pragma unroll
      for (int e = 0; e < DPL; ++e) store_out<OT>(op + e, acc[e] * inv);
    }
    return;
  }
  if (kg == 0) {
    if (dl == 0) { sm_m[wave] = mrun; sm_l[wave] = lrun; }
#pragma unroll
    for (int e = 0; e < DPL; ++e) sm_acc[wave][dl * DPL + e] = acc[e];
  }
  __syncthreads();
  if (tid < HDIM) {
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) M = fmaxf(M, sm_m[w]);
    float L = 0.f, o = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float sc = (sm_m[w] == -INFINITY) ? 0.f : expf(sm_m[w] - M);
      L += sm_l[w] * sc;
      o += sm_acc[w][tid] * sc;
    }
    if (SPLIT && n_piece > 1) {
      // hand-off through memory: partial -> write-through stores -> drain -> ticket; the last arriver merges
      unsigned* P = reinterpret_cast<unsigned*>(rm.sp_part + ((size_t)su * ATT_SPLIT_MAX + piece) * 66);
      __hip_atomic_store(P + tid, __float_as_uint(o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tid == 0) {
        __hip_atomic_store(P + 64, __float_as_uint(M), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(P + 65, __float_as_uint(L), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's partial has left (tid < 64: wave 0 only)
      int ticket = 0;
      if (tid == 0) ticket = __hip_atomic_fetch_add(rm.sp_cnt + su, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ticket = __builtin_amdgcn_readfirstlane(ticket);
      if (ticket != n_piece - 1) return;
      const unsigned* Q = reinterpret_cast<const unsigned*>(rm.sp_part + (size_t)su * ATT_SPLIT_MAX * 66);
      float pm[ATT_SPLIT_MAX], pl[ATT_SPLIT_MAX], po[ATT_SPLIT_MAX];
      float Mx = -INFINITY;
#pragma unroll
      for (int p = 0; p < ATT_SPLIT_MAX; ++p) {
        if (p < n_piece) {
          pm[p] = __uint_as_float(__hip_atomic_load(Q + p * 66 + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          pl[p] = __uint_as_float(__hip_atomic_load(Q + p * 66 + 65, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          po[p] = __uint_as_float(__hip_atomic_load(Q + p * 66 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          Mx = fmaxf(Mx, pm[p]);
        }
      }
      L = 0.f; o = 0.f;
#pragma unroll
      for (int p = 0; p < ATT_SPLIT_MAX; ++p) {
        if (p < n_piece) {
          const float sc = (pm[p] == -INFINITY) ? 0.f : expf(pm[p] - Mx);
          L += pl[p] * sc;
          o += po[p] * sc;
        }
      }
      if (tid == 0) __hip_atomic_store(rm.sp_cnt + su, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    if constexpr (!OPJ) {
      if constexpr (PKO) store_pko<OT>(out, m, h * HDIM + tid, o / L, rm.x3_plane);
      else store_out<OT>(out + (size_t)m * HID + h * HDIM + tid, o / L);
    }
    else sm_on[tid] = f32_to_bf16(o / L);   // the same bf16 rounding the o_proj kernel's A operand had
  }
  if constexpr (OPJ) {
    // ---- o_proj + residual inside the attention launch (HF Llama: self_attn.o_proj, examples/onnx/modeling_llama.py:500,557) ----
    // Every (utterance, head) unit multiplies its 64 outputs by its 64 columns of Wo (bf16 x bf16 -> f32: v_dot2_f32_bf16, k ascending)
    // and publishes a 768-wide f32 partial WRITE-THROUGH (sc1, MI355X guide Guideline 16 R1); every storing wave drains, one lane
    // draws a ticket from the row's counter; the unit that draws 11 is the row's last: it reads the 12 partials with sc1 loads,
    // adds them in head order 0..11 onto the residual (fixed order: deterministic and independent of arrival order), and writes what
    // the o_proj kernel's epilogue wrote: the f32 residual, its bf16 copy in fragment order, the 48 partial sums of squares.
    // Replaces one launch per layer (the one at 0.04 of the HBM roof); the counter is left at 0 for the next launch.
    __syncthreads();
    float pj[3] = {0.f, 0.f, 0.f};
    const u128* onv = reinterpret_cast<const u128*>(sm_on);
#pragma unroll
    for (int g8 = 0; g8 < 8; ++g8) {
      const u128 a = onv[g8];   // 8 bf16 outputs, the same address in every lane (LDS broadcast)
#pragma unroll
      for (int jc = 0; jc < 3; ++jc) {
        pj[jc] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, wv[jc][g8].x), __builtin_bit_cast(bf16x2_t, a.x), pj[jc], false);
        pj[jc] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, wv[jc][g8].y), __builtin_bit_cast(bf16x2_t, a.y), pj[jc], false);
        pj[jc] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, wv[jc][g8].z), __builtin_bit_cast(bf16x2_t, a.z), pj[jc], false);
        pj[jc] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, wv[jc][g8].w), __builtin_bit_cast(bf16x2_t, a.w), pj[jc], false);
      }
    }
#pragma unroll
    for (int jc = 0; jc < 3; ++jc) sm_p[tid + 256 * jc] = pj[jc];
    __syncthreads();
    const auto prs = __builtin_amdgcn_make_buffer_rsrc((void*)rm.op_part, 0, rm.op_part_bytes, 0x00020000);
    if (tid < 192)   // 16-byte write-through stores: columns 4 tid .. 4 tid + 3
      __builtin_amdgcn_raw_buffer_store_b128(reinterpret_cast<const u32x4_t*>(sm_p)[tid], prs, ((m * NHEAD + h) * HID + 4 * tid) * 4, 0, 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave: its partial has left
    __syncthreads();
    if (tid == 0) sm_last = __hip_atomic_fetch_add(rm.op_cnt + m, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == NHEAD - 1;
    __syncthreads();
    if (!sm_last) return;
    if (tid == 0) __hip_atomic_store(rm.op_cnt + m, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    if (tid < 192) {
      u32x4_t pv[NHEAD];
#pragma unroll
      for (int hh = 0; hh < NHEAD; ++hh) pv[hh] = __builtin_amdgcn_raw_buffer_load_b128(prs, ((m * NHEAD + hh) * HID + 4 * tid) * 4, 0, 16);
      float4 sres = *reinterpret_cast<const float4*>(rm.x32 + (size_t)m * HID + 4 * tid);
#pragma unroll
      for (int hh = 0; hh < NHEAD; ++hh) {
        sres.x += __uint_as_float(pv[hh].x); sres.y += __uint_as_float(pv[hh].y);
        sres.z += __uint_as_float(pv[hh].z); sres.w += __uint_as_float(pv[hh].w);
      }
      emit_row(sres, tid, rm.x32 + (size_t)m * HID, xb_row_ptr(rm.xp, m, tid, 1), rm.ssq + (size_t)m * SSQ_PARTS);
    }
  }
  ASTAMP(5);
#undef ASTAMP
}

template <typename KT, int NW, typename OT, bool PKO = false, bool SPLIT = false, bool PF = false, int NBUF = 2, bool OPJ = false>
__global__ __launch_bounds__(64 * NW + (PF ? 64 : 0)) void attention_k(const float* __restrict__ qkv, const KT* __restrict__ kc,
                                                       const KT* __restrict__ vc, int cmax, OT* __restrict__ out, GptRowMap rm) {
  if (NW > 1) CTTS_PROBE_RETURN();
  attention_body<KT, NW, OT, PKO, SPLIT, PF, NBUF, OPJ, false>(qkv, kc, vc, cmax, out, rm, blockIdx.x, blockIdx.y, blockIdx.y * gridDim.x + blockIdx.x);
}

// HPW heads of one utterance per workgroup (round 5 A/B, CTTS_ATT_HPW=2|3|4; ctts_k_attention_heads_per_wg): the SAME 4-wave unit,
// arithmetic and bits as attention_k<KT, 4>, but 12 / HPW workgroups of 4 HPW waves per utterance instead of 12 of 4 -- the question is
// whether a launch's fixed cost (5.0 us by bench.py's roofline.fit; an empty 768-workgroup stage measures 5.1 us, a 192-workgroup one
// 2.6) follows the number of WORKGROUPS dispatched or the number of waves.  The heads of one utterance have the same context, so the
// units of a workgroup run the same number of blocks and meet at the merge's barrier together.
template <typename KT, typename OT, int HPW>
__global__ __launch_bounds__(256 * HPW) void attention_hpw_k(const float* __restrict__ qkv, const KT* __restrict__ kc, const KT* __restrict__ vc, int cmax,
                                                              OT* __restrict__ out, GptRowMap rm) {
  attention_body<KT, 4, OT, true, false, false, 2, false, false, HPW>(qkv, kc, vc, cmax, out, rm, blockIdx.x * HPW, blockIdx.y, blockIdx.y * gridDim.x + blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// Persistent decode attention (round 5; both modes, packed output).  attention_k launches one workgroup per (utterance, head)
// unit of the captured batch -- 768 workgroups at batch 64 whether 64 or 6 utterances are still alive -- and a launch of 768
// workgroups costs ~3 us more than one of 256 on this part before any byte moves (profiles/r4a_overlap_probe.log E rows;
// attention_k's MINIMUM over a C3 run is 5.9 us).  Here the grid is G <= 256 workgroups of 4 waves -- one per CU -- and workgroup w
// walks the units w, 2G-1-w, 2G+w, ... ("snake": the rows are ordered by descending context, so a workgroup's long unit of an even
// round is paired with a short one of the odd round) until the unit index leaves the step's live list (RowDesc.b < 0, or
// 12 * *n_active).  A unit is computed EXACTLY as attention_k<KT, 4> computes it -- the same split of the visible keys over the 4 waves, the
// same 32-key (f32: 16-key) blocks in the same order, the same online softmax, the same LDS merge -- so the result is the same bit
// for bit in both modes; what changes is how the bytes are kept in flight.  One workgroup per CU has only 4 waves to cover the HBM
// round trip (attention_k had two or three resident workgroups), so every wave runs a RING of D blocks (D - 1 blocks = 8 KB each
// in flight while one is consumed) over its FLATTENED block sequence: when a unit's last block has been requested the ring goes
// on with the first blocks of the workgroup's next unit (its descriptor was fetched a unit ahead, its q row rides behind the
// current unit's blocks), so the merge, the barrier and the store of a unit overlap the next unit's first round trip.
// The ring is branch-free on the load side: every iteration issues 8 loads; a slot with nothing left to fetch re-reads the
// last key (L1 hit) and is skipped on the consuming side behind a register touch (hipcc's vmcnt bookkeeping stays exact that way).
// Reference op: softmax(q K^T / 8 + mask) V of HF Llama attention, examples/onnx/modeling_llama.py:455-475.
// ------------------------------------------------------------------------------------------------
template <typename KT, typename OT, int D>
__global__ __launch_bounds__(256) void attention_persist_k(const float* __restrict__ qkv, const KT* __restrict__ kc, const KT* __restrict__ vc,
                                                           int cmax, OT* __restrict__ out, const RowDesc* __restrict__ desc, int n_units,
                                                           const int32_t* __restrict__ n_active, size_t x3_plane) {
  CTTS_PROBE_RETURN();
  constexpr int NW = 4;
  constexpr int DPL = KTraits<KT>::DPL;
  constexpr int LPK = HDIM / DPL, KPI = 64 / LPK, NI = 4, KB = KPI * NI;
  __shared__ float sm_m[2][NW], sm_l[2][NW], sm_acc[2][NW][HDIM];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = lane / LPK, dl = lane % LPK;
  const int G = gridDim.x, w = blockIdx.x;
  if (n_active != nullptr) n_units = min(n_units, *n_active * NHEAD);   // host-compacted batches: descriptors beyond the live count are stale
  const int n_rounds = (n_units + G - 1) / G;
  if (n_rounds == 0) return;
  auto unit_of = [&](int r) { return (r & 1) ? (r + 1) * G - 1 - w : r * G + w; };
  auto desc_of = [&](int r) { return desc[min(unit_of(r), n_units - 1) / NHEAD]; };

  // ---- issue side: the unit whose blocks are being requested --------------------------------------------------------
  int ir = -1, im = 0, ih = 0, ij = 0, ijbeg = 0, ijend = 0, ijlast = 0;
  bool iv = false;
  const KT* ikb = kc + dl * DPL;
  const KT* ivb = vc + dl * DPL;
  RowDesc dpre = desc_of(0);
  float qn[DPL];
#pragma unroll
  for (int e = 0; e < DPL; ++e) qn[e] = 0.f;
  auto load_qn = [&]() {
    constexpr int NV = DPL / 4;
    const float* qp = qkv + (size_t)im * (3 * HID) + ih * HDIM + dl * DPL;
    float4 qv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) qv[i] = reinterpret_cast<const float4*>(qp)[i];
#pragma unroll
    for (int i = 0; i < NV; ++i) { qn[4 * i] = qv[i].x; qn[4 * i + 1] = qv[i].y; qn[4 * i + 2] = qv[i].z; qn[4 * i + 3] = qv[i].w; }
  };
  auto advance_issue = [&](bool with_q) {
    ++ir;
    const RowDesc d = dpre;
    const int u = unit_of(ir);
    if (ir + 1 < n_rounds) dpre = desc_of(ir + 1);
    iv = u < n_units && d.b >= 0;
    ijbeg = ijend = ij = 0;
    if (iv) {
      im = u / NHEAD;
      ih = u - im * NHEAD;
      const int rbeg = d.jlo, rend = d.slot + 1;
      const int nkeys = max(rend - rbeg, 0);
      const int per = ((nkeys + NW - 1) / NW + KPI - 1) / KPI * KPI;
      ijbeg = rbeg + wave * per;
      ijend = max(min(ijbeg + per, rend), ijbeg);
      ij = ijbeg;
      ijlast = max(min(ijbeg + per, rend) - 1, d.jlo);
      const size_t base = ((size_t)d.b * NHEAD + ih) * cmax * HDIM + dl * DPL;
      ikb = kc + base;
      ivb = vc + base;
      if (with_q) load_qn();
    }
  };

  // ---- consuming side ------------------------------------------------------------------------------------------------
  int cr = -1, cm = 0, ch = 0, cj = 0, cjend = 0, par = 0, nfin = 0;
  bool cv = false;
  float q[DPL], acc[DPL], mrun = -INFINITY, lrun = 0.f;
#pragma unroll
  for (int e = 0; e < DPL; ++e) { acc[e] = 0.f; q[e] = 0.f; }

  u128 kR[D][NI], vR[D][NI];
  int sj[D];
  auto issue = [&](const int s) {
    if (ij >= ijend && ir == cr && ir + 1 < n_rounds) advance_issue(true);
    const bool okb = ij < ijend;
    sj[s] = okb ? ij : -1;
    const int j0 = ij;
#pragma unroll
    for (int i = 0; i < NI; ++i) kR[s][i] = load16_nt(ikb + (size_t)min(j0 + i * KPI + kg, ijlast) * HDIM);
#pragma unroll
    for (int i = 0; i < NI; ++i) vR[s][i] = load16_nt(ivb + (size_t)min(j0 + i * KPI + kg, ijlast) * HDIM);
    if (okb) ij += KB;
  };
  auto use_blk = [&](const u128* kr, const u128* vr, const int j0) {   // attention_body's use_blk_e, operation for operation
    float s[NI];
    bool ok[NI];
    float bmax = -INFINITY;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      ok[i] = (j0 + i * KPI + kg) < cjend;
      float kf[DPL];
      float d = 0.f;
      unpack16<KT, DPL>(kr[i], kf);
#pragma unroll
      for (int e = 0; e < DPL; ++e) d = fmaf(q[e], kf[e], d);
      d = group_sum<LPK>(d);
      d *= 0.125f;
      s[i] = ok[i] ? d : -INFINITY;
      bmax = fmaxf(bmax, s[i]);
    }
    bmax = groups_max<LPK>(bmax);
    const float mnew = fmaxf(mrun, bmax);
    const float alpha = expf(mrun - mnew);
    lrun *= alpha;
#pragma unroll
    for (int e = 0; e < DPL; ++e) acc[e] *= alpha;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float p = expf(s[i] - mnew);
      lrun += p;
      float vf[DPL];
      unpack16<KT, DPL>(vr[i], vf);
#pragma unroll
      for (int e = 0; e < DPL; ++e) acc[e] = fmaf(p, vf[e], acc[e]);
    }
    mrun = mnew;
  };
  auto finish_unit = [&]() {   // attention_body's merge: key groups by shuffles, waves through LDS (double-buffered by unit parity)
    lrun = groups_sum<LPK>(lrun, lane);
#pragma unroll
    for (int e = 0; e < DPL; ++e) acc[e] = groups_sum<LPK>(acc[e], lane);
    if (kg == 0) {
      if (dl == 0) { sm_m[par][wave] = mrun; sm_l[par][wave] = lrun; }
#pragma unroll
      for (int e = 0; e < DPL; ++e) sm_acc[par][wave][dl * DPL + e] = acc[e];
    }
    __syncthreads();
    if (wave == (nfin & (NW - 1))) {   // the merging wave rotates; lane = output column
      float M = -INFINITY;
#pragma unroll
      for (int x = 0; x < NW; ++x) M = fmaxf(M, sm_m[par][x]);
      float L = 0.f, o = 0.f;
#pragma unroll
      for (int x = 0; x < NW; ++x) {
        const float sc = (sm_m[par][x] == -INFINITY) ? 0.f : expf(sm_m[par][x] - M);
        L += sm_l[par][x] * sc;
        o += sm_acc[par][x][lane] * sc;
      }
      store_pko<OT>(out, cm, ch * HDIM + lane, o / L, x3_plane);
    }
    mrun = -INFINITY; lrun = 0.f;
#pragma unroll
    for (int e = 0; e < DPL; ++e) acc[e] = 0.f;
    par ^= 1;
    ++nfin;
  };
  auto switch_consume = [&]() -> bool {
    if (cr + 1 >= n_rounds) return false;
    if (ir == cr) advance_issue(true);
    ++cr;
    cv = iv; cm = im; ch = ih; cj = ijbeg; cjend = ijend;
#pragma unroll
    for (int e = 0; e < DPL; ++e) q[e] = qn[e];
    return true;
  };

  // prologue: round 0's first block, THEN its q (an L2 miss in front of the KV stream costs more than one behind it), then the
  // rest of the ring
  advance_issue(false);
  issue(0);
  if (iv) load_qn();
#pragma unroll
  for (int s = 1; s < D - 1; ++s) issue(s);
  // No `break` out of the unrolled ring: hipcc routes every such exit through the loop latch, and its vmcnt bookkeeping then drains
  // the ring at the head of every D-th block.  A workgroup that has run out of units finishes the current turn of the ring on empty
  // slots instead (<= D - 1 re-reads of its last key).
  bool live = switch_consume();
  while (live && cj >= cjend) { if (cv) finish_unit(); live = switch_consume(); }
  while (live) {
#pragma unroll
    for (int p = 0; p < D; ++p) {
      issue((p + D - 1) % D);
      __builtin_amdgcn_sched_barrier(0);   // keep the 8 requests ahead of the consumer (hipcc sinks them otherwise)
      if (sj[p] >= 0) {
        use_blk(kR[p], vR[p], sj[p]);
        cj += KB;
      } else {
        // an empty slot: nothing to compute, but its (last) load must count as waited-for on this path too -- otherwise hipcc drains the
        // whole ring before it reuses the slot's registers
        asm volatile("" ::"v"(vR[p][NI - 1].w));
      }
      while (live && cj >= cjend) { if (cv) finish_unit(); live = switch_consume(); }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Fused QKV + attention launch (round 4; decode, perf mode, <= 64 rows).  512-thread workgroups: [0, 144) are the QKV tiles of
// decode_dev.hpp (RMSNorm scale + q/k/v_proj + RoPE + KV append, all rows each, 8 waves splitting K), workgroups 144 + 6 m + p hold the
// attention units (utterance m, heads 2p and 2p + 1), 4 waves each -- 144 + 6 * 64 = 528 workgroups of <= 128 VGPRs: (almost) all
// resident at once, which the overlap needs.  The dependency q / newest K,V -> attention is NOT all-to-all: a unit needs the 12 column tiles of its head only, and it
// has work that does not depend on them -- the requests for its old keys, the first memory round trip of the attention launch
// (3.0 us from request to first use, profiles/r3y_attn_phase_probe.log).  So the units start WITH the QKV tiles, put two blocks of
// old keys per wave in flight, and a fifth wave polls the head's arrival word (written behind write-through stores by the QKV tiles'
// finishing waves, decode_dev.hpp) from wave 0 of each unit; q and the newest key / value are then read with sc1 loads.  Removes one kernel boundary per layer
// and hides the KV stream's start-up latency behind the QKV tiles' weight stream.  Deadlock-free: QKV tiles have the lowest
// workgroup ids (dispatched first; and two 8-wave workgroups per CU hold nearly the whole grid), the spin is bounded (20 ms, then trap).
// Reference ops: HF Llama attention, examples/onnx/modeling_llama.py:415-417,239-256,455-475.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 4) void qkv_attention_k(DecGemmArgs d, const bf16_t* __restrict__ kc, const bf16_t* __restrict__ vc, int cmax,
                                                          bf16_t* __restrict__ out, GptRowMap rm, int n_qkv) {
  CTTS_PROBE_RETURN();
  if ((int)blockIdx.x < n_qkv) {   // QKV tile: 16 columns x all (<= 64) rows, the 8 waves split K (3 chunks of 32 each, one round)
    gemm_dec_wg<4, 8, true, FEPI_QKV_ROPE, true, 3>(d, blockIdx.x, 0, 1, blockIdx.x, n_qkv);
    return;
  }
  const int u = blockIdx.x - n_qkv;   // (utterance, head pair)
  attention_body<bf16_t, 4, bf16_t, true, false, false, 2, false, true>(d.C32, kc, vc, cmax, out, rm, 2 * (u % (NHEAD / 2)), u / (NHEAD / 2), blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// G5 (prefill, perf mode, long prompts): flash-style attention on the matrix cores.  The one-wave-per-(row, head) kernel
// above re-streams a row's keys for every query row -- O(T^2) cache traffic and scalar FMAs; fine for the 16-48 token text
// prompts of the benchmark configs (35 us per layer), not for an `spk_smp` audio-code prompt of several hundred tokens
// (core.py:435-453).  Here a workgroup owns 64 consecutive query rows of one (utterance, head): each of its 4 waves holds 16
// query rows as the A operand of v_mfma_f32_16x16x32_bf16 (q * 1/8, rounded to bf16), the workgroup walks the visible keys in
// blocks of 32 staged through LDS once for all 4 waves (K row-major for S = Q K^T, V TRANSPOSED so that a B fragment of
// P V is one 16-byte LDS read), keeps the running row max / sum of the online softmax in the MFMA C layout
// (row = 4 (lane >> 4) + r), turns P into an A operand through a per-wave LDS tile, and accumulates O[16 x 64] in 4 MFMA tiles.
// Mask = the reference's causal + left-pad mask (keys kv_start[b] .. slot; a pad query row sees only itself, its output is never
// consumed).  Reference math: /root/reference/examples/onnx/modeling_llama.py:455-475 (softmax(q k^T / 8 + mask) v in f32;
// here P and V enter the second product as bf16, like every other activation of the perf mode).
// ------------------------------------------------------------------------------------------------
template <typename OT>
__global__ __launch_bounds__(256) void attention_prefill_mfma_k(const float* __restrict__ qkv, const bf16_t* __restrict__ kc,
                                                                const bf16_t* __restrict__ vc, int cmax, OT* __restrict__ out, GptRowMap rm) {
  constexpr int KB = 32;                 // keys per block
  constexpr int KLD = HDIM + 8;          // K tile row stride (bf16): 144 B, conflict-free 16-byte fragment reads
  constexpr int VLD = KB + 8;            // V^T tile row stride (bf16): 80 B
  __shared__ __attribute__((aligned(16))) bf16_t Ks[KB][KLD];
  __shared__ __attribute__((aligned(16))) bf16_t Vt[HDIM][VLD];
  __shared__ __attribute__((aligned(16))) bf16_t Ps[4][16][VLD];

  const int T = rm.q_per_b;
  const int qt = blockIdx.x, h = blockIdx.y, bq = blockIdx.z;        // query tile of 64 rows, head, prompt row group
  const int b = rm.row_map ? rm.row_map[bq] : bq;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int kvs = rm.kv_start[b];
  const int t0 = qt * 64;                                             // first query index (within the chunk) of this workgroup
  if (t0 >= T) return;
  const int tq = t0 + wave * 16;                                      // this wave's first query index
  // A operand: lane (li, g) holds q[row li][c * 32 + g * 8 .. + 8] for the two 32-wide d chunks, scaled by 1/8
  // (q stays f32-accurate: q/8 = hi + lo in bf16, two MFMAs per product -- the decode kernel keeps q in f32 as well)
  bf16x8 qa[2], ql[2];
  {
    const int t = min(tq + li, T - 1);
    const float* qp = qkv + ((size_t)bq * T + t) * (3 * HID) + h * HDIM + g * 8;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = qp[c * 32 + e] * 0.125f;
        const __bf16 hi = (__bf16)v;
        qa[c][e] = hi;
        ql[c][e] = (__bf16)(v - (float)hi);
      }
  }
  // rows of this lane in the C layout: r -> query index tq + 4 g + r, its KV slot and first visible key
  int slot[4], jlo[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    slot[r] = rm.slot0 + min(tq + 4 * g + r, T - 1);
    jlo[r] = min(kvs, slot[r]);
  }
  float mrun[4], lrun[4];
  f32x4 o[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { mrun[r] = -INFINITY; lrun[r] = 0.f; }
#pragma unroll
  for (int d = 0; d < 4; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bf16_t* kbase = kc + ((size_t)b * NHEAD + h) * cmax * HDIM;
  const bf16_t* vbase = vc + ((size_t)b * NHEAD + h) * cmax * HDIM;
  const int slot_first = rm.slot0 + t0, slot_last = rm.slot0 + min(t0 + 63, T - 1);
  const int jbeg = min(kvs, slot_first) / KB * KB;
  for (int j0 = jbeg; j0 <= slot_last; j0 += KB) {
    __syncthreads();   // the previous block's K / V^T tiles have been consumed
    {  // stage K [32 keys][64 d] and V^T [64 d][32 keys]: thread -> (key = tid / 8, 8 d starting at (tid % 8) * 8)
      const int key = tid >> 3, d0 = (tid & 7) * 8;
      const int j = min(j0 + key, cmax - 1);
      const u128 kv = load16(kbase + (size_t)j * HDIM + d0);
      u128 vv = load16(vbase + (size_t)j * HDIM + d0);
      if (j0 + key > slot_last) vv = u128{0u, 0u, 0u, 0u};   // never-written cache rows: 0 * garbage must not become NaN in P V
      *reinterpret_cast<u128*>(&Ks[key][d0]) = kv;
      const bf16_t* ve = reinterpret_cast<const bf16_t*>(&vv);
#pragma unroll
      for (int e = 0; e < 8; ++e) Vt[d0 + e][key] = ve[e];
    }
    __syncthreads();
    // S = Q K^T for the two 16-key halves of the block
    f32x4 sc[2];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      sc[hb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const bf16x8 kb = *reinterpret_cast<const bf16x8*>(&Ks[hb * 16 + li][c * 32 + g * 8]);
        sc[hb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ql[c], kb, sc[hb], 0, 0, 0);
        sc[hb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[c], kb, sc[hb], 0, 0, 0);
      }
    }
    // mask + online softmax per query row (a row's 32 scores sit in the 16 lanes of its lane group, 2 per lane)
    float p[2][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s0 = sc[0][r], s1 = sc[1][r];
      const int ja = j0 + li, jb = j0 + 16 + li;
      if (ja < jlo[r] || ja > slot[r]) s0 = -INFINITY;
      if (jb < jlo[r] || jb > slot[r]) s1 = -INFINITY;
      float bm = fmaxf(s0, s1);
#pragma unroll
      for (int x = 1; x < 16; x <<= 1) bm = fmaxf(bm, __shfl_xor(bm, x, 64));
      const float mnew = fmaxf(mrun[r], bm);
      const float alpha = (mnew == -INFINITY) ? 1.f : expf(mrun[r] - mnew);   // nothing visible yet: keep the zeros
      const float p0 = (s0 == -INFINITY) ? 0.f : expf(s0 - mnew), p1 = (s1 == -INFINITY) ? 0.f : expf(s1 - mnew);
      float ps = p0 + p1;
#pragma unroll
      for (int x = 1; x < 16; x <<= 1) ps += __shfl_xor(ps, x, 64);
      lrun[r] = lrun[r] * alpha + ps;
      mrun[r] = mnew;
#pragma unroll
      for (int d = 0; d < 4; ++d) o[d][r] *= alpha;
      p[0][r] = p0; p[1][r] = p1;
    }
    // P (C layout: row 4g+r, key li / 16+li) -> LDS tile [16 rows][32 keys] -> A operand (row li, keys g*8..)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      Ps[wave][4 * g + r][li] = f32_to_bf16(p[0][r]);
      Ps[wave][4 * g + r][16 + li] = f32_to_bf16(p[1][r]);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes have landed (same-wave read-after-write)
    __builtin_amdgcn_wave_barrier();
    const bf16x8 pa = *reinterpret_cast<const bf16x8*>(&Ps[wave][li][g * 8]);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const bf16x8 vb = *reinterpret_cast<const bf16x8*>(&Vt[d * 16 + li][g * 8]);
      o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, vb, o[d], 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int t = tq + 4 * g + r;
    if (t < T) {
      const float inv = 1.0f / lrun[r];
      OT* op = out + ((size_t)bq * T + t) * HID + h * HDIM + li;
#pragma unroll
      for (int d = 0; d < 4; ++d) store_out<OT>(op + d * 16, o[d][r] * inv);
    }
  }
}

// persistent decode attention: [0] on / off, [1] workgroups, [2] ring depth -- environment at first use, ctts_k_attention_cfg afterwards
static int att_hpw_ = -1;   // heads per workgroup of the decode attention (attention_hpw_k): CTTS_ATT_HPW at first use, ctts_k_attention_heads_per_wg afterwards
static int att_hpw_cfg() {
  if (att_hpw_ < 0) {
    const char* e = getenv("CTTS_ATT_HPW");
    const int v = e ? atoi(e) : 1;
    att_hpw_ = (v == 2 || v == 3 || v == 4) ? v : 1;
  }
  return att_hpw_;
}
void attention_hpw_override(int hpw) { att_hpw_ = (hpw == 2 || hpw == 3 || hpw == 4) ? hpw : 1; }
static int att_cfg_[3] = {-1, 0, 4};
static int att_persist_cfg(int i) {
  if (att_cfg_[0] < 0) {
    const char* e = getenv("CTTS_ATT_PERSIST"); att_cfg_[0] = e ? (atoi(e) != 0) : 0;   // measured: 9.7-10.9 us per launch vs 8.9 (profiles/r5b_ab_persist_dpp.log)
    const char* eg = getenv("CTTS_ATT_G"); if (eg) att_cfg_[1] = atoi(eg);
    const char* ed = getenv("CTTS_ATT_D"); if (ed) att_cfg_[2] = atoi(ed);
    if (att_cfg_[1] <= 0) {
      int dev = 0, cus = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      att_cfg_[1] = cus > 0 ? cus : 256;
    }
  }
  return att_cfg_[i];
}
void attention_persist_override(int persist, int g, int d) {
  (void)att_persist_cfg(0);
  if (persist >= 0) att_cfg_[0] = persist != 0;
  if (g > 0) att_cfg_[1] = g;
  if (d > 0) att_cfg_[2] = d;
}

hipError_t launch_attention(const float* qkv, const void* kcache, const void* vcache, int kv_wt, int cmax, void* out, int out_bf16,
                            GptRowMap rm, int M, hipStream_t st) {
  dim3 grid(NHEAD, M);
  const bool decode = rm.q_per_b == 1;
  static int flash_min = -1;   // CTTS_FLASH_MIN_T: prompt (chunk) length from which prefill uses the MFMA kernel (0 = never)
  if (flash_min < 0) { const char* e = getenv("CTTS_FLASH_MIN_T"); flash_min = e ? atoi(e) : 128; }
  if (!decode && kv_wt == WT_BF16 && flash_min > 0 && rm.q_per_b >= flash_min && M % rm.q_per_b == 0) {
    dim3 g3((rm.q_per_b + 63) / 64, NHEAD, M / rm.q_per_b);
    if (out_bf16 == 1) CTTS_LAUNCH((attention_prefill_mfma_k<bf16_t>), g3, dim3(256), st, qkv, (const bf16_t*)kcache, (const bf16_t*)vcache, cmax, (bf16_t*)out, rm);
    else if (out_bf16 == 0) CTTS_LAUNCH((attention_prefill_mfma_k<float>), g3, dim3(256), st, qkv, (const bf16_t*)kcache, (const bf16_t*)vcache, cmax, (float*)out, rm);
    else return hipErrorInvalidValue;
    return hipGetLastError();
  }
  static int nw8 = -1;  // CTTS_ATT_NW=8: 8 waves per (utterance, head) in decode (A/B knob)
  if (nw8 < 0) { const char* e = getenv("CTTS_ATT_NW"); nw8 = (e && atoi(e) == 8) ? 1 : 0; }
  // CTTS_ATT_LDS=<bytes>: dynamic LDS the decode attention workgroups declare (and never touch).  It bounds the workgroups a CU
  // holds at once (160 KiB / bytes), which turns the dispatcher into a greedy list scheduler: with the rows ordered by descending
  // context (ctts_gen_state.order) the longest units start first and the short ones fill the CUs that free up.  0 = no bound.
  static int att_lds = -1, att_small_m = 0, nw_packed = 4, att_nbuf = 2;
  if (att_lds < 0) {
    { const char* e4 = getenv("CTTS_ATT_NBUF"); if (e4) att_nbuf = atoi(e4); }          // KV blocks in flight per wave of the packed decode attention: 2 | 3
    { const char* e3 = getenv("CTTS_ATT_NW_PACKED"); if (e3) nw_packed = atoi(e3); }   // waves per (utterance, head) unit of the perf-mode decode attention: 4 | 8 | 16
    { const char* e2 = getenv("CTTS_ATT_SMALL_M"); if (e2) att_small_m = atoi(e2); }   // batches up to this many rows: 16-wave units (0 = never)
    const char* e = getenv("CTTS_ATT_LDS");
    att_lds = e ? atoi(e) : 0;
    if (att_lds > 65536) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_k<bf16_t, 4, bf16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, att_lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_k<float, 4, float, true>), hipFuncAttributeMaxDynamicSharedMemorySize, att_lds);
    }
  }
  // round 5: persistent grid (attention_persist_k) for the decode step of both modes.  CTTS_ATT_PERSIST=0: one workgroup per unit
  // (attention_k); CTTS_ATT_G=<workgroups> (default: the device's CU count), CTTS_ATT_D=2|3|4 KV blocks per wave ring (default 4);
  // ctts_k_attention_cfg overrides all three (tests, probes)
  const int persist = att_persist_cfg(0), att_g = att_persist_cfg(1), att_d = att_persist_cfg(2);
  if (persist && decode && rm.desc != nullptr && (out_bf16 == 2 || out_bf16 == 3 || out_bf16 == 4) && rm.dbg == nullptr &&
      !(out_bf16 == 2 && rm.sp_cus > 0 && rm.sp_part != nullptr && rm.sp_cnt != nullptr)) {
    if ((out_bf16 == 2) != (kv_wt == WT_BF16) || (out_bf16 == 4 && rm.x3_plane == 0)) return hipErrorInvalidValue;
    const int n_units = NHEAD * M;
    const dim3 pg(min(att_g, n_units));
    const int32_t* nact = rm.desc_covers_all ? nullptr : rm.n_active;
#define ATTP(KT, OT, D) CTTS_LAUNCH((attention_persist_k<KT, OT, D>), pg, dim3(256), st, qkv, (const KT*)kcache, (const KT*)vcache, cmax, (OT*)out, rm.desc, n_units, nact, rm.x3_plane)
    if (out_bf16 == 2) {
      if (att_d == 2) ATTP(bf16_t, bf16_t, 2); else if (att_d == 3) ATTP(bf16_t, bf16_t, 3); else ATTP(bf16_t, bf16_t, 4);
    } else if (out_bf16 == 3) {
      if (att_d == 2) ATTP(float, float, 2); else if (att_d == 3) ATTP(float, float, 3); else ATTP(float, float, 4);
    } else {
      if (att_d == 2) ATTP(float, x3p_t, 2); else if (att_d == 3) ATTP(float, x3p_t, 3); else ATTP(float, x3p_t, 4);
    }
#undef ATTP
    return hipGetLastError();
  }
  const int hpw = att_hpw_cfg();
  if (hpw > 1 && decode && rm.desc != nullptr && (out_bf16 == 2 || out_bf16 == 3 || out_bf16 == 4) && rm.dbg == nullptr &&
      !(out_bf16 == 2 && rm.sp_cus > 0 && rm.sp_part != nullptr && rm.sp_cnt != nullptr)) {
    if ((out_bf16 == 2) != (kv_wt == WT_BF16) || (out_bf16 == 4 && rm.x3_plane == 0)) return hipErrorInvalidValue;
    const dim3 hg(NHEAD / hpw, M), hb(256 * hpw);
#define ATTH(KT, OT, H) CTTS_LAUNCH((attention_hpw_k<KT, OT, H>), hg, hb, st, qkv, (const KT*)kcache, (const KT*)vcache, cmax, (OT*)out, rm)
    if (out_bf16 == 2) { if (hpw == 2) ATTH(bf16_t, bf16_t, 2); else if (hpw == 3) ATTH(bf16_t, bf16_t, 3); else ATTH(bf16_t, bf16_t, 4); }
    else if (out_bf16 == 3) { if (hpw == 2) ATTH(float, float, 2); else if (hpw == 3) ATTH(float, float, 3); else ATTH(float, float, 4); }
    else { if (hpw == 2) ATTH(float, x3p_t, 2); else if (hpw == 3) ATTH(float, x3p_t, 3); else ATTH(float, x3p_t, 4); }
#undef ATTH
    return hipGetLastError();
  }
  if (out_bf16 == 2) {   // decode, perf mode: bf16 output in the fragment-packed order the o_proj kernel of decode.hip reads
    if (!decode || kv_wt != WT_BF16) return hipErrorInvalidValue;
    if (rm.sp_cus > 0 && rm.sp_part != nullptr && rm.sp_cnt != nullptr)
      CTTS_LAUNCH((attention_k<bf16_t, 4, bf16_t, true, true>), dim3(NHEAD * M + rm.sp_cus), dim3(256), st, qkv, (const bf16_t*)kcache,
                  (const bf16_t*)vcache, cmax, (bf16_t*)out, rm);
    else if (M <= att_small_m)
      // A/B knob, OFF by default (CTTS_ATT_SMALL_M=<rows>): 16 waves per unit for tiny batches (BASELINE C2: batch 1 = 12 units on
      // 256 CUs), so a long context is in flight in one round trip.  MEASURED at batch 1 (profiles/r3b_c2_ab.log): 0.482 ms per
      // step with it, 0.478 without -- the 16-way LDS merge costs what the shorter stream saves -- and it would make a row's
      // perf-mode bits depend on the batch size.
      CTTS_LAUNCH((attention_k<bf16_t, 16, bf16_t, true>), grid, dim3(1024), st, qkv, (const bf16_t*)kcache, (const bf16_t*)vcache, cmax, (bf16_t*)out, rm);
#ifdef CTTS_PF_BUILD
    else if (rm.pf.base != nullptr)
      CTTS_LAUNCH_SMEM((attention_k<bf16_t, 4, bf16_t, true, false, true>), grid, dim3(320), att_lds, st, qkv, (const bf16_t*)kcache, (const bf16_t*)vcache, cmax, (bf16_t*)out, rm);
#endif
    else if (nw_packed == 2)
      CTTS_LAUNCH_SMEM((attention_k<bf16_t, 2, bf16_t, true>), grid, dim3(128), att_lds, st, qkv, (const bf16_t*)kcache, (const bf16_t*)vcache, cmax, (bf16_t*)out, rm);
    else if (nw_packed == 8)
      CTTS_LAUNCH_SMEM((attention_k<bf16_t, 8, bf16_t, true>), grid, dim3(512), att_lds, st, qkv, (const bf16_t*)kcache, (const bf16_t*)vcache, cmax, (bf16_t*)out, rm);
    else if (nw_packed == 16)
      CTTS_LAUNCH_SMEM((attention_k<bf16_t, 16, bf16_t, true>), grid, dim3(1024), att_lds, st, qkv, (const bf16_t*)kcache, (const bf16_t*)vcache, cmax, (bf16_t*)out, rm);
    else if (att_nbuf == 3)
      CTTS_LAUNCH_SMEM((attention_k<bf16_t, 4, bf16_t, true, false, false, 3>), grid, dim3(256), att_lds, st, qkv, (const bf16_t*)kcache, (const bf16_t*)vcache, cmax, (bf16_t*)out, rm);
    else
      CTTS_LAUNCH_SMEM((attention_k<bf16_t, 4, bf16_t, true>), grid, dim3(256), att_lds, st, qkv, (const bf16_t*)kcache, (const bf16_t*)vcache, cmax, (bf16_t*)out, rm);
    return hipGetLastError();
  }
  if (out_bf16 == 3 && !decode) {   // prefill, f32 parity mode: one wave per (row, head), output in the packed f32 order
    if (kv_wt == WT_BF16) return hipErrorInvalidValue;
    CTTS_LAUNCH((attention_k<float, 1, float, true>), grid, dim3(64), st, qkv, (const float*)kcache, (const float*)vcache, cmax, (float*)out, rm);
    return hipGetLastError();
  }
  if (out_bf16 == 4) {   // decode, split-bf16 parity mode: f32 KV cache, output as hi / lo bf16 planes in decode.hip's fragment order
    if (!decode || kv_wt == WT_BF16 || rm.x3_plane == 0) return hipErrorInvalidValue;
    CTTS_LAUNCH_SMEM((attention_k<float, 4, x3p_t, true>), grid, dim3(256), att_lds, st, qkv, (const float*)kcache, (const float*)vcache, cmax, (x3p_t*)out, rm);
    return hipGetLastError();
  }
  if (out_bf16 == 3) {   // decode, f32 parity mode: f32 output in the fragment-packed order o_proj of decode32.hip reads
    if (!decode || kv_wt == WT_BF16) return hipErrorInvalidValue;
    if (att_nbuf == 3)
      CTTS_LAUNCH_SMEM((attention_k<float, 4, float, true, false, false, 3>), grid, dim3(256), att_lds, st, qkv, (const float*)kcache, (const float*)vcache, cmax, (float*)out, rm);
    else
      CTTS_LAUNCH_SMEM((attention_k<float, 4, float, true>), grid, dim3(256), att_lds, st, qkv, (const float*)kcache, (const float*)vcache, cmax, (float*)out, rm);
    return hipGetLastError();
  }
  if (decode && nw8 && kv_wt == WT_BF16 && out_bf16) {
    CTTS_LAUNCH((attention_k<bf16_t, 8, bf16_t>), grid, dim3(512), st, qkv, (const bf16_t*)kcache, (const bf16_t*)vcache, cmax, (bf16_t*)out, rm);
    return hipGetLastError();
  }
#define ATT(KT, NW, OT) CTTS_LAUNCH((attention_k<KT, NW, OT>), grid, dim3(64 * NW), st, qkv, (const KT*)kcache, (const KT*)vcache, cmax, (OT*)out, rm)
  if (kv_wt == WT_BF16) {
    if (out_bf16) { if (decode) ATT(bf16_t, 4, bf16_t); else ATT(bf16_t, 1, bf16_t); }
    else { if (decode) ATT(bf16_t, 4, float); else ATT(bf16_t, 1, float); }
  } else {
    if (out_bf16) { if (decode) ATT(float, 4, bf16_t); else ATT(float, 1, bf16_t); }
    else { if (decode) ATT(float, 4, float); else ATT(float, 1, float); }
  }
#undef ATT
  return hipGetLastError();
}

hipError_t launch_qkv_attention(const DecGemmArgs& d_in, const void* kcache, const void* vcache, int cmax, void* out_packed, GptRowMap rm, int M,
                                hipStream_t st) {
  DecGemmArgs d = d_in;
  if (rm.q_per_b != 1 || rm.desc == nullptr || M <= 0 || M > 64 || d.M != M || d.epi != FEPI_QKV_ROPE || d.N != 3 * HID || d.K != HID || !d.ssq_in ||
      !d.desc || !d.ho_flag || rm.qf_flag != d.ho_flag || !d.rope_cs)
    return hipErrorInvalidValue;
  static int nt = -1, a_early = 1;
  if (nt < 0) { const char* e = getenv("CTTS_W_NT"); nt = e ? atoi(e) : 1; const char* e2 = getenv("CTTS_DEC_A_EARLY"); a_early = e2 ? atoi(e2) : 1; }
  d.w_nt = d.force_nt ? (d.force_nt == 2) : nt;
  d.a_early = a_early;
  rm.sp_cus = 0; rm.sp_part = nullptr; rm.sp_cnt = nullptr;
  rm.qf_rows = M;
  const int n_qkv = 3 * HID / 16;
  CTTS_LAUNCH(qkv_attention_k, dim3(n_qkv + (NHEAD / 2) * M), dim3(512), st, d, (const bf16_t*)kcache, (const bf16_t*)vcache, cmax, (bf16_t*)out_packed, rm, n_qkv);
  return hipGetLastError();
}

hipError_t launch_attention_oproj(const float* qkv, const void* kcache, const void* vcache, int cmax, GptRowMap rm, int M, hipStream_t st) {
  if (rm.q_per_b != 1 || rm.desc == nullptr || !rm.wo_h || !rm.op_part || !rm.op_cnt || !rm.x32 || !rm.xp || !rm.ssq) return hipErrorInvalidValue;
  rm.sp_cus = 0; rm.sp_part = nullptr; rm.sp_cnt = nullptr;   // no remainder splitting on this path
  CTTS_LAUNCH((attention_k<bf16_t, 4, bf16_t, true, false, false, 2, true>), dim3(NHEAD, M), dim3(256), st, qkv, (const bf16_t*)kcache,
              (const bf16_t*)vcache, cmax, (bf16_t*)nullptr, rm);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// G8  final RMSNorm of the last position of every row; the f32 result is the step's hidden state
//     (gpt.py:430-436) -> hiddens[b, gen] and the staging row read by the heads GEMM.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(192) void final_norm_k(const float* __restrict__ x, int q_per_b, const float* __restrict__ w, float eps,
                                                    float* __restrict__ hfin, float* __restrict__ hiddens, int max_new,
                                                    const int32_t* __restrict__ len, int T, const int32_t* __restrict__ row_map,
                                                    const int32_t* __restrict__ n_active, const int32_t* __restrict__ prompt_len,
                                                    float* __restrict__ hfin_p, const int32_t* __restrict__ last_row) {
  __shared__ float part[3];
  CTTS_PROBE_RETURN();
  const int m = blockIdx.x, t = threadIdx.x;
  if (row_absent(n_active, m)) return;
  const int b = row_map ? row_map[m] : m;   // compact activation row m belongs to utterance b
  // (last_row: the prompt pass ran over the valid tokens only -- prefill_compact_k says where utterance m's last token sits)
  const float* row = x + (last_row != nullptr ? (size_t)last_row[m] : (size_t)m * q_per_b + (q_per_b - 1)) * HID;
  const float4 v = *reinterpret_cast<const float4*>(row + t * 4);
  float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  ss = wave_sum(ss);
  if ((t & 63) == 0) part[t >> 6] = ss;
  __syncthreads();
  const float rstd = 1.0f / sqrtf(((part[0] + part[1]) + part[2]) / (float)HID + eps);
  const float4 g = *reinterpret_cast<const float4*>(w + t * 4);
  float4 o;
  o.x = g.x * (v.x * rstd); o.y = g.y * (v.y * rstd); o.z = g.z * (v.z * rstd); o.w = g.w * (v.w * rstd);
  *reinterpret_cast<float4*>(hfin + (size_t)m * HID + t * 4) = o;
  if (hfin_p != nullptr) *reinterpret_cast<float4*>(hfin_p + pk32_off(m, 4 * t, HID / 16)) = o;   // A operand of the packed heads GEMM
  const int gen = len[b] - (prompt_len ? prompt_len[b] : T);
  if (hiddens != nullptr && gen >= 0 && gen < max_new)
    *reinterpret_cast<float4*>(hiddens + ((size_t)b * max_new + gen) * HID + t * 4) = o;
}

hipError_t launch_final_norm(const float* x, int q_per_b, const float* w, float eps, float* hfin, float* hiddens, int max_new,
                             const int32_t* len, int T, int B, const int32_t* row_map, const int32_t* n_active,
                             const int32_t* prompt_len, hipStream_t st, float* hfin_packed, const int32_t* last_row) {
  CTTS_LAUNCH(final_norm_k, dim3(B), dim3(192), st, x, q_per_b, w, eps, hfin, hiddens, max_new, len, T, row_map, n_active, prompt_len, hfin_packed,
              last_row);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// G10 + G11  fused sampling chain, one wave per (b, k) row, one workgroup per batch row b.
//   logits / temperature                                   gpt.py:487
//   repetition penalty over the last <=16 generated ids    processors.py:18-35 (rows >= 625: none)
//   TopP(min_keep 3) then TopK(min_keep 3)                 processors.py:38-58 + transformers warpers
//   EOS mask while gen < min_new_token                     gpt.py:494-495
//   softmax, argmax(p / q), q ~ Exp(1) from the CPU stream gpt.py:497-508 (host draws q)
//   finish |= any(tok == eos); ids_buf[:, T+gen] = tok; end_idx += !finish; len += 1   gpt.py:512-577
//
// The sort the reference does per row (626 logits) is replaced by an exact equivalent that needs
// no sort: both warpers keep a PREFIX of the descending order, so the kept set is found by
// repeatedly extracting the wave-wide maximum (ties: lowest index first) while accumulating the
// probability mass above it in double -- cum_ascending(v) = fl32(S_all - mass_above(v)), the same
// value ATen's double-accumulated cumsum rounds to float.  At most max(top_k,3)+ties extractions.
// ------------------------------------------------------------------------------------------------
#define SLOTS 10  // ceil(626 / 64)

__device__ __forceinline__ void wave_argmax(float v, int idx, float& bv, int& bi) {
  // max value, ties -> lowest index; result uniform across the wave.  Two DPP reductions (max of the values,
  // then min of the indices that hold it) instead of a 6-step (value, index) butterfly on ds_bpermute.
  bv = wave_max_dpp(v);
  bi = wave_min_dpp(v == bv ? idx : 0x7fffffff);
}

// Exp(1) draws of the opt-in DEVICE generator (ctts_gen_state.rng_device): Philox4x32-10 keyed on the call's seed, counter =
// (token group, global sampling row, step, stream tag).  One call yields the draws of 4 consecutive tokens of one row:
// q = -log(u), u = (x + 1) * 2^-32 in (0, 1].  The reference on a GPU device draws from the device generator too (gpt.py:39:
// torch.Generator(device=device)); its CPU stream stays the parity default (rng.py).
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float exp1_from_bits(uint32_t x) {
  const float u = ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f);   // 24 bits -> (0, 1], exactly representable
  return -logf(u);
}
// the 4 draws of tokens 4 grp .. 4 grp + 3 of global sampling row `grow` at generation step `step`
#define CTTS_RNG_WORD3 0x43545453u   // fourth counter word when the caller gives no per-slot nonce
__device__ __forceinline__ void device_exp_draws4(unsigned long long seed, int grow, int step, int grp, float (&q)[4], uint32_t w3 = CTTS_RNG_WORD3) {
  uint32_t r[4];
  philox4x32_10((uint32_t)grp, (uint32_t)grow, (uint32_t)step, w3, (uint32_t)seed, (uint32_t)(seed >> 32), r);
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = exp1_from_bits(r[i]);
}

__global__ __launch_bounds__(256) void exp_draws_k(unsigned long long seed, int step, int row0, int V, float* __restrict__ out) {
  const int row = blockIdx.x, grp = threadIdx.x + blockIdx.y * 256;   // tests: the generator's draws as a [rows, V] tensor
  if (4 * grp >= V) return;
  float q[4];
  device_exp_draws4(seed, row0 + row, step, grp, q);
#pragma unroll
  for (int i = 0; i < 4; ++i) if (4 * grp + i < V) out[(size_t)row * V + 4 * grp + i] = q[i];
}
hipError_t launch_exp_draws(unsigned long long seed, int step, int row0, int rows, int V, float* out, hipStream_t st) {
  hipLaunchKernelGGL(exp_draws_k, dim3(rows, (V / 4 + 255) / 256), dim3(256), 0, st, seed, step, row0, V, out);
  return hipGetLastError();
}

// wave-wide minimum of a float (DPP max of the negation), result uniform
__device__ __forceinline__ float wave_minf_dpp(float v) { return -wave_max_dpp(-v); }

// PARITY CERTIFICATE (ctts_gen_state.margin): the smallest distance, in tempered-logit units, by which this step's outcome was decided.
//   c_arg = log(r_best / r_second) of argmax(p / q); c_cut = value gap between the last kept and the first dropped token of the warpers'
//   prefix; c_p = |log(cum / thr)| of the top-p test at the last kept rank and at the rank top-p removed first; all divided by the
//   largest factor the repetition penalty applies to a perturbation (alpha for negative, 1 / alpha for positive scores).
// Moving every tempered logit by less than half of min(c_arg, c_cut, c_p) changes neither the kept set nor the argmax: p / q moves by a
// factor e^(+-2 eps) between any two tokens, a cumulative probability by e^(+-2 eps), a value gap by 2 eps.
__global__ __launch_bounds__(256) void sample_k(SampleArgs a) {
  __shared__ int tok_s[NVQ];
  __shared__ float marg_s[NVQ];
  __shared__ int fin_s;
  __shared__ float cand_v[NVQ][64], cand_e[NVQ][64];
  __shared__ int cand_i[NVQ][64];
  CTTS_PROBE_RETURN();
  const int m = blockIdx.x;                       // compact logits row
  const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
  long long* dbg = a.dbg ? a.dbg + (size_t)m * 8 : nullptr;
#define SSTAMP(i) do { if (dbg && threadIdx.x == 0) dbg[i] = wall_clock64(); } while (0)
  SSTAMP(0);
  int b, len;
  if (a.desc != nullptr) {    // decode with device-side compaction: ONE load gives the utterance and its length (and says whether
    const RowDesc d = a.desc[m];   // the row exists this step) instead of the n_active -> row_map -> len chain
    if (d.b < 0) return;
    b = d.b; len = d.slot + 1;
  } else {
    if (row_absent(a.n_active, m)) return;
    b = a.row_map ? a.row_map[m] : m;     // utterance (batch slot)
    len = a.len[b];
  }
  if (len >= a.tcap) {  // slot full (slot pools only; generate() never steps past max_new_token): stop, write nothing
    if (threadIdx.x == 0) a.finish[b] = 1;
    return;
  }
  const int gen = len - (a.prompt_len ? a.prompt_len[b] : a.T);  // tokens generated so far == step index i of gpt.py:394
  const float* lrow = a.logits + ((size_t)m * NVQ + k) * NAUDIO;
  SSTAMP(1);   // row known
  const float temp = a.temperature[k];
  // global sampling row (multi-GPU shards keep the reference's numbering; row_base: shards that are not contiguous row blocks)
  const int grow = (a.row_base != nullptr ? a.row_base[b] : a.row_offset + b * NVQ) + k;
  const unsigned long long seed = a.rng_device ? *a.rng_seed : 0ull;
  const uint32_t w3 = (a.rng_device && a.rng_nonce != nullptr) ? a.rng_nonce[b] : CTTS_RNG_WORD3;

  // everything that depends only on (b, k, gen) is requested NOW, in one round: the logits, the Exp(1) draws consumed at the very
  // end, the <= 16 history tokens, the harness hooks
  float x[SLOTS], qv[SLOTS];
  int cnt[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int v = s * 64 + lane;
    x[s] = (v < NAUDIO) ? lrow[v] : 0.f;
    cnt[s] = 0;
  }
  if (a.rng_device == 0) {
    const float* qrow = a.q + ((size_t)(gen % a.nq) * a.q_rows * NVQ + (size_t)b * NVQ + k) * NAUDIO;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) qv[s] = (s * 64 + lane < NAUDIO) ? qrow[s * 64 + lane] : 1.f;
  }
  const bool penal = a.pow_table != nullptr && grow < a.max_input_ids;
  const int nh = min(gen, 16);
  // the <=16 history tokens are fetched by 16 lanes in ONE load round and broadcast (a serial loop of
  // dependent global loads costs one L2 round trip per token)
  const int mine = (penal && lane < nh) ? (int)a.ids_buf[((size_t)b * a.tcap + (len - 1 - lane)) * NVQ + k] : -1;
  const float ptab_reg = penal ? a.pow_table[min(lane, 16)] : 1.f;   // the 17-entry table penalty^count, one entry per lane
  const int sa = a.stop_at != nullptr ? a.stop_at[b] : -1;
  const bool forced_t = a.teacher != nullptr && gen < a.teacher_stride;
  const int64_t teach = forced_t ? a.teacher[((size_t)b * a.teacher_stride + gen) * NVQ + k] : 0;
  if (a.rng_device != 0) {   // device generator: the draws of this lane's tokens (token v = 64 s + lane sits in group v / 4)
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int v = s * 64 + lane;
      float q4[4];
      device_exp_draws4(seed, grow, a.rng_per_step ? gen : 0, v >> 2, q4, w3);
      qv[s] = q4[v & 3];
    }
  }

#pragma unroll
  for (int s = 0; s < SLOTS; ++s) x[s] = (s * 64 + lane < NAUDIO) ? x[s] / temp : -INFINITY;   // gpt.py:487
  if (dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  SSTAMP(2);   // every load landed
  // repetition penalty
  float pen_amp = 1.f;
  if (penal) {
    // occurrences of this lane's 10 tokens among the <= 16 history tokens, 5 bits per slot in one 64-bit word: history token t
    // (wave-uniform) belongs to lane t % 64, slot t / 64
    unsigned long long packed = 0ull;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int t = __builtin_amdgcn_readlane(mine, j);  // -1 beyond the history: matches no vocabulary slot
      const int d = t - lane;
      const bool hit = (d & 63) == 0 && (unsigned)d < (unsigned)(SLOTS * 64);
      packed += hit ? (1ull << (5 * (d >> 6))) : 0ull;
    }
    // alpha = penalty^count from the table held in registers (one cross-lane read per slot instead of a dependent global load
    // per slot), then BOTH candidates -- logit * alpha and logit / alpha -- and a select: no divergent branches
    float al[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      cnt[s] = (int)((packed >> (5 * s)) & 31ull);
      al[s] = __shfl(ptab_reg, cnt[s], 64);
    }
    if (a.margin != nullptr) {   // certificate: the largest factor the penalty applies to a perturbation of a tempered logit
      float am = 1.f;
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) if (s * 64 + lane < NAUDIO) am = fmaxf(am, (x[s] < 0.f) ? al[s] : 1.0f / al[s]);
      pen_amp = wave_max_dpp(am);
    }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const float mu = x[s] * al[s], dv = x[s] / al[s];
      x[s] = (x[s] < 0.f) ? mu : dv;
    }
  }

  // softmax statistics over the whole row (needed by top-p)
  float mx = -INFINITY;
  SSTAMP(3);   // penalty done
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) mx = fmaxf(mx, x[s]);
  const float lane_max = mx;
  mx = wave_max_dpp(mx);
  float e[SLOTS];
  float zs = 0.f;
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) { e[s] = (s * 64 + lane < NAUDIO) ? expf(x[s] - mx) : 0.f; zs += e[s]; }
  zs = wave_sum(zs);
  const float rz = 1.0f / zs;
  double sall = 0.0;
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) { e[s] = e[s] * rz; sall += (double)e[s]; }  // e = softmax prob (f32)
  sall = wave_sum_d(sall);
  SSTAMP(4);   // softmax statistics done

  // ---- the kept set.  Both warpers keep a PREFIX of the descending order (value desc, ties: lowest index first), so it is
  // described by its last element (v_last, i_last): kept = everything at or before it in that order.
  const int kk = a.use_top_k ? min(max(a.top_k, 3), NAUDIO) : NAUDIO;
  const float thr = a.top_p_thr;  // float32(1 - top_p): `cum <= (1 - top_p)` on a float tensor casts the scalar to float
  const bool any_filter = a.use_top_p || a.use_top_k;
  float v_last = INFINITY; int i_last = -1;   // nothing kept yet
  int n_kept = 0;
  bool done = !any_filter;
  const bool cert = a.margin != nullptr;   // uniform
  float c_cut = INFINITY, c_p = INFINITY;
  // FAST PATH (top-k <= 64, the reference's default 20): the prefix can only end inside the top kk (+ ties with the kk-th value).
  // t = the kk-th largest LANE maximum bounds the kk-th largest value from below (kk elements are >= t), so the candidates
  // {x >= t} contain the whole prefix; they are compacted to one per lane and ranked by counting (value desc, index asc -- the same
  // total order as the serial extraction) together with the probability mass before them (double); then every candidate applies the
  // warpers' tests to itself and the prefix ends before the first one removed.  Falls back to the serial loop when more than 64
  // candidates survive.
  if (any_filter && a.use_top_k && kk <= 64) {
    // t: the smallest lane maximum that has fewer than kk lane maxima strictly above it = the kk-th largest lane maximum
    int gtc = 0;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
      const float o = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lane_max), j));
      gtc += (o > lane_max) ? 1 : 0;
    }
    const float t = -wave_max_dpp(gtc < kk ? -lane_max : -INFINITY);
    int base = 0;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const bool c = x[s] >= t;   // padded slots hold -inf
      const unsigned long long ms = __ballot(c);
      const int pos = base + (int)__popcll(ms & ((1ull << lane) - 1ull));
      if (c && pos < 64) { cand_v[k][pos] = x[s]; cand_e[k][pos] = e[s]; cand_i[k][pos] = s * 64 + lane; }
      base += (int)__popcll(ms);
    }
    const int C = base;   // >= kk
    if (C <= 64) {
      __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes have landed (same-wave read-after-write)
      __builtin_amdgcn_wave_barrier();
      const bool act = lane < C;
      const float cv = act ? cand_v[k][lane] : -INFINITY, ce = act ? cand_e[k][lane] : 0.f;
      const int ci = act ? cand_i[k][lane] : 0x7fffffff;
      // rank of this lane's candidate in the order (value desc, index asc), and the probability mass of everything before it
      int rank = 0;
      double mass_above = 0.0;
      for (int j = 0; j < C; ++j) {
        const float o = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cv), j));
        const int oi = __builtin_amdgcn_readlane(ci, j);
        const float oe = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ce), j));
        const bool before = (o > cv) || (o == cv && oi < ci);
        rank += before ? 1 : 0;
        mass_above += before ? (double)oe : 0.0;
      }
      // every candidate judges itself; the kept prefix ends before the first (lowest-rank) candidate a warper removes
      const unsigned long long whok = __ballot(act && rank == kk - 1);     // exists: C >= kk
      const float kth_val = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cv), (int)__ffsll((long long)whok) - 1));
      const bool ok_p = !(a.use_top_p && rank >= 3 && (float)(sall - mass_above) <= thr);   // ascending cumulative prob incl. itself <= 1 - top_p
      const bool ok_k = rank < kk || cv == kth_val;                                          // ties with the k-th largest value survive
      const int n = wave_min_dpp((act && !(ok_p && ok_k)) ? rank : C);
      const unsigned long long wlast = __ballot(act && rank == n - 1);      // n >= 3 (min_tokens_to_keep)
      const int src = (int)__ffsll((long long)wlast) - 1;
      v_last = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cv), src));
      i_last = __builtin_amdgcn_readlane(ci, src);
      n_kept = n;
      done = true;
      if (cert) {
        // first dropped value: the candidate of rank n, or (every candidate kept) the largest non-candidate
        float nxt;
        if (n < C) {
          const unsigned long long wn = __ballot(act && rank == n);
          nxt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cv), (int)__ffsll((long long)wn) - 1));
        } else {
          float bm = -INFINITY;
#pragma unroll
          for (int s = 0; s < SLOTS; ++s) bm = fmaxf(bm, x[s] < t ? x[s] : -INFINITY);
          nxt = wave_max_dpp(bm);
        }
        c_cut = (a.use_top_k && n > kk) ? 0.f : v_last - nxt;   // (ties with the k-th value were kept: an exact tie decided the set)
        if (a.use_top_p) {
          // the top-p test of the last kept rank (if it was tested at all: rank >= 3) and of the rank top-p removed first
          const float cum = fmaxf((float)(sall - mass_above), 1e-38f);
          const bool mine = act && rank >= 3 && (rank == n - 1 || (rank == n && !ok_p));
          c_p = wave_minf_dpp(mine ? fabsf(__logf(cum / thr)) : INFINITY);
        }
      }
    }
  }
  if (!done) {
    // SERIAL PATH: repeated extraction of the wave-wide maximum (ties: lowest index first) while accumulating the probability
    // mass above it in double -- cum_ascending(v) = fl32(S_all - mass_above(v)), the value ATen's double-accumulated cumsum rounds to
    unsigned taken = 0;                // bit s: slot s of this lane already extracted
    double mass_above = 0.0;
    float kth_val = 0.f;
    int n = 0;
    float nxt = -INFINITY, cum_last = INFINITY, cum_drop = -1.f;   // certificate: first dropped value, cum at the last kept / first p-dropped rank
    while (n < NAUDIO) {
      float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        const bool avail = !((taken >> s) & 1u) && (s * 64 + lane < NAUDIO);
        if (avail && (x[s] > bv)) { bv = x[s]; bi = s * 64 + lane; }  // ascending s => lowest index on ties
      }
      float wv; int wi;
      wave_argmax(bv, bi, wv, wi);
      float cum = INFINITY;
      if (a.use_top_p && n >= 3) {
        cum = (float)(sall - mass_above);
        if (cum <= thr) { nxt = wv; cum_drop = cum; break; }
      }
      if (a.use_top_k && n >= kk && !(wv == kth_val)) { nxt = wv; break; }
      cum_last = cum;
      const int ws = wi >> 6, wl = wi & 63;
      float pe = 0.f;
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) if (s == ws) pe = e[s];
      pe = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pe), wl));  // wl is wave-uniform
      mass_above += (double)pe;
      if (lane == wl) taken |= 1u << ws;
      v_last = wv; i_last = wi;
      if (n == kk - 1) kth_val = wv;
      ++n;
    }
    n_kept = n;
    if (cert) {
      c_cut = (a.use_top_k && n > kk) ? 0.f : v_last - nxt;
      if (cum_last < INFINITY) c_p = fabsf(__logf(fmaxf(cum_last, 1e-38f) / thr));
      if (cum_drop >= 0.f) c_p = fminf(c_p, fabsf(__logf(fmaxf(cum_drop, 1e-38f) / thr)));
    }
  }

  // EOS handling (min_new_token and the bench harness's stop_at hook)
  bool mask_eos = gen < a.min_new;
  SSTAMP(5);   // kept set known
  bool force_eos = false;
  if (sa >= 0) { mask_eos = mask_eos || (gen < sa); force_eos = gen >= sa; }
  // final softmax over the kept set and argmax(p / q)
  float m2 = -INFINITY;
  bool live[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int v = s * 64 + lane;
    const bool kept = !any_filter || (n_kept > 0 && (x[s] > v_last || (x[s] == v_last && v <= i_last)));
    live[s] = kept && v < NAUDIO && !(mask_eos && v == a.eos);
    if (live[s]) m2 = fmaxf(m2, x[s]);
  }
  m2 = wave_max_dpp(m2);
  float z2 = 0.f, p2[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) { p2[s] = live[s] ? expf(x[s] - m2) : 0.f; z2 += p2[s]; }
  z2 = wave_sum(z2);
  const float rz2 = 1.0f / z2;
  float bv = -1.f, bv2 = -1.f; int bi = 0x7fffffff;   // bv2: this lane's second best (certificate)
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int v = s * 64 + lane;
    if (v < NAUDIO) {
      const float r = (p2[s] * rz2) / qv[s];
      if (r > bv) { bv2 = bv; bv = r; bi = v; }
      else bv2 = fmaxf(bv2, r);
    }
  }
  float wv; int wi;
  wave_argmax(bv, bi, wv, wi);
  if (cert) {
    const float r2 = wave_max_dpp(bi == wi ? bv2 : bv);   // best of everything but the winner (a tie with the winner: margin 0)
    const float c_arg = !(wv > 0.f) ? 0.f : (r2 > 0.f ? __logf(wv / r2) : INFINITY);
    if (lane == 0) marg_s[k] = fminf(c_arg, fminf(c_cut, c_p)) / pen_amp;   // in units of the PRE-penalty tempered logit
  }
  if (force_eos) wi = a.eos;
  SSTAMP(6);   // token drawn
  if (a.sampled != nullptr && gen < a.teacher_stride && lane == 0) a.sampled[((size_t)b * a.teacher_stride + gen) * NVQ + k] = (int64_t)wi;
  if (forced_t) wi = (int)teach;
  if (lane == 0) {
    a.ids_buf[((size_t)b * a.tcap + len) * NVQ + k] = (int64_t)wi;
    tok_s[k] = wi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    bool fin = a.finish[b] != 0;
#pragma unroll
    for (int c = 0; c < NVQ; ++c) fin = fin || (tok_s[c] == a.eos);
    a.finish[b] = fin ? 1 : 0;
    if (!fin) a.end_idx[b] += 1;
    a.len[b] = len + 1;
    // (a forced EOS -- the bench harness's stop_at -- decides nothing; one workgroup per utterance and step: no atomics needed)
    if (cert && !force_eos) a.margin[b] = fminf(a.margin[b], fminf(fminf(marg_s[0], marg_s[1]), fminf(marg_s[2], marg_s[3])));
    fin_s = fin ? 1 : 0;
  }
  if (a.next.x != nullptr) {
    // the next step's input row of this utterance, at the same compact row m (EmbedNext, kernels.hpp): what embed_codes_k writes
    __syncthreads();
    const int t = threadIdx.x;
    const bool fin = fin_s != 0;
    if (t == 0) {
      const int ks = a.next.sp.kv_start[b];
      a.next.sp.desc[m] = RowDesc{fin ? -1 : b, len, len - ks < 0 ? 1 : len - ks, ks > len ? len : ks};   // as write_desc, slot = len
    }
    if (!fin && t < 192) {
      write_rope_cs(a.next.sp, m, b, len, t);
      float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int c = 0; c < NVQ; ++c) {
        const int id = min(max(tok_s[c], 0), NAUDIO - 1);
        const float4 v = *reinterpret_cast<const float4*>(a.next.emb_code + ((size_t)c * NAUDIO + id) * HID + t * 4);
        if (c == 0) s4 = v; else { s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w; }
      }
      emit_row(s4, t, a.next.x + (size_t)m * HID, xb_row_ptr(a.next.xb, m, t, a.next.sp.xb_packed),
               a.next.ssq ? a.next.ssq + (size_t)m * SSQ_PARTS : nullptr,
               a.next.sp.xp32 ? a.next.sp.xp32 + pk32_off(m, 4 * t, HID / 16) : nullptr, a.next.sp.xb_lo_plane);
    }
  }
  SSTAMP(7);
#undef SSTAMP
}

hipError_t launch_sample(const SampleArgs& a, hipStream_t st) {
  CTTS_LAUNCH(sample_k, dim3(a.B), dim3(256), st, a);
  return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// refine-text mode (infer_text=True, gpt.py:406-407,439-440,477-485,519-525): the decode input is
// emb_text[last token], the head is the 21178-way text head, one sampling row per utterance.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(192) void embed_text_k(const float* __restrict__ emb_text, int n_text, const int64_t* __restrict__ ids_buf,
                                                    int tcap, const int32_t* __restrict__ len, float* __restrict__ x,
                                                    uint16_t* __restrict__ xb, float* __restrict__ ssq,
                                                    const int32_t* __restrict__ row_map, const int32_t* __restrict__ n_active,
                                                    StepPrep sp) {
  step_zero(sp);
  const int m = blockIdx.x, t = threadIdx.x;
  int b;
  if (sp.row_map_out != nullptr) {   // device-side compaction (see embed_codes_k)
    int total;
    b = nth_unfinished(sp.finish, sp.order, gridDim.x, m, total);
    if (m == 0 && t == 0) *sp.n_active_out = total;
    if (b < 0) {   // row m does not exist this step: say so in its descriptor (the attention kernel reads nothing else)
      if (sp.desc != nullptr && t == 0) sp.desc[m] = RowDesc{-1, 0, 0, 0};
      return;
    }
    if (t == 0) sp.row_map_out[m] = b;
  } else {
    if (row_absent(n_active, m)) return;
    b = row_map ? row_map[m] : m;
  }
  const int slot = len[b] - 1;
  if (sp.desc != nullptr && t == 0) write_desc(sp, m, b, slot);
  write_rope_cs(sp, m, b, slot, t);
  int id = (int)ids_buf[((size_t)b * tcap + slot) * NVQ];  // slot 0 (gpt.py:407)
  id = min(max(id, 0), n_text - 1);
  const float4 s = *reinterpret_cast<const float4*>(emb_text + (size_t)id * HID + t * 4);
  emit_row(s, t, x + (size_t)m * HID, xb_row_ptr(xb, m, t, sp.xb_packed), ssq ? ssq + (size_t)m * SSQ_PARTS : nullptr,
           sp.xp32 ? sp.xp32 + pk32_off(m, 4 * t, HID / 16) : nullptr, sp.xb_lo_plane);
}

hipError_t launch_embed_text(const float* emb_text, int n_text, const int64_t* ids_buf, int tcap, const int32_t* len, float* x,
                             uint16_t* xb, float* ssq, int B, const int32_t* row_map, const int32_t* n_active, hipStream_t st,
                             const StepPrep* prep) {
  CTTS_LAUNCH(embed_text_k, dim3(B), dim3(192), st, emb_text, n_text, ids_buf, tcap, len, x, xb, ssq, row_map, n_active, prep_or_none(prep));
  return hipGetLastError();
}

#define TEXT_VMAX 21248  // >= num_text_tokens (21178), multiple of 256
#define TEXT_NT 1024     // threads of a sample_text_k workgroup
#define TEXT_NW (TEXT_NT / 64)
#define TEXT_PER ((TEXT_VMAX + TEXT_NT - 1) / TEXT_NT)   // 21 elements per thread

struct BlockRed {
  float f[TEXT_NW]; int i[TEXT_NW]; double d[TEXT_NW];
};
// block-wide reductions of the TEXT_NW waves: wave results meet in LDS and EVERY thread folds them in the same fixed order
__device__ __forceinline__ float block_maxN(float v, BlockRed& r, int wave, int lane) {
  v = wave_max(v);
  __syncthreads();
  if (lane == 0) r.f[wave] = v;
  __syncthreads();
  float o = r.f[0];
#pragma unroll
  for (int w = 1; w < TEXT_NW; ++w) o = fmaxf(o, r.f[w]);
  return o;
}
__device__ __forceinline__ float block_sumN(float v, BlockRed& r, int wave, int lane) {
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) r.f[wave] = v;
  __syncthreads();
  float o = r.f[0];
#pragma unroll
  for (int w = 1; w < TEXT_NW; ++w) o += r.f[w];
  return o;
}
__device__ __forceinline__ double block_sumNd(double v, BlockRed& r, int wave, int lane) {
  v = wave_sum_d(v);
  __syncthreads();
  if (lane == 0) r.d[wave] = v;
  __syncthreads();
  double o = r.d[0];
#pragma unroll
  for (int w = 1; w < TEXT_NW; ++w) o += r.d[w];
  return o;
}
__device__ __forceinline__ int block_minNi(int v, BlockRed& r, int wave, int lane) {
  v = wave_min_dpp(v);
  __syncthreads();
  if (lane == 0) r.i[wave] = v;
  __syncthreads();
  int o = r.i[0];
#pragma unroll
  for (int w = 1; w < TEXT_NW; ++w) o = min(o, r.i[w]);
  return o;
}
// max value, ties -> lowest index, uniform over the block
__device__ __forceinline__ void block_argmaxN(float v, int idx, BlockRed& r, int wave, int lane, float& bv, int& bi) {
  float wv; int wi;
  wave_argmax(v, idx, wv, wi);
  __syncthreads();
  if (lane == 0) { r.f[wave] = wv; r.i[wave] = wi; }
  __syncthreads();
  bv = r.f[0]; bi = r.i[0];
#pragma unroll
  for (int w = 1; w < TEXT_NW; ++w)
    if (r.f[w] > bv || (r.f[w] == bv && r.i[w] < bi)) { bv = r.f[w]; bi = r.i[w]; }
}

// One 1024-thread workgroup per utterance; the row lives in REGISTERS: thread t holds the tempered logits v = t + 1024 i, i < 21 (every
// loop below is fully unrolled over them; the global loads of the row are ONE round trip).  Round 6 history of this kernel
// (profiles/r6b_bench.log, r6c_reftext.log): the round-5 version extracted up to top_K maxima one by one, each a 21178-wide LDS scan with
// two barriers; 256 threads with the row in LDS and a rolled load loop took 64-84 us per launch, 256 threads with the row in registers
// 44-60 us (one wave per SIMD grinding through 83 elements x ~100 instructions); 1024 threads cut that serial stream by four.
// Same sort-free prefix description of the kept set as sample_k:
//   FAST PATH (top-k <= 64, the reference's refine default 20, core.py:182-193): t = the kk-th largest of the 64 column maxima (column =
//   the same lane of all 16 waves) -- kk elements are >= t, so {x >= t} holds the whole prefix (~25 candidates for kk = 20).  They are
//   compacted into LDS (deterministic order: a block prefix sum of the per-thread counts), ranked by
//   counting in the order (value desc, index asc) with the probability mass before them in double, and every candidate applies the warpers'
//   tests to itself.
//   SERIAL PATH (no top-k, top-k > 64, or more than TEXT_CAND candidates): repeated extraction of the block-wide maximum.
// Both leave (v_last, i_last, n_kept); the last three passes (max, sum, argmax(p / q)) are shared and read q only for kept tokens.
#define TEXT_CAND 256
__global__ __launch_bounds__(TEXT_NT) void sample_text_k(SampleArgs a, int V) {
  __shared__ BlockRed red;
  __shared__ float cand_v[TEXT_CAND], cand_e[TEXT_CAND];
  __shared__ int cand_i[TEXT_CAND];
  __shared__ int wave_cnt[TEXT_NW];
  __shared__ float lane_max[TEXT_NW][64];
  __shared__ float sh_f[6];   // kth_val, v_last, nxt, c_p(last kept), c_p(first p-dropped), spare
  __shared__ int sh_i[2];     // i_last
  const int m = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  long long* dbg = a.dbg ? a.dbg + (size_t)m * 8 : nullptr;   // probes only (tools/text_phase_probe.py)
#define TSTAMP(i) do { if (dbg && tid == 0) dbg[i] = wall_clock64(); } while (0)
  TSTAMP(0);
  if (row_absent(a.n_active, m)) return;
  const int b = a.row_map ? a.row_map[m] : m;
  const int len = a.len[b];
  if (len >= a.tcap) {
    if (tid == 0) a.finish[b] = 1;
    return;
  }
  const int gen = len - (a.prompt_len ? a.prompt_len[b] : a.T);
  const float* lrow = a.logits + (size_t)m * V;
  const float temp = a.temperature[0];
  const bool cert = a.margin != nullptr;
  TSTAMP(1);   // row known

  float x[TEXT_PER];
#pragma unroll
  for (int i = 0; i < TEXT_PER; ++i) { const int v = tid + TEXT_NT * i; x[i] = (v < V) ? lrow[v] : 0.f; }   // all loads in flight at once
  if (dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  TSTAMP(2);   // row landed
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < TEXT_PER; ++i) {
    x[i] = (tid + TEXT_NT * i < V) ? x[i] / temp : -INFINITY;   // gpt.py:487 (repetition penalty is not supported in this mode)
    mx = fmaxf(mx, x[i]);
  }
  const float tmax = mx;   // this thread's maximum
  if (tid < 6) sh_f[tid] = INFINITY;
  mx = block_maxN(mx, red, wave, lane);
  float zs = 0.f, ex[TEXT_PER];
#pragma unroll
  for (int i = 0; i < TEXT_PER; ++i) { ex[i] = (tid + TEXT_NT * i < V) ? expf(x[i] - mx) : 0.f; zs += ex[i]; }
  zs = block_sumN(zs, red, wave, lane);
  const float rz = 1.0f / zs;
  double sall = 0.0;
#pragma unroll
  for (int i = 0; i < TEXT_PER; ++i) if (tid + TEXT_NT * i < V) sall += (double)(ex[i] * rz);
  sall = block_sumNd(sall, red, wave, lane);
  TSTAMP(3);   // softmax statistics

  const int kk = a.use_top_k ? min(max(a.top_k, 3), V) : V;
  const float thr = a.top_p_thr;
  const bool any_filter = a.use_top_p || a.use_top_k;
  float v_last = INFINITY; int i_last = -1, n_kept = 0;
  float c_cut = INFINITY, c_p = INFINITY;
  bool done = !any_filter;
  bool fast = false;                     // the kept set was found on the candidate list: (cand_rank, cand_val, cand_idx) of this thread's candidate
  int cand_rank = 0x7fffffff, cand_idx = 0x7fffffff;
  float cand_val = -INFINITY;
  if (any_filter && a.use_top_k && kk <= 64) {
    // threshold: the kk-th largest of the 64 COLUMN maxima (column l = lane l of every wave, 16 x 21 elements).  kk columns hold an
    // element >= t, so {x >= t} contains the prefix -- and only ~1.2 kk candidates survive (the kk-th largest thread maximum of ONE wave,
    // a 1/16 sample of the row, left ~370 of them, and ranking those by counting was 38 of the kernel's 50 us: profiles/r6g_text_phase.log)
    lane_max[wave][lane] = tmax;
    __syncthreads();
    float sm = lane_max[0][lane];
#pragma unroll
    for (int w = 1; w < TEXT_NW; ++w) sm = fmaxf(sm, lane_max[w][lane]);
    int gtc = 0;
#pragma unroll 8
    for (int j = 0; j < 64; ++j) {
      const float o = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sm), j));
      gtc += (o > sm) ? 1 : 0;
    }
    const float t = -wave_max_dpp(gtc < kk ? -sm : -INFINITY);   // the same value in every wave
    // deterministic compaction of {x >= t}: per-thread counts -> block exclusive prefix sum -> each thread writes its own, ascending index
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < TEXT_PER; ++i) cnt += (x[i] >= t) ? 1 : 0;   // (padded slots hold -inf)
    int inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    if (lane == 63) wave_cnt[wave] = inc;
    __syncthreads();
    int base = inc - cnt, C = 0;
#pragma unroll
    for (int w = 0; w < TEXT_NW; ++w) { if (w < wave) base += wave_cnt[w]; C += wave_cnt[w]; }   // C >= kk
    if (C <= TEXT_CAND) {
      if (cnt > 0) {
#pragma unroll
        for (int i = 0; i < TEXT_PER; ++i)
          if (x[i] >= t) { cand_v[base] = x[i]; cand_i[base] = tid + TEXT_NT * i; cand_e[base] = ex[i] * rz; ++base; }
      }
      __syncthreads();
      const bool act = tid < C;
      const float cv = act ? cand_v[tid] : -INFINITY;
      const int ci = act ? cand_i[tid] : 0x7fffffff;
      int rank = 0;
      double mass_above = 0.0;
      if (act) {
        for (int j = 0; j < C; ++j) {
          const float o = cand_v[j];
          const int oi = cand_i[j];
          const bool before = (o > cv) || (o == cv && oi < ci);
          rank += before ? 1 : 0;
          mass_above += before ? (double)cand_e[j] : 0.0;
        }
      }
      if (act && rank == kk - 1) sh_f[0] = cv;   // exists: C >= kk
      __syncthreads();
      const float kth_val = sh_f[0];
      const float cum = (float)(sall - mass_above);   // ascending cumulative probability including itself
      const bool ok_p = !(a.use_top_p && rank >= 3 && cum <= thr);
      const bool ok_k = rank < kk || cv == kth_val;   // ties with the k-th largest value survive
      const int n = block_minNi((act && !(ok_p && ok_k)) ? rank : C, red, wave, lane);   // >= 3 (min_tokens_to_keep)
      if (act && rank == n - 1) {
        sh_f[1] = cv; sh_i[0] = ci;
        if (a.use_top_p && rank >= 3) sh_f[3] = fabsf(__logf(fmaxf(cum, 1e-38f) / thr));
      }
      if (act && rank == n) {
        sh_f[2] = cv;
        if (a.use_top_p && rank >= 3 && !ok_p) sh_f[4] = fabsf(__logf(fmaxf(cum, 1e-38f) / thr));
      }
      float bm = -INFINITY;   // every candidate kept: the first dropped value is the largest non-candidate
      if (cert && n == C) {
#pragma unroll
        for (int i = 0; i < TEXT_PER; ++i) bm = fmaxf(bm, x[i] < t ? x[i] : -INFINITY);
        bm = block_maxN(bm, red, wave, lane);
      } else {
        __syncthreads();
      }
      v_last = sh_f[1]; i_last = sh_i[0]; n_kept = n;
      if (cert) {
        const float nxt = (n == C) ? bm : sh_f[2];
        c_cut = (n > kk) ? 0.f : v_last - nxt;
        c_p = fminf(sh_f[3], sh_f[4]);
      }
      done = true;
      fast = true;
      if (act) { cand_rank = rank; cand_val = cv; cand_idx = ci; }
    }
  }
  if (!done) {
    unsigned taken = 0u;   // bit i: this thread's element i already extracted (TEXT_PER <= 32)
    double mass_above = 0.0;
    float kth_val = 0.f, nxt = -INFINITY, cum_last = INFINITY, cum_drop = -1.f;
    int n = 0;
    while (n < V) {
      float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
      for (int i = 0; i < TEXT_PER; ++i) {
        const bool avail = !((taken >> i) & 1u);
        if (avail && x[i] > bv) { bv = x[i]; bi = tid + TEXT_NT * i; }   // ascending i => lowest index on ties
      }
      float wv; int wi;
      block_argmaxN(bv, bi, red, wave, lane, wv, wi);
      float cum = INFINITY;
      if (a.use_top_p && n >= 3) {
        cum = (float)(sall - mass_above);
        if (cum <= thr) { nxt = wv; cum_drop = cum; break; }
      }
      if (a.use_top_k && n >= kk && !(wv == kth_val)) { nxt = wv; break; }
      cum_last = cum;
      mass_above += (double)(expf(wv - mx) * rz);
      if ((wi & (TEXT_NT - 1)) == tid) taken |= 1u << (wi / TEXT_NT);
      v_last = wv; i_last = wi;
      if (n == kk - 1) kth_val = wv;
      ++n;
    }
    n_kept = n;
    if (cert) {
      c_cut = (a.use_top_k && n > kk) ? 0.f : v_last - nxt;
      if (cum_last < INFINITY) c_p = fabsf(__logf(fmaxf(cum_last, 1e-38f) / thr));
      if (cum_drop >= 0.f) c_p = fminf(c_p, fabsf(__logf(fmaxf(cum_drop, 1e-38f) / thr)));
    }
  }

  bool mask_eos = gen < a.min_new;
  bool force_eos = false;
  if (a.stop_at != nullptr) {
    const int sa = a.stop_at[b];
    if (sa >= 0) { mask_eos = mask_eos || (gen < sa); force_eos = gen >= sa; }
  }
  const float* qrow = a.q + ((size_t)(gen % a.nq) * a.q_rows + b) * V;
  float wv; int wi;
  float c_arg = INFINITY;
  TSTAMP(4);   // kept set known
  if (dbg && tid == 0) dbg[7] = fast ? 1 : 0;
  if (fast) {
    // the kept tokens ARE candidates: softmax over them and argmax(p / q) on the candidate list, q gathered for the kept tokens only, in
    // ONE parallel round trip (measured: the same gather as conditional loads inside an unrolled loop over the row was ~20 SERIAL misses,
    // 40 of the kernel's 57 us, profiles/r6e_kernel_stats_text.csv).  Every other token has p = 0, hence p / q = 0 for any draw: it can only
    // win when every kept token's p / q underflowed to 0 too, and then the lowest index of the row (0) wins, as in the row-wide argmax.
    const bool live = cand_rank < n_kept && !(mask_eos && cand_idx == a.eos);
    const float qj = live ? qrow[cand_idx] : 1.f;
    const float m2 = block_maxN(live ? cand_val : -INFINITY, red, wave, lane);
    const float z2 = block_sumN(live ? expf(cand_val - m2) : 0.f, red, wave, lane);
    const float rz2 = 1.0f / z2;
    const float r = live ? (expf(cand_val - m2) * rz2) / qj : -1.f;
    block_argmaxN(r, live ? cand_idx : 0x7fffffff, red, wave, lane, wv, wi);
    if (cert) {
      const float r2 = fmaxf(block_maxN((live && cand_idx != wi) ? r : -1.f, red, wave, lane), 0.f);   // (every removed token: p / q = 0)
      c_arg = !(wv > 0.f) ? 0.f : (r2 > 0.f ? __logf(wv / r2) : INFINITY);
    }
    if (!(wv > 0.f)) wi = 0;
  } else {
    float qv[TEXT_PER];
#pragma unroll
    for (int i = 0; i < TEXT_PER; ++i) { const int v = tid + TEXT_NT * i; qv[i] = (v < V) ? qrow[v] : 1.f; }   // one round trip
#define TEXT_LIVE(x_, v_) ((v_) < V && (!any_filter || (n_kept > 0 && ((x_) > v_last || ((x_) == v_last && (v_) <= i_last)))) && !(mask_eos && (v_) == a.eos))
    float m2 = -INFINITY;
#pragma unroll
    for (int i = 0; i < TEXT_PER; ++i) if (TEXT_LIVE(x[i], tid + TEXT_NT * i)) m2 = fmaxf(m2, x[i]);
    m2 = block_maxN(m2, red, wave, lane);
    float z2 = 0.f;
#pragma unroll
    for (int i = 0; i < TEXT_PER; ++i) if (TEXT_LIVE(x[i], tid + TEXT_NT * i)) z2 += expf(x[i] - m2);
    z2 = block_sumN(z2, red, wave, lane);
    const float rz2 = 1.0f / z2;
    float bv = -1.f, bv2 = -1.f; int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < TEXT_PER; ++i) {
      const int v = tid + TEXT_NT * i;
      if (v < V) {
        const float r = (TEXT_LIVE(x[i], v) ? expf(x[i] - m2) * rz2 : 0.f) / qv[i];
        if (r > bv) { bv2 = bv; bv = r; bi = v; }
        else bv2 = fmaxf(bv2, r);
      }
    }
#undef TEXT_LIVE
    block_argmaxN(bv, bi, red, wave, lane, wv, wi);
    if (cert) {
      const float r2 = block_maxN(bi == wi ? bv2 : bv, red, wave, lane);
      c_arg = !(wv > 0.f) ? 0.f : (r2 > 0.f ? __logf(wv / r2) : INFINITY);
    }
  }
  TSTAMP(5);   // token drawn
  if (force_eos) wi = a.eos;
  if (tid < NVQ) a.ids_buf[((size_t)b * a.tcap + len) * NVQ + tid] = (int64_t)wi;   // gpt.py:522-525: replicated over the 4 slots
  if (tid == 0) {
    const bool fin = (a.finish[b] != 0) || (wi == a.eos);
    a.finish[b] = fin ? 1 : 0;
    if (!fin) a.end_idx[b] += 1;
    a.len[b] = len + 1;
    if (cert && !force_eos) a.margin[b] = fminf(a.margin[b], fminf(c_arg, fminf(c_cut, c_p)));
  }
  TSTAMP(6);
#undef TSTAMP
}

hipError_t launch_sample_text(const SampleArgs& a, int V, hipStream_t st) {
  if (V > TEXT_VMAX || a.pow_table != nullptr) return hipErrorInvalidValue;
  CTTS_LAUNCH(sample_text_k, dim3(a.B), dim3(TEXT_NT), st, a, V);
  return hipGetLastError();
}
