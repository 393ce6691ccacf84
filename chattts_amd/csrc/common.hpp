// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits in HBM

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define WAVE 64

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN kept quiet (same rule as torch's .to(bfloat16))
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)0x7fc0;
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

// two floats -> packed bf16 pair (first value in the low half) with gfx950's v_cvt_pk_bf16_f32 (round-to-nearest-even)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// two f32 -> two fp16 (round to nearest even), saturated to the largest finite half: an activation beyond +-65504 must not become inf
__device__ __forceinline__ uint32_t pack_f16x2(float a, float b) {
  a = fminf(fmaxf(a, -65504.f), 65504.f);
  b = fminf(fmaxf(b, -65504.f), 65504.f);
  const _Float16 ha = (_Float16)a, hb = (_Float16)b;
  return (uint32_t)__builtin_bit_cast(uint16_t, ha) | ((uint32_t)__builtin_bit_cast(uint16_t, hb) << 16);
}
// ---- the "f32x3" operand format of the GPT projections (round 6: split-FP16; rounds 5-6a: split-bf16) ------------------------------
// x = hi + lo' * 2^-11 with hi = fp16(x) and lo' = fp16((x - hi) * 2^11): 11 + 11 significant bits (2^-22 relative; the bf16 split of
// round 5 kept 8 + 8, 2^-17).  The residual is scaled into fp16's normal range (an unscaled residual of a 0.02-sized weight would be a
// subnormal); a product is hi*hi on one accumulator and lo'*hi + hi*lo' on a second one that enters the result with 2^-11 (the lo'*lo'
// term, 2^-22 of the product, is dropped).  Same bytes and the same three MFMAs (v_mfma_f32_*_f16) as the bf16 split, 32x closer to
// float32: measured |dlogit| and the divergence rate against the f32 MFMA kernels in DESIGN.md section 2.  Inputs are saturated at the
// fp16 range (+-65504) where they are rounded.
#define X3_LO_SCALE 2048.0f
#define X3_LO_INV (1.0f / 2048.0f)
__device__ __forceinline__ void x3_split(float v, uint16_t& hi, uint16_t& lo) {
  v = v > 65504.0f ? 65504.0f : (v < -65504.0f ? -65504.0f : v);   // (a NaN stays a NaN)
  const _Float16 h = (_Float16)v;
  const _Float16 l = (_Float16)((v - (float)h) * X3_LO_SCALE);
  hi = __builtin_bit_cast(uint16_t, h);
  lo = __builtin_bit_cast(uint16_t, l);
}

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  const bf16x2_t r = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);
  return __builtin_bit_cast(uint32_t, r);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- DPP wave reductions (no LDS crossbar): __shfl_xor lowers to ds_bpermute_b32 (~50+ cycles each, 12 in a
// dependent chain per argmax); the DPP forms below are plain VALU ops.  Pattern: xor-1, xor-2 (quad_perm),
// row_half_mirror, row_mirror reduce each 16-lane row; row_bcast15 / row_bcast31 fold the 4 rows into lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false);  // unwritten lanes keep v
}
__device__ __forceinline__ float wave_max_dpp(float v) {
#define CTTS_FMAX_DPP(C, M) v = fmaxf(v, __int_as_float(dpp_i<C, M>(__float_as_int(v))))
  CTTS_FMAX_DPP(0xB1, 0xf);   // quad_perm [1,0,3,2]
  CTTS_FMAX_DPP(0x4E, 0xf);   // quad_perm [2,3,0,1]
  CTTS_FMAX_DPP(0x141, 0xf);  // row_half_mirror
  CTTS_FMAX_DPP(0x140, 0xf);  // row_mirror
  CTTS_FMAX_DPP(0x142, 0xa);  // row_bcast15 -> rows 1, 3
  CTTS_FMAX_DPP(0x143, 0xc);  // row_bcast31 -> rows 2, 3
#undef CTTS_FMAX_DPP
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_min_dpp(int v) {
#define CTTS_IMIN_DPP(C, M) v = min(v, dpp_i<C, M>(v))
  CTTS_IMIN_DPP(0xB1, 0xf);
  CTTS_IMIN_DPP(0x4E, 0xf);
  CTTS_IMIN_DPP(0x141, 0xf);
  CTTS_IMIN_DPP(0x140, 0xf);
  CTTS_IMIN_DPP(0x142, 0xa);
  CTTS_IMIN_DPP(0x143, 0xc);
#undef CTTS_IMIN_DPP
  return __builtin_amdgcn_readlane(v, 63);
}

// erf-based GELU (nn.GELU default, approximate='none')
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// The same function for an output that is rounded to fp16 anyway (gemm_h1p_k): Phi(x) from Abramowitz-Stegun 7.1.26,
// erfc(z) = (a1 t + ... + a5 t^5) exp(-z^2), t = 1 / (1 + p z), |error| <= 1.5e-7 (3.2e-7 |x| on the result in f32 arithmetic, checked on 2e6 points) -- branch-free, one v_rcp_f32 and one v_exp_f32
// (libm's erff is ~45 VALU instructions per element behind two divergent branches, and the 256 x 256 tile's GELU epilogue was
// costing more than its 8 k stages of fp16 MFMAs).  |gelu_fast - gelu_erf| <= 1.5e-7 |x|, a 2^-11 rounding follows.
__device__ __forceinline__ float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float half_erfc = 0.5f * (p * t) * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);   // 0.5 erfc(|x| / sqrt 2) = Phi(-|x|)
  const float phi = x < 0.f ? half_erfc : 1.0f - half_erfc;
  return x * phi;
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }

struct alignas(16) u128 {
  uint32_t x, y, z, w;
};

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// 16-byte load with the non-temporal hint (`global_load_dwordx4 ... nt`): for operands that ONE workgroup
// reads once (decode weights, KV cache) -- MI355X guide, price-list row "nt-weights": issued->landed -18 %.
__device__ __forceinline__ u128 load16_nt(const void* p) {
  const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
  u128 r;
  r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
  return r;
}
__device__ __forceinline__ u128 load16(const void* p) { return *reinterpret_cast<const u128*>(p); }
// 16-byte load that is coherent at agent scope (`global_load_dwordx4 ... sc1`): data another workgroup of the SAME launch published with
// write-through (sc1) stores -- MI355X guide Guideline 16.  Two 8-byte relaxed agent atomics (the widest the builtin lowers to sc1).
__device__ __forceinline__ u128 load16_sc1(const void* p) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  u128 r;
  r.x = (uint32_t)a; r.y = (uint32_t)(a >> 32); r.z = (uint32_t)b; r.w = (uint32_t)(b >> 32);
  return r;
}
// the XCD this wave runs on (0..7); used for speed only (which copy of a flag to poll), never for correctness
__device__ __forceinline__ unsigned xcc_id() {
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x;
#else
  return 0;
#endif
}

// Fragment-packed bf16 operand order of the decode projections (decode.hip): a [rows][C] matrix is stored as
// [rows/16][C/32][lane = (c%32)/8 * 16 + row%16][c%8], i.e. every (16-row tile, 32-column chunk) is one contiguous KiB in
// exactly the lane order v_mfma_f32_16x16x32_bf16 wants its A / B operand.  kch = C / 32.
__device__ __forceinline__ size_t pk_off(int m, int c, int kch) {
  return ((size_t)((m >> 4) * kch + (c >> 5)) * 64 + (((c & 31) >> 3) << 4) + (m & 15)) * 8 + (c & 7);
}
// The float32 twin (decode32.hip, parity mode): [rows/16][C/16][lane = (c%16)/4 * 16 + row%16][c%4] -- one contiguous KiB per
// (16-row tile, 16-column chunk) in the lane order of four consecutive v_mfma_f32_16x16x4_f32 steps.  kch = C / 16.
__device__ __forceinline__ size_t pk32_off(int m, int c, int kch) {
  return ((size_t)((m >> 4) * kch + (c >> 4)) * 64 + (((c & 15) >> 2) << 4) + (m & 15)) * 4 + (c & 3);
}
// Rotate-half RoPE of one (x[d], x[d+32]) pair in the f32 parity mode, pinned to what hipcc made of rope_append_k's
// `x1 * c - x2 * s` / `x2 * c + x1 * s` when the goldens were established: one rounded product, then one fma.
__device__ __forceinline__ float rope_lo(float x1, float x2, float c, float s) {
#pragma clang fp contract(off)
  const float p = x2 * s;
  return __builtin_fmaf(x1, c, -p);
}
__device__ __forceinline__ float rope_hi(float x1, float x2, float c, float s) {
#pragma clang fp contract(off)
  const float p = x1 * s;
  return __builtin_fmaf(x2, c, p);
}
// RMSNorm statistics of the f32 parity mode.  gemm_skinny_k, gemm_dec32_k (and any later kernel) must produce the SAME BITS for
// 1 / rms of a row, so the operation order is pinned (contraction off, explicit parentheses) instead of left to the optimiser:
// lane l owns columns 4l..4l+3 of every 256-column block; a block contributes ((x*x + y*y) + z*z) + w*w (four IEEE multiplies,
// three adds, no fma -- what hipcc emitted for the plain expression when the goldens were established); blocks are added in
// ascending order; lanes are summed by the xor butterfly of wave_sum (32, 16, ..., 1); rstd = 1 / sqrt(ss / K + eps) with the
// correctly rounded divide and square root hipcc uses by default.
__device__ __forceinline__ float rms_acc4(float ss, const float4 v) {
#pragma clang fp contract(off)   // honoured under hipcc's default -ffp-contract=fast-honor-pragmas: no mul+add fusion in here
  const float xx = v.x * v.x, yy = v.y * v.y, zz = v.z * v.z, ww = v.w * v.w;
  const float t = ((xx + yy) + zz) + ww;
  return ss + t;
}
__device__ __forceinline__ float rms_rstd_of(float ss, int K, float eps) { return 1.0f / sqrtf(ss / (float)K + eps); }
__device__ __forceinline__ float rms_finish(float ss, int K, float eps) { return rms_rstd_of(wave_sum(ss), K, eps); }
// one row, any K (a multiple of 4): every lane returns the value
__device__ __forceinline__ float wave_row_rstd(const float* __restrict__ row, int K, float eps, int lane) {
  float ss = 0.f;
  for (int k = lane * 4; k < K; k += 256) ss = rms_acc4(ss, *reinterpret_cast<const float4*>(row + k));
  return rms_finish(ss, K, eps);
}
// NR rows of K = 768 at once: all 3 NR loads of the lane are in flight together (one memory round trip instead of 3 NR)
template <int NR>
__device__ __forceinline__ void wave_rows_rstd_768(const float* const (&rows)[NR], float eps, int lane, float (&rstd)[NR]) {
  float4 v[NR][3];
#pragma unroll
  for (int q = 0; q < NR; ++q)
#pragma unroll
    for (int t = 0; t < 3; ++t) v[q][t] = *reinterpret_cast<const float4*>(rows[q] + lane * 4 + 256 * t);
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    float ss = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t) ss = rms_acc4(ss, v[q][t]);
    rstd[q] = rms_finish(ss, 768, eps);
  }
}

#ifdef CTTS_PF_BUILD   // probe builds only (python -m chattts_amd.build --variant pf -DCTTS_PF_BUILD=1, env CTTS_PF=<mask>): a recorded negative result
// ---- cross-kernel weight prefetch ------------------------------------------------------------------------------------------
// A decode step is a chain of short, latency-bound kernels; each projection starts with an HBM round trip for weights that nothing
// upstream has touched (phase probe: +0.6 ... 1.5 us per launch with cold weights).  The NEXT kernel's weights are known when the
// current one starts, so one extra wave per workgroup touches them -- one 4-byte load per 64-byte granule, data discarded -- which
// pulls them into the L2 of the XCD this workgroup runs on.  Workgroup b runs on XCD b % 8 (observed placement, used for speed only)
// and weight tile t of the next launch is consumed by a workgroup with linear id = t (mod 8), so a prefetching workgroup on XCD x takes
// its share of the tiles t = x, x + 8, ...  The HBM traffic is the same bytes, moved one kernel earlier, while the pipe is idle.
struct PfDesc {
  const char* base;      // first tile, or null: nothing to prefetch
  unsigned tile_bytes;   // contiguous bytes of one weight tile (a multiple of 64)
  unsigned n_tiles;
};
__device__ __forceinline__ void prefetch_weight_tiles(const PfDesc& pf, int lane, unsigned wg, unsigned n_wg) {
  if (pf.base == nullptr) return;
  const unsigned x = wg & 7u, q = wg >> 3;
  if (x >= pf.n_tiles || x >= n_wg) return;
  const unsigned nq = (n_wg - x + 7u) >> 3;            // workgroups of this launch on XCD x
  const unsigned nt = (pf.n_tiles - x + 7u) >> 3;      // tiles the next launch consumes on XCD x
  const unsigned gpt = pf.tile_bytes >> 6;             // 64-byte granules per tile
  const unsigned G = nt * gpt, per = (G + nq - 1u) / nq;
  const unsigned beg = q * per, end = min(G, beg + per);
  // The loads return asynchronously: their destination register must stay reserved until they have landed, or late data would
  // overwrite whatever the compiler put there next (it believes an asm's output is written when the asm ends).  One sink register,
  // read-write in every asm so that it is live across the loop, and a final wait that consumes it.
  unsigned sink = 0;
  for (unsigned g = beg + (unsigned)lane; g < end; g += 64u) {
    const unsigned ti = g / gpt;
    const char* p = pf.base + (size_t)(x + 8u * ti) * pf.tile_bytes + ((size_t)(g - ti * gpt) << 6);
    asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(p) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink) : : "memory");
}
#endif   // CTTS_PF_BUILD
