// Prompt pass of the "f32x3" parity mode (round 6): the four projections of a layer over M = B * T prompt rows on SPLIT-fp16 operands,
// LDS-tiled.  Until round 5 the f32x3 mode ran its prompt on the f32-input MFMA kernels of prefill32.hip (v_mfma_f32_16x16x4_f32, a
// sixteenth of the bf16 rate): 694 us per layer = 13.9 ms per pass at 3072 prompt rows, a fifth of the time to the first streamed chunk.
// Here both operands are read as FLOAT32, row-major (the residual stream / activations the row-major parity path keeps, and the plain
// [N, K] weight matrices the loader keeps for it), split into hi = fp16(x), lo' = fp16((x - hi) 2^11) (common.hpp x3_split: 22
// significant bits) by the tile loader on their way into LDS, and a product is hi*hi on one accumulator and lo'*hi + hi*lo' on a second
// one (x 2^-11) on v_mfma_f32_32x32x16_f16 -- the arithmetic of the mode's decode step (decode32x.hip), no extra copy of any weight.  The RMSNorm gain multiplies the activation BEFORE the split (a' = a * g[k], one f32
// multiply in the loader); the row's 1 / rms (launch_rows_rstd32, the f32 kernels' bits) scales the accumulator.
//   EPI_STORE     C = rstd * acc                      (QKV; RoPE + KV append stay the row-major path's rope_append_k)
//   EPI_RES       C = res + acc                       (o_proj / down_proj)
//   EPI_SILU_MUL  C = silu(rstd * acc_gate) * (rstd * acc_up)      (W = [gate rows; up rows])
// 128 x 128 x 32 tiles (64-row tiles when the launch would not fill the chip), 256 threads = 2 x 2 waves, register-staged global loads
// one tile ahead, two LDS buffers of four planes (A hi | A lo | W hi | W lo), one barrier per k-step.  Reference ops: HF Llama
// q/k/v/o_proj, gate/up/down_proj (examples/onnx/modeling_llama.py:259-295,455-505).
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"

namespace {

constexpr int XBK = 32, XLD = XBK + 8;   // 80-byte LDS rows
constexpr int XNT = 256;

typedef float xf2 __attribute__((ext_vector_type(2)));
typedef _Float16 xb2 __attribute__((ext_vector_type(2)));

// The split runs once per element and k-step in the tile loader, so its VALU cost is on the critical path beside 24 MFMAs per wave:
// hardware conversions, two values per instruction (the first version, a software bf16 conversion of 7 integer ops per value, made the
// loader VALU-bound: 257 us per gate/up launch against 170, profiles/r6h / r6i_kernel_stats_f32x3.csv).
__device__ __forceinline__ void split2(const float a, const float b, uint32_t& hi, uint32_t& lo) {
  // the split-fp16 format of common.hpp x3_split, two values per packed conversion (v_cvt_pk / v_cvt_f16_f32, round to nearest even)
  const xf2 v = {a > 65504.0f ? 65504.0f : (a < -65504.0f ? -65504.0f : a), b > 65504.0f ? 65504.0f : (b < -65504.0f ? -65504.0f : b)};
  const xb2 h = __builtin_convertvector(v, xb2);
  hi = __builtin_bit_cast(uint32_t, h);
  const xf2 hf = __builtin_convertvector(h, xf2);
  const xb2 l = __builtin_convertvector((v - hf) * X3_LO_SCALE, xb2);
  lo = __builtin_bit_cast(uint32_t, l);
}
__device__ __forceinline__ void split4(const float4 v, ushort4& hi, ushort4& lo) {
  uint32_t h0, l0, h1, l1;
  split2(v.x, v.y, h0, l0);
  split2(v.z, v.w, h1, l1);
  hi = __builtin_bit_cast(ushort4, make_uint2(h0, h1));
  lo = __builtin_bit_cast(ushort4, make_uint2(l0, l1));
}

template <int EPI, int MBLK>
__global__ __launch_bounds__(XNT) void gemm_pre_x3_k(GemmArgs a, const float* __restrict__ rstd) {
  constexpr bool SILU = EPI == EPI_SILU_MUL;
  constexpr int BM = 64 * MBLK;
  constexpr int BN = 128;                 // W rows per tile (SILU: 64 gate rows + the 64 matching up rows)
  constexpr int BNO = SILU ? 64 : 128;    // output columns per tile
  __shared__ __attribute__((aligned(16))) uint16_t As[2][2][BM][XLD], Ws[2][2][BN][XLD];   // [buffer][plane hi | lo][row][k]
  __shared__ float rstd_s[BM];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave & 1, wn = wave >> 1;
  const int M = a.M, N = a.N, K = a.K;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BNO;
  const float* Wf = reinterpret_cast<const float*>(a.W);

  if (rstd != nullptr && tid < BM) rstd_s[tid] = rstd[min(m0 + tid, M - 1)];

  // loaders: 8 lanes x 16 B = one 128-byte row segment (32 floats), 32 rows per pass
  const int lr = tid >> 3, lk = (tid & 7) * 4;
  constexpr int PA = BM / 32;
  const float* ap[PA];
  const float* wp[4];
#pragma unroll
  for (int p = 0; p < PA; ++p) ap[p] = a.A + (size_t)min(m0 + lr + 32 * p, M - 1) * a.lda + lk;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = lr + 32 * p;
    const int gr = SILU ? (r < 64 ? n0 + r : N + n0 + (r - 64)) : min(n0 + r, N - 1);
    wp[p] = Wf + (size_t)gr * K + lk;
  }
  float4 ra[PA], rw[4], rg;
#define X_FETCH(K0)                                                                                               \
  do {                                                                                                            \
    _Pragma("unroll") for (int p = 0; p < PA; ++p) ra[p] = *reinterpret_cast<const float4*>(ap[p] + (K0));         \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) rw[p] = *reinterpret_cast<const float4*>(wp[p] + (K0));          \
    rg = a.norm_w != nullptr ? *reinterpret_cast<const float4*>(a.norm_w + (K0) + lk) : make_float4(1.f, 1.f, 1.f, 1.f); \
  } while (0)
#define X_STAGE(SB)                                                                                               \
  do {                                                                                                            \
    _Pragma("unroll") for (int p = 0; p < PA; ++p) {                                                              \
      ushort4 h, l;                                                                                               \
      split4(make_float4(ra[p].x * rg.x, ra[p].y * rg.y, ra[p].z * rg.z, ra[p].w * rg.w), h, l);                  \
      *reinterpret_cast<ushort4*>(&As[SB][0][lr + 32 * p][lk]) = h;                                               \
      *reinterpret_cast<ushort4*>(&As[SB][1][lr + 32 * p][lk]) = l;                                               \
    }                                                                                                             \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                                               \
      ushort4 h, l;                                                                                               \
      split4(rw[p], h, l);                                                                                        \
      *reinterpret_cast<ushort4*>(&Ws[SB][0][lr + 32 * p][lk]) = h;                                               \
      *reinterpret_cast<ushort4*>(&Ws[SB][1][lr + 32 * p][lk]) = l;                                               \
    }                                                                                                             \
  } while (0)

  f32x16 acc[MBLK][2], accx[MBLK][2];   // hi*hi | the cross terms (they enter with 2^-11)
#pragma unroll
  for (int i = 0; i < MBLK; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; accx[i][j][r] = 0.f; }

  X_FETCH(0);
  X_STAGE(0);
  if (XBK < K) X_FETCH(XBK);
  __syncthreads();
  const int ri = lane & 31, kg = (lane >> 5) * 8;
  // W rows of this wave's two 32-column blocks: plain: columns wn*64 + j*32; SILU: j = 0 gate block, j = 1 the up block of the SAME 32
  // output columns (tile rows 64..127 hold the up rows)
  const int wrow0 = SILU ? wn * 32 : wn * 64, wrow1 = SILU ? 64 + wn * 32 : wn * 64 + 32;
  int sb = 0;
  for (int k0 = 0; k0 < K; k0 += XBK) {
    if (k0 + XBK < K) X_STAGE(sb ^ 1);            // tile k+1 (requested one step ago) -> the other buffer
    if (k0 + 2 * XBK < K) X_FETCH(k0 + 2 * XBK);
#pragma unroll
    for (int kk = 0; kk < XBK; kk += 16) {
      const f16x8 wh0 = *reinterpret_cast<const f16x8*>(&Ws[sb][0][wrow0 + ri][kk + kg]);
      const f16x8 wl0 = *reinterpret_cast<const f16x8*>(&Ws[sb][1][wrow0 + ri][kk + kg]);
      const f16x8 wh1 = *reinterpret_cast<const f16x8*>(&Ws[sb][0][wrow1 + ri][kk + kg]);
      const f16x8 wl1 = *reinterpret_cast<const f16x8*>(&Ws[sb][1][wrow1 + ri][kk + kg]);
#pragma unroll
      for (int i = 0; i < MBLK; ++i) {
        const f16x8 ah = *reinterpret_cast<const f16x8*>(&As[sb][0][(wm * MBLK + i) * 32 + ri][kk + kg]);
        const f16x8 al = *reinterpret_cast<const f16x8*>(&As[sb][1][(wm * MBLK + i) * 32 + ri][kk + kg]);
        f32x16 x0 = accx[i][0], x1 = accx[i][1];
        x0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh0, x0, 0, 0, 0);
        x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh1, x1, 0, 0, 0);
        x0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl0, x0, 0, 0, 0);
        x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl1, x1, 0, 0, 0);
        accx[i][0] = x0; accx[i][1] = x1;
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh0, acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh1, acc[i][1], 0, 0, 0);
      }
    }
    __syncthreads();
    sb ^= 1;
  }
#undef X_FETCH
#undef X_STAGE

  // ---- epilogue: lane holds column (lane & 31) of a 32-column block, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----
#pragma unroll
  for (int i = 0; i < MBLK; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] += accx[i][j][r] * X3_LO_INV;
  const int cl = lane & 31;
#pragma unroll
  for (int i = 0; i < MBLK; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lrow = (wm * MBLK + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int row = m0 + lrow;
      if (row >= M) continue;
      const float rs = rstd != nullptr ? rstd_s[lrow] : 1.0f;
      if (EPI == EPI_SILU_MUL) {
        const int col = n0 + wn * 32 + cl;
        a.C[(size_t)row * a.ldc + col] = silu_f(acc[i][0][r] * rs) * (acc[i][1][r] * rs);
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = n0 + wn * 64 + j * 32 + cl;
          if (col >= N) continue;
          const float v = acc[i][j][r] * rs;
          a.C[(size_t)row * a.ldc + col] = (EPI == EPI_RES) ? a.res[(size_t)row * a.ldr + col] + v : v;
        }
      }
    }
  }
}

}  // namespace

// C = epi(rstd[row] * ((A * diag(norm_w)) W^T)) on split-bf16 operands; A [M, lda] f32, W [N (2N for SILU_MUL), K] f32 row-major.
// rstd / norm_w: both or neither (a.norm_w set: launch_rows_rstd32 gives rstd).  K % 32 == 0, lda % 4 == 0, N % 128 == 0 (SILU: % 64).
hipError_t launch_gemm_pre_x3(const GemmArgs& a, const float* rstd, hipStream_t st) {
  if (a.M <= 0 || a.N <= 0 || a.K <= 0 || (a.K % XBK) || (a.lda & 3) || !a.A || !a.W || !a.C || a.wt != WT_F32 || a.taps != 1) return hipErrorInvalidValue;
  if ((a.norm_w != nullptr) != (rstd != nullptr)) return hipErrorInvalidValue;
  if (a.epi == EPI_SILU_MUL ? (a.N % 64) != 0 : (a.N % 128) != 0) return hipErrorInvalidValue;
  if (a.epi == EPI_RES && !a.res) return hipErrorInvalidValue;
  const dim3 block(XNT);
  const int nx = a.epi == EPI_SILU_MUL ? a.N / 64 : a.N / 128;
  static int force_small = -1;
  if (force_small < 0) { const char* e = getenv("CTTS_PREX3_SMALL"); force_small = e ? atoi(e) : 0; }   // A/B: 64-row tiles everywhere (61 KB of LDS: 2 workgroups per CU)
  // 64-row tiles (61 KB of LDS: two workgroups per CU) unless 128-row tiles fill the chip four times over: at the 2048 valid prompt rows of
  // the bench workload QKV 82.8 -> 66.2 us, gate/up 136 -> 127 us (profiles/r6v_ab_prefill_x3_tiles.log); CTTS_PREX3_SMALL=1 / 2: always / never
  const bool small = force_small == 1 || (force_small != 2 && (long)nx * ((a.M + 127) / 128) < 1024);
  const dim3 grid(nx, small ? (a.M + 63) / 64 : (a.M + 127) / 128);
  switch (a.epi) {
    case EPI_SILU_MUL:
      if (small) CTTS_LAUNCH((gemm_pre_x3_k<EPI_SILU_MUL, 1>), grid, block, st, a, rstd); else CTTS_LAUNCH((gemm_pre_x3_k<EPI_SILU_MUL, 2>), grid, block, st, a, rstd);
      break;
    case EPI_RES:
      if (small) CTTS_LAUNCH((gemm_pre_x3_k<EPI_RES, 1>), grid, block, st, a, rstd); else CTTS_LAUNCH((gemm_pre_x3_k<EPI_RES, 2>), grid, block, st, a, rstd);
      break;
    case EPI_STORE:
      if (small) CTTS_LAUNCH((gemm_pre_x3_k<EPI_STORE, 1>), grid, block, st, a, rstd); else CTTS_LAUNCH((gemm_pre_x3_k<EPI_STORE, 2>), grid, block, st, a, rstd);
      break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
