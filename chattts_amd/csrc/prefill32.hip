// Prompt-pass projections of the GPT path in the float32 PARITY mode: register-blocked, on the fragment-packed f32 operands of
// decode32.hip, bit-identical to gemm_skinny_k<float> / gemm_dec32_k (the kernels the parity goldens were established with).
//
// Reference op: the four nn.Linear calls of a HF Llama decoder layer over the B x T prompt rows (step 0 of GPT.generate,
// /root/reference/ChatTTS/model/gpt.py:396-427; in-tree twin /root/reference/examples/onnx/modeling_llama.py:415-417, :500, :293),
// with RMSNorm as prologue and RoPE + KV append / residual / SiLU(gate)*up as epilogue.
//
// The arithmetic those kernels define per output element: the 16-wide k chunks c = 0, 1, 2, ... fall into 4 classes c % 4 (the four
// waves of a decode workgroup); each class accumulates its chunks in ascending order, a chunk being four k-ordered
// v_mfma_f32_16x16x4_f32 steps; the four class sums are added ((s0 + s1) + s2) + s3.  A decode workgroup spends a wave per class and
// meets in LDS; with thousands of prompt rows there is no need to split K across waves: here ONE wave keeps all four class
// accumulators of its 32 x 32 output block (2 x 2 MFMA tiles x 4 classes = 64 VGPRs) in registers, walks the chunks in order and adds
// the classes at the end -- same operands, same order, same bits, no LDS, no barrier, and every A / W fragment it loads (one
// contiguous KiB per wave instruction) feeds two MFMA tiles instead of one.  A workgroup is 2 x 2 such waves (64 rows x 64 columns;
// gate/up: 64 rows x 32 output columns, each wave holding the gate AND the up tile of its columns).
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"

// 1 / rms of every prompt row, once per RMSNorm (wave_row_rstd: the pinned summation order of gemm_skinny_k, common.hpp)
__global__ __launch_bounds__(256) void rows_rstd32_k(const float* __restrict__ X, int ldx, int M, float eps, float* __restrict__ rstd) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float r = wave_row_rstd(X + (size_t)row * ldx, 768, eps, lane);
  if (lane == 0) rstd[row] = r;
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_pre32_k(Dec32Args a, const float* __restrict__ rstd) {
  constexpr bool SILU = EPI == EPI_SILU_MUL;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int M = a.M, KCH = a.K >> 4, MT = (M + 15) >> 4;
  const int mt0 = blockIdx.y * 4 + wr * 2;                                   // this wave's first row tile
  // column tiles of this wave's two B fragments: gate/up -> (gate tile t, up tile t); else two neighbouring tiles
  const int ct0 = SILU ? blockIdx.x * 2 + wc : blockIdx.x * 4 + wc * 2;
  const int nt[2] = {ct0, SILU ? ct0 + (a.N >> 4) : ct0 + 1};
  const u128* ap[2];
  const u128* wp[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) ap[i] = reinterpret_cast<const u128*>(a.Ap) + (size_t)min(mt0 + i, MT - 1) * KCH * 64 + lane;
#pragma unroll
  for (int j = 0; j < 2; ++j) wp[j] = reinterpret_cast<const u128*>(a.Wp) + (size_t)nt[j] * KCH * 64 + lane;
  const bool rms = a.norm_w != nullptr;
  float rs[2] = {1.f, 1.f};
  if (rms) {
#pragma unroll
    for (int i = 0; i < 2; ++i) rs[i] = rstd[min((mt0 + i) * 16 + li, M - 1)];
  }

  f32x4 acc[4][2][2];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[c][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  u128 fa[2][2], fw[2][2];   // [buffer][tile]
  float4 nw[2];
  auto load = [&](const int buf, const int c) {
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[buf][i] = load16(ap[i] + (size_t)c * 64);
#pragma unroll
    for (int j = 0; j < 2; ++j) fw[buf][j] = load16(wp[j] + (size_t)c * 64);
    if (rms) nw[buf] = *reinterpret_cast<const float4*>(a.norm_w + c * 16 + g * 4);
  };
  auto mul = [&](const int buf, f32x4 (&cls)[2][2]) {
    float4 a0[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a0[i] = *reinterpret_cast<const float4*>(&fa[buf][i]);
      if (rms) {
        const float s = rs[i];
        a0[i].x = nw[buf].x * (a0[i].x * s); a0[i].y = nw[buf].y * (a0[i].y * s);
        a0[i].z = nw[buf].z * (a0[i].z * s); a0[i].w = nw[buf].w * (a0[i].w * s);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float4 b = *reinterpret_cast<const float4*>(&fw[buf][j]);
        f32x4 c = cls[i][j];
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i].x, b.x, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i].y, b.y, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i].z, b.z, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i].w, b.w, c, 0, 0, 0);
        cls[i][j] = c;
      }
  };
  // chunks in order, four at a time (one per class), the next chunk's fragments requested before the current one is multiplied
  load(0, 0);
  for (int c4 = 0; c4 < KCH; c4 += 4) {
#pragma unroll
    for (int cl = 0; cl < 4; ++cl) {
      const int c = c4 + cl;
      if (c + 1 < KCH) load((cl + 1) & 1, c + 1);
      __builtin_amdgcn_sched_barrier(0);
      mul(cl & 1, acc[cl]);
    }
  }

  // QKV tiles (weight rows permuted by the loader, engine.py rope_row_perm): columns 0..7 of a q/k tile are dims d0..d0+7 of one
  // head, columns 8..15 are dims d0+32..d0+39, so a rotate-half pair sits 8 lanes apart (decode32.hip dec32_body)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = (mt0 + i) * 16 + 4 * g + r;      // C/D map of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + reg
      float v[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) v[j] = ((acc[0][i][j][r] + acc[1][i][j][r]) + acc[2][i][j][r]) + acc[3][i][j][r];   // fixed order
      if (EPI == EPI_SILU_MUL) {
        const int col = ct0 * 16 + li;
        const float o = silu_f(v[0]) * v[1];
        if (row < M) a.Cp[pk32_off(row, col, a.kch_out)] = o;
      } else if (EPI == EPI_RES) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = nt[j] * 16 + li;
          if (row < M) {
            const float o = a.res[(size_t)row * a.ldr + col] + v[j];
            a.C[(size_t)row * a.ldc + col] = o;
            if (a.Cp != nullptr) a.Cp[pk32_off(row, col, a.kch_out)] = o;
          }
        }
      } else {   // D32_EPI_QKV_ROPE
        const RowDesc rd = a.desc[min(row, M - 1)];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n0 = nt[j] * 16;
          const int sect = n0 / 768, hcol = n0 % 768, head = hcol >> 6, t4 = (hcol & 63) >> 4;
          const int dlo = 8 * t4 + (li & 7);
          const float rc = a.cos_t[rd.pos * 32 + dlo], rsn = a.sin_t[rd.pos * 32 + dlo];
          const float other = __shfl_xor(v[j], 8, 64);
          const bool hi = li >= 8;
          const float roped = hi ? rope_hi(other, v[j], rc, rsn) : rope_lo(v[j], other, rc, rsn);
          const int d = dlo + (hi ? 32 : 0);
          if (row < M && rd.b >= 0) {
            const size_t cbase = (((size_t)rd.b * 12 + head) * a.cmax + rd.slot) * 64;
            if (sect == 0) a.C[(size_t)row * a.ldc + head * 64 + d] = roped;
            else if (sect == 1) a.kc[cbase + d] = roped;
            else a.vc[cbase + (hcol & 63) + li] = v[j];
          }
        }
      }
    }
}

hipError_t launch_rows_rstd32(const float* X, int ldx, int M, float eps, float* rstd, hipStream_t st) {
  CTTS_LAUNCH(rows_rstd32_k, dim3((M + 3) / 4), dim3(256), st, X, ldx, M, eps, rstd);
  return hipGetLastError();
}

hipError_t launch_gemm_pre32(const Dec32Args& a, const float* rstd, hipStream_t st) {
  if (a.M <= 0 || (a.K & 63) || a.n_active != nullptr) return hipErrorInvalidValue;
  if (a.norm_w != nullptr && rstd == nullptr) return hipErrorInvalidValue;
  const int my = (a.M + 63) / 64;
  if (a.epi == EPI_SILU_MUL) {
    if ((a.N & 31) || !a.Cp) return hipErrorInvalidValue;
    CTTS_LAUNCH((gemm_pre32_k<EPI_SILU_MUL>), dim3(a.N / 32, my), dim3(256), st, a, rstd);
  } else if (a.epi == EPI_RES) {
    if ((a.N & 63) || !a.res || !a.C) return hipErrorInvalidValue;
    CTTS_LAUNCH((gemm_pre32_k<EPI_RES>), dim3(a.N / 64, my), dim3(256), st, a, rstd);
  } else if (a.epi == D32_EPI_QKV_ROPE) {
    if (a.N != 2304 || a.K != 768 || !a.desc || !a.kc || !a.vc) return hipErrorInvalidValue;
    CTTS_LAUNCH((gemm_pre32_k<D32_EPI_QKV_ROPE>), dim3(a.N / 64, my), dim3(256), st, a, rstd);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
