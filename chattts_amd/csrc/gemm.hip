// GEMM kernels for gfx950:  C[M,N] = epi( pro(A)[M,K] * W[N,K]^T ),  A/C float32, W float32 or bf16.
//
//  gemm_skinny : the decode/prefill projection kernel of the GPT path (reference: the nn.Linear calls
//     inside HF LlamaModel.forward reached from /root/reference/ChatTTS/model/gpt.py:419-427 and the
//     heads of gpt.py:438-454).  M is the number of live token rows (<= 64 per tile), so the kernel is
//     weight-streaming bound: one workgroup owns 16 output columns, its 4 waves split K (interleaved
//     chunks, so the workgroup reads 256 contiguous bytes of every W row per iteration), every lane
//     pulls its W fragment straight from HBM with one 16-byte load per chunk (no LDS round trip for an
//     operand that is used once), MFMA 16x16x32 bf16 / 16x16x4 f32 accumulates, the 4 partial tiles are
//     reduced through LDS in a fixed order (deterministic), and RMSNorm / residual / SiLU*up are fused
//     as prologue / epilogue so a decoder layer is 6 launches.
//  gemm_tiled  : 64x64x32 LDS-tiled f32 MFMA (32x32x2) kernel for the large-M dense layers of the
//     acoustic decoder (DVAE /root/reference/ChatTTS/model/dvae.py:145-161, Vocos backbone/head), with
//     conv-as-GEMM gather on the A side and bias / GELU / layer-scale+residual / coef epilogues.
//
// f32-input MFMA is an exact k-ordered fmaf chain on gfx950 (MI355X guide), which is what the f32
// "parity" mode relies on.
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"

// ------------------------------------------------------------------------------------------------
// skinny
// ------------------------------------------------------------------------------------------------
template <typename WT> struct WTraits;
template <> struct WTraits<float>  { static constexpr int EPL = 4; };  // elements per 16-byte lane load
template <> struct WTraits<bf16_t> { static constexpr int EPL = 8; };

template <typename WT, int MB, bool RMS, int EPI>
__global__ __launch_bounds__(256) void gemm_skinny_k(GemmArgs a) {
  constexpr int EPL = WTraits<WT>::EPL;
  constexpr int KC = EPL * 4;  // k covered by one lane-load step of the wave (4 lane groups)
  constexpr int NACC = (EPI == EPI_SILU_MUL) ? 2 : 1;
  constexpr int U = 3;         // chunks in flight per wave
  __shared__ float red[4][NACC][MB][64][4];
  __shared__ float rstd_s[16 * MB];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16 * MB;
  const int N = a.N, K = a.K;
  // decode: only the first *n_active rows exist (compact active utterances); rows beyond are neither read nor written
  const int M = a.n_active ? min(*a.n_active, a.M) : a.M;
  if (m0 >= M) return;

  if (RMS) {
    for (int r = wave; r < 16 * MB; r += 4) {
      const int m = min(m0 + r, M - 1);
      const float rstd = wave_row_rstd(a.A + (size_t)m * a.lda, K, a.eps, lane);
      if (lane == 0) rstd_s[r] = rstd;
    }
    __syncthreads();
  }

  const WT* W = reinterpret_cast<const WT*>(a.W);
  const int n = min(n0 + li, N - 1);
  const WT* wrow = W + (size_t)n * K;
  const WT* wrow2 = wrow + (size_t)N * K;  // "up" rows (EPI_SILU_MUL only)
  const float* arow[MB];
  float rs[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    arow[mb] = a.A + (size_t)min(m0 + 16 * mb + li, M - 1) * a.lda;
    rs[mb] = RMS ? rstd_s[16 * mb + li] : 1.0f;
  }

  f32x4 acc[NACC][MB];
#pragma unroll
  for (int na = 0; na < NACC; ++na)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[na][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nper = K / (KC * 4);  // chunks per wave (launcher guarantees divisibility by U)
  for (int i = 0; i < nper; i += U) {
    u128 wf[NACC][U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int k0 = ((i + j) * 4 + wave) * KC + g * EPL;
      wf[0][j] = *reinterpret_cast<const u128*>(wrow + k0);
      if (NACC == 2) wf[1][j] = *reinterpret_cast<const u128*>(wrow2 + k0);
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int k0 = ((i + j) * 4 + wave) * KC + g * EPL;
      if constexpr (EPL == 8) {
        float4 nw0, nw1;
        if (RMS) {
          nw0 = *reinterpret_cast<const float4*>(a.norm_w + k0);
          nw1 = *reinterpret_cast<const float4*>(a.norm_w + k0 + 4);
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          float4 a0 = *reinterpret_cast<const float4*>(arow[mb] + k0);
          float4 a1 = *reinterpret_cast<const float4*>(arow[mb] + k0 + 4);
          if (RMS) {
            const float s = rs[mb];
            a0.x = nw0.x * (a0.x * s); a0.y = nw0.y * (a0.y * s); a0.z = nw0.z * (a0.z * s); a0.w = nw0.w * (a0.w * s);
            a1.x = nw1.x * (a1.x * s); a1.y = nw1.y * (a1.y * s); a1.z = nw1.z * (a1.z * s); a1.w = nw1.w * (a1.w * s);
          }
          bf16x8 af;
          af[0] = (__bf16)a0.x; af[1] = (__bf16)a0.y; af[2] = (__bf16)a0.z; af[3] = (__bf16)a0.w;
          af[4] = (__bf16)a1.x; af[5] = (__bf16)a1.y; af[6] = (__bf16)a1.z; af[7] = (__bf16)a1.w;
#pragma unroll
          for (int na = 0; na < NACC; ++na) {
            const bf16x8 bfv = *reinterpret_cast<const bf16x8*>(&wf[na][j]);
            acc[na][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfv, acc[na][mb], 0, 0, 0);
          }
        }
      } else {
        float4 nw0;
        if (RMS) nw0 = *reinterpret_cast<const float4*>(a.norm_w + k0);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          float4 a0 = *reinterpret_cast<const float4*>(arow[mb] + k0);
          if (RMS) {
            const float s = rs[mb];
            a0.x = nw0.x * (a0.x * s); a0.y = nw0.y * (a0.y * s); a0.z = nw0.z * (a0.z * s); a0.w = nw0.w * (a0.w * s);
          }
#pragma unroll
          for (int na = 0; na < NACC; ++na) {
            const float4 b = *reinterpret_cast<const float4*>(&wf[na][j]);
            f32x4 c = acc[na][mb];
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b.x, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b.y, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b.z, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b.w, c, 0, 0, 0);
            acc[na][mb] = c;
          }
        }
      }
    }
  }

  // cross-wave (split-K) reduction, fixed order w = 0..3
#pragma unroll
  for (int na = 0; na < NACC; ++na)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][na][mb][lane][r] = acc[na][mb][r];
  __syncthreads();
  if (wave < MB) {
    const int mb = wave;
    const int col = n0 + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + 16 * mb + 4 * g + r;  // C/D map of 16x16 MFMA: col = lane&15, row = 4*(lane>>4)+reg
      float v = ((red[0][0][mb][lane][r] + red[1][0][mb][lane][r]) + red[2][0][mb][lane][r]) + red[3][0][mb][lane][r];
      if (row < M && col < N) {
        if (EPI == EPI_SILU_MUL) {
          const float u = ((red[0][NACC - 1][mb][lane][r] + red[1][NACC - 1][mb][lane][r]) + red[2][NACC - 1][mb][lane][r]) +
                          red[3][NACC - 1][mb][lane][r];
          v = silu_f(v) * u;
        } else if (EPI == EPI_RES) {
          v = a.res[(size_t)row * a.ldr + col] + v;
        }
        a.C[(size_t)row * a.ldc + col] = v;
      }
    }
  }
}

template <typename WT, int MB>
static hipError_t skinny_dispatch(const GemmArgs& a, hipStream_t st) {
  dim3 grid((a.N + 15) / 16, (a.M + 16 * MB - 1) / (16 * MB)), block(256);
  const bool rms = a.norm_w != nullptr;
  if (a.epi == EPI_STORE && rms) CTTS_LAUNCH((gemm_skinny_k<WT, MB, true, EPI_STORE>), grid, block, st, a);
  else if (a.epi == EPI_STORE) CTTS_LAUNCH((gemm_skinny_k<WT, MB, false, EPI_STORE>), grid, block, st, a);
  else if (a.epi == EPI_RES && !rms) CTTS_LAUNCH((gemm_skinny_k<WT, MB, false, EPI_RES>), grid, block, st, a);
  else if (a.epi == EPI_SILU_MUL && rms) CTTS_LAUNCH((gemm_skinny_k<WT, MB, true, EPI_SILU_MUL>), grid, block, st, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_gemm_skinny(const GemmArgs& a, hipStream_t st) {
  const int epl = (a.wt == WT_BF16) ? 8 : 4;
  if (a.M <= 0 || a.N <= 0 || a.K % (epl * 4 * 4 * 3) != 0 || (a.lda % 4) != 0) return hipErrorInvalidValue;
  if (a.wt == WT_BF16) {
    if (a.M <= 16) return skinny_dispatch<bf16_t, 1>(a, st);
    if (a.M <= 32) return skinny_dispatch<bf16_t, 2>(a, st);
    return skinny_dispatch<bf16_t, 4>(a, st);
  }
  if (a.M <= 16) return skinny_dispatch<float, 1>(a, st);
  if (a.M <= 32) return skinny_dispatch<float, 2>(a, st);
  return skinny_dispatch<float, 4>(a, st);
}

// ------------------------------------------------------------------------------------------------
// tiled (f32 weights)
// ------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256) void gemm_tiled_f32_k(GemmArgs a) {
  constexpr int BM = 64, BN = 64, BK = 32, LD = BK + 1;  // +1 pad: conflict-free ds_read_b32 of MFMA fragments
  __shared__ float As[BM][LD];
  __shared__ float Ws[BN][LD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave & 1, wn = wave >> 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int M = a.M, N = a.N, K = a.K;
  const float* W = reinterpret_cast<const float*>(a.W);

  // loader coordinates: 8 lanes cover one 128-byte row segment (32 floats)
  const int lr = tid >> 3, lk = (tid & 7) * 4;
  // per-thread gather bases for its two A rows (rows lr and lr+32)
  int ab[2], af[2];
  bool aval[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int m = m0 + lr + 32 * p;
    aval[p] = m < M;
    if (a.taps > 1) { ab[p] = m / a.frames; af[p] = m - ab[p] * a.frames; } else { ab[p] = 0; af[p] = m; }
  }

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  for (int k0 = 0; k0 < K; k0 += BK) {
    const int k = k0 + lk;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (aval[p] && k < K) {
        if (a.taps > 1) {
          const int tap = k / a.cin, c = k - tap * a.cin;
          const int fs = af[p] + (tap - a.pad) * a.dil;
          if (fs >= 0 && fs < a.frames) v = *reinterpret_cast<const float4*>(a.A + ((size_t)ab[p] * a.frames + fs) * a.lda + c);
        } else {
          v = *reinterpret_cast<const float4*>(a.A + (size_t)af[p] * a.lda + k);
        }
      }
      float* dst = &As[lr + 32 * p][lk];
      dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
      float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
      const int nn = n0 + lr + 32 * p;
      if (nn < N && k < K) w = *reinterpret_cast<const float4*>(W + (size_t)nn * K + k);
      float* dw = &Ws[lr + 32 * p][lk];
      dw[0] = w.x; dw[1] = w.y; dw[2] = w.z; dw[3] = w.w;
    }
    __syncthreads();
    const int ri = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float av = As[wm * 32 + ri][kk + kh];
      const float bv = Ws[wn * 32 + ri][kk + kh];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    __syncthreads();
  }

  const int col = n0 + wn * 32 + (lane & 31);
  if (col < N) {
    float bias = 0.f, gam = 1.f;
    if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_SCALE_RES) bias = a.bias[col];
    if (EPI == EPI_BIAS_SCALE_RES || EPI == EPI_SCALE || EPI == EPI_LOG_DIV) gam = a.gamma[col];
    // the residual is updated IN PLACE (C == res): read element by element, every load would wait for the previous store (may-alias),
    // one memory round trip per element.  A thread reads and writes only its own elements: its 16 residuals are requested first.
    float rv[16];
    if (EPI == EPI_BIAS_SCALE_RES || EPI == EPI_RES) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        rv[r] = a.res[(size_t)min(row, M - 1) * a.ldr + col];
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);  // C/D map of 32x32 MFMA
      if (row < M) {
        float v = acc[r];
        if (EPI == EPI_BIAS) v = v + bias;
        else if (EPI == EPI_BIAS_GELU) v = gelu_erf(v + bias);
        else if (EPI == EPI_BIAS_SCALE_RES) v = rv[r] + gam * (v + bias);
        else if (EPI == EPI_RES) v = rv[r] + v;
        else if (EPI == EPI_SCALE) v = v * gam;
        else if (EPI == EPI_LOG_DIV) v = logf(fmaxf(v, 1e-5f)) / gam;
        a.C[(size_t)row * a.ldc + col] = v;
      }
    }
  }
}

hipError_t launch_gemm_tiled(const GemmArgs& a, hipStream_t st) {
  if (a.wt != WT_F32 || a.K % 4 != 0 || a.lda % 4 != 0 || a.norm_w != nullptr) return hipErrorInvalidValue;
  if (a.taps > 1 && (a.cin % 4 != 0 || a.K != a.taps * a.cin)) return hipErrorInvalidValue;
  dim3 grid((a.N + 63) / 64, (a.M + 63) / 64), block(256);
  switch (a.epi) {
    case EPI_STORE: CTTS_LAUNCH((gemm_tiled_f32_k<EPI_STORE>), grid, block, st, a); break;
    case EPI_RES: CTTS_LAUNCH((gemm_tiled_f32_k<EPI_RES>), grid, block, st, a); break;
    case EPI_BIAS: CTTS_LAUNCH((gemm_tiled_f32_k<EPI_BIAS>), grid, block, st, a); break;
    case EPI_BIAS_GELU: CTTS_LAUNCH((gemm_tiled_f32_k<EPI_BIAS_GELU>), grid, block, st, a); break;
    case EPI_BIAS_SCALE_RES: CTTS_LAUNCH((gemm_tiled_f32_k<EPI_BIAS_SCALE_RES>), grid, block, st, a); break;
    case EPI_SCALE: CTTS_LAUNCH((gemm_tiled_f32_k<EPI_SCALE>), grid, block, st, a); break;
    case EPI_LOG_DIV: CTTS_LAUNCH((gemm_tiled_f32_k<EPI_LOG_DIV>), grid, block, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// fast bf16 path (perf mode): bf16 activations in, every load of a round issued before the first
// MFMA, operands the epilogue needs (residual, RoPE position/cos/sin, row sums of squares) requested
// at kernel entry so that no dependent memory round trip is left on the tail of the kernel.
// ------------------------------------------------------------------------------------------------
template <int MB, int NW, bool SCALE, int EPI>
__global__ __launch_bounds__(64 * (NW + (EPI == FEPI_QKV_ROPE ? 1 : 0)))
void gemm_fast_k(FastGemmArgs a) {
  constexpr int NACC = (EPI == FEPI_SILU) ? 2 : 1;
  constexpr int U = 6;    // k-chunks (of 32) per wave per round; every lane keeps U*(NACC+MB) 16-byte loads in flight
  constexpr int KC = 32;
  __shared__ float red[NW][NACC][MB][64][4];
  __shared__ float rstd_s[16 * MB];
  __shared__ float cs_s[(EPI == FEPI_QKV_ROPE) ? 16 * MB : 1][16];   // per row: cos[8], sin[8] of this tile's dims
  __shared__ int meta_s[(EPI == FEPI_QKV_ROPE) ? 16 * MB : 1][2];    // per row: b, slot

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16 * MB;
  const int N = a.N, K = a.K;
  // decode: only the first *n_active rows exist (compact active utterances, see GptRowMap); rows beyond are neither
  // loaded (their A fragments clamp onto the last live row: L1 hits) nor stored, and whole M tiles beyond exit
  const int M = a.n_active ? min(*a.n_active, a.M) : a.M;
  if (m0 >= M) return;
  // optional phase stamps (tools/gemm_phase_probe.py): 100 MHz s_memrealtime, wave 0 lane 0 of every workgroup
  long long* dbg = a.dbg ? a.dbg + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;
#define STAMP(i) do { if (dbg && tid == 0) dbg[i] = wall_clock64(); } while (0)
  STAMP(0);
  // ROPE tiles (weights permuted by the loader): columns 0..7 of a q/k tile are dims d0..d0+7 of one
  // head, columns 8..15 are dims d0+32..d0+39, so a rotate-half pair sits 8 lanes apart.
  const int sect = n0 / 768, hcol = n0 % 768, head = hcol >> 6, t4 = (hcol & 63) >> 4;
  const int dlo = 8 * t4 + (li & 7);

  if (EPI == FEPI_QKV_ROPE && wave == NW) {
    // helper wave: the dependent chain len -> position -> cos/sin runs here, beside the main waves' load
    // phase (a vector-memory wait in a main wave would sit behind its 30+ operand loads: vmcnt is in-order)
    if (lane < 16 * MB) {
      const int row = min(m0 + lane, M - 1);
      int b, slot;
      if (a.q_per_b == 1) { b = a.row_map ? a.row_map[row] : row; slot = a.len[b] - 1; }
      else { b = row / a.q_per_b; slot = a.slot0 + row - b * a.q_per_b; if (a.row_map) b = a.row_map[b]; }
      int pos = slot - a.kv_start[b];
      if (pos < 0) pos = 1;
      const float4 c0 = *reinterpret_cast<const float4*>(a.cos_t + pos * 32 + 8 * t4);
      const float4 c1 = *reinterpret_cast<const float4*>(a.cos_t + pos * 32 + 8 * t4 + 4);
      const float4 s0 = *reinterpret_cast<const float4*>(a.sin_t + pos * 32 + 8 * t4);
      const float4 s1 = *reinterpret_cast<const float4*>(a.sin_t + pos * 32 + 8 * t4 + 4);
      float* o = cs_s[lane];
      o[0] = c0.x; o[1] = c0.y; o[2] = c0.z; o[3] = c0.w; o[4] = c1.x; o[5] = c1.y; o[6] = c1.z; o[7] = c1.w;
      o[8] = s0.x; o[9] = s0.y; o[10] = s0.z; o[11] = s0.w; o[12] = s1.x; o[13] = s1.y; o[14] = s1.z; o[15] = s1.w;
      meta_s[lane][0] = b; meta_s[lane][1] = slot;
    }
    __syncthreads();
    return;
  }

  // per-row sum of squares: 4 threads x 12 partials per row, consumed only in the epilogue
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0;
  const int srow = tid >> 2, spart = tid & 3;
  if (SCALE && srow < 16 * MB) {
    const float* sp = a.ssq_in + (size_t)min(m0 + srow, M - 1) * SSQ_PARTS + spart * 12;
    s0 = *reinterpret_cast<const float4*>(sp);
    s1 = *reinterpret_cast<const float4*>(sp + 4);
    s2 = *reinterpret_cast<const float4*>(sp + 8);
  }

  // epilogue operands of the finishing waves (wave mb finishes m-block mb)
  // Finishing work = 4*MB (m-block, accumulator register) pairs of 64 outputs each, dealt round-robin to the
  // first NF waves, so the LDS reduction + epilogue of a tile is spread over up to 4 waves instead of MB.
  constexpr int NPAIR = 4 * MB;
  constexpr int NF = NW < 4 ? NW : 4;
  constexpr int PPW = NPAIR / NF;        // pairs per finishing wave
  float pre0[PPW];                        // RES: residual, requested before the operand loads
#pragma unroll
  for (int q = 0; q < PPW; ++q) pre0[q] = 0.f;
  if (EPI == FEPI_RES && wave < NF) {
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
      const int p = wave + q * NF, mb = p >> 2, r = p & 3;
      const int row = min(m0 + 16 * mb + 4 * g + r, M - 1);
      pre0[q] = a.C32[(size_t)row * a.ldc + min(n0 + li, N - 1)];
    }
  }

  const int n = min(n0 + li, N - 1);
  const uint16_t* wrow = a.W + (size_t)n * K + g * 8;
  const uint16_t* wrow2 = wrow + (size_t)N * K;
  const uint16_t* arow[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) arow[mb] = a.A + (size_t)min(m0 + 16 * mb + li, M - 1) * a.lda + g * 8;

  f32x4 acc[NACC][MB];
#pragma unroll
  for (int na = 0; na < NACC; ++na)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[na][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nper = K / (KC * NW);
  const bool w_once = gridDim.y == 1 && a.w_nt;
  for (int i = 0; i < nper; i += U) {
    u128 wf[NACC][U], af[MB][U];
    // W is streamed once when a single M tile covers all rows (decode): non-temporal then; when several M
    // tiles re-read it (o/down at 16-row tiles, prefill) the normal policy keeps it in L2 for the others
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int k0 = (wave * nper + i + j) * KC;  // contiguous K range per wave: whole 128-B lines of a row stay in one wave
      if (w_once) {
        wf[0][j] = load16_nt(wrow + k0);
        if (NACC == 2) wf[1][j] = load16_nt(wrow2 + k0);
      } else {
        wf[0][j] = load16(wrow + k0);
        if (NACC == 2) wf[1][j] = load16(wrow2 + k0);
      }
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int k0 = (wave * nper + i + j) * KC;  // contiguous K range per wave: whole 128-B lines of a row stay in one wave
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) af[mb][j] = *reinterpret_cast<const u128*>(arow[mb] + k0);
    }
    // keep every load of the round in flight: hipcc otherwise sinks the loads next to their MFMA and
    // waits vmcnt(1) per fragment (one L2/HBM round trip per MFMA pair)
    __builtin_amdgcn_sched_barrier(0);
    if (i == 0) STAMP(1);
#pragma unroll
    for (int j = 0; j < U; ++j)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int na = 0; na < NACC; ++na)
          acc[na][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&af[mb][j]),
                                                                *reinterpret_cast<const bf16x8*>(&wf[na][j]), acc[na][mb], 0, 0, 0);
  }

  STAMP(2);
  if (SCALE) {
    float s = (((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w))) + ((s2.x + s2.y) + (s2.z + s2.w));
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (spart == 0 && srow < 16 * MB) rstd_s[srow] = 1.0f / sqrtf(s / 768.0f + a.eps);
  }
#pragma unroll
  for (int na = 0; na < NACC; ++na)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][na][mb][lane][r] = acc[na][mb][r];
  __syncthreads();
  STAMP(3);
  if (wave < NF) {
    const int col = n0 + li;
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
      const int p = wave + q * NF, mb = p >> 2, r = p & 3;
      const int row = m0 + 16 * mb + 4 * g + r;
      float v = 0.f, u = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += red[w][0][mb][lane][r];
      if (EPI == FEPI_SILU) {
#pragma unroll
        for (int w = 0; w < NW; ++w) u += red[w][NACC - 1][mb][lane][r];
      }
      const bool ok = row < M && col < N;
      if (SCALE) {
        const float rs = rstd_s[16 * mb + 4 * g + r];
        v *= rs;
        u *= rs;
      }
      if (EPI == FEPI_STORE32) {
        if (ok) a.C32[(size_t)row * a.ldc + col] = v;
      } else if (EPI == FEPI_SILU) {
        if (ok) a.Cb[(size_t)row * a.ldcb + col] = f32_to_bf16(silu_f(v) * u);
      } else if (EPI == FEPI_RES) {
        float xn = 0.f;
        if (ok) {
          xn = pre0[q] + v;
          a.C32[(size_t)row * a.ldc + col] = xn;
          a.Cb[(size_t)row * a.ldcb + col] = f32_to_bf16(xn);
        }
        float sq = xn * xn;
        sq += __shfl_xor(sq, 1, 64);
        sq += __shfl_xor(sq, 2, 64);
        sq += __shfl_xor(sq, 4, 64);
        sq += __shfl_xor(sq, 8, 64);
        if (li == 0 && row < M) a.ssq_out[(size_t)row * SSQ_PARTS + blockIdx.x] = sq;
      } else {  // FEPI_QKV_ROPE: q -> roped, in the f32 qkv buffer; k -> roped, KV cache; v -> KV cache
        const float other = __shfl_xor(v, 8, 64);
        const bool hi = li >= 8;
        const int lr = 16 * mb + 4 * g + r;
        const float cc = cs_s[lr][li & 7], ss = cs_s[lr][8 + (li & 7)];
        // rotate-half: out[d] = x[d] c - x[d+32] s ; out[d+32] = x[d+32] c + x[d] s
        const float roped = hi ? (v * cc + other * ss) : (v * cc - other * ss);
        const int d = dlo + (hi ? 32 : 0);
        if (ok) {
          const size_t cbase = (((size_t)meta_s[lr][0] * 12 + head) * a.cmax + meta_s[lr][1]) * 64;
          if (sect == 0) a.C32[(size_t)row * a.ldc + head * 64 + d] = roped;
          else if (sect == 1) a.kc[cbase + d] = f32_to_bf16(roped);
          else a.vc[cbase + (hcol & 63) + li] = f32_to_bf16(v);
        }
      }
    }
  }
  STAMP(4);
#undef STAMP
}

template <int MB>
static hipError_t fast_dispatch(const FastGemmArgs& a, hipStream_t st) {
  dim3 grid((a.N + 15) / 16, (a.M + 16 * MB - 1) / (16 * MB));
  const bool scale = a.ssq_in != nullptr;
  if (a.K == 768) {
    if (a.epi == FEPI_STORE32 && scale) CTTS_LAUNCH((gemm_fast_k<MB, 4, true, FEPI_STORE32>), grid, dim3(256), st, a);
    else if (a.epi == FEPI_QKV_ROPE && scale) CTTS_LAUNCH((gemm_fast_k<MB, 4, true, FEPI_QKV_ROPE>), grid, dim3(320), st, a);
    else if (a.epi == FEPI_SILU && scale) CTTS_LAUNCH((gemm_fast_k<MB, 4, true, FEPI_SILU>), grid, dim3(256), st, a);
    else if (a.epi == FEPI_RES && !scale) CTTS_LAUNCH((gemm_fast_k<MB, 4, false, FEPI_RES>), grid, dim3(256), st, a);
    else return hipErrorInvalidValue;
  } else if (a.K == 3072) {
    if (a.epi == FEPI_RES && !scale) {
      // one load round per wave when the tile is a single m-block (decode): 16 waves x 6 chunks = K
      if (MB == 1) CTTS_LAUNCH((gemm_fast_k<1, 16, false, FEPI_RES>), grid, dim3(1024), st, a);
      else CTTS_LAUNCH((gemm_fast_k<MB, 8, false, FEPI_RES>), grid, dim3(512), st, a);
    } else return hipErrorInvalidValue;
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_gemm_fast(const FastGemmArgs& a_in, hipStream_t st) {
  FastGemmArgs a = a_in;
  {
    static int nt = -1;
    if (nt < 0) { const char* e = getenv("CTTS_W_NT"); nt = e ? atoi(e) : 1; }
    a.w_nt = nt;
  }
  if (a.M <= 0 || a.N <= 0 || (a.lda % 8) != 0) return hipErrorInvalidValue;
  if (a.epi == FEPI_RES && a.N != 16 * SSQ_PARTS) return hipErrorInvalidValue;
  if (a.epi == FEPI_QKV_ROPE && (a.N != 2304 || a.K != 768)) return hipErrorInvalidValue;
  {
    // prompt-sized M: the LDS-tiled kernel of prefill.hip (this one is a weight-streaming shape for M <= 64 and
    // re-reads its operands from L2 for every 16 output columns when there are many M tiles)
    static int tiled = -1;   // CTTS_PREFILL_TILED=0: keep gemm_fast_k for every M (A/B)
    if (tiled < 0) { const char* e = getenv("CTTS_PREFILL_TILED"); tiled = e ? atoi(e) : 1; }
    if (tiled && a.M >= 256 && !a.force_mb && gemm_prefill_supported(a)) return launch_gemm_prefill(a, st);
  }
  // rows per workgroup: the per-workgroup latency is set by fixed round trips, not bytes, so prefer
  // enough workgroups to cover the 256 CUs over big M tiles (decode: M <= 64)
  // measured (tools/gemm_phase_probe.py): these kernels are bound by each CU's vector-memory path, so pick
  // the M tile that minimises  ceil(workgroups / 256 CUs) x (W tile + A tile bytes)  of the busiest CU
  const int ntiles = (a.N + 15) / 16;
  const int nacc = a.epi == FEPI_SILU ? 2 : 1;
  int mb = 1;
  long best = -1;
  for (int c = 1; c <= 4; c <<= 1) {
    if (c > 1 && a.M <= 16 * (c / 2)) break;
    const long wgs = (long)ntiles * ((a.M + 16 * c - 1) / (16 * c));
    const long per = (long)(16 * nacc + 16 * c) * a.K * 2;
    const long cost = ((wgs + 255) / 256) * per;
    if (best < 0 || cost < best) { best = cost; mb = c; }
  }
  if (a.force_mb) mb = a.force_mb;
  {  // tuning hook: CTTS_MB_<epi> = 1|2|4 overrides the heuristic for that epilogue kind (read once)
    static int env_mb[4] = {-1, -1, -1, -1};
    if (env_mb[0] < 0) {
      const char* names[4] = {"CTTS_MB_STORE", "CTTS_MB_RES", "CTTS_MB_SILU", "CTTS_MB_QKV"};
      for (int i = 0; i < 4; ++i) { const char* e = getenv(names[i]); env_mb[i] = e ? atoi(e) : 0; }
    }
    int e = env_mb[a.epi & 3];
    if (a.epi == FEPI_RES && a.K == 3072) { static int dn = -1; if (dn < 0) { const char* v = getenv("CTTS_MB_DOWN"); dn = v ? atoi(v) : 0; } e = dn; }
    if ((e == 1 || e == 2 || e == 4) && a.M > 16 * (e / 2)) mb = e;
  }
  if (mb == 1) return fast_dispatch<1>(a, st);
  if (mb == 2) return fast_dispatch<2>(a, st);
  return fast_dispatch<4>(a, st);
}

// ------------------------------------------------------------------------------------------------
// tiled, split-bf16 ("bf16x3"): near-f32 accuracy at the bf16 MFMA rate for the acoustic decoder's
// dense layers.  Every f32 operand x is written as x = hi + lo with hi = bf16(x), lo = bf16(x - hi)
// (residual <= 2^-17 |x|), and a.w ~= a_hi w_hi + a_hi w_lo + a_lo w_hi (the dropped lo.lo term is
// <= 2^-16 relative), three v_mfma_f32_32x32x16_bf16 per product, f32 accumulation.  Weights are split
// once at load ([N][Kp/32][2][32] bf16: per row and 32-wide k block the hi values then the lo values -- one k-step of
// a row is one 128-byte line --, K zero-padded to a multiple of 32), activations
// are split while they are staged into LDS.  128x128x32 tile, 4 waves as 2x2, each wave 2x2 MFMA blocks.
// f32-input MFMA peaks at 157 TF on gfx950 and the f32 tiled kernel above already runs at ~115 TF, so
// this is the only way to make the 157 MFLOP/token decoder cheaper without giving up the 1e-4 RMS bar.
// ------------------------------------------------------------------------------------------------
// WM x WN waves, each wave MBLK x NBLK MFMA blocks of 32x32: tile = (WM*MBLK*32) x (WN*NBLK*32).
//   <2,2,2,2>: 128x128, 256 threads (narrow layers);  <4,2,2,4>: 256x256, 512 threads, one workgroup per CU --
//   the kernel is bound by each CU's L1 fill, and a 256x256x32 step moves 64 KB for 4x the flops of a 128x128 one.
// NBUF = 2 (256x256 tile only: 2 x 80 KB = the CU's whole 160 KB LDS): the staged tile k+1 goes to the other LDS buffer
// while tile k is multiplied, so a k-step has ONE barrier and the waves drift apart -- one wave of a SIMD converts /
// writes LDS / sits blocked on its load issue while the other one feeds the MFMA pipe.
template <int EPI, int WM, int WN, int MBLK, int NBLK, int NBUF = 1>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 4) ? 2 : 1) void gemm_tiled_bf16x3_k(GemmArgs a) {
  constexpr int NT = 64 * WM * WN;
  constexpr int BM = WM * MBLK * 32, BN = WN * NBLK * 32, BK = 32, LD = 40;  // LD: 80-byte rows -> conflict-free ds_read_b128
  constexpr int PA = BM / (NT / 8);   // A loader passes: NT/8 rows per pass (8 lanes = one 128-byte row segment)
  constexpr int PW = BN / (NT / 8);   // W loader passes: NT/8 rows per pass (8 lanes = one 128-byte line: hi 64 B | lo 64 B)
  __shared__ __attribute__((aligned(16))) uint16_t Ah[NBUF][BM][LD], Al[NBUF][BM][LD], Wh[NBUF][BN][LD], Wl[NBUF][BN][LD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave % WM, wn = wave / WM;
  const int M = a.M, N = a.N, K = a.K;
  // XCD-aware tile order: the dispatcher places workgroup L on XCD L % 8 (each XCD has its own L2), so give
  // every XCD a contiguous run of tiles; the N-tiles that share one A row-panel then hit the same L2.
  const int nx = (N + BN - 1) / BN, ny = (M + BM - 1) / BM, T = nx * ny, per = (T + 7) / 8;
  const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || t >= T) return;
  const int m0 = (t / nx) * BM, n0 = (t % nx) * BN;
  const int Kp = (K + 31) & ~31;
  const uint16_t* Wb = reinterpret_cast<const uint16_t*>(a.W);   // [N][Kp/32][2][32]: per row and k block, hi then lo

  const int ar = tid >> 3, ak = (tid & 7) * 4;
  int ab[PA], af[PA];
  bool aval[PA];
#pragma unroll
  for (int p = 0; p < PA; ++p) {
    const int m = m0 + ar + (NT / 8) * p;
    aval[p] = m < M;
    if (a.taps > 1) { ab[p] = m / a.frames; af[p] = m - ab[p] * a.frames; } else { ab[p] = 0; af[p] = m; }
  }
  const int wr = tid >> 3, wk = (tid & 3) * 8, wlo = (tid >> 2) & 1;   // lanes 0-3 of a row: hi chunks, lanes 4-7: lo chunks

  f32x16 acc[MBLK][NBLK];
#pragma unroll
  for (int i = 0; i < MBLK; ++i)
#pragma unroll
    for (int j = 0; j < NBLK; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // register staging (one tile ahead): the global loads of tile k+1 are in flight while tile k is multiplied
  float4 ra[PA];
  u128 rw[PW];
#define X3_FETCH_A(P, K0)                                                                                   \
  do {                                                                                                      \
    const int p_ = (P), k = (K0) + ak;                                                                      \
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                             \
    if (aval[p_] && k < K) {                                                                                \
      if (a.taps > 1) {                                                                                     \
        const int tap = k / a.cin, c = k - tap * a.cin;                                                     \
        const int fs = af[p_] + (tap - a.pad) * a.dil;                                                      \
        if (fs >= 0 && fs < a.frames)                                                                       \
          v = *reinterpret_cast<const float4*>(a.A + ((size_t)ab[p_] * a.frames + fs) * a.lda + c);         \
      } else {                                                                                              \
        v = *reinterpret_cast<const float4*>(a.A + (size_t)af[p_] * a.lda + k);                             \
      }                                                                                                     \
    }                                                                                                       \
    ra[p_] = v;                                                                                             \
  } while (0)
#define X3_FETCH_W(P, K0)                                                                                   \
  do {                                                                                                      \
    const int p_ = (P);                                                                                     \
    const int nn = min(n0 + wr + (NT / 8) * p_, N - 1); /* clamped: rows >= N are never stored */           \
    rw[p_] = *reinterpret_cast<const u128*>(Wb + ((size_t)nn * Kp + (K0)) * 2 + wlo * 32 + wk);             \
  } while (0)
#define X3_FETCH(K0)                                                                                        \
  do {                                                                                                      \
    _Pragma("unroll") for (int p = 0; p < PA; ++p) X3_FETCH_A(p, (K0));                                     \
    _Pragma("unroll") for (int p = 0; p < PW; ++p) X3_FETCH_W(p, (K0));                                     \
  } while (0)

#define X3_STAGE(SB)                                                                                        \
  do {                                                                                                      \
    const int sb_ = (SB);                                                                                   \
    _Pragma("unroll") for (int p = 0; p < PA; ++p) {                                                        \
      const float4 v = ra[p];                                                                               \
      /* x = hi + lo: hi = bf16(x), lo = bf16(x - hi), two values per v_cvt_pk_bf16_f32 */                  \
      const uint32_t h01 = pack_bf16x2(v.x, v.y), h23 = pack_bf16x2(v.z, v.w);                              \
      const uint32_t l01 = pack_bf16x2(v.x - __uint_as_float(h01 << 16), v.y - __uint_as_float(h01 & 0xffff0000u)); \
      const uint32_t l23 = pack_bf16x2(v.z - __uint_as_float(h23 << 16), v.w - __uint_as_float(h23 & 0xffff0000u)); \
      *reinterpret_cast<uint2*>(&Ah[sb_][ar + (NT / 8) * p][ak]) = make_uint2(h01, h23);                    \
      *reinterpret_cast<uint2*>(&Al[sb_][ar + (NT / 8) * p][ak]) = make_uint2(l01, l23);                    \
    }                                                                                                       \
    _Pragma("unroll") for (int p = 0; p < PW; ++p) {                                                        \
      *reinterpret_cast<u128*>(wlo ? &Wl[sb_][wr + (NT / 8) * p][wk] : &Wh[sb_][wr + (NT / 8) * p][wk]) = rw[p]; \
    }                                                                                                       \
  } while (0)

#ifdef CTTS_X3_PROBE
  // phase probe (tools/x3_phase_probe.py): wave 0 accumulates s_memrealtime deltas: fetch issue, MFMA, barrier, stage, barrier
  long long tacc[5] = {0, 0, 0, 0, 0}, tprev = wall_clock64();
#define X3_MARK(i) do { if (a.dbg && wave == 0) { const long long tn = wall_clock64(); tacc[i] += tn - tprev; tprev = tn; } } while (0)
#else
#define X3_MARK(i) do { } while (0)
#endif
  X3_FETCH(0);
  X3_STAGE(0);
  if (NBUF == 2 && BK < Kp) X3_FETCH(BK);
  __syncthreads();
  X3_MARK(3);
  const int ri = lane & 31, kg = (lane >> 5) * 8;
  int sb = 0;
  for (int k0 = 0; k0 < Kp; k0 += BK) {
    const bool more = k0 + BK < Kp;
    const bool more2 = k0 + 2 * BK < Kp;
    if (NBUF == 2) {
      // registers hold tile k+1 (requested one whole step ago): park it in the other buffer, then request tile k+2
      if (more) X3_STAGE(sb ^ 1);
      X3_MARK(3);
      if (more2) X3_FETCH(k0 + 2 * BK);   // in bulk: spreading these loads over the MFMA phase measured 12 % slower
    } else {
      if (more) X3_FETCH(k0 + BK);
    }
    X3_MARK(0);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      bf16x8 fwh[NBLK], fwl[NBLK];
#pragma unroll
      for (int j = 0; j < NBLK; ++j) {
        fwh[j] = *reinterpret_cast<const bf16x8*>(&Wh[sb][(wn * NBLK + j) * 32 + ri][kk + kg]);
        fwl[j] = *reinterpret_cast<const bf16x8*>(&Wl[sb][(wn * NBLK + j) * 32 + ri][kk + kg]);
      }
#pragma unroll
      for (int i = 0; i < MBLK; ++i) {
        const bf16x8 fah = *reinterpret_cast<const bf16x8*>(&Ah[sb][(wm * MBLK + i) * 32 + ri][kk + kg]);
        const bf16x8 fal = *reinterpret_cast<const bf16x8*>(&Al[sb][(wm * MBLK + i) * 32 + ri][kk + kg]);
#pragma unroll
        for (int j = 0; j < NBLK; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal, fwh[j], acc[i][j], 0, 0, 0);  // small terms first
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah, fwl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah, fwh[j], acc[i][j], 0, 0, 0);
        }
      }
    }
    X3_MARK(1);
    __syncthreads();
    X3_MARK(2);
    if (NBUF == 2) {
      sb ^= 1;
    } else if (more) {
      X3_STAGE(0);
      X3_MARK(3);
      __syncthreads();
      X3_MARK(4);
    }
  }
#ifdef CTTS_X3_PROBE
  if (a.dbg && tid == 0) {
    long long* d = a.dbg + (size_t)blockIdx.x * 8;
    for (int i = 0; i < 5; ++i) d[i] = tacc[i];
  }
#endif

#pragma unroll
  for (int j = 0; j < NBLK; ++j) {
    const int col = n0 + (wn * NBLK + j) * 32 + (lane & 31);
    if (col >= N) continue;
    float bias = 0.f, gam = 1.f;
    if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_SCALE_RES) bias = a.bias[col];
    if (EPI == EPI_BIAS_SCALE_RES || EPI == EPI_SCALE) gam = a.gamma[col];
    float rv[MBLK][16];   // in place (C == res): a column block's residuals are requested before its results are stored (see gemm_tiled_k)
    if (EPI == EPI_BIAS_SCALE_RES || EPI == EPI_RES) {
#pragma unroll
      for (int i = 0; i < MBLK; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + (wm * MBLK + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          rv[i][r] = a.res[(size_t)min(row, M - 1) * a.ldr + col];
        }
    }
#pragma unroll
    for (int i = 0; i < MBLK; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wm * MBLK + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M) {
          float v = acc[i][j][r];
          if (EPI == EPI_BIAS) v = v + bias;
          else if (EPI == EPI_BIAS_GELU) v = gelu_erf(v + bias);
          else if (EPI == EPI_BIAS_SCALE_RES) v = rv[i][r] + gam * (v + bias);
          else if (EPI == EPI_RES) v = rv[i][r] + v;
          else if (EPI == EPI_SCALE) v = v * gam;
          a.C[(size_t)row * a.ldc + col] = v;
        }
      }
    }
  }
}

#undef X3_FETCH
#undef X3_FETCH_A
#undef X3_FETCH_W
#undef X3_STAGE
#undef X3_MARK

template <int WM, int WN, int MBLK, int NBLK, int NBUF = 1>
static hipError_t x3_dispatch(const GemmArgs& a, hipStream_t st) {
  constexpr int BM = WM * MBLK * 32, BN = WN * NBLK * 32;
  const int tiles = ((a.N + BN - 1) / BN) * ((a.M + BM - 1) / BM);
  dim3 grid(((tiles + 7) / 8) * 8), block(64 * WM * WN);  // 1-D grid, remapped XCD-aware inside the kernel
  switch (a.epi) {
    case EPI_STORE: CTTS_LAUNCH((gemm_tiled_bf16x3_k<EPI_STORE, WM, WN, MBLK, NBLK, NBUF>), grid, block, st, a); break;
    case EPI_RES: CTTS_LAUNCH((gemm_tiled_bf16x3_k<EPI_RES, WM, WN, MBLK, NBLK, NBUF>), grid, block, st, a); break;
    case EPI_BIAS: CTTS_LAUNCH((gemm_tiled_bf16x3_k<EPI_BIAS, WM, WN, MBLK, NBLK, NBUF>), grid, block, st, a); break;
    case EPI_BIAS_GELU: CTTS_LAUNCH((gemm_tiled_bf16x3_k<EPI_BIAS_GELU, WM, WN, MBLK, NBLK, NBUF>), grid, block, st, a); break;
    case EPI_BIAS_SCALE_RES: CTTS_LAUNCH((gemm_tiled_bf16x3_k<EPI_BIAS_SCALE_RES, WM, WN, MBLK, NBLK, NBUF>), grid, block, st, a); break;
    case EPI_SCALE: CTTS_LAUNCH((gemm_tiled_bf16x3_k<EPI_SCALE, WM, WN, MBLK, NBLK, NBUF>), grid, block, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_gemm_tiled_bf16x3(const GemmArgs& a, hipStream_t st) {
  if (a.K % 4 != 0 || a.lda % 4 != 0 || a.norm_w != nullptr) return hipErrorInvalidValue;
  if (a.taps > 1 && (a.cin % 4 != 0 || a.K != a.taps * a.cin)) return hipErrorInvalidValue;
  static int big = -1;  // CTTS_X3_TILE=128 forces the small tile, =2 the 256x128 tile (A/B experiments)
  if (big < 0) { const char* e = getenv("CTTS_X3_TILE"); big = e ? (atoi(e) == 128 ? 0 : atoi(e)) : 1; }
  if (big == 2 && a.N >= 512 && a.M >= 2048) return x3_dispatch<2, 2, 4, 2>(a, st);  // 256x128 tile, 256 threads, 2 per CU
  // 256x256 tiles pay off once they fill the chip about twice over; below that the one-workgroup-per-CU rounds quantise badly
  // (tools/codec_small_probe.py, whole decoder: 2304 frames 6.8 -> 4.1 ms, 9216 frames 8.2 -> 7.2 ms with 128x128 tiles;
  // 18432 frames 10.5 vs 12.2 ms and 65536 frames in favour of 256x256)
  if (big == 1 && a.N >= 512 && a.M >= 12288) return x3_dispatch<4, 2, 2, 4, 2>(a, st);   // 256x256 tile, 512 threads, 2 LDS buffers
  if (big && a.N >= 512 && a.M >= 2048) return x3_dispatch<4, 2, 2, 4>(a, st);   // CTTS_X3_TILE=256: single LDS buffer (A/B)
  return x3_dispatch<2, 2, 2, 2>(a, st);                                          // 128x128 tile, 256 threads
}
