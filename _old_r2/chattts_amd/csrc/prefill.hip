// Prefill projections of the perf (bf16) mode: C[M,N] = epi(A[M,K] * W[N,K]^T) with M = B*T prompt rows (hundreds to
// thousands), bf16 operands, f32 accumulation on v_mfma_f32_32x32x16_bf16.  The decode kernel (gemm.hip: gemm_fast_k) is
// a weight-streaming shape for M <= 64; with many rows it re-reads operands from L2 for every 16 output columns.  This
// one is the classic LDS-tiled GEMM: 128 x 128 x 64 tiles (128 x 64 gate|up pairs for the SiLU epilogue; 64-row tiles
// when the launch would not fill the chip), 256 threads = 2 x 2 waves, register-staged global loads one tile ahead, two
// LDS buffers, one barrier per k-step.
// Same operands, layouts and epilogue semantics as gemm_fast_k (FastGemmArgs), so the two are interchangeable per
// launch; the launcher picks this one for M >= 256.
//   RES       x32 += acc; bf16 copy; per-row partial sums of squares (one per 16 columns) for the next RMSNorm
//   SILU      act = silu(rstd * acc_gate) * (rstd * acc_up)                  (W = [gate rows; up rows])
//   QKV_ROPE  rstd * acc, rotate-half RoPE on q / k (weight rows permuted by the loader so that a pair sits 8 lanes
//             apart), q -> f32 buffer, k / v -> this layer's bf16 KV cache at (b, head, slot)
#include "common.hpp"
#include "kernels.hpp"

namespace {

constexpr int BK = 64, LD = BK + 8;   // 144-byte LDS rows: conflict-free ds_read_b128 of MFMA fragments
constexpr int NTHR = 256;

// MBLK = 32-row MFMA blocks per wave along M: 2 -> 128-row tiles; 1 -> 64-row tiles (twice the workgroups when a
// launch would otherwise not fill the chip: N = 768 outputs, short prompts)
template <int EPI, int MBLK>
__global__ __launch_bounds__(NTHR, 2) void gemm_prefill_k(FastGemmArgs a) {
  constexpr bool SILU = EPI == FEPI_SILU;
  constexpr int BM = 64 * MBLK;
  constexpr int BN = 128;                        // W rows per tile (SILU: 64 gate rows + the 64 matching up rows)
  constexpr int BNO = SILU ? 64 : 128;           // output columns per tile
  __shared__ __attribute__((aligned(16))) uint16_t As[2][BM][LD], Ws[2][BN][LD];
  __shared__ float rstd_s[BM];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave & 1, wn = wave >> 1;
  const int M = a.M, N = a.N, K = a.K;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BNO;

  // per-row 1/rms from the producer's 48 partial sums of squares (RMSNorm gain is folded into W)
  if (a.ssq_in != nullptr && tid < BM) {
    const float* sp = a.ssq_in + (size_t)min(m0 + tid, M - 1) * SSQ_PARTS;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < SSQ_PARTS / 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(sp + 4 * q);
      s += (v.x + v.y) + (v.z + v.w);
    }
    rstd_s[tid] = 1.0f / sqrtf(s / 768.0f + a.eps);
  }

  // loaders: 8 lanes x 16 B = one 128-byte row segment (64 bf16), 32 rows per pass, 4 passes for 128 rows
  const int lr = tid >> 3, lk = (tid & 7) * 8;
  constexpr int PA = BM / 32;
  const uint16_t* ap[PA];
  const uint16_t* wp[4];
#pragma unroll
  for (int p = 0; p < PA; ++p) ap[p] = a.A + (size_t)min(m0 + lr + 32 * p, M - 1) * a.lda + lk;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = lr + 32 * p;                    // tile row of W
    const int gr = SILU ? (r < 64 ? n0 + r : N + n0 + (r - 64)) : min(n0 + r, N - 1);
    wp[p] = a.W + (size_t)gr * K + lk;
  }
  u128 ra[PA], rw[4];
#define PF_FETCH(K0)                                                                  \
  do {                                                                                \
    _Pragma("unroll") for (int p = 0; p < PA; ++p) ra[p] = load16(ap[p] + (K0));      \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) rw[p] = load16(wp[p] + (K0));       \
  } while (0)
#define PF_STAGE(SB)                                                                                  \
  do {                                                                                                \
    _Pragma("unroll") for (int p = 0; p < PA; ++p) *reinterpret_cast<u128*>(&As[SB][lr + 32 * p][lk]) = ra[p]; \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) *reinterpret_cast<u128*>(&Ws[SB][lr + 32 * p][lk]) = rw[p];  \
  } while (0)

  f32x16 acc[MBLK][2];
#pragma unroll
  for (int i = 0; i < MBLK; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  PF_FETCH(0);
  PF_STAGE(0);
  if (BK < K) PF_FETCH(BK);
  __syncthreads();
  const int ri = lane & 31, kg = (lane >> 5) * 8;
  // W rows of this wave's two 32-column blocks: plain: columns wn*64 + j*32; SILU: j = 0 gate block, j = 1 the up block
  // of the SAME 32 output columns (tile rows 64..127 hold the up rows)
  const int wrow0 = SILU ? wn * 32 : wn * 64, wrow1 = SILU ? 64 + wn * 32 : wn * 64 + 32;
  int sb = 0;
  for (int k0 = 0; k0 < K; k0 += BK) {
    if (k0 + BK < K) PF_STAGE(sb ^ 1);            // tile k+1 (requested one step ago) -> the other buffer
    if (k0 + 2 * BK < K) PF_FETCH(k0 + 2 * BK);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      const bf16x8 fw0 = *reinterpret_cast<const bf16x8*>(&Ws[sb][wrow0 + ri][kk + kg]);
      const bf16x8 fw1 = *reinterpret_cast<const bf16x8*>(&Ws[sb][wrow1 + ri][kk + kg]);
#pragma unroll
      for (int i = 0; i < MBLK; ++i) {
        const bf16x8 fa = *reinterpret_cast<const bf16x8*>(&As[sb][(wm * MBLK + i) * 32 + ri][kk + kg]);
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fw0, acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fw1, acc[i][1], 0, 0, 0);
      }
    }
    __syncthreads();
    sb ^= 1;
  }
#undef PF_FETCH
#undef PF_STAGE

  // ---- epilogue: lane holds column (lane & 31) of a 32-column block, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----
  const int cl = lane & 31;
#pragma unroll
  for (int i = 0; i < MBLK; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lrow = (wm * MBLK + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int row = m0 + lrow;
      const bool rok = row < M;
      const float rs = a.ssq_in != nullptr ? rstd_s[lrow] : 1.0f;
      if (EPI == FEPI_SILU) {
        const int col = n0 + wn * 32 + cl;
        const float gv = acc[i][0][r] * rs, uv = acc[i][1][r] * rs;
        if (rok) a.Cb[(size_t)row * a.ldcb + col] = f32_to_bf16(silu_f(gv) * uv);
      } else if (EPI == FEPI_RES) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = n0 + wn * 64 + j * 32 + cl;
          float xn = 0.f;
          if (rok) {
            xn = a.C32[(size_t)row * a.ldc + col] + acc[i][j][r] * rs;
            a.C32[(size_t)row * a.ldc + col] = xn;
            a.Cb[(size_t)row * a.ldcb + col] = f32_to_bf16(xn);
          }
          float sq = xn * xn;     // per 16-column group (SSQ_PARTS = 768 / 16), lanes of a group are 16 consecutive
          sq += __shfl_xor(sq, 1, 64);
          sq += __shfl_xor(sq, 2, 64);
          sq += __shfl_xor(sq, 4, 64);
          sq += __shfl_xor(sq, 8, 64);
          if ((cl & 15) == 0 && rok) a.ssq_out[(size_t)row * SSQ_PARTS + (col >> 4)] = sq;
        }
      } else {  // FEPI_QKV_ROPE
        // row -> (utterance, slot, rotary position), as GptRowMap / the decode kernel's helper wave
        int b, slot;
        const int rr = min(row, M - 1);
        if (a.q_per_b == 1) { b = a.row_map ? a.row_map[rr] : rr; slot = a.len[b] - 1; }
        else { b = rr / a.q_per_b; slot = a.slot0 + rr - b * a.q_per_b; if (a.row_map) b = a.row_map[b]; }
        int pos = slot - a.kv_start[b];
        if (pos < 0) pos = 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = n0 + wn * 64 + j * 32 + cl;
          const int sect = col / 768, hcol = col - sect * 768, head = hcol >> 6, hd = hcol & 63;
          const int t4 = hd >> 4, li = hd & 15;
          const float v = acc[i][j][r] * rs;
          const float other = __shfl_xor(v, 8, 64);     // rotate-half partner: 8 columns away inside the 16-group
          const bool hi = li >= 8;
          const int dlo = 8 * t4 + (li & 7);
          const float cc = a.cos_t[pos * 32 + dlo], ss = a.sin_t[pos * 32 + dlo];
          const float roped = hi ? (v * cc + other * ss) : (v * cc - other * ss);
          const int d = dlo + (hi ? 32 : 0);
          if (rok) {
            const size_t cbase = (((size_t)b * 12 + head) * a.cmax + slot) * 64;
            if (sect == 0) a.C32[(size_t)row * a.ldc + head * 64 + d] = roped;
            else if (sect == 1) a.kc[cbase + d] = f32_to_bf16(roped);
            else a.vc[cbase + hd] = f32_to_bf16(v);
          }
        }
      }
    }
  }
}

}  // namespace

bool gemm_prefill_supported(const FastGemmArgs& a) {
  if (a.K % BK != 0 || (a.lda % 8) != 0 || a.n_active != nullptr) return false;
  if (a.epi == FEPI_SILU) return a.ssq_in != nullptr && a.N % 64 == 0;
  if (a.epi == FEPI_RES) return a.ssq_in == nullptr && a.N % 128 == 0 && a.N == 16 * SSQ_PARTS;
  if (a.epi == FEPI_QKV_ROPE) return a.ssq_in != nullptr && a.N == 2304 && a.K == 768;
  return false;
}

hipError_t launch_gemm_prefill(const FastGemmArgs& a, hipStream_t st) {
  if (!gemm_prefill_supported(a)) return hipErrorInvalidValue;
  const dim3 block(NTHR);
  const int nx = a.epi == FEPI_SILU ? a.N / 64 : a.N / 128;
  // 64-row tiles when 128-row tiles would leave CUs without a workgroup (measured at 3072 rows: N = 768 -> 144
  // workgroups: o 42 -> 28 us, down 70 -> 57 us with 64-row tiles; qkv at 432 workgroups is 10 % faster with 128 rows)
  const bool small = (long)nx * ((a.M + 127) / 128) < 256;
  const dim3 grid(nx, small ? (a.M + 63) / 64 : (a.M + 127) / 128);
  switch (a.epi) {
    case FEPI_SILU:
      if (small) CTTS_LAUNCH((gemm_prefill_k<FEPI_SILU, 1>), grid, block, st, a); else CTTS_LAUNCH((gemm_prefill_k<FEPI_SILU, 2>), grid, block, st, a);
      break;
    case FEPI_RES:
      if (small) CTTS_LAUNCH((gemm_prefill_k<FEPI_RES, 1>), grid, block, st, a); else CTTS_LAUNCH((gemm_prefill_k<FEPI_RES, 2>), grid, block, st, a);
      break;
    case FEPI_QKV_ROPE:
      if (small) CTTS_LAUNCH((gemm_prefill_k<FEPI_QKV_ROPE, 1>), grid, block, st, a); else CTTS_LAUNCH((gemm_prefill_k<FEPI_QKV_ROPE, 2>), grid, block, st, a);
      break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
