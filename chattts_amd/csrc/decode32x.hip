// Decode-step projections of the GPT path in the float32 PARITY mode on SPLIT operands (round 5: split-bf16, VERDICT r4 item 4;
// ROUND 6: SPLIT-FP16 -- hi = fp16(x), lo' = fp16((x - hi) 2^11), 22 significant bits, common.hpp x3_split: the same bytes and the same
// three MFMAs, now v_mfma_f32_16x16x32_f16 on two accumulators (hi*hi | the cross terms, which enter with 2^-11).  Measured against the
// same model in float64: rms relative hidden error 2.3e-7, the f32 MFMA kernels' 3.3e-7 (profiles/r6p_f64_distance.log) -- this IS
// float32-class arithmetic.  The text below describes the round-5 bf16 format; read "fp16, scaled lo plane" for "bf16".)
//
// Reference op: the four nn.Linear calls of a HF Llama decoder layer reached from /root/reference/ChatTTS/model/gpt.py:419-427
// (in-tree twin /root/reference/examples/onnx/modeling_llama.py:415-417 q/k/v_proj, :500 o_proj, :293 gate/up/down) with the
// RMSNorm of :76-84 as prologue and the residual add / SiLU(gate)*up as epilogue -- float32 in the reference.
//
// Why: the parity mode (token ids bit-identical to the reference's CPU run) multiplied on v_mfma_f32_16x16x4_f32 -- 256 flop per
// clock and CU, a sixteenth of the bf16 rate: 7.7 us of matrix-pipe time per layer at batch 64 before a byte of latency
// (decode32.hip).  Here every operand is held as TWO bf16 planes, x = hi + lo with hi = bf16(x), lo = bf16(x - hi)
// (16-17 significant bits), and a product is three bf16 MFMAs, lo*hi + hi*lo + hi*hi, accumulated in float32 (the lo*lo term, 2^-18
// of the product, is dropped) -- the arithmetic the acoustic decoder's dense layers have used since round 1 (codec_gemm.hip).  The
// bytes are the float32 bytes (2 x 2 instead of 4 per value), the matrix pipe runs at 3/16 of the f32 cost.
//
// Parity: NOT bit-identical to gemm_dec32_k; the bar of this mode is "the reference's token ids on every golden".  Priced before it
// was built: the same arithmetic EMULATED inside the f32 kernels (build variant emux3, decode32.hip mfma4_proj) reproduces the
// reference's ids on the bench workload's 85,752 draws and on every reference-generated golden (profiles/r5d_x3_emulation.log); the
// e2e goldens run in this mode (tests/test_gpu_e2e.py) and bench.py's parity leg compares its sha256 with the reference's.
// The host picks the arithmetic (GptEngine dtype "f32x3" loads the planes; ctts_gen_state.proj_exact = 1 runs a call on decode32.hip).
//
// Layout: a plane is the bf16 fragment order of decode.hip, [rows/16][K/32][lane = (k%32)/8 * 16 + row%16][k%8]: one contiguous KiB
// per (16-row tile, 32-wide k chunk) in exactly the lane order v_mfma_f32_16x16x32_bf16 wants; the lo plane lies `plane` elements
// behind the hi plane.  Weights are split once at load (engine.py pack_frag_x3) with the RMSNorm gain folded in (W' = W diag(g), as
// the perf mode does): x g / rms . W = (x . W') / rms, so the activations reach the matrix core un-normalised and the row's 1 / rms
// -- taken from the float32 residual rows in gemm_skinny_k's order (common.hpp wave_rows_rstd_768), the same bits the f32 kernels
// use -- scales the accumulator.  Activations are split by their producers: embed_codes_k, attention_k (x3p_t output), the RES /
// SILU epilogues below.
//
// One kernel body, as decode32.hip's generic one: a workgroup = 16 output columns x (16 NMB) rows, its 4 waves split K into
// contiguous quarters, all loads of a round are issued before the first MFMA, the K partials meet in LDS and are added in the fixed
// order ((w0 + w1) + w2) + w3.
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"

// k chunks of 32 per wave and round, by the KERNEL's row tiles and K (both known before the live-row count).  The rounds are not
// double-buffered -- every round is one exposed memory round trip -- so a workgroup takes its whole quarter of K in ONE round wherever the
// registers allow: 6 chunks at K = 768 (16-row workgroups: 24-36 16-byte loads in flight per lane; 64-row workgroups: up to 72, 404
// VGPRs, one workgroup per CU, which is all a 144- / 192-tile launch has anyway), 24 chunks at K = 3072 for the 16-row workgroups of
// down_proj (96 loads, 400 VGPRs, 192 workgroups).  Measured (profiles/r5i_ab_x3_rounds.log, parity leg of the bench): 3 -> 6 chunks for
// the 64-row RMSNorm launches 995 -> 1016 audio-s/s, 12 -> 24 for down 978 -> 986.
// (one round everywhere by default: chunks per wave and round = (K / 32) / waves; only the wide workgroups at K = 3072 -- not a shipped
// choice -- keep rounds of 3)
template <int MBT, int KT, int NW> struct DxU { static constexpr int v = (MBT > 1 && KT == 3072) ? 3 : (KT / 32) / NW; };

__device__ __forceinline__ void x3_store(uint16_t* __restrict__ hi_at, const size_t plane, const float v) {
  uint16_t h, l;
  x3_split(v, h, l);   // split-fp16 (common.hpp)
  hi_at[0] = h;
  hi_at[plane] = l;
}

template <int NMB, int MBT, int KT, bool RMS, int EPI, int NW>
__device__ __forceinline__ void dec32x_body(const Dec32xArgs& a, const int M, const int tile, const int mt0,
                                            u128 (&wh)[(EPI == EPI_SILU_MUL) ? 2 : 1][DxU<MBT, KT, NW>::v], u128 (&wl)[(EPI == EPI_SILU_MUL) ? 2 : 1][DxU<MBT, KT, NW>::v],
                                            float (*red)[(EPI == EPI_SILU_MUL) ? 2 : 1][MBT][64][4], float* rstd_s) {
  constexpr int NACC = (EPI == EPI_SILU_MUL) ? 2 : 1;
  constexpr int U = DxU<MBT, KT, NW>::v;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int n0 = tile * 16, m0 = mt0 * 16;
  const int N = a.N;
  constexpr int KCH = KT >> 5;

  constexpr int PPW = NMB >= 3 ? 4 : NMB;
  constexpr int NF = NMB >= 3 ? NMB : 4;
  const int fmb = (wave * PPW) >> 2, fr0 = (wave * PPW) & 3;   // meaningful for wave < NF

  constexpr int nper = KCH / NW;   // chunks per wave (nper % U == 0); wave w owns the contiguous share w of K
  const u128* wp = reinterpret_cast<const u128*>(a.Wp) + ((size_t)tile * KCH + wave * nper) * 64 + lane;
  const u128* wp2 = wp + (size_t)(N >> 4) * KCH * 64;  // SILU_MUL: the "up" tile of the same columns
  const u128* ap = reinterpret_cast<const u128*>(a.Ap) + ((size_t)mt0 * KCH + wave * nper) * 64 + lane;
  const size_t wpl = a.w_plane >> 3, apl = a.a_plane >> 3;   // plane strides in 16-byte units
  const bool w_once = a.w_nt && gridDim.y == 1;
  u128 ah[NMB][U], al[NMB][U];
  auto load_a = [&](const int i) {
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
      for (int j = 0; j < U; ++j) {
        ah[mb][j] = load16(ap + ((size_t)mb * KCH + i + j) * 64);
        al[mb][j] = load16(ap + apl + ((size_t)mb * KCH + i + j) * 64);
      }
  };
  // Every load the workgroup can issue is requested here, up front, and none depends on another: the activation fragments of the first
  // round (the weights' were requested at kernel entry), then the small operands of the prologue / epilogue -- residual, row
  // descriptors, RoPE factors, the rows' partial sums of squares.  One memory round trip, not a chain of them.
  load_a(0);
  float pre0[PPW];  // RES: residual
#pragma unroll
  for (int q = 0; q < PPW; ++q) pre0[q] = 0.f;
  if (EPI == EPI_RES && wave < NF) {
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
      const int row = min(m0 + 16 * fmb + 4 * g + fr0 + q, M - 1);
      pre0[q] = a.res[(size_t)row * a.ldr + n0 + li];
    }
  }
  // QKV_ROPE tiles (weight rows permuted by the loader, engine.py rope_row_perm): see decode32.hip.  The RoPE factors come from the
  // per-row table the step's first kernel wrote (StepPrep.rope_cs: cos[32] | sin[32] of the row's position -- the same table entries
  // a.cos_t[pos * 32 + d] would deliver, without the row -> position -> table chain), or through the descriptor when there is none.
  const int sect = n0 / 768, hcol = n0 % 768, head = hcol >> 6, t4 = (hcol & 63) >> 4;
  const int dlo = 8 * t4 + (li & 7);
  RowDesc rd[PPW];
  float rc[PPW], rsn[PPW];
  if (EPI == D32_EPI_QKV_ROPE && wave < NF) {
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
      const int row = min(m0 + 16 * fmb + 4 * g + fr0 + q, M - 1);
      rd[q] = a.desc[row];
      if (a.rope_cs != nullptr) {
        rc[q] = a.rope_cs[(size_t)row * 64 + dlo];
        rsn[q] = a.rope_cs[(size_t)row * 64 + 32 + dlo];
      } else {
        rc[q] = a.cos_t[rd[q].pos * 32 + dlo];
        rsn[q] = a.sin_t[rd[q].pos * 32 + dlo];
      }
    }
  }
  // RMSNorm launches: 1 / rms of the workgroup's rows.  With `ssq_in` (the decode step): from the 48 partial sums of squares per row
  // the producers of the residual stream left (embed_codes_k, the RES epilogue below) -- 4 threads x 12 partials per row, as decode.hip;
  // without (tests): from the float32 rows themselves, gemm_skinny_k's arithmetic.
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0;
  const int srow = tid >> 2, spart = tid & 3;
  if (RMS && a.ssq_in != nullptr && srow < 16 * NMB) {
    const float* sp = a.ssq_in + (size_t)min(m0 + srow, M - 1) * SSQ_PARTS + spart * 12;
    s0 = *reinterpret_cast<const float4*>(sp);
    s1 = *reinterpret_cast<const float4*>(sp + 4);
    s2 = *reinterpret_cast<const float4*>(sp + 8);
  }
  if (RMS && a.ssq_in == nullptr && wave < 4) {
#pragma unroll
    for (int r0 = 0; r0 < 16 * NMB; r0 += 16) {
      const float* rows[4];
      float rstd[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) rows[q] = a.X + (size_t)min(m0 + r0 + wave + 4 * q, M - 1) * a.ldx;
      wave_rows_rstd_768<4>(rows, a.eps, lane, rstd);
      if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) rstd_s[r0 + wave + 4 * q] = rstd[q];
      }
    }
  }

  f32x4 acc[NACC][NMB], accx[NACC][NMB];   // hi*hi | the cross terms lo'*hi + hi*lo' (they enter with 2^-11)
#pragma unroll
  for (int na = 0; na < NACC; ++na)
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) { acc[na][mb] = (f32x4){0.f, 0.f, 0.f, 0.f}; accx[na][mb] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  for (int i = 0; i < nper; i += U) {
    if (i > 0) {   // the first round's weight fragments were requested at kernel entry (before *n_active was known)
      if (w_once) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
          wh[0][j] = load16_nt(wp + (size_t)(i + j) * 64);
          wl[0][j] = load16_nt(wp + wpl + (size_t)(i + j) * 64);
          if (NACC == 2) { wh[1][j] = load16_nt(wp2 + (size_t)(i + j) * 64); wl[1][j] = load16_nt(wp2 + wpl + (size_t)(i + j) * 64); }
        }
      } else {
#pragma unroll
        for (int j = 0; j < U; ++j) {
          wh[0][j] = load16(wp + (size_t)(i + j) * 64);
          wl[0][j] = load16(wp + wpl + (size_t)(i + j) * 64);
          if (NACC == 2) { wh[1][j] = load16(wp2 + (size_t)(i + j) * 64); wl[1][j] = load16(wp2 + wpl + (size_t)(i + j) * 64); }
        }
      }
      load_a(i);
    }
    // every load of the round in flight before the first MFMA (hipcc otherwise sinks each load next to its use)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < U; ++j)
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int na = 0; na < NACC; ++na) {
          f32x4 cx = accx[na][mb];
          cx = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&al[mb][j]), *reinterpret_cast<const f16x8*>(&wh[na][j]), cx, 0, 0, 0);
          cx = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&ah[mb][j]), *reinterpret_cast<const f16x8*>(&wl[na][j]), cx, 0, 0, 0);
          accx[na][mb] = cx;
          acc[na][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&ah[mb][j]), *reinterpret_cast<const f16x8*>(&wh[na][j]), acc[na][mb], 0, 0, 0);
        }
  }

  if (RMS && a.ssq_in != nullptr) {
    float sq = (((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w))) + ((s2.x + s2.y) + (s2.z + s2.w));
    sq += __shfl_xor(sq, 1, 64);
    sq += __shfl_xor(sq, 2, 64);
    if (spart == 0 && srow < 16 * NMB) rstd_s[srow] = 1.0f / sqrtf(sq / 768.0f + a.eps);
  }
#pragma unroll
  for (int na = 0; na < NACC; ++na)
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) *reinterpret_cast<f32x4*>(&red[wave][na][mb][lane][0]) = acc[na][mb] + accx[na][mb] * X3_LO_INV;
  __syncthreads();   // (also publishes rstd_s)
  if (wave >= NF) return;

  float t[NW][NACC][PPW];   // the NW waves' partials of this wave's outputs
#pragma unroll
  for (int w = 0; w < NW; ++w)
#pragma unroll
    for (int na = 0; na < NACC; ++na) {
      if constexpr (PPW == 4) {
        *reinterpret_cast<f32x4*>(t[w][na]) = *reinterpret_cast<const f32x4*>(&red[w][na][fmb][lane][0]);
      } else if constexpr (PPW == 2) {
        *reinterpret_cast<float2*>(t[w][na]) = *reinterpret_cast<const float2*>(&red[w][na][fmb][lane][fr0]);
      } else {
        t[w][na][0] = red[w][na][fmb][lane][fr0];
      }
    }

  const int col = n0 + li;
#pragma unroll
  for (int q = 0; q < PPW; ++q) {
    const int rloc = 16 * fmb + 4 * g + fr0 + q;   // C/D map of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + reg
    const int row = m0 + rloc;
    float v = t[0][0][q], u = 0.f;   // fixed order: ((w0 + w1) + w2) + ...
#pragma unroll
    for (int w = 1; w < NW; ++w) v += t[w][0][q];
    if (EPI == EPI_SILU_MUL) {
      u = t[0][NACC - 1][q];
#pragma unroll
      for (int w = 1; w < NW; ++w) u += t[w][NACC - 1][q];
    }
    if (RMS) {
      const float rs = rstd_s[rloc];
      v *= rs;
      u *= rs;
    }
    if (EPI == EPI_SILU_MUL) v = silu_f(v) * u;
    else if (EPI == EPI_RES) v = pre0[q] + v;
    if (EPI == D32_EPI_QKV_ROPE) {   // q -> roped, qkv buffer; k -> roped, KV cache; v -> KV cache  (rope_append_k, gpt.hip)
      const float other = __shfl_xor(v, 8, 64);   // every lane of the wave is here (rows are skipped below, not above)
      const bool hi = li >= 8;
      const float roped = hi ? rope_hi(other, v, rc[q], rsn[q]) : rope_lo(v, other, rc[q], rsn[q]);
      const int d = dlo + (hi ? 32 : 0);
      if (row < M && rd[q].b >= 0) {
        const size_t cbase = (((size_t)rd[q].b * 12 + head) * a.cmax + rd[q].slot) * 64;
        if (sect == 0) a.C[(size_t)row * a.ldc + head * 64 + d] = roped;
        else if (sect == 1) a.kc[cbase + d] = roped;
        else a.vc[cbase + (hcol & 63) + li] = v;
      }
      continue;
    }
    if (EPI == EPI_RES && a.ssq_out != nullptr) {   // this tile's share of the new row's sum of squares (every lane of the wave is here)
      float sq = row < M ? v * v : 0.f;
      sq += __shfl_xor(sq, 1, 64);
      sq += __shfl_xor(sq, 2, 64);
      sq += __shfl_xor(sq, 4, 64);
      sq += __shfl_xor(sq, 8, 64);
      if (li == 0 && row < M) a.ssq_out[(size_t)row * SSQ_PARTS + tile] = sq;
    }
    if (row >= M) continue;
    if (EPI == EPI_RES) a.C[(size_t)row * a.ldc + col] = v;
    if (a.Cp != nullptr) x3_store(a.Cp + pk_off(row, col, a.kch_out), a.c_plane, v);
    if (EPI == EPI_RES && a.Cp32 != nullptr) a.Cp32[pk32_off(row, col, a.kch32_out)] = v;
  }
}

template <int MBT, int KT, bool RMS, int EPI, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_dec32x_k(Dec32xArgs a) {
  constexpr int NACC = (EPI == EPI_SILU_MUL) ? 2 : 1;
  constexpr int U = DxU<MBT, KT, NW>::v;
  __shared__ __attribute__((aligned(16))) float red[NW][NACC][MBT][64][4];
  __shared__ float rstd_s[16 * MBT];
  CTTS_PROBE_RETURN();

  const int tile = blockIdx.x, mt0 = blockIdx.y * MBT;
  // the weight fragments of the first round depend on nothing but the kernel arguments: request them before the live-row count (a
  // dependent scalar load) is known
  u128 wh[NACC][U], wl[NACC][U];
  {
    constexpr int KCH = KT >> 5, nper = KCH / NW;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u128* wp = reinterpret_cast<const u128*>(a.Wp) + ((size_t)tile * KCH + wave * nper) * 64 + lane;
    const u128* wp2 = wp + (size_t)(a.N >> 4) * KCH * 64;
    const size_t wpl = a.w_plane >> 3;
    if (a.w_nt && gridDim.y == 1) {
#pragma unroll
      for (int j = 0; j < U; ++j) {
        wh[0][j] = load16_nt(wp + (size_t)j * 64); wl[0][j] = load16_nt(wp + wpl + (size_t)j * 64);
        if (NACC == 2) { wh[1][j] = load16_nt(wp2 + (size_t)j * 64); wl[1][j] = load16_nt(wp2 + wpl + (size_t)j * 64); }
      }
    } else {
#pragma unroll
      for (int j = 0; j < U; ++j) {
        wh[0][j] = load16(wp + (size_t)j * 64); wl[0][j] = load16(wp + wpl + (size_t)j * 64);
        if (NACC == 2) { wh[1][j] = load16(wp2 + (size_t)j * 64); wl[1][j] = load16(wp2 + wpl + (size_t)j * 64); }
      }
    }
  }
  const int M = a.n_active ? min(*a.n_active, a.M) : a.M;   // live (compact) rows
  if (mt0 * 16 >= M) return;
  const int nmb = min(MBT, (M - mt0 * 16 + 15) >> 4);
  if constexpr (MBT == 1) {
    dec32x_body<1, MBT, KT, RMS, EPI, NW>(a, M, tile, mt0, wh, wl, red, rstd_s);
  } else if constexpr (MBT == 2) {
    if (nmb == 1) dec32x_body<1, MBT, KT, RMS, EPI, NW>(a, M, tile, mt0, wh, wl, red, rstd_s);
    else dec32x_body<2, MBT, KT, RMS, EPI, NW>(a, M, tile, mt0, wh, wl, red, rstd_s);
  } else {
    if (nmb == 1) dec32x_body<1, MBT, KT, RMS, EPI, NW>(a, M, tile, mt0, wh, wl, red, rstd_s);
    else if (nmb == 2) dec32x_body<2, MBT, KT, RMS, EPI, NW>(a, M, tile, mt0, wh, wl, red, rstd_s);
    else if (nmb == 3) dec32x_body<3, MBT, KT, RMS, EPI, NW>(a, M, tile, mt0, wh, wl, red, rstd_s);
    else dec32x_body<4, MBT, KT, RMS, EPI, NW>(a, M, tile, mt0, wh, wl, red, rstd_s);
  }
}

static int env_i(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

template <int MBT, int NW>
static hipError_t dec32x_dispatch(const Dec32xArgs& a, hipStream_t st) {
  const int mt = (a.M + 15) / 16;
  dim3 grid(a.N / 16, (mt + MBT - 1) / MBT), block(64 * NW);
  if (a.epi == EPI_RES && !a.rms && a.K == 768) CTTS_LAUNCH((gemm_dec32x_k<MBT, 768, false, EPI_RES, NW>), grid, block, st, a);
  else if (a.epi == EPI_RES && !a.rms && a.K == 3072) CTTS_LAUNCH((gemm_dec32x_k<MBT, 3072, false, EPI_RES, NW>), grid, block, st, a);
  else if (a.epi == EPI_SILU_MUL && a.rms && a.K == 768) CTTS_LAUNCH((gemm_dec32x_k<MBT, 768, true, EPI_SILU_MUL, NW>), grid, block, st, a);
  else if (a.epi == D32_EPI_QKV_ROPE && a.rms && a.K == 768) CTTS_LAUNCH((gemm_dec32x_k<MBT, 768, true, D32_EPI_QKV_ROPE, NW>), grid, block, st, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_gemm_dec32x(const Dec32xArgs& a_in, hipStream_t st) {
  Dec32xArgs a = a_in;
  static int nt = -1, mb_qkv = 4, mb_silu = 4, mb_o = 1, mb_down = 1, nw_qkv = 8, nw_silu = 8, nw_o = 4, nw_down = 8;
  if (nt < 0) {
    nt = env_i("CTTS_W_NT", 1);
    // rows per workgroup (A/B knobs; profiles/r5h_ab_x3_mb.log, r5i_ab_x3_rounds.log): the RMSNorm launches take all <= 64 rows per weight
    // tile -- every weight fragment crosses L2 -> CU once instead of once per row tile (16-row workgroups: 110 MB of operand traffic per
    // gate/up launch for 19 MB of weights) --, o / down (48 weight tiles) stay 16-row workgroups (2 rows per workgroup: 906 vs 981)
    mb_qkv = env_i("CTTS_D32X_MB_QKV", 4); mb_silu = env_i("CTTS_D32X_MB_SILU", 4);
    mb_o = env_i("CTTS_D32X_MB_O", 1); mb_down = env_i("CTTS_D32X_MB_DOWN", 1);
    // waves per workgroup (4 | 8): with 8 a wave requests half as many chunks and twice as many waves have requests in flight
    // (profiles/r5s_ab_x3_nw.log, parity leg: 4 waves everywhere 1011-1021 audio-s/s; QKV + gate/up on 8: 1023-1046; + down: 1026-1052; + o: no change)
    nw_qkv = env_i("CTTS_D32X_NW_QKV", 8); nw_silu = env_i("CTTS_D32X_NW_SILU", 8); nw_o = env_i("CTTS_D32X_NW_O", 4); nw_down = env_i("CTTS_D32X_NW_DOWN", 8);
  }
  a.w_nt = nt;
  // K: chunks of 32, 4 waves, rounds of 6 (3) chunks
  if (a.M <= 0 || a.N <= 0 || (a.N & 15) || (a.K != 768 && a.K != 3072) || !a.Ap || !a.Wp || (a.a_plane & 7) || (a.w_plane & 7) || (a.c_plane & 7))
    return hipErrorInvalidValue;
  if (a.rms && a.K != 768) return hipErrorInvalidValue;
  if (a.rms && a.ssq_in == nullptr && (a.X == nullptr || (a.ldx & 3))) return hipErrorInvalidValue;
  if (a.epi == D32_EPI_QKV_ROPE && (a.N != 2304 || a.K != 768 || !a.desc || !a.kc || !a.vc || !a.C)) return hipErrorInvalidValue;
  if (a.epi == EPI_RES && (!a.res || !a.C)) return hipErrorInvalidValue;
  if (a.epi == EPI_SILU_MUL && !a.Cp) return hipErrorInvalidValue;
  int mb = a.epi == EPI_SILU_MUL ? mb_silu : a.epi == EPI_RES ? (a.K > 768 ? mb_down : mb_o) : mb_qkv;
  int nw = a.epi == EPI_SILU_MUL ? nw_silu : a.epi == EPI_RES ? (a.K > 768 ? nw_down : nw_o) : nw_qkv;
  if (a.force_mb) { mb = a.force_mb & 7; nw = (a.force_mb & 16) ? 16 : (a.force_mb & 8) ? 8 : 4; }   // tests: rows per workgroup in the low bits, +8 = eight waves, +16 = sixteen (down_proj, 16 rows)
  if (nw == 16 && a.epi == EPI_RES && !a.rms && a.K == 3072 && mb == 1) {   // down_proj only: 16 waves x 6 chunks, as the perf mode's down_proj
    dim3 grid(a.N / 16, (a.M + 15) / 16), block(1024);
    CTTS_LAUNCH((gemm_dec32x_k<1, 3072, false, EPI_RES, 16>), grid, block, st, a);
    return hipGetLastError();
  }
  if (nw >= 8) {
    if (mb >= 4) return dec32x_dispatch<4, 8>(a, st);
    if (mb == 2) return dec32x_dispatch<2, 8>(a, st);
    return dec32x_dispatch<1, 8>(a, st);
  }
  if (mb >= 4) return dec32x_dispatch<4, 4>(a, st);
  if (mb == 2) return dec32x_dispatch<2, 4>(a, st);
  return dec32x_dispatch<1, 4>(a, st);
}
