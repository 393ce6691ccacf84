// Device-side body of the decode-step projections on fragment-packed operands: see decode.hip for the design notes.  Lives in a
// header because TWO kernels run it: gemm_dec_k (decode.hip) and the fused QKV + attention launch (gpt.hip qkv_attention_k), whose
// first workgroups are QKV tiles that hand q / k / v to the attention units of the same launch (HO = true).
#pragma once
#include "common.hpp"
#include "kernels.hpp"

// (HO_STRIDE, the spacing of the fused QKV + attention launch's arrival words, lives in kernels.hpp: capi.hip carves the buffer with it)
constexpr int DEC_U = 6;  // k-chunks of 32 per wave and round: DEC_U * (NACC + NMB) 16-byte loads in flight per lane

template <int NMB, int MBT, int NW, bool SCALE, int EPI, bool HO = false, int U_ = DEC_U>
__device__ __forceinline__ void dec_body(const DecGemmArgs& a, const int M, const int tile, const int mt0, const int row_groups,
                                         const int wg_linear,
                                         u128 (&wf)[(EPI == FEPI_SILU) ? 2 : 1][U_],
                                         float (*red)[(EPI == FEPI_SILU) ? 2 : 1][MBT][64][4], float* rstd_s,
                                         float (*cs_s)[16], int (*meta_s)[2], const u128 (&af0)[U_], const bool a_pre, int* ho_cnt = nullptr) {
  constexpr int NACC = (EPI == FEPI_SILU) ? 2 : 1;
  constexpr int U = U_;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int n0 = tile * 16, m0 = mt0 * 16;
  const int N = a.N, KCH = a.K >> 5;
  // optional phase stamps (tools/dec_phase_probe.py): 100 MHz realtime counter, thread 0 of every workgroup
  long long* dbg = a.dbg ? a.dbg + (size_t)wg_linear * 8 : nullptr;
#define STAMP(i) do { if (dbg && tid == 0) dbg[i] = wall_clock64(); } while (0)
  STAMP(1);

  // per-row sum of squares: 4 threads x 12 partials per row, consumed only in the epilogue
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0;
  const int srow = tid >> 2, spart = tid & 3;
  if (SCALE && srow < 16 * NMB) {
    const float* sp = a.ssq_in + (size_t)min(m0 + srow, a.M - 1) * SSQ_PARTS + spart * 12;   // a.M: no wait on *n_active
    s0 = *reinterpret_cast<const float4*>(sp);
    s1 = *reinterpret_cast<const float4*>(sp + 4);
    s2 = *reinterpret_cast<const float4*>(sp + 8);
  }

  // finishing work = 4*NMB (row tile, accumulator register) pairs of 64 outputs.  Finishing wave w owns PPW CONSECUTIVE pairs, so
  // its LDS reads below are whole 16- / 8-byte vectors (reading one float of every lane's float4 is an 8-way bank conflict):
  // NMB >= 3: wave w finishes row tile w (all 4 registers); NMB = 2: tile w/2, registers 2(w&1)..+1; NMB = 1: register w
  constexpr int PPW = NMB >= 3 ? 4 : NMB;
  constexpr int NF = NMB >= 3 ? NMB : 4;
  static_assert(NW >= 4, "needs at least 4 waves");
  const int fmb = (wave * PPW) >> 2, fr0 = (wave * PPW) & 3;   // meaningful for wave < NF
  float pre0[PPW];  // RES: residual, requested before the operand loads
#pragma unroll
  for (int q = 0; q < PPW; ++q) pre0[q] = 0.f;
  if (EPI == FEPI_RES && wave < NF) {
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
      const int row = min(m0 + 16 * fmb + 4 * g + fr0 + q, a.M - 1);
      pre0[q] = a.C32[(size_t)row * a.ldc + n0 + li];
    }
  }

  // HO (QKV tile inside the fused QKV + attention launch): no helper wave.  Wave 0 requests the rows' RoPE factors and KV destinations
  // here, ahead of the activation loads, from the per-row table the step's first kernel wrote (StepPrep.rope_cs: cos[32] then sin[32]
  // of the row's position; desc: utterance, KV slot) -- one hop, nothing dependent -- and parks them in LDS once the activation
  // requests are out (they return in order, ahead of those).
  RowDesc hdsc = RowDesc{-1, 0, 0, 0};
  float4 hc0 = make_float4(0.f, 0.f, 0.f, 0.f), hc1 = hc0, hs0 = hc0, hs1 = hc0;
  const bool hmeta = HO && EPI == FEPI_QKV_ROPE && wave == 0 && lane < 16 * NMB;
  if (hmeta) {
    const int row = min(m0 + lane, a.M - 1);
    const int t4h = ((tile * 16) % 768 & 63) >> 4;
    hdsc = a.desc[row];
    const float* cs = a.rope_cs + (size_t)row * 64 + 8 * t4h;
    hc0 = *reinterpret_cast<const float4*>(cs); hc1 = *reinterpret_cast<const float4*>(cs + 4);
    hs0 = *reinterpret_cast<const float4*>(cs + 32); hs1 = *reinterpret_cast<const float4*>(cs + 36);
  }
  const int nper = KCH / NW;  // chunks per wave (launcher guarantees nper % U == 0)
  const u128* wp = reinterpret_cast<const u128*>(a.Wp) + ((size_t)tile * KCH + wave * nper) * 64 + lane;
  const u128* wp2 = wp + (size_t)(N >> 4) * KCH * 64;  // SILU: the "up" tile of the same columns
  const u128* ap = reinterpret_cast<const u128*>(a.Ap) + ((size_t)mt0 * KCH + wave * nper) * 64 + lane;
  const bool w_once = a.w_nt && row_groups == 1;  // a single row group reads W: stream it past the caches

  f32x4 acc[NACC][NMB];
#pragma unroll
  for (int na = 0; na < NACC; ++na)
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) acc[na][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int i = 0; i < nper; i += U) {
    u128 af[NMB][U];
    if (i > 0) {   // the first round's weight fragments were requested at kernel entry (before *n_active was known)
      // (the policy test is hoisted out of the unrolled loads: a per-load select makes hipcc branch around every load)
      if (w_once) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
          wf[0][j] = load16_nt(wp + (size_t)(i + j) * 64);
          if (NACC == 2) wf[1][j] = load16_nt(wp2 + (size_t)(i + j) * 64);
        }
      } else {
#pragma unroll
        for (int j = 0; j < U; ++j) {
          wf[0][j] = load16(wp + (size_t)(i + j) * 64);
          if (NACC == 2) wf[1][j] = load16(wp2 + (size_t)(i + j) * 64);
        }
      }
    }
    if (NMB == 1 && a_pre && i == 0) {   // small batches: row tile 0's fragments were requested at kernel entry
#pragma unroll
      for (int j = 0; j < U; ++j) af[0][j] = af0[j];
    } else {
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int j = 0; j < U; ++j) af[mb][j] = load16(ap + ((size_t)mb * KCH + i + j) * 64);
    }
    // every load of the round in flight before the first MFMA (hipcc otherwise sinks each load next to its use)
    __builtin_amdgcn_sched_barrier(0);
    if (i == 0) STAMP(2);
    if constexpr (HO) {   // (registers: 2 workgroups of 8 waves per CU need <= 128) everything that only waits for the epilogue goes to LDS now
      if (i == 0) {
        if (hmeta) {
          float* o = cs_s[lane];
          o[0] = hc0.x; o[1] = hc0.y; o[2] = hc0.z; o[3] = hc0.w; o[4] = hc1.x; o[5] = hc1.y; o[6] = hc1.z; o[7] = hc1.w;
          o[8] = hs0.x; o[9] = hs0.y; o[10] = hs0.z; o[11] = hs0.w; o[12] = hs1.x; o[13] = hs1.y; o[14] = hs1.z; o[15] = hs1.w;
          meta_s[lane][0] = hdsc.b; meta_s[lane][1] = hdsc.slot;
        }
        if (SCALE) {
          float sq = (((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w))) + ((s2.x + s2.y) + (s2.z + s2.w));
          sq += __shfl_xor(sq, 1, 64);
          sq += __shfl_xor(sq, 2, 64);
          if (spart == 0 && srow < 16 * NMB) rstd_s[srow] = 1.0f / sqrtf(sq / 768.0f + a.eps);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int j = 0; j < U; ++j)
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int na = 0; na < NACC; ++na)
          acc[na][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&af[mb][j]),
                                                                *reinterpret_cast<const bf16x8*>(&wf[na][j]), acc[na][mb], 0, 0, 0);
  }

  STAMP(3);
  if (SCALE && !HO) {
    float s = (((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w))) + ((s2.x + s2.y) + (s2.z + s2.w));
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (spart == 0 && srow < 16 * NMB) rstd_s[srow] = 1.0f / sqrtf(s / 768.0f + a.eps);
  }
#pragma unroll
  for (int na = 0; na < NACC; ++na)
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) *reinterpret_cast<f32x4*>(&red[wave][na][mb][lane][0]) = acc[na][mb];
  __syncthreads();
  STAMP(4);
  if (wave >= NF) return;

  float vsum[PPW], usum[PPW];
#pragma unroll
  for (int q = 0; q < PPW; ++q) { vsum[q] = 0.f; usum[q] = 0.f; }
#pragma unroll
  for (int w = 0; w < NW; ++w) {      // fixed order: deterministic
    float tv[4], tu[4];
    if constexpr (PPW == 4) {
      *reinterpret_cast<f32x4*>(tv) = *reinterpret_cast<const f32x4*>(&red[w][0][fmb][lane][0]);
      if (EPI == FEPI_SILU) *reinterpret_cast<f32x4*>(tu) = *reinterpret_cast<const f32x4*>(&red[w][NACC - 1][fmb][lane][0]);
    } else if constexpr (PPW == 2) {
      *reinterpret_cast<float2*>(tv) = *reinterpret_cast<const float2*>(&red[w][0][fmb][lane][fr0]);
      if (EPI == FEPI_SILU) *reinterpret_cast<float2*>(tu) = *reinterpret_cast<const float2*>(&red[w][NACC - 1][fmb][lane][fr0]);
    } else {
      tv[0] = red[w][0][fmb][lane][fr0];
      if (EPI == FEPI_SILU) tu[0] = red[w][NACC - 1][fmb][lane][fr0];
    }
#pragma unroll
    for (int q = 0; q < PPW; ++q) { vsum[q] += tv[q]; if (EPI == FEPI_SILU) usum[q] += tu[q]; }
  }

  const int col = n0 + li;
  // ROPE tiles (weights permuted by the loader): columns 0..7 of a q/k tile are dims d0..d0+7 of one head, columns
  // 8..15 are dims d0+32..d0+39, so a rotate-half pair sits 8 lanes apart.
  const int sect = n0 / 768, hcol = n0 % 768, head = hcol >> 6, t4 = (hcol & 63) >> 4;
  const int dlo = 8 * t4 + (li & 7);
#pragma unroll
  for (int q = 0; q < PPW; ++q) {
    const int mb = fmb, r = fr0 + q;
    const int row = m0 + 16 * mb + 4 * g + r;
    float v = vsum[q], u = usum[q];
    const bool ok = row < M;
    if (SCALE) {
      const float rs = rstd_s[16 * mb + 4 * g + r];
      v *= rs;
      u *= rs;
    }
    if (EPI == FEPI_STORE32) {
      if (ok) a.C32[(size_t)row * a.ldc + col] = v;
    } else if (EPI == FEPI_SILU) {
      if (ok) a.Cp[pk_off(row, col, a.kch_out)] = f32_to_bf16(silu_f(v) * u);
    } else if (EPI == FEPI_RES) {
      float xn = 0.f;
      if (ok) {
        xn = pre0[q] + v;
        a.C32[(size_t)row * a.ldc + col] = xn;
        a.Cp[pk_off(row, col, a.kch_out)] = f32_to_bf16(xn);
        if (a.Cp32 != nullptr) a.Cp32[pk32_off(row, col, 768 / 16)] = xn;
      }
      float sq = xn * xn;
      sq += __shfl_xor(sq, 1, 64);
      sq += __shfl_xor(sq, 2, 64);
      sq += __shfl_xor(sq, 4, 64);
      sq += __shfl_xor(sq, 8, 64);
      if (li == 0 && ok) a.ssq_out[(size_t)row * SSQ_PARTS + tile] = sq;
    } else {  // FEPI_QKV_ROPE: q -> roped, f32 qkv buffer; k -> roped, KV cache; v -> KV cache
      const float other = __shfl_xor(v, 8, 64);
      const bool hi = li >= 8;
      const int lr = 16 * mb + 4 * g + r;
      const float cc = cs_s[lr][li & 7], ss = cs_s[lr][8 + (li & 7)];
      const int mb_b = meta_s[lr][0], mb_slot = meta_s[lr][1];
      // rotate-half: out[d] = x[d] c - x[d+32] s ; out[d+32] = x[d+32] c + x[d] s
      const float roped = hi ? (v * cc + other * ss) : (v * cc - other * ss);
      const int d = dlo + (hi ? 32 : 0);
      if (ok && mb_b >= 0) {
        const size_t cbase = (((size_t)mb_b * 12 + head) * a.cmax + mb_slot) * 64;
        if constexpr (HO) {   // consumed by attention units of the SAME launch, on other XCDs: write-through (sc1) stores
          if (sect == 0) __hip_atomic_store(a.C32 + (size_t)row * a.ldc + head * 64 + d, roped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else if (sect == 1) __hip_atomic_store(a.kc + cbase + d, f32_to_bf16(roped), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else __hip_atomic_store(a.vc + cbase + (hcol & 63) + li, f32_to_bf16(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
        if (sect == 0) a.C32[(size_t)row * a.ldc + head * 64 + d] = roped;
        else if (sect == 1) a.kc[cbase + d] = f32_to_bf16(roped);
        else a.vc[cbase + (hcol & 63) + li] = f32_to_bf16(v);
        }
      }
    }
  }
  if constexpr (HO && EPI == FEPI_QKV_ROPE) {
    // hand-off (MI355X guide, Guideline 16 R1 in its counter form): every finishing wave drains its sc1 stores and checks in at the
    // workgroup's LDS counter; the last one adds ONE arrival to the head's word (one relaxed agent atomic per tile: a per-wave, per-copy
    // add was 18 k atomics per launch on one memory channel -- 46 us per launch, profiles/r4c_ab_qkv_att.log).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int last = 0;
    if (lane == 0) last = atomicAdd(ho_cnt, 1) == NF - 1;
    if (__builtin_amdgcn_readfirstlane(last))
      if (lane == 0) __hip_atomic_fetch_add(a.ho_flag + head * HO_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  STAMP(5);
#undef STAMP
}

// one workgroup of the launch: weight tile `tile`, row tiles mt0 .. mt0 + MBT - 1 (mt0 = MBT * row group); `row_groups` = row groups
// of the launch (1: the weights are read once -> non-temporal), `wg_linear` indexes the probes' stamp buffer
template <int MBT, int NW, bool SCALE, int EPI, bool HO = false, int U_ = DEC_U>
__device__ __forceinline__ void gemm_dec_wg(const DecGemmArgs& a, const int tile, const int mt0, const int row_groups, const int wg_linear,
                                            const int n_wg) {
  constexpr int NACC = (EPI == FEPI_SILU) ? 2 : 1;
  __shared__ __attribute__((aligned(16))) float red[NW][NACC][MBT][64][4];
  __shared__ float rstd_s[16 * MBT];
  __shared__ float cs_s[(EPI == FEPI_QKV_ROPE) ? 16 * MBT : 1][16];   // per row: cos[8], sin[8] of this tile's dims
  __shared__ int meta_s[(EPI == FEPI_QKV_ROPE) ? 16 * MBT : 1][2];    // per row: utterance b (-1: finished), KV slot
  __shared__ int ho_cnt_s;   // HO: finishing waves that have drained their stores (zeroed here, read behind the body's __syncthreads)
  if (HO && threadIdx.x == 0) ho_cnt_s = 0;

  // QKV: the RoPE helper wave (index NW) first pulls weights of later launches of the step towards this XCD's L2 (common.hpp).  An
  // extra wave for this in the gate/up kernel cost that kernel +0.6 us by itself (profiles/r3s_ab_prefetch.log): not there.
#ifdef CTTS_PF_BUILD
  if (!HO && EPI == FEPI_QKV_ROPE && (threadIdx.x >> 6) == NW) {
    prefetch_weight_tiles(a.pf[0], threadIdx.x & 63, (unsigned)wg_linear, (unsigned)n_wg);
    prefetch_weight_tiles(a.pf[1], threadIdx.x & 63, (unsigned)wg_linear, (unsigned)n_wg);
  }
#else
  (void)n_wg;
#endif
  if (a.dbg && threadIdx.x == 0) a.dbg[(size_t)wg_linear * 8] = wall_clock64();
  // The weight fragments of the first round do not depend on anything but the kernel arguments: request them before
  // the live-row count (a dependent scalar load) is known.  Decode weights are read by one row group (<= 64 live
  // rows with MBT = 4; the 16-row workgroups of o/down re-read them from L2), streamed non-temporal when so.
  u128 wf[NACC][U_], af0[U_];
  const bool is_helper = !HO && EPI == FEPI_QKV_ROPE && (threadIdx.x >> 6) == NW;
  // Batches of <= 16 utterances (BASELINE C2: batch 1) are ONE row tile whose buffer exists whatever the live count is: its
  // activation fragments are requested here too, so the kernel's critical path is one memory round trip (weights || activations)
  // instead of live-count -> activations.  (With more row tiles this measured slower: profiles/r2d_*, r2v_*.)
  const bool a_pre = a.a_early && a.M <= 16 && mt0 == 0;
  if (!is_helper) {
    const int KCH = a.K >> 5, nper = KCH / NW, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (a_pre) {
      const u128* ap0 = reinterpret_cast<const u128*>(a.Ap) + ((size_t)wave * nper) * 64 + lane;
#pragma unroll
      for (int j = 0; j < U_; ++j) af0[j] = load16(ap0 + (size_t)j * 64);
    }
    const u128* wp = reinterpret_cast<const u128*>(a.Wp) + ((size_t)tile * KCH + wave * nper) * 64 + lane;
    const u128* wp2 = wp + (size_t)(a.N >> 4) * KCH * 64;
    const bool w_once = a.w_nt && row_groups == 1;
    if (w_once) {
#pragma unroll
      for (int j = 0; j < U_; ++j) {
        wf[0][j] = load16_nt(wp + (size_t)j * 64);
        if (NACC == 2) wf[1][j] = load16_nt(wp2 + (size_t)j * 64);
      }
    } else {
#pragma unroll
      for (int j = 0; j < U_; ++j) {
        wf[0][j] = load16(wp + (size_t)j * 64);
        if (NACC == 2) wf[1][j] = load16(wp2 + (size_t)j * 64);
      }
    }
  }
  const int M = a.n_active ? min(*a.n_active, a.M) : a.M;   // live (compact) rows
  if (mt0 * 16 >= M) return;
  const int nmb = min(MBT, (M - mt0 * 16 + 15) >> 4);

  if (is_helper) {
    // helper wave: row descriptor -> cos/sin of the row's position, beside the main waves' load phase
    const int lane = threadIdx.x & 63;
    if (lane < 16 * nmb) {
      const int row = min(mt0 * 16 + lane, M - 1);
      const RowDesc d = a.desc[row];
      const int t4 = ((tile * 16) % 768 & 63) >> 4;
      const float4 c0 = *reinterpret_cast<const float4*>(a.cos_t + d.pos * 32 + 8 * t4);
      const float4 c1 = *reinterpret_cast<const float4*>(a.cos_t + d.pos * 32 + 8 * t4 + 4);
      const float4 s0 = *reinterpret_cast<const float4*>(a.sin_t + d.pos * 32 + 8 * t4);
      const float4 s1 = *reinterpret_cast<const float4*>(a.sin_t + d.pos * 32 + 8 * t4 + 4);
      float* o = cs_s[lane];
      o[0] = c0.x; o[1] = c0.y; o[2] = c0.z; o[3] = c0.w; o[4] = c1.x; o[5] = c1.y; o[6] = c1.z; o[7] = c1.w;
      o[8] = s0.x; o[9] = s0.y; o[10] = s0.z; o[11] = s0.w; o[12] = s1.x; o[13] = s1.y; o[14] = s1.z; o[15] = s1.w;
      meta_s[lane][0] = d.b; meta_s[lane][1] = d.slot;
    }
    __syncthreads();
    return;
  }

  if constexpr (MBT == 1) {
    dec_body<1, MBT, NW, SCALE, EPI, HO, U_>(a, M, tile, mt0, row_groups, wg_linear, wf, red, rstd_s, cs_s, meta_s, af0, a_pre, &ho_cnt_s);
  } else if constexpr (MBT == 2) {
    if (nmb == 1) dec_body<1, MBT, NW, SCALE, EPI, HO, U_>(a, M, tile, mt0, row_groups, wg_linear, wf, red, rstd_s, cs_s, meta_s, af0, a_pre, &ho_cnt_s);
    else dec_body<2, MBT, NW, SCALE, EPI, HO, U_>(a, M, tile, mt0, row_groups, wg_linear, wf, red, rstd_s, cs_s, meta_s, af0, a_pre, &ho_cnt_s);
  } else {
    if (nmb == 1) dec_body<1, MBT, NW, SCALE, EPI, HO, U_>(a, M, tile, mt0, row_groups, wg_linear, wf, red, rstd_s, cs_s, meta_s, af0, a_pre, &ho_cnt_s);
    else if (nmb == 2) dec_body<2, MBT, NW, SCALE, EPI, HO, U_>(a, M, tile, mt0, row_groups, wg_linear, wf, red, rstd_s, cs_s, meta_s, af0, a_pre, &ho_cnt_s);
    else if (nmb == 3) dec_body<3, MBT, NW, SCALE, EPI, HO, U_>(a, M, tile, mt0, row_groups, wg_linear, wf, red, rstd_s, cs_s, meta_s, af0, a_pre, &ho_cnt_s);
    else dec_body<4, MBT, NW, SCALE, EPI, HO, U_>(a, M, tile, mt0, row_groups, wg_linear, wf, red, rstd_s, cs_s, meta_s, af0, a_pre, &ho_cnt_s);
  }
}

