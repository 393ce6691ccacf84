// Split-bf16 ("bf16x3") GEMM of the acoustic decoder's ConvNeXt point-wise layers on PRE-SPLIT, FRAGMENT-PACKED operands, staged by
// LDS-DMA.  Reference ops: `ConvNeXtBlock.pwconv1 -> GELU -> pwconv2 -> * gamma -> + residual`
// (/root/reference/ChatTTS/model/dvae.py:46-66 and vocos.modules.ConvNeXtBlock) -- 40 of the 45 GEMM launches and 95 % of the
// flops of DVAE decode + Vocos.
//
// gemm_tiled_bf16x3_k (gemm.hip) takes f32 activations: every k-step each wave pulls its share of the tile into registers,
// splits x = hi + lo on the VALU, writes four LDS planes, and only then multiplies -- and the 8 waves of a workgroup do these
// phases in lock-step, so a 256x256x32 step costs about their SUM (4.4 us against 1.3 us of MFMA issue, round-1 phase
// probe).  Here nothing but the multiply is left to the waves:
//   * the PRODUCERS write both bf16 planes (the depthwise-conv + LayerNorm kernel, and this kernel's own GELU epilogue), in
//     MFMA fragment order:   plane[row / 32][k / 16][hi | lo][lane = (k % 16) / 8 * 32 + row % 32][k % 8]
//     -- one (32-row tile, 16-wide k block, plane) is one contiguous KiB, exactly what v_mfma_f32_32x32x16_bf16 takes as A / B;
//     weights are packed the same way once at load;
//   * a 16-wide k block of the 256x256 tile is 32 such KiB fragments (16 A + 16 W) copied global -> LDS by
//     `global_load_lds_dwordx4` (4 per wave, no VGPR round trip, no ds_write) into a ring of 4 slots: the DMA runs THREE k blocks
//     ahead of the MFMAs (96 KiB in flight per CU), kept in flight across the per-block barrier by counted `s_waitcnt vmcnt(8)`
//     + raw `s_barrier` (a __syncthreads would drain it to one block: measured 3.8 us per 32-wide step that way);
//   * fragment reads are lane-linear ds_read_b128 (conflict free), 12 per wave and k block for 24 MFMAs.
// Numerics are those of gemm_tiled_bf16x3_k: a.w ~= a_lo w_hi + a_hi w_lo + a_hi w_hi, f32 accumulation over k in order.
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"

__device__ __forceinline__ size_t x3p_off(int r, int k, int p, int kb16) {
  return ((((size_t)(r >> 5) * kb16 + (k >> 4)) * 2 + p) * 64 + (((k & 15) >> 3) << 5) + (r & 31)) * 8 + (k & 7);
}

// the fp16 plane of gemm_h1p_k (below): the x3p layout without the hi | lo axis
__device__ __forceinline__ size_t h1p_off(int r, int k, int kb16) {
  return (((size_t)(r >> 5) * kb16 + (k >> 4)) * 64 + (((k & 15) >> 3) << 5) + (r & 31)) * 8 + (k & 7);
}

// ---- epilogues, shared by the 256 x 256 tiles (one workgroup of 8 waves per CU) and the 128 x 256 tiles (two workgroups of 4 waves
// per CU, round 6).  A wave owns 64 rows x 128 columns = acc[2][4] MFMA blocks of 32 x 32 whose first row / column are mb / nb.
// C layout of a block: one column per lane (lane & 31), 16 rows: rr(r) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
template <bool FULL>
__device__ __forceinline__ void epi_scale_res(const X3pArgs& a, const f32x16 (&acc)[2][4], int mb, int nb, int lane, int M) {
  // X3P_SCALE_RES: C = res + gamma * (acc + bias), f32 row-major (the residual stream the depthwise conv reads).  C and res are the
  // SAME buffer (the residual stream is updated in place): written element by element, every load would have to wait for the
  // previous store (may-alias), one memory round trip per element -- 62 of a 122 us tile (profiles/r3ag_h1p_phase_probe.log).  Each
  // thread reads and writes only its own elements, so a column block's 32 residuals are requested together, then the 32 results are
  // stored.  FULL: every row of the tile exists -> straight-line loads and stores (the ragged last tile clamps and predicates).
  // ADDRESSING (round 6): one 32-bit byte offset per lane and column block + a UNIFORM row offset per element (scalar base +
  // vector offset form of global_load / global_store).  With a 64-bit address per element the 64 addresses of a column block sat
  // in 128 VGPRs next to the 128 accumulators: since round 3 every SCALE_RES kernel had been spilling 131-135 VGPRs (472-496 bytes
  // of scratch per lane, -Rpass-analysis=kernel-resource-usage) around its epilogue.  The launcher checks M * ld * 4 < 2^32.
  // ORDER (round 6): the vector-memory counter is in order, stores included -- a wait for a load drains every store issued before
  // it.  The 34 loads of column block j + 1 used to follow the 32 stores of block j: four store drains per tile, and the same in
  // the GELU epilogue (a bias load per block behind the previous block's stores: eight drains, 10 us of a 27 us tile,
  // profiles/r3ah_h1p_phase_probe.log).  Now bias / gamma are loaded once, up front, and the residuals of unit u + 1 (16 rows of
  // one column block) are requested BEFORE the results of unit u are stored: a wait only ever covers loads.
  const char* resb = reinterpret_cast<const char*>(a.res);
  char* cb = reinterpret_cast<char*>(a.C);
  const int lrow = mb + 4 * (lane >> 5);
  float bias[4], gam[4];
  uint32_t off_r[4], off_c[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = nb + j * 32 + (lane & 31);
    bias[j] = a.bias[col];
    gam[j] = a.gamma[col];
    off_r[j] = ((uint32_t)lrow * (uint32_t)a.ldr + (uint32_t)col) * 4u;
    off_c[j] = ((uint32_t)lrow * (uint32_t)a.ldc + (uint32_t)col) * 4u;
  }
  float rv[2][16];
  auto load_unit = [&](int u, float* v) {   // unit u = (column block u / 2, row block u % 2)
    const int j = u >> 1, i = u & 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dr = i * 32 + (r & 3) + 8 * (r >> 2);   // compile-time row offset inside the wave's 64 rows
      if (FULL) {
        v[r] = __builtin_nontemporal_load(reinterpret_cast<const float*>(resb + (size_t)dr * a.ldr * 4 + off_r[j]));
      } else {   // clamped, never predicated
        const uint32_t o = ((uint32_t)min(lrow + dr, M - 1) * (uint32_t)a.ldr + (uint32_t)(nb + j * 32 + (lane & 31))) * 4u;
        v[r] = __builtin_nontemporal_load(reinterpret_cast<const float*>(resb + o));
      }
    }
  };
  load_unit(0, rv[0]);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int j = u >> 1, i = u & 1;
    if (u + 1 < 8) load_unit(u + 1, rv[(u + 1) & 1]);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dr = i * 32 + (r & 3) + 8 * (r >> 2);
      if (FULL || lrow + dr < M) *reinterpret_cast<float*>(cb + (size_t)dr * a.ldc * 4 + off_c[j]) = rv[u & 1][r] + gam[j] * (acc[i][j][r] + bias[j]);
    }
  }
}

// X3P_GELU_PACKED: C layout -> 8 consecutive columns of one row per lane through a wave-private LDS tile (`scr`, 32 x 36 floats):
// bias + GELU, then the operand planes of the NEXT GEMM (K' = N) in fragment order.  H1P: one fp16 plane (gelu_fast: within
// 1.5e-7 |x| of the erf form, common.hpp; a 2^-11 rounding follows); else the hi | lo bf16 planes (libm erff: 6 % of the pass's
// point-wise GEMM time over gelu_fast, profiles/r6z_x3p_gelu_ab.log, kept -- this is the f32-class decoder).
// Rows >= M land in the buffer's padding (allocated to a multiple of 256 rows).
template <bool H1P>
__device__ __forceinline__ void epi_gelu_packed(const X3pArgs& a, const f32x16 (&acc)[2][4], float* scr, int mb, int nb, int lane) {
  constexpr int FRAG = 512;
  const int nb16 = a.N >> 4;
  float bias4[4];   // loaded up front: a load between two blocks' stores would drain the stores (in-order vmcnt, see epi_scale_res)
#pragma unroll
  for (int j = 0; j < 4; ++j) bias4[j] = a.bias[nb + j * 32 + (lane & 31)];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cb = nb + j * 32;
      const float bias = bias4[j];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        scr[rr * 36 + (lane & 31)] = H1P ? gelu_fast(acc[i][j][r] + bias) : gelu_erf(acc[i][j][r] + bias);
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int item = lane + 64 * it, rr = item >> 2, cg = item & 3;
        const float4 v0 = *reinterpret_cast<const float4*>(scr + rr * 36 + cg * 8);
        const float4 v1 = *reinterpret_cast<const float4*>(scr + rr * 36 + cg * 8 + 4);
        const int row = mb + i * 32 + rr;
        if (H1P) {
          *reinterpret_cast<uint4*>(a.Cp + h1p_off(row, cb + cg * 8, nb16)) =
              make_uint4(pack_f16x2(v0.x, v0.y), pack_f16x2(v0.z, v0.w), pack_f16x2(v1.x, v1.y), pack_f16x2(v1.z, v1.w));
        } else {
          const uint32_t h0 = pack_bf16x2(v0.x, v0.y), h1 = pack_bf16x2(v0.z, v0.w), h2 = pack_bf16x2(v1.x, v1.y), h3 = pack_bf16x2(v1.z, v1.w);
          const uint32_t l0 = pack_bf16x2(v0.x - __uint_as_float(h0 << 16), v0.y - __uint_as_float(h0 & 0xffff0000u));
          const uint32_t l1 = pack_bf16x2(v0.z - __uint_as_float(h1 << 16), v0.w - __uint_as_float(h1 & 0xffff0000u));
          const uint32_t l2 = pack_bf16x2(v1.x - __uint_as_float(h2 << 16), v1.y - __uint_as_float(h2 & 0xffff0000u));
          const uint32_t l3 = pack_bf16x2(v1.z - __uint_as_float(h3 << 16), v1.w - __uint_as_float(h3 & 0xffff0000u));
          const size_t o = x3p_off(row, cb + cg * 8, 0, nb16);
          *reinterpret_cast<uint4*>(a.Cp + o) = make_uint4(h0, h1, h2, h3);
          *reinterpret_cast<uint4*>(a.Cp + o + FRAG) = make_uint4(l0, l1, l2, l3);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
}

template <int EPI, int VAR>
__global__ __launch_bounds__(512, 2) void gemm_x3p_k(X3pArgs a) {
  constexpr int BM = 256, BN = 256;
  constexpr int FRAG = 512;                 // bf16 elements of one fragment (1 KiB)
  constexpr int SLOT = 32 * FRAG;           // one ring slot = one 16-wide k block of the tile: 16 A + 16 W fragments = 32 KiB
  constexpr int NSLOT = 4;                  // ring of 4 slots (128 KiB): the DMA runs 3 k blocks ahead of the MFMAs
  __shared__ __attribute__((aligned(16))) uint16_t lds[NSLOT * SLOT];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave & 3, wn = wave >> 2;  // 4 x 2 waves, each 64 rows x 128 columns = 2 x 4 MFMA blocks of 32 x 32
  const int M = a.M, N = a.N, K = a.K;
  // XCD-aware tile order (workgroup L runs on XCD L % 8, each XCD has its own L2): a contiguous run of tiles per XCD, so the
  // column tiles that share one activation row panel hit the same L2
  const int nx = N / BN, ny = (M + BM - 1) / BM, T = nx * ny, per = (T + 7) / 8;
  const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || t >= T) return;
  const int m0 = (t / nx) * BM, n0 = (t % nx) * BN;
  const int kb16 = K >> 4;                  // k blocks = ring steps

  // wave w stages row tile w of the A panel and column tile w of the W panel: per k block the hi and the lo fragment, 2 contiguous KiB
  const uint16_t* ag = a.Ap + ((size_t)((m0 >> 5) + wave) * kb16) * 2 * FRAG + lane * 8;
  const uint16_t* wg = a.Wp + ((size_t)((n0 >> 5) + wave) * kb16) * 2 * FRAG + lane * 8;
  auto issue = [&](int kb) {   // 4 LDS-DMA pieces of 1 KiB per wave
    uint16_t* la = lds + (kb & (NSLOT - 1)) * SLOT + wave * 2 * FRAG;
    uint16_t* lw = la + 16 * FRAG;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ag + ((size_t)kb * 2 + p) * FRAG),
                                       (__attribute__((address_space(3))) void*)(la + p * FRAG), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wg + ((size_t)kb * 2 + p) * FRAG),
                                       (__attribute__((address_space(3))) void*)(lw + p * FRAG), 16, 0, 0);
    }
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Ring protocol (MI355X guide, "pipelining across barriers": counted vmcnt + raw s_barrier, never a draining __syncthreads in
  // the loop).  Iteration s: (1) wait until THIS wave's 4 pieces of k block s have landed -- the pieces of blocks s+1, s+2 stay
  // in flight (vmcnt counts in order); (2) barrier: every wave's pieces of block s are in LDS, and every wave is done reading
  // slot (s-1) % 4; (3) refill that slot with block s+3; (4) multiply block s.
  // VAR 3 (probe): wave 0 accumulates 100 MHz realtime deltas: DMA wait, barrier, DMA issue, fragment reads, MFMAs
  long long tacc[5] = {0, 0, 0, 0, 0}, tprev = 0;
#define X3P_MARK(i) do { if (VAR == 3) { const long long tn = wall_clock64(); tacc[i] += tn - tprev; tprev = tn; } } while (0)
  if (VAR == 4) {
    // STAGES OF TWO k blocks (slots {0,1} / {2,3}), double-buffered: one barrier per 32-wide k step instead of one per 16 -- the phase
    // probe put 0.47 of a k block's 1.3 us on the barrier (profiles/r2k_x3p_phase_probe.log).  The refill of a stage is issued
    // right after the barrier that proves every wave has finished reading it (it was consumed in the previous iteration), so a
    // stage is in flight for exactly one iteration (48 MFMAs per wave); same accumulation order as the ring variants, same bits.
    const int np = kb16 >> 1;   // K % 32 == 0
    issue(0); issue(1);
    for (int p = 0; p < np; ++p) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (p + 1 < np) { issue(2 * p + 2); issue(2 * p + 3); }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int sl = (2 * p + h) & (NSLOT - 1);
        const uint16_t* la = lds + sl * SLOT + lane * 8;
        const uint16_t* lw = la + 16 * FRAG;
        bf16x8 fah[2], fal[2], fwh[4], fwl[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fah[i] = *reinterpret_cast<const bf16x8*>(la + ((wm * 2 + i) * 2 + 0) * FRAG);
          fal[i] = *reinterpret_cast<const bf16x8*>(la + ((wm * 2 + i) * 2 + 1) * FRAG);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          fwh[j] = *reinterpret_cast<const bf16x8*>(lw + ((wn * 4 + j) * 2 + 0) * FRAG);
          fwl[j] = *reinterpret_cast<const bf16x8*>(lw + ((wn * 4 + j) * 2 + 1) * FRAG);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[i], fwh[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fwl[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fwh[j], acc[i][j], 0, 0, 0);
      }
    }
  } else {
  issue(0);
  if (kb16 > 1) issue(1);
  if (kb16 > 2) issue(2);
  if (VAR == 3) tprev = wall_clock64();
  for (int s = 0; s < kb16; ++s) {
    if (s + 2 < kb16) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (s + 1 < kb16) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    X3P_MARK(0);
    __builtin_amdgcn_s_barrier();
    X3P_MARK(1);
    if (s + 3 < kb16) issue(s + 3);
    X3P_MARK(2);
    const uint16_t* la = lds + (s & (NSLOT - 1)) * SLOT + lane * 8;
    const uint16_t* lw = la + 16 * FRAG;
    bf16x8 fah[2], fal[2], fwh[4], fwl[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      fah[i] = *reinterpret_cast<const bf16x8*>(la + ((wm * 2 + i) * 2 + 0) * FRAG);
      fal[i] = *reinterpret_cast<const bf16x8*>(la + ((wm * 2 + i) * 2 + 1) * FRAG);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      fwh[j] = *reinterpret_cast<const bf16x8*>(lw + ((wn * 4 + j) * 2 + 0) * FRAG);
      fwl[j] = *reinterpret_cast<const bf16x8*>(lw + ((wn * 4 + j) * 2 + 1) * FRAG);
    }
    if (VAR == 3) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); X3P_MARK(3); }
    // per accumulator the order is lo.hi, hi.lo, hi.hi (small terms first).  VAR 1: term-major issue order, so that consecutive
    // MFMAs write DIFFERENT accumulators (8 independent ones between two dependent ones); VAR 2: + raised wave priority
    if (VAR == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[i], fwh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fwl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fwh[j], acc[i][j], 0, 0, 0);
        }
    } else {
      if (VAR == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[i], fwh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fwl[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fwh[j], acc[i][j], 0, 0, 0);
      if (VAR == 2) __builtin_amdgcn_s_setprio(0);
    }
    if (VAR == 3) {   // make the MFMAs' completion visible to the clock: touch one result register of the last MFMA group
      asm volatile("s_nop 15\n\ts_nop 15" :: "v"(acc[1][3][0]));
      X3P_MARK(4);
    }
  }
  }
  if (VAR == 3 && a.dbg != nullptr && tid == 0) {
    long long* d = a.dbg + (size_t)blockIdx.x * 8;
    for (int i = 0; i < 5; ++i) d[i] = tacc[i];
  }
#undef X3P_MARK
  __syncthreads();   // everybody is done with the ring: the epilogue below reuses it as scratch

  if (EPI == X3P_GELU_PACKED) epi_gelu_packed<false>(a, acc, reinterpret_cast<float*>(lds) + wave * (32 * 36), m0 + wm * 64, n0 + wn * 128, lane);
  else if (m0 + BM <= M) epi_scale_res<true>(a, acc, m0 + wm * 64, n0 + wn * 128, lane, M);
  else epi_scale_res<false>(a, acc, m0 + wm * 64, n0 + wn * 128, lane, M);
}

// ------------------------------------------------------------------------------------------------
// Round 6: the same two GEMMs on 128 x 256 tiles, TWO workgroups of four waves per CU.
// What the micro-benchmarks said (tools/fill_probe.hip, tools/loop_probe.hip; profiles/r6z_fill_probe.log, r6z_loop_probe.log) about
// the 256 x 256 kernel at the roofline shape (M 65,536, N 2048, K 512, fp16 planes: 268 us, MFMA floor 55): its operand movement alone
// runs at 94 GB/s per CU (24 TB/s over the chip: the L2 -> LDS fill is NOT the wall round 3 took it for), its k loop without an
// epilogue at 0.50 of the dense fp16 peak (110 us) -- the other 160 us are prologue and epilogue, during which the ONE workgroup a CU
// holds (128 KiB of ring) leaves the matrix pipe idle.  Here a workgroup holds a ring of 3 slots x 24 KiB (72 KiB), so two are resident
// per CU with independent barriers.  A slot = (4 A + 8 W row tiles) x 2 fragments: 32 of k on the fp16 plane (h1p), 16 of k x {hi, lo}
// on the split-bf16 planes (x3p); one barrier per slot, the slot after the one being multiplied already landed and the one after
// that in flight (counted vmcnt).  The loop moves 1.5 x the operand bytes per flop of the 256 x 256 tile and runs as fast (loop_probe
// variant 4: 114 vs 110 us).  MEASURED GAIN: modest on the large launches -- loop, GELU arithmetic (+42 us), transposes + stores
// (+56 us) add up almost linearly even with two workgroups per CU (193 us in the probe, staggering them changes nothing), the epilogue's
// VALU and store work is simply there -- and 25-30 % on launches whose 256 x 256 tiling leaves CUs empty (streaming windows, the DVAE
// decoder's narrow outputs).  The larger win of the round sits in the shared epilogue (epi_scale_res: no spills, loads ahead of stores).
// Same MFMAs in the same k order per accumulator, same epilogues: bit-identical to the 256 x 256 kernels
// (tests/test_gpu_kernels.py::test_codec_gemm_tilings_are_bit_identical).  Which launch takes which tiling: codec_tile128 below.
// ------------------------------------------------------------------------------------------------
#define T128_GLL(g, l) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g), (__attribute__((address_space(3))) void*)(l), 16, 0, 0)

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_h1p128_k(X3pArgs a) {
  constexpr int BM = 128, BN = 256, FRAG = 512, SLOT = 24 * FRAG, NSLOT = 3;
  __shared__ __attribute__((aligned(16))) uint16_t lds[NSLOT * SLOT];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave & 1, wn = wave >> 1;   // 2 x 2 waves, each 64 rows x 128 columns
  const int M = a.M, N = a.N, K = a.K;
  const int nx = N / BN, ny = (M + BM - 1) / BM, T = nx * ny, per = (T + 7) / 8;   // XCD-aware tile order, as gemm_x3p_k
  const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || t >= T) return;
  const int m0 = (t / nx) * BM, n0 = (t % nx) * BN;
  const int kb16 = K >> 4, nq = K >> 5;
  // wave w stages A row tile w and W row tiles w, 4 + w: per 32-wide k block two consecutive fragments each = 6 KiB-pieces
  const uint16_t* ag = a.Ap + ((size_t)((m0 >> 5) + wave) * kb16) * FRAG + lane * 8;
  const uint16_t* wg0 = a.Wp + ((size_t)((n0 >> 5) + wave) * kb16) * FRAG + lane * 8;
  const uint16_t* wg1 = a.Wp + ((size_t)((n0 >> 5) + 4 + wave) * kb16) * FRAG + lane * 8;
  auto issue = [&](int q) {   // slot layout: A row tile r at fragments 2 r + h, W row tile c at 8 + 2 c + h
    uint16_t* l = lds + (q % NSLOT) * SLOT + wave * 2 * FRAG;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      T128_GLL(ag + ((size_t)q * 2 + h) * FRAG, l + h * FRAG);
      T128_GLL(wg0 + ((size_t)q * 2 + h) * FRAG, l + (8 + h) * FRAG);
      T128_GLL(wg1 + ((size_t)q * 2 + h) * FRAG, l + (16 + h) * FRAG);
    }
  };
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  f16x8 fa0[2], fw0[4], fa1[2], fw1[4];
  auto rd = [&](f16x8* fa, f16x8* fw, int u) {   // fragments of 16-wide k block u: slot u / 2, half u % 2
    const uint16_t* l = lds + ((u >> 1) % NSLOT) * SLOT + lane * 8;
    const int h = u & 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const f16x8*>(l + ((wm * 2 + i) * 2 + h) * FRAG);
#pragma unroll
    for (int j = 0; j < 4; ++j) fw[j] = *reinterpret_cast<const f16x8*>(l + (8 + (wn * 4 + j) * 2 + h) * FRAG);
  };
  // as gemm_h1p_k: the block's first MFMA, then the next block's reads, then the other seven (which cover the reads' latency)
  auto mm_a = [&](const f16x8* fa, const f16x8* fw) { acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0], fw[0], acc[0][0], 0, 0, 0); };
  auto mm_b = [&](const f16x8* fa, const f16x8* fw) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (i + j > 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fw[j], acc[i][j], 0, 0, 0);
  };
#define T128_SB() __builtin_amdgcn_sched_barrier(0)
  issue(0);
  if (nq > 1) issue(1);
  if (nq > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (nq > 2) issue(2);
  rd(fa0, fw0, 0);
  for (int q = 0; q < nq; ++q) {
    T128_SB(); mm_a(fa0, fw0); T128_SB(); rd(fa1, fw1, 2 * q + 1); T128_SB(); mm_b(fa0, fw0);
    T128_SB(); mm_a(fa1, fw1); T128_SB();
    if (q + 1 < nq) {
      // this wave's pieces of slot q + 1 (its 6 of slot q + 2 may stay in flight); the barrier: everybody's pieces of q + 1 are in,
      // and everybody holds their last fragments of slot q in registers -> slot q is refilled with q + 3
      if (q + 2 < nq) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (q + 3 < nq) issue(q + 3);
      rd(fa0, fw0, 2 * q + 2);
      T128_SB();
    }
    mm_b(fa1, fw1);
  }
  __syncthreads();   // the ring becomes the epilogue's scratch
  if (EPI == X3P_GELU_PACKED) epi_gelu_packed<true>(a, acc, reinterpret_cast<float*>(lds) + wave * (32 * 36), m0 + wm * 64, n0 + wn * 128, lane);
  else if (m0 + BM <= M) epi_scale_res<true>(a, acc, m0 + wm * 64, n0 + wn * 128, lane, M);
  else epi_scale_res<false>(a, acc, m0 + wm * 64, n0 + wn * 128, lane, M);
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_x3p128_k(X3pArgs a) {
  constexpr int BM = 128, BN = 256, FRAG = 512, SLOT = 24 * FRAG, NSLOT = 3;
  __shared__ __attribute__((aligned(16))) uint16_t lds[NSLOT * SLOT];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave & 1, wn = wave >> 1;
  const int M = a.M, N = a.N, K = a.K;
  const int nx = N / BN, ny = (M + BM - 1) / BM, T = nx * ny, per = (T + 7) / 8;
  const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || t >= T) return;
  const int m0 = (t / nx) * BM, n0 = (t % nx) * BN;
  const int kb16 = K >> 4;
  // wave w stages A row tile w and W row tiles w, 4 + w: per 16-wide k block the hi and the lo fragment, 2 contiguous KiB each
  const uint16_t* ag = a.Ap + ((size_t)((m0 >> 5) + wave) * kb16) * 2 * FRAG + lane * 8;
  const uint16_t* wg0 = a.Wp + ((size_t)((n0 >> 5) + wave) * kb16) * 2 * FRAG + lane * 8;
  const uint16_t* wg1 = a.Wp + ((size_t)((n0 >> 5) + 4 + wave) * kb16) * 2 * FRAG + lane * 8;
  auto issue = [&](int kb) {   // slot layout: A row tile r at fragments 2 r + {hi, lo}, W row tile c at 8 + 2 c + {hi, lo}
    uint16_t* l = lds + (kb % NSLOT) * SLOT + wave * 2 * FRAG;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      T128_GLL(ag + ((size_t)kb * 2 + p) * FRAG, l + p * FRAG);
      T128_GLL(wg0 + ((size_t)kb * 2 + p) * FRAG, l + (8 + p) * FRAG);
      T128_GLL(wg1 + ((size_t)kb * 2 + p) * FRAG, l + (16 + p) * FRAG);
    }
  };
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  issue(0);
  if (kb16 > 1) issue(1);
  for (int s = 0; s < kb16; ++s) {
    // this wave's pieces of block s (its 6 of block s + 1 may stay in flight); the barrier: everybody's pieces of s are in, and
    // everybody is done reading slot (s - 1) % 3 -> it is refilled with block s + 2
    if (s + 1 < kb16) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (s + 2 < kb16) issue(s + 2);
    const uint16_t* l = lds + (s % NSLOT) * SLOT + lane * 8;
    bf16x8 fah[2], fal[2], fwh[4], fwl[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      fah[i] = *reinterpret_cast<const bf16x8*>(l + ((wm * 2 + i) * 2 + 0) * FRAG);
      fal[i] = *reinterpret_cast<const bf16x8*>(l + ((wm * 2 + i) * 2 + 1) * FRAG);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      fwh[j] = *reinterpret_cast<const bf16x8*>(l + (8 + (wn * 4 + j) * 2 + 0) * FRAG);
      fwl[j] = *reinterpret_cast<const bf16x8*>(l + (8 + (wn * 4 + j) * 2 + 1) * FRAG);
    }
    // per accumulator lo.hi, hi.lo, hi.hi (small terms first), term-major: gemm_x3p_k's order
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[i], fwh[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fwl[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fwh[j], acc[i][j], 0, 0, 0);
  }
  __syncthreads();   // everybody is done with the ring: the epilogue reuses it as scratch
  if (EPI == X3P_GELU_PACKED) epi_gelu_packed<false>(a, acc, reinterpret_cast<float*>(lds) + wave * (32 * 36), m0 + wm * 64, n0 + wn * 128, lane);
  else if (m0 + BM <= M) epi_scale_res<true>(a, acc, m0 + wm * 64, n0 + wn * 128, lane, M);
  else epi_scale_res<false>(a, acc, m0 + wm * 64, n0 + wn * 128, lane, M);
}
#undef T128_SB
#undef T128_GLL

// Which tiling (profiles/r6z_tile_real_shapes_ab.log, both bit-identical): the fp16-plane kernel is faster on 128 x 256 tiles at every
// shape of the decoder (-3 ... -25 %, streaming windows -25 ... -30 %); the split-bf16 kernel (three MFMAs per staged byte, the erf epilogue)
// only where the 256 x 256 tiling leaves CUs without a tile (streaming windows, the DVAE decoder's 256-wide output: -25 ... -30 %) -- on
// the large launches its 256 x 256 tiles stay 3-6 % ahead.  CTTS_CODEC_TILE=256|128 forces one (A/B, the bit-identity test; read at every launch).
static bool codec_tile128(const X3pArgs& a, bool h1p) {
  const char* e = getenv("CTTS_CODEC_TILE");
  if (e && atoi(e) == 256) return false;
  if (e && atoi(e) == 128) return true;
  return h1p || (a.N / 256) * ((a.M + 255) / 256) < 256;
}
static hipError_t launch_tile128(const X3pArgs& a, bool h1p, hipStream_t st) {
  const int tiles = (a.N / 256) * ((a.M + 127) / 128);
  const dim3 grid(((tiles + 7) / 8) * 8);
  if (h1p) {
    if (a.epi == X3P_GELU_PACKED) CTTS_LAUNCH((gemm_h1p128_k<X3P_GELU_PACKED>), grid, dim3(256), st, a);
    else CTTS_LAUNCH((gemm_h1p128_k<X3P_SCALE_RES>), grid, dim3(256), st, a);
  } else {
    if (a.epi == X3P_GELU_PACKED) CTTS_LAUNCH((gemm_x3p128_k<X3P_GELU_PACKED>), grid, dim3(256), st, a);
    else CTTS_LAUNCH((gemm_x3p128_k<X3P_SCALE_RES>), grid, dim3(256), st, a);
  }
  return hipGetLastError();
}

template <int VAR>
static void x3p_launch(const X3pArgs& a, dim3 grid, hipStream_t st) {
  if (a.epi == X3P_GELU_PACKED) CTTS_LAUNCH((gemm_x3p_k<X3P_GELU_PACKED, VAR>), grid, dim3(512), st, a);
  else CTTS_LAUNCH((gemm_x3p_k<X3P_SCALE_RES, VAR>), grid, dim3(512), st, a);
}

hipError_t launch_gemm_x3p(const X3pArgs& a, hipStream_t st) {
  if (a.M <= 0 || (a.N % 256) != 0 || (a.K % 32) != 0 || a.K < 32) return hipErrorInvalidValue;
  if (a.epi != X3P_GELU_PACKED && a.epi != X3P_SCALE_RES) return hipErrorInvalidValue;
  if (a.epi == X3P_SCALE_RES && ((unsigned long long)a.M * (unsigned)a.ldc >= (1ull << 30) || (unsigned long long)a.M * (unsigned)a.ldr >= (1ull << 30)))
    return hipErrorInvalidValue;   // the residual epilogue's 32-bit byte offsets
  const int tiles = (a.N / 256) * ((a.M + 255) / 256);
  dim3 grid(((tiles + 7) / 8) * 8);
  static int var = -1;   // CTTS_X3P_VAR: MFMA issue order / priority variant (A/B, tools/x3p_probe.py)
  if (var < 0) { const char* e = getenv("CTTS_X3P_VAR"); var = e ? atoi(e) : 4; }
  if (var == 4 && codec_tile128(a, false)) return launch_tile128(a, false, st);
  // default 4: two k blocks per barrier (-1.7 ... -2.6 % per GEMM against the 4-slot ring of variant 1, profiles/r3q_x3p_probe.log; bit-identical)
  if (var < 0) { const char* e = getenv("CTTS_X3P_VAR"); var = e ? atoi(e) : 4; }
  if (var == 0) x3p_launch<0>(a, grid, st);
  else if (var == 2) x3p_launch<2>(a, grid, st);
  else if (var == 3) x3p_launch<3>(a, grid, st);
  else if (var == 4) x3p_launch<4>(a, grid, st);
  else x3p_launch<1>(a, grid, st);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// gemm_h1p_k (round 3, gemm_mode 2 = "f16"): the same 256 x 256 tile, LDS-DMA ring and epilogues on ONE fp16 plane per operand.
// The north-star bar for the acoustic decoder is a waveform within 1e-4 RMS of the float32 path; the split-bf16 kernel above lands
// at 4e-7 and pays three MFMAs per product and 4 bytes per staged element for it.  fp16 keeps 11 significant bits of both operands
// (f32 accumulation): 6e-6 RMS on the waveform (tests/test_gpu_e2e.py states the bound), one MFMA per product, 2 bytes per element
// in HBM, in the DMA and in the fragment reads.  Activations are saturated to +-65504 where they are rounded (pack_f16x2).
//   plane[row / 32][k / 16][lane = (k % 16) / 8 * 32 + row % 32][k % 8]   -- the x3p layout without the hi | lo axis
// A ring slot holds a 32-wide k block (two 16-wide fragments per 32-row tile where x3p holds hi | lo of one), a stage is two
// slots = 64 of k: one barrier per 32 MFMAs of a wave.
// ------------------------------------------------------------------------------------------------

// PROBE (CTTS_H1P_PROBE=1 + CTTS_X3_DBG_PTR, tools/x3p_phase_probe.py --h1p): wave 0 accumulates 100 MHz phase times.
// NS: ring slots.  4 = stages of two slots, one in flight (64 KiB) while the other is multiplied; 5 (CTTS_H1P_RING=5, all 160 KiB of
// the CU's LDS) = one slot per barrier, FOUR in flight (128 KiB) -- the A/B of "is the loop bound by the fill's latency or by its rate".
template <int EPI, bool PROBE = false, int NS = 4>
__global__ __launch_bounds__(512, 2) void gemm_h1p_k(X3pArgs a) {
  constexpr int BM = 256, BN = 256;
  constexpr int FRAG = 512;                 // fp16 elements of one fragment (1 KiB): 32 rows x 16 of k
  constexpr int SLOT = 32 * FRAG;           // one ring slot = a 32-wide k block of the tile: (8 A + 8 W row tiles) x 2 fragments = 32 KiB
  constexpr int NSLOT = NS;
  __shared__ __attribute__((aligned(16))) uint16_t lds[NSLOT * SLOT];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave & 3, wn = wave >> 2;  // 4 x 2 waves, each 64 rows x 128 columns = 2 x 4 MFMA blocks of 32 x 32
  const int M = a.M, N = a.N, K = a.K;
  const int nx = N / BN, ny = (M + BM - 1) / BM, T = nx * ny, per = (T + 7) / 8;   // XCD-aware tile order, as gemm_x3p_k
  const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || t >= T) return;
  const int m0 = (t / nx) * BM, n0 = (t % nx) * BN;
  const int kb16 = K >> 4;

  // wave w stages row tile w of the A panel and column tile w of the W panel: per 32-wide k block two consecutive fragments = 2 KiB each
  const uint16_t* ag = a.Ap + ((size_t)((m0 >> 5) + wave) * kb16) * FRAG + lane * 8;
  const uint16_t* wg = a.Wp + ((size_t)((n0 >> 5) + wave) * kb16) * FRAG + lane * 8;
  auto issue = [&](int q) {   // 32-wide k block q -> slot q % 4: 4 LDS-DMA pieces of 1 KiB per wave
    uint16_t* la = lds + (q % NSLOT) * SLOT + wave * 2 * FRAG;
    uint16_t* lw = la + 16 * FRAG;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ag + ((size_t)q * 2 + h) * FRAG),
                                       (__attribute__((address_space(3))) void*)(la + h * FRAG), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wg + ((size_t)q * 2 + h) * FRAG),
                                       (__attribute__((address_space(3))) void*)(lw + h * FRAG), 16, 0, 0);
    }
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Stages of two slots (64 of k), double-buffered, as gemm_x3p_k's variant 4: wait for this wave's pieces of the stage, barrier
  // (everybody's pieces are in, everybody has READ the other stage), refill the other stage, multiply this one.  On top of that the
  // fragment reads run ONE 16-wide k block ahead of the MFMAs (two register sets): with one product per MFMA the LDS pipe (384
  // cycles of fragment reads + 128 of DMA writes per k block and CU) is as busy as the matrix pipe (512), so a wave that reads,
  // waits, then multiplies leaves both idle half the time.  The reads of a stage's first block are issued right behind the
  // barrier and land under the previous stage's last 8 MFMAs, whose operands are already in registers.
  const int np = K >> 6;   // K % 64 == 0
  f16x8 fa0[2], fw0[4], fa1[2], fw1[4];
  auto rd = [&](f16x8* fa, f16x8* fw, int u) {   // fragments of 16-wide k block u
    const uint16_t* la = lds + ((u >> 1) % NSLOT) * SLOT + lane * 8;
    const uint16_t* lw = la + 16 * FRAG;
    const int h = u & 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const f16x8*>(la + ((wm * 2 + i) * 2 + h) * FRAG);
#pragma unroll
    for (int j = 0; j < 4; ++j) fw[j] = *reinterpret_cast<const f16x8*>(lw + ((wn * 4 + j) * 2 + h) * FRAG);
  };
  // mm_a: the block's first MFMA (hipcc puts its `s_waitcnt lgkmcnt(0)` for the block's fragments in front of it -- they were
  // requested a whole block earlier); the NEXT block's reads are issued behind it, so nothing younger is outstanding at that wait;
  // mm_b: the other 7, which cover the reads' latency
  auto mm_a = [&](const f16x8* fa, const f16x8* fw) { acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0], fw[0], acc[0][0], 0, 0, 0); };
  auto mm_b = [&](const f16x8* fa, const f16x8* fw) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (i + j > 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fw[j], acc[i][j], 0, 0, 0);
  };
#define H1P_SB() __builtin_amdgcn_sched_barrier(0)
  long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0, t_begin = 0;
#define H1P_MARK(i) do { if (PROBE) { const long long tn = wall_clock64(); tacc[i] += tn - tprev; tprev = tn; } } while (0)
  if (PROBE) t_begin = tprev = wall_clock64();
  if constexpr (NS != 4) {
    // one 32-wide slot per barrier, NS - 1 slots in flight behind it.  Waiting for slot t leaves the pieces of the slots issued
    // after it outstanding: min(nq - 1, t + NS - 2) - t slots x 4 pieces of this wave (vmcnt counts in order).
    const int nq = K >> 5;
    auto wait_slot = [&](int t) {
      const int younger = min(nq - 1, t + NS - 2) - t;
      if (younger >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    for (int q = 0; q < NS - 1 && q < nq; ++q) issue(q);
    wait_slot(0);
    __builtin_amdgcn_s_barrier();
    if (NS - 1 < nq) issue(NS - 1);
    rd(fa0, fw0, 0);
    H1P_MARK(0);
    for (int q = 0; q < nq; ++q) {
      H1P_SB(); mm_a(fa0, fw0); H1P_SB(); rd(fa1, fw1, 2 * q + 1); H1P_SB(); mm_b(fa0, fw0);
      H1P_SB(); mm_a(fa1, fw1); H1P_SB();
      H1P_MARK(1);
      if (q + 1 < nq) {   // slot boundary: this wave's fragments of slot q are in registers -> after the barrier the slot is refilled
        wait_slot(q + 1);
        H1P_MARK(2);
        __builtin_amdgcn_s_barrier();
        H1P_MARK(3);
        if (q + NS < nq) issue(q + NS);
        rd(fa0, fw0, 2 * q + 2);
        H1P_SB();
        H1P_MARK(4);
      }
      mm_b(fa1, fw1);
    }
  } else {
  issue(0); issue(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (np > 1) { issue(2); issue(3); }
  rd(fa0, fw0, 0);
  H1P_MARK(0);   // prologue: first stage requested, landed, barrier, second stage requested
  for (int p = 0; p < np; ++p) {
    const int u = 4 * p;
    H1P_SB(); mm_a(fa0, fw0); H1P_SB(); rd(fa1, fw1, u + 1); H1P_SB(); mm_b(fa0, fw0);
    H1P_SB(); mm_a(fa1, fw1); H1P_SB(); rd(fa0, fw0, u + 2); H1P_SB(); mm_b(fa1, fw1);
    H1P_SB(); mm_a(fa0, fw0); H1P_SB(); rd(fa1, fw1, u + 3); H1P_SB(); mm_b(fa0, fw0);
    H1P_SB(); mm_a(fa1, fw1); H1P_SB();
    H1P_MARK(1);   // 25 MFMAs issued, 18 fragment reads
    if (p + 1 < np) {   // stage boundary: every wave holds its last fragments of stage p in registers -> its slots may be refilled
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      H1P_MARK(2);   // this wave's DMA pieces of the next stage
      __builtin_amdgcn_s_barrier();
      H1P_MARK(3);   // everybody else
      if (p + 2 < np) { issue(2 * p + 4); issue(2 * p + 5); }
      rd(fa0, fw0, u + 4);
      H1P_SB();
      H1P_MARK(4);   // 8 DMA pieces + 6 reads issued
    }
    mm_b(fa1, fw1);
  }
  }
  if (PROBE) { asm volatile("s_nop 15\n\ts_nop 15" :: "v"(acc[1][3][0])); H1P_MARK(1); }
#undef H1P_SB
  __syncthreads();   // the ring becomes the epilogue's scratch

  if (EPI == X3P_GELU_PACKED) epi_gelu_packed<true>(a, acc, reinterpret_cast<float*>(lds) + wave * (32 * 36), m0 + wm * 64, n0 + wn * 128, lane);
  else if (m0 + BM <= M) epi_scale_res<true>(a, acc, m0 + wm * 64, n0 + wn * 128, lane, M);
  else epi_scale_res<false>(a, acc, m0 + wm * 64, n0 + wn * 128, lane, M);
  if (PROBE && a.dbg != nullptr && tid == 0) {
    H1P_MARK(5);   // epilogue (from the end of the k loop)
    long long* d = a.dbg + (size_t)blockIdx.x * 8;
    for (int i = 0; i < 6; ++i) d[i] = tacc[i];
    d[6] = wall_clock64() - t_begin;
  }
#undef H1P_MARK
}

hipError_t launch_gemm_h1p(const X3pArgs& a, hipStream_t st) {
  if (a.M <= 0 || (a.N % 256) != 0 || (a.K % 64) != 0 || a.K < 64) return hipErrorInvalidValue;
  if (a.epi != X3P_GELU_PACKED && a.epi != X3P_SCALE_RES) return hipErrorInvalidValue;
  if (a.epi == X3P_SCALE_RES && ((unsigned long long)a.M * (unsigned)a.ldc >= (1ull << 30) || (unsigned long long)a.M * (unsigned)a.ldr >= (1ull << 30)))
    return hipErrorInvalidValue;   // the residual epilogue's 32-bit byte offsets
  const int tiles = (a.N / 256) * ((a.M + 255) / 256);
  dim3 grid(((tiles + 7) / 8) * 8);
  static int probe = -1, ring = 4;
  if (probe < 0) {
    const char* e = getenv("CTTS_H1P_PROBE"); probe = (e && atoi(e) > 0) ? 1 : 0;
    const char* r = getenv("CTTS_H1P_RING"); if (r && atoi(r) == 5) ring = 5;
  }
  if (ring == 4 && !probe && codec_tile128(a, true)) return launch_tile128(a, true, st);
  if (ring == 5) {
    if (probe && a.dbg != nullptr) {
      if (a.epi == X3P_GELU_PACKED) CTTS_LAUNCH((gemm_h1p_k<X3P_GELU_PACKED, true, 5>), grid, dim3(512), st, a);
      else CTTS_LAUNCH((gemm_h1p_k<X3P_SCALE_RES, true, 5>), grid, dim3(512), st, a);
    } else {
      if (a.epi == X3P_GELU_PACKED) CTTS_LAUNCH((gemm_h1p_k<X3P_GELU_PACKED, false, 5>), grid, dim3(512), st, a);
      else CTTS_LAUNCH((gemm_h1p_k<X3P_SCALE_RES, false, 5>), grid, dim3(512), st, a);
    }
    return hipGetLastError();
  }
  if (probe && a.dbg != nullptr) {
    if (a.epi == X3P_GELU_PACKED) CTTS_LAUNCH((gemm_h1p_k<X3P_GELU_PACKED, true>), grid, dim3(512), st, a);
    else CTTS_LAUNCH((gemm_h1p_k<X3P_SCALE_RES, true>), grid, dim3(512), st, a);
    return hipGetLastError();
  }
  if (a.epi == X3P_GELU_PACKED) CTTS_LAUNCH((gemm_h1p_k<X3P_GELU_PACKED>), grid, dim3(512), st, a);
  else CTTS_LAUNCH((gemm_h1p_k<X3P_SCALE_RES>), grid, dim3(512), st, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Round 6: the ConvNeXt MLP in ONE launch (fp16 plane).  out = res + gamma * (GELU(A W1^T + b1) W2^T + b2) for a tile of 128 frames.
// VERDICT r5 item 4c.  BUILT, MEASURED, NOT THE DEFAULT (mlp_fused_pays below): bit-identical to the two launches, 3-10 % ahead of them at
// 65,536 frames, behind at every other shape.
//
// Why it was tried: as two launches the pair moves the 2048-wide activation through HBM (268 MB written, 537-805 MB read at 65,536
// frames) and each launch pays a prologue and an epilogue during which the matrix pipe idles.  Here a workgroup (8 waves, one per CU: the
// 128 x 512 f32 output tile is 128 accumulator registers per lane) walks the hidden dimension in CHUNKS of 128:
//   P1  H^T[128 hidden x 128 frames] = W1[chunk] A^T   (K = 512: 16 ring steps of 32)     2 MFMA blocks per wave (1 x 2)
//       bias + GELU + fp16 in registers -> the chunk's H as phase 3's A fragments in LDS (32 KiB), no transpose:
//       W1's rows enter the MFMA with bits 2 and 3 of the row index swapped (a lane permutation of the DMA's SOURCE addresses),
//       so that a lane's 16 accumulators of H^T are hidden units 16 (r / 8) + 8 (lane / 32) + r % 8 -- exactly the eight consecutive
//       k values per 16-block an A fragment holds; one ds_write_b128 per block and k block
//   P3  out[128 frames x 512] += H[chunk] W2[:, chunk]^T (K = 128: 8 ring steps of 16)    8 MFMA blocks per wave (2 x 4)
// and after the last chunk the residual epilogue of the two-launch path (epi_scale_res).  ONE ring of 7 slots x 16 KiB carries both
// phases' operands (P1 step: 8 A + 8 W1 fragments; P3 step: 16 W2 fragments; two pieces per wave and step), the DMA six steps (96 KiB)
// ahead of the MFMAs across the phase and chunk boundaries (counted vmcnt), fragment reads one k block ahead of the MFMAs across the
// step's barrier.  Every output element sees the same products in the same k order as the two launches: bit-identical
// (tests/test_gpu_kernels.py::test_mlp_fused_equals_the_two_launches).
// What the compiler taught (kept in the code as comments): several __shared__ objects, or any LDS access it can see between an LDS-DMA
// issue and its counted wait, get a `s_waitcnt vmcnt(0)` -- one DMA latency per ring step.
// ------------------------------------------------------------------------------------------------
#define MLP_GLL(g, l) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g), (__attribute__((address_space(3))) void*)(l), 16, 0, 0)
#define MLP_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
template <bool PROBE>
__global__ __launch_bounds__(512) void mlp_fused_h1p_k(MlpArgs a) {
  constexpr int BM = 128, BH = 128, FRAG = 512, SLOT = 16 * FRAG, NSLOT = 7, DEPTH = NSLOT - 1;
  constexpr int KB1 = 512 / 16;            // k blocks of GEMM 1
  constexpr int S1 = 16, S3 = BH / 16, SC = S1 + S3;   // ring steps per chunk: P1 (32 of k each), P3 (16 of k each)
  // ONE LDS object (H | bias | ring): with several, LDS lowering tags them with alias scopes and the waitcnt pass then puts a
  // `s_waitcnt vmcnt(0)` between every LDS-DMA issue and the next fragment read of the ring (measured: the DMA's latency per step)
  constexpr int HELEMS = (BM / 32) * (BH / 16) * FRAG;
  __shared__ __attribute__((aligned(16))) uint16_t lds_all[HELEMS + 2 * 2048 + NSLOT * SLOT];
  uint16_t* const hbuf = lds_all;                                   // 32 KiB: [frame tile][k block of the chunk] fragments (first: its offsets fit ds_read's immediate)
  float* const b1s = reinterpret_cast<float*>(lds_all + HELEMS);    // 8 KiB
  uint16_t* const ring = lds_all + HELEMS + 2 * 2048;               // 7 x 16 KiB
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // wave: an SGPR (fragment offsets, DMA bases)
  const int M = a.M, inter = a.inter, NC = inter / BH, kb2 = inter >> 4;
  const int m0 = blockIdx.x * BM;
  const int wh = wave & 3, wf = wave >> 2;   // P1: hidden tile wh x frame tiles 2 wf, 2 wf + 1
  const int wm = wave & 1, wn = wave >> 1;   // P3: frame tiles 2 wm, 2 wm + 1 x column tiles 4 wn .. 4 wn + 3

  for (int i = tid; i < inter; i += 512) b1s[i] = a.b1[i];

  // DMA pieces of this wave: fragments 2 wave, 2 wave + 1 of the slot.  P1 step (c, j): waves 0-3 bring A (frame tile wave, the two k halves of
  // the 32-wide block), waves 4-7 W1 (hidden tile wave - 4 of the chunk) with the row permutation in the SOURCE lane; P3 step (c, j): W2 column
  // tiles 2 wave, 2 wave + 1.  Wave-uniform bases (SGPRs) + the lane's 16 bytes.
  const int sl = (lane & 32) | (lane & 19) | ((lane & 4) << 1) | ((lane & 8) >> 1);   // bits 2 <-> 3
  const bool w1w = wave >= 4;
  const uint16_t* src1 = w1w ? a.W1p + (size_t)(wave - 4) * KB1 * FRAG : a.Ap + (size_t)((m0 >> 5) + wave) * KB1 * FRAG;
  const size_t cs1 = w1w ? (size_t)(BH / 32) * KB1 * FRAG : 0;
  const int lo1 = (w1w ? sl : lane) * 8;
  const uint16_t* src3 = a.W2p + (size_t)(2 * wave) * kb2 * FRAG;
  const int ln8 = lane * 8;

  // PROBE (CTTS_MLP_PROBE=1 + MlpArgs.dbg, tools/mlp_phase_probe.py): wave 0 accumulates 100 MHz phase times:
  // 0 DMA wait, 1 barrier, 2 DMA issue, 3 P1 reads + MFMAs, 4 GELU + H writes, 5 P3 reads + MFMAs, 6 epilogue, 7 whole kernel
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0, t_begin = 0;
#define MLP_MARK(i) do { if (PROBE) { const long long tn = wall_clock64(); tacc[i] += tn - tprev; tprev = tn; } } while (0)
  int slot_w = 0;   // slot the next issue fills
  auto issue = [&](int c, int s) {   // ring step s of chunk c
    uint16_t* l = ring + slot_w * SLOT + 2 * wave * FRAG;
    if (s < S1) {
      const uint16_t* g = src1 + (size_t)c * cs1 + (size_t)s * 2 * FRAG + lo1;
      MLP_GLL(g, l);
      MLP_GLL(g + FRAG, l + FRAG);
    } else {
      const uint16_t* g = src3 + (size_t)(c * S3 + s - S1) * FRAG + ln8;
      MLP_GLL(g, l);
      MLP_GLL(g + (size_t)kb2 * FRAG, l + FRAG);
    }
    slot_w = slot_w == NSLOT - 1 ? 0 : slot_w + 1;
  };
  // step s of chunk c: wait for this wave's two pieces of it (the pieces of the next DEPTH - 1 steps stay in flight: vmcnt counts in order),
  // barrier (everybody's pieces are in; everybody is done with the slot of the previous step), refill that slot with step + DEPTH
  auto step_sync = [&](int c, int s) {
    const bool last = c == NC - 1;
    const int left = SC - 1 - s;   // steps behind this one in the chunk
    if (!last || left >= DEPTH - 1) MLP_VMCNT(10);
    else if (left == 4) MLP_VMCNT(8);
    else if (left == 3) MLP_VMCNT(6);
    else if (left == 2) MLP_VMCNT(4);
    else if (left == 1) MLP_VMCNT(2);
    else MLP_VMCNT(0);
    MLP_MARK(0);
    __builtin_amdgcn_s_barrier();
    MLP_MARK(1);
  };
  // WHERE a wave issues its two refill pieces inside the ring step (MlpArgs.ipos_mode, CTTS_MLP_IPOS): 0 (default) position 2 (w / 4) + w % 2 of
  // the step's four (between the read / MFMA groups), so that the two waves of a SIMD (w, w + 4) never issue together; 1 position w % 4; 2 all
  // eight waves right behind the barrier.  The idea: the 16 pieces of a step are 420 cycles of the CU's address path, and the phase probe shows
  // 55 of a tile's 240 us in the issue and 61 in the barrier skew behind it.  MEASURED: no difference between the three
  // (profiles/r6Q_mlp_ipos.log) -- the issue cost is per instruction, not queueing behind the other waves.
  const int ipos_mode = a.ipos_mode;
  const int ipos = ipos_mode == 1 ? (wave & 3) : ipos_mode == 2 ? 0 : 2 * (wave >> 2) + (wave & 1);
  auto refill = [&](int c, int s, int pos) {
    if (pos != ipos) return;
    if (s + DEPTH < SC) issue(c, s + DEPTH);
    else if (c != NC - 1) issue(c + 1, s + DEPTH - SC);
    MLP_MARK(2);
  };

  f32x16 acc2[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;

  MLP_VMCNT(0);   // the bias loads: the counted waits below count DMA pieces only
  __syncthreads();
#pragma unroll
  for (int s = 0; s < DEPTH; ++s) issue(0, s);
  if (PROBE) t_begin = tprev = wall_clock64();
  int slot_r = 0;   // slot of the step being multiplied
  // LDS byte offsets for the two places that touch LDS through inline asm (below)
  const uint32_t hb_off = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) void*)hbuf);
  const uint32_t b1_off = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) void*)b1s);
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x4v __attribute__((ext_vector_type(4)));
#define MLP_SB() __builtin_amdgcn_sched_barrier(0)
  for (int c = 0; c < NC; ++c) {
    f32x16 acc1[2];
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[y][r] = 0.f;
    // ---- P1: the fragment reads run one 16-wide k block ahead of the MFMAs (two register sets), across the step's barrier: the
    // second half's MFMAs of step j - 1 are issued behind the barrier of step j and the first reads of its slot, whose latency they cover
    f16x8 fa0[2], fw0, fa1[2], fw1;
    auto rd1 = [&](f16x8* fa, f16x8& fw, const uint16_t* l, int h) {
#pragma unroll
      for (int y = 0; y < 2; ++y) fa[y] = *reinterpret_cast<const f16x8*>(l + ((2 * wf + y) * 2 + h) * FRAG);
      fw = *reinterpret_cast<const f16x8*>(l + (8 + wh * 2 + h) * FRAG);
    };
    auto mm1 = [&](const f16x8* fa, const f16x8& fw) {
#pragma unroll
      for (int y = 0; y < 2; ++y) acc1[y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw, fa[y], acc1[y], 0, 0, 0);
    };
#pragma unroll
    for (int j = 0; j < S1; ++j) {
      step_sync(c, j);
      const uint16_t* l = ring + slot_r * SLOT + lane * 8;
      slot_r = slot_r == NSLOT - 1 ? 0 : slot_r + 1;
      refill(c, j, 0);
      MLP_SB(); rd1(fa0, fw0, l, 0); MLP_SB();
      refill(c, j, 1);
      if (j > 0) mm1(fa1, fw1);
      MLP_SB(); rd1(fa1, fw1, l, 1); MLP_SB();
      refill(c, j, 2);
      mm1(fa0, fw0);
      MLP_SB();
      refill(c, j, 3);
      if (PROBE) { asm volatile("s_nop 15\n\ts_nop 15" :: "v"(acc1[1][0])); MLP_MARK(3); }
    }
    mm1(fa1, fw1);
    // ---- bias + GELU + fp16 -> the chunk's H as A fragments.  The bias reads and the H writes go through inline asm: an LDS access
    // the compiler can see gets a `s_waitcnt vmcnt(0)` in front of it while LDS-DMA is in flight (the waitcnt pass takes every LDS
    // access for a possible reader of the DMA's destination), which would drain the ring once per chunk
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const uint32_t ba = b1_off + (uint32_t)(c * BH + 32 * wh + 16 * s2 + 8 * (lane >> 5)) * 4u;
      f32x4v bA, bB;
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(bA), "=&v"(bB) : "v"(ba) : "memory");
      const float bb[8] = {bA[0], bA[1], bA[2], bA[3], bB[0], bB[1], bB[2], bB[3]};
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_fast(acc1[y][8 * s2 + e] + bb[e]);
        u32x4 pk;
        pk[0] = pack_f16x2(v[0], v[1]); pk[1] = pack_f16x2(v[2], v[3]); pk[2] = pack_f16x2(v[4], v[5]); pk[3] = pack_f16x2(v[6], v[7]);
        const uint32_t ha = hb_off + (uint32_t)(((2 * wf + y) * S3 + 2 * wh + s2) * 64 + lane) * 16u;
        asm volatile("ds_write_b128 %0, %1" :: "v"(ha), "v"(pk) : "memory");
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // H is in LDS before the barrier of P3's first step
    MLP_MARK(4);
    // ---- P3: the same pipeline, one 16-wide k block per step ----
    f16x8 fh0[2], fv0[4], fh1[2], fv1[4];
    auto rd3 = [&](f16x8* fh, f16x8* fv, const uint16_t* l, int j) {
#pragma unroll
      for (int i = 0; i < 2; ++i) fh[i] = *reinterpret_cast<const f16x8*>(hbuf + ((size_t)((2 * wm + i) * S3 + j) * 64 + lane) * 8);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) fv[jj] = *reinterpret_cast<const f16x8*>(l + (4 * wn + jj) * FRAG);
    };
    auto mm3h = [&](const f16x8* fh, const f16x8* fv, int i) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) acc2[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[i], fv[jj], acc2[i][jj], 0, 0, 0);
    };
    auto mm3 = [&](const f16x8* fh, const f16x8* fv) { mm3h(fh, fv, 0); mm3h(fh, fv, 1); };
#pragma unroll
    for (int j = 0; j < S3; j += 2) {
      step_sync(c, S1 + j);
      {
        const uint16_t* l = ring + slot_r * SLOT + lane * 8;
        slot_r = slot_r == NSLOT - 1 ? 0 : slot_r + 1;
        refill(c, S1 + j, 0);
        MLP_SB(); rd3(fh0, fv0, l, j); MLP_SB();
        refill(c, S1 + j, 1);
        if (j > 0) mm3h(fh1, fv1, 0);
        MLP_SB();
        refill(c, S1 + j, 2);
        if (j > 0) mm3h(fh1, fv1, 1);
        MLP_SB();
        refill(c, S1 + j, 3);
        if (PROBE) { asm volatile("s_nop 15\n\ts_nop 15" :: "v"(acc2[1][3][0])); MLP_MARK(5); }
      }
      step_sync(c, S1 + j + 1);
      {
        const uint16_t* l = ring + slot_r * SLOT + lane * 8;
        slot_r = slot_r == NSLOT - 1 ? 0 : slot_r + 1;
        refill(c, S1 + j + 1, 0);
        MLP_SB(); rd3(fh1, fv1, l, j + 1); MLP_SB();
        refill(c, S1 + j + 1, 1);
        mm3h(fh0, fv0, 0);
        MLP_SB();
        refill(c, S1 + j + 1, 2);
        mm3h(fh0, fv0, 1);
        MLP_SB();
        refill(c, S1 + j + 1, 3);
        if (PROBE) { asm volatile("s_nop 15\n\ts_nop 15" :: "v"(acc2[1][3][0])); MLP_MARK(5); }
      }
    }
    mm3(fh1, fv1);
  }
#undef MLP_SB
  X3pArgs e;
  e.bias = a.b2; e.gamma = a.gamma; e.res = a.C; e.ldr = 512; e.C = a.C; e.ldc = 512;
  if (m0 + BM <= M) epi_scale_res<true>(e, acc2, m0 + wm * 64, wn * 128, lane, M);
  else epi_scale_res<false>(e, acc2, m0 + wm * 64, wn * 128, lane, M);
  if (PROBE && a.dbg != nullptr && tid == 0) {
    MLP_MARK(6);
    long long* d = a.dbg + (size_t)blockIdx.x * 8;
    for (int i = 0; i < 7; ++i) d[i] = tacc[i];
    d[7] = wall_clock64() - t_begin;
  }
}
#undef MLP_MARK
#undef MLP_GLL
#undef MLP_VMCNT

bool mlp_fused_pays(int M) {
  // CTTS_MLP_FUSED=1 always, 2 where the launch fills whole rounds of 256 workgroups; DEFAULT 0 = never: measured (tools/mlp_ab.py,
  // profiles/r6M_mlp_ab.log) the one launch is 3-10 % ahead of the two at 65,536 frames and behind everywhere else -- one workgroup of 8
  // waves per CU runs its phases in lock-step (phase probe, profiles/r6L_mlp_phase.log: of a tile's 240 us, 61 in the barrier, 55 issuing the
  // LDS-DMA, 53 in P1's LDS-bound reads + MFMAs, 28 in P3's MFMAs, 26 in the GELU), where the two-launch kernels keep TWO workgroups
  // per CU whose phases overlap.  Kept as an A/B switch and as the bit-identity test's subject.
  const char* e = getenv("CTTS_MLP_FUSED");   // read at every launch, like CTTS_CODEC_TILE (A/B inside one process, the tests)
  const int mode = e ? atoi(e) : 0;
  if (mode == 0) return false;
  if (mode == 1) return true;
  const int tiles = (M + 127) / 128;
  return tiles >= 256 && (tiles % 256 == 0 || tiles % 256 >= 192 || tiles >= 2048);
}

hipError_t launch_mlp_fused_h1p(const MlpArgs& a, hipStream_t st) {
  if (a.M <= 0 || a.inter % 128 != 0 || a.inter < 128 || a.inter > 2048) return hipErrorInvalidValue;
  if ((unsigned long long)a.M * 512ull >= (1ull << 30)) return hipErrorInvalidValue;   // epi_scale_res: 32-bit byte offsets
  MlpArgs b = a;
  { const char* e = getenv("CTTS_MLP_IPOS"); b.ipos_mode = e ? atoi(e) : 0; }
  if (a.dbg != nullptr) CTTS_LAUNCH(mlp_fused_h1p_k<true>, dim3((a.M + 127) / 128), dim3(512), st, b);
  else CTTS_LAUNCH(mlp_fused_h1p_k<false>, dim3((a.M + 127) / 128), dim3(512), st, b);
  return hipGetLastError();
}
