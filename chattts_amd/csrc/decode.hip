// Decode-step projections of the GPT path on FRAGMENT-PACKED operands (perf mode, M <= 64 live rows per tile).
//
// Reference op: the four nn.Linear calls of a HF Llama decoder layer reached from
// /root/reference/ChatTTS/model/gpt.py:419-427 (in-tree twin /root/reference/examples/onnx/modeling_llama.py:415-417
// q/k/v_proj, :500 o_proj, :293 gate/up/down), with RMSNorm / RoPE / KV append / residual / SiLU fused as in gemm_fast_k.
//
// Why a second kernel: gemm_fast_k (gemm.hip) loads its MFMA fragments straight from row-major [rows][K] operands, i.e.
// every 16-byte lane load of a wave instruction touches a different 128-byte line (16 rows x 4 k-groups = 64 line visits
// for 1 KiB).  Round-1 phase stamps put ~58 cycles on every such instruction -- the CU's address/tag path handles about
// one line visit per cycle -- so a 64-row tile spent 2.9-3.6 us just ISSUING its 120 loads (profiles/r1_gemm_phase_probe*).
// Here both operands live in HBM in the exact order the MFMA wants them:
//
//     packed[tile of 16 rows][k chunk of 32][lane = (k%32)/8 * 16 + row%16][k%8]        (bf16, 16 bytes per lane)
//
// so one wave instruction reads ONE contiguous KiB (8 line visits), the weight stream of a workgroup is a single
// sequential run of K/32 KiB, and the activation tile is re-read from L2 in whole lines.  Producers write activations in
// this order (embedding gather, attention output, the RES / SILU epilogues below): element (m, c) of a [M][C] activation
// lives at pk_off(m, c, C/32).  Weights are packed once at load (engine.py pack_frag).
//
// A workgroup owns 16 output columns (one weight tile) and MBT row tiles; its NW waves split K; the partial tiles are
// reduced through LDS in a fixed order (deterministic).  Row tiles beyond the live rows (*n_active) are neither loaded
// nor multiplied: the body is instantiated per live-tile count.
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"
#include "decode_dev.hpp"

template <int MBT, int NW, bool SCALE, int EPI, int U = DEC_U>
__global__ __launch_bounds__(64 * (NW + (EPI == FEPI_QKV_ROPE ? 1 : 0)))
void gemm_dec_k(DecGemmArgs a) {
  CTTS_PROBE_RETURN();
  gemm_dec_wg<MBT, NW, SCALE, EPI, false, U>(a, blockIdx.x, blockIdx.y * MBT, gridDim.y, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
}

// Waves per workgroup of the K = 768 launches.  Rounds 2-4 ran all of them on 4 waves (6 chunks of 32 each, one round).  Round 5 (the
// probe VERDICT r4 item 8 asked for at batch 1 -- "more requesters per launch" -- which moved batch 1 by 1 % and batch 64 by more): with
// 8 waves a wave requests 3 chunks instead of 6, twice as many waves have requests in flight and the K partials meet through 8 instead of 4
// LDS tiles.  Measured per launch on the C3 bench (profiles/r5r_ab_nw768.log): QKV 5.25 -> 5.14 us, gate/up 5.70 -> 5.37, o_proj 4.76 ->
// 4.85 (its 16-row workgroups already are 192 x 4 waves: stays at 4); 6 and 12 waves measure like 8 / slightly worse; 1413 -> 1436 audio-s/s.
// (A different split of K changes the order of the partial sums: the perf mode's bits move, its bounds -- DESIGN 2 -- are re-measured.)
#ifndef CTTS_DEC_NW_QKV
#define CTTS_DEC_NW_QKV 8
#endif
#ifndef CTTS_DEC_NW_SILU
#define CTTS_DEC_NW_SILU 8
#endif
#ifndef CTTS_DEC_NW_O
#define CTTS_DEC_NW_O 4
#endif

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

template <int MBT>
static hipError_t dec_dispatch_k768(const DecGemmArgs& a, hipStream_t st) {
  constexpr int NQ = CTTS_DEC_NW_QKV, NS = CTTS_DEC_NW_SILU, NO = CTTS_DEC_NW_O;   // K = 768: 24 chunks of 32, ONE round: U = 24 / waves
  const int mt = (a.M + 15) / 16;
  dim3 grid(a.N / 16, (mt + MBT - 1) / MBT);
  const bool scale = a.ssq_in != nullptr;
  if (a.K != 768) {   // K = 1536 (probes): the 4-wave shape
    if (a.epi == FEPI_SILU && scale) CTTS_LAUNCH((gemm_dec_k<MBT, 4, true, FEPI_SILU>), grid, dim3(256), st, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
  }
  if (a.epi == FEPI_QKV_ROPE && scale) CTTS_LAUNCH((gemm_dec_k<MBT, NQ, true, FEPI_QKV_ROPE, 24 / NQ>), grid, dim3(64 * NQ + 64), st, a);
  else if (a.epi == FEPI_SILU && scale) CTTS_LAUNCH((gemm_dec_k<MBT, NS, true, FEPI_SILU, 24 / NS>), grid, dim3(64 * NS), st, a);
  else if (a.epi == FEPI_RES && !scale) CTTS_LAUNCH((gemm_dec_k<MBT, NO, false, FEPI_RES, 24 / NO>), grid, dim3(64 * NO), st, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
template <int MBT, int NW>
static hipError_t dec_dispatch_k3072(const DecGemmArgs& a, hipStream_t st) {
  const int mt = (a.M + 15) / 16;
  dim3 grid(a.N / 16, (mt + MBT - 1) / MBT);
  if (a.epi == FEPI_RES && a.ssq_in == nullptr) CTTS_LAUNCH((gemm_dec_k<MBT, NW, false, FEPI_RES>), grid, dim3(64 * NW), st, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_gemm_dec(const DecGemmArgs& a_in, hipStream_t st) {
  DecGemmArgs a = a_in;
  static int nt = -1, mb_qkv = -1, mb_silu = -1, mb_o = -1, mb_down = -1, a_early = 1;
  if (nt < 0) {
    nt = env_int("CTTS_W_NT", 1);
    a_early = env_int("CTTS_DEC_A_EARLY", 1);
    mb_qkv = env_int("CTTS_DEC_MB_QKV", 4); mb_silu = env_int("CTTS_DEC_MB_SILU", 4);
    mb_o = env_int("CTTS_DEC_MB_O", 1); mb_down = env_int("CTTS_DEC_MB_DOWN", 1);
  }
  a.w_nt = a.force_nt ? (a.force_nt == 2) : nt;
  a.a_early = a_early;
  if (a.M <= 0 || a.N <= 0 || (a.N & 15) || !(a.K == 768 || a.K == 1536 || a.K == 3072)) return hipErrorInvalidValue;   // 1536: probes (tools/ogu_probe.py)
  if (a.epi == FEPI_RES && a.N != 16 * SSQ_PARTS) return hipErrorInvalidValue;
  if (a.epi == FEPI_QKV_ROPE && (a.N != 2304 || a.K != 768 || !a.desc)) return hipErrorInvalidValue;
  // Rows per workgroup.  Every workgroup pulls its whole activation tile besides its weight tile, and a CU's load path
  // moves at most 64 B per clock: QKV / gate-up (N >= 2304: 144 / 192 weight tiles) take all <= 64 rows per workgroup;
  // o_proj / down_proj have only 48 weight tiles, so 16-row workgroups (192 at 64 rows) keep the CUs busy instead.
  int mb = a.epi == FEPI_QKV_ROPE ? mb_qkv : a.epi == FEPI_SILU ? mb_silu : a.K == 3072 ? mb_down : mb_o;
  if (a.force_mb) mb = a.force_mb;
  if (a.K == 768 || a.K == 1536) {
    if (mb == 4) return dec_dispatch_k768<4>(a, st);
    if (mb == 2) return dec_dispatch_k768<2>(a, st);
    return dec_dispatch_k768<1>(a, st);
  }
  if (mb == 4) return dec_dispatch_k3072<4, 8>(a, st);
  if (mb == 2) return dec_dispatch_k3072<2, 8>(a, st);
  return dec_dispatch_k3072<1, 16>(a, st);
}
