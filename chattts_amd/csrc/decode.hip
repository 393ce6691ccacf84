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

template <int MBT, int NW, bool SCALE, int EPI>
__global__ __launch_bounds__(64 * (NW + (EPI == FEPI_QKV_ROPE ? 1 : 0)))
void gemm_dec_k(DecGemmArgs a) {
  CTTS_PROBE_RETURN();
  gemm_dec_wg<MBT, NW, SCALE, EPI>(a, blockIdx.x, blockIdx.y * MBT, gridDim.y, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
}

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

template <int MBT>
static hipError_t dec_dispatch_k768(const DecGemmArgs& a, hipStream_t st) {
  constexpr int NW = 4;
  const int mt = (a.M + 15) / 16;
  dim3 grid(a.N / 16, (mt + MBT - 1) / MBT);
  const bool scale = a.ssq_in != nullptr;
  if (a.epi == FEPI_QKV_ROPE && scale) CTTS_LAUNCH((gemm_dec_k<MBT, NW, true, FEPI_QKV_ROPE>), grid, dim3(64 * NW + 64), st, a);
  else if (a.epi == FEPI_SILU && scale) CTTS_LAUNCH((gemm_dec_k<MBT, NW, true, FEPI_SILU>), grid, dim3(64 * NW), st, a);
  else if (a.epi == FEPI_RES && !scale) CTTS_LAUNCH((gemm_dec_k<MBT, NW, false, FEPI_RES>), grid, dim3(64 * NW), st, a);
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
