// Internal launcher interface between the kernel translation units and capi.hip.
// Every launcher enqueues on `st` and returns hipGetLastError(); none allocates or synchronises,
// so all of them are legal inside hipStreamBeginCapture/EndCapture.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "common.hpp"

// Per-kernel timing for bench.py's roofline leg: when capi.hip arms these two events, the NEXT launch goes
// through hipExtLaunchKernelGGL, which stamps them with the dispatch's own start/stop timestamps (the same
// clock rocprofv3's kernel trace reads) -- no marker packets, so the figure agrees with the rocprof summary.
extern thread_local hipEvent_t ctts_prof_start, ctts_prof_stop;
#define CTTS_LAUNCH(kern, grid, block, st, ...)                                                                   \
  do {                                                                                                            \
    if (ctts_prof_start) {                                                                                        \
      hipExtLaunchKernelGGL(kern, grid, block, 0, st, ctts_prof_start, ctts_prof_stop, 0, __VA_ARGS__);           \
      ctts_prof_start = nullptr; ctts_prof_stop = nullptr;                                                        \
    } else {                                                                                                      \
      hipLaunchKernelGGL(kern, grid, block, 0, st, __VA_ARGS__);                                                  \
    }                                                                                                             \
  } while (0)

// the same with `smem` bytes of dynamic LDS (used only to bound how many workgroups a CU holds at once)
#define CTTS_LAUNCH_SMEM(kern, grid, block, smem, st, ...)                                                        \
  do {                                                                                                            \
    if (ctts_prof_start) {                                                                                        \
      hipExtLaunchKernelGGL(kern, grid, block, smem, st, ctts_prof_start, ctts_prof_stop, 0, __VA_ARGS__);        \
      ctts_prof_start = nullptr; ctts_prof_stop = nullptr;                                                        \
    } else {                                                                                                      \
      hipLaunchKernelGGL(kern, grid, block, smem, st, __VA_ARGS__);                                               \
    }                                                                                                             \
  } while (0)

// Probe builds only (python -m chattts_amd.build --variant probe -DCTTS_PROBE_EXIT=1, tools/step_floor_probe.py): every kernel of
// the decode step returns at entry WITHOUT touching memory -- what a replay of the captured step then costs is launch + dispatch
// alone, to be set against the "every utterance finished" replay of the normal build (launch + one dependent load of the live count).
#ifdef CTTS_PROBE_EXIT
#define CTTS_PROBE_RETURN() do { return; } while (0)
#else
#define CTTS_PROBE_RETURN() do { } while (0)
#endif

enum { WT_F32 = 0, WT_BF16 = 1 };

// arrival words of the fused QKV + attention launch (decode_dev.hpp, gpt.hip qkv_attention_k): one int per (layer, head), HO_STRIDE ints
// apart (4 KiB + 256 B: neighbouring words on different memory channels -- they are polled by every attention unit of the head)
constexpr int HO_STRIDE = 1088;

// ---- GEMM  C[M,N] = epi( pro(A)[M,K] * W[N,K]^T ) ------------------------------------------
enum GemmEpi {
  EPI_STORE = 0,       // C = acc
  EPI_RES = 1,         // C = res + acc                               (o_proj / down_proj)
  EPI_SILU_MUL = 2,    // C[:, n] = silu(acc_gate[n]) * acc_up[n]     (W = [gate; up], N = rows of gate)
  EPI_BIAS = 3,        // C = acc + bias[n]
  EPI_BIAS_GELU = 4,   // C = gelu(acc + bias[n])
  EPI_BIAS_SCALE_RES = 5,  // C = res + gamma[n] * (acc + bias[n])    (ConvNeXt pwconv2)
  EPI_SCALE = 6,       // C = acc * scale[n]                          (DVAE out_conv * coef)
  EPI_LOG_DIV = 7,     // C = log(max(acc, 1e-5)) / gamma[n]          (log-mel / coef, DVAE encode; f32 tiles only)
};

struct GemmArgs {
  const float* A;      // [rows, lda] f32
  const void* W;       // [N(, 2N for SILU), K] f32 or bf16, row-major
  float* C;            // [M, ldc]
  int M, N, K;
  int lda, ldc;
  int wt;              // WT_F32 | WT_BF16
  int epi;             // GemmEpi
  // prologue: RMSNorm over K of each A row (skinny kernel only): a' = norm_w[k] * (a * rsqrt(mean(a^2)+eps))
  const float* norm_w;
  float eps;
  // epilogue operands
  const float* res; int ldr;
  const float* bias;   // [N]
  const float* gamma;  // [N] (EPI_BIAS_SCALE_RES) or scale (EPI_SCALE)
  const int32_t* n_active;  // skinny kernel, decode: device scalar, rows >= *n_active are not computed; null = M
  // conv-as-GEMM gather (tiled kernel only): logical row m = b*F + f, k = tap*Cin + c reads
  // X[b, f + (tap - pad)*dil, c] (zero outside [0,F)); taps == 1 -> plain GEMM.
  int taps, cin, frames, pad, dil;
  long long* dbg;      // -DCTTS_X3_PROBE builds only: [n_workgroups][8] accumulated phase times of the split-bf16 tiles, or null
};

hipError_t launch_gemm_skinny(const GemmArgs& a, hipStream_t st);  // weight-streaming, M-tiles of <=64 rows
// prompt pass of the "f32x3" parity mode (prefill32x.hip): C = epi(rstd[row] * ((A diag(norm_w)) W^T)) on split-bf16 operands, LDS-tiled;
// A / W float32 row-major, split by the tile loader; epi: EPI_STORE | EPI_RES | EPI_SILU_MUL
hipError_t launch_gemm_pre_x3(const GemmArgs& a, const float* rstd, hipStream_t st);

// ---- bf16 fast path of the GPT projections (perf mode) ------------------------------------------
// Activations travel between kernels as bf16 (the f32 residual stream is kept beside its bf16
// copy), the RMSNorm gain is folded into the weights at load and the per-row 1/rms is applied in
// the epilogue from 48 per-row partial sums of squares that the PRODUCER of the residual wrote,
// so no consumer ever re-reads a full f32 row to normalise it.
enum FastEpi { FEPI_STORE32 = 0, FEPI_RES = 1, FEPI_SILU = 2, FEPI_QKV_ROPE = 3 };
#define SSQ_PARTS 48  // 768 / 16: one partial per 16-column tile of the residual stream
struct FastGemmArgs {
  const uint16_t* A; int lda;   // [M,K] bf16
  const uint16_t* W;            // [N (2N for SILU: gate rows then up rows), K] bf16
  int M, N, K;
  const float* ssq_in;          // [M,48] or null (no row scale)
  float eps;
  int epi;
  float* C32; int ldc;          // STORE32: out; RES: f32 residual, updated in place
  uint16_t* Cb; int ldcb;       // RES: bf16 copy of the new residual; SILU: activation
  float* ssq_out;               // RES: [M,48] partial sums of squares of the new residual
  // QKV_ROPE: RoPE + KV append fused into the epilogue (weights row-permuted by the loader, see engine.py)
  int q_per_b; const int32_t* len; const int32_t* kv_start;   // row -> (b, slot) map, as GptRowMap
  const int32_t* row_map;       // decode: compact row -> batch slot (QKV_ROPE epilogue), or null
  const int32_t* n_active;      // decode: device scalar, rows >= *n_active are not computed; null = M
  const float* cos_t; const float* sin_t;                     // [max_pos, 32]
  uint16_t* kc; uint16_t* vc; int cmax;                       // this layer's bf16 K / V cache [B,12,cmax,64]
  int slot0;                    // prefill in chunks: first prompt slot of this chunk (row m -> slot0 + m % q_per_b)
  int force_mb;                 // tests only: 0 = heuristic
  int w_nt;                     // 1: non-temporal W loads when a single M tile reads W (set by the launcher, env CTTS_W_NT=0 disables)
  long long* dbg;               // probes only: [n_workgroups][8] phase stamps, or null
};
hipError_t launch_gemm_fast(const FastGemmArgs& a, hipStream_t st);
// prefill.hip: 128x128x64 LDS-tiled bf16 GEMM with the RES / SILU / QKV_ROPE epilogues, for M = B*T prompt rows
// (launch_gemm_fast routes M >= 256 here; CTTS_PREFILL_TILED=0 keeps the A-stationary gemm_fast_k variant)
bool gemm_prefill_supported(const FastGemmArgs& a);
hipError_t launch_gemm_prefill(const FastGemmArgs& a, hipStream_t st);
// x32 row -> bf16 copy + partial sums of squares (prefill entry); optional code-embedding gather
hipError_t launch_rows_prep(const float* x32, uint16_t* xb, float* ssq, int M, hipStream_t st);
hipError_t launch_gemm_tiled(const GemmArgs& a, hipStream_t st);   // 64x64 LDS-tiled f32 MFMA, large M
hipError_t launch_gemm_tiled_bf16x3(const GemmArgs& a, hipStream_t st);  // 128x128 split-bf16 (3 MFMA / product), W = [N][Kp/32][2][32] bf16

// ---- decode-step projections on fragment-packed operands (decode.hip) --------------------------
// Compact decode row m -> everything the layer kernels need about it, written ONCE per step by the embedding kernel
// (which walks row_map -> len -> kv_start anyway), so that the QKV epilogue and the attention kernel of each of the 20
// layers start from one 16-byte load instead of a 3-hop dependent chain.
struct alignas(16) RowDesc {
  int32_t b;     // utterance (batch slot), or -1: already finished (sampled EOS) -- nothing downstream is ever read
  int32_t slot;  // KV slot of this step's token = len[b] - 1
  int32_t pos;   // RoPE position = slot - kv_start[b]
  int32_t jlo;   // first visible key = kv_start[b]
};
struct DecGemmArgs {
  const uint16_t* Ap;           // activations, packed [ceil(M/16)][K/32][64][8] bf16
  const uint16_t* Wp;           // weights, packed [N/16 (SILU: gate tiles then up tiles)][K/32][64][8] bf16
  int M, N, K;                  // M = rows the buffers hold (utterances); live rows = *n_active
  const int32_t* n_active;      // device scalar or null (M)
  const float* ssq_in;          // [M,48] or null (no row scale)
  float eps;
  int epi;                      // FastEpi
  float* C32; int ldc;          // QKV_ROPE: f32 qkv buffer; RES: f32 residual, updated in place
  uint16_t* Cp; int kch_out;    // packed bf16 output (RES: new residual, SILU: activation) with kch_out = columns / 32
  float* Cp32;                  // RES, optional: the new residual once more in the packed f32 order of decode32.hip (pk32_off, 768 / 16
                                // chunks): the last layer's down_proj feeds the fused final-norm + heads launch with it
  float* ssq_out;               // RES: [M,48]
  const RowDesc* desc;          // QKV_ROPE
  const float* cos_t; const float* sin_t;
  uint16_t* kc; uint16_t* vc; int cmax;
  int force_mb;                 // tests only
  int w_nt;                     // set by the launcher
  int a_early;                  // set by the launcher (CTTS_DEC_A_EARLY): batches of <= 16 rows request their activation tile at entry
  int force_nt;                 // 0: the launcher's policy (CTTS_W_NT); 1: plain (temporal) weight loads; 2: non-temporal (A/B: CTTS_W_TEMPORAL_LAYERS)
#ifdef CTTS_PF_BUILD
  PfDesc pf[2];                 // weights of later launches of the step this launch's auxiliary wave pulls towards L2 (QKV_ROPE, SILU), or {null}
#endif
  long long* dbg;               // probes only (tools/dec_phase_probe.py): [n_workgroups][8] phase stamps, or null
  const float* rope_cs;         // QKV_ROPE inside the fused QKV + attention launch: [rows][64] cos[32] | sin[32] of each row's position (StepPrep)
  int32_t* ho_flag;             // QKV_ROPE inside the fused QKV + attention launch: this layer's arrival words, head h at [h * HO_STRIDE]
};
hipError_t launch_gemm_dec(const DecGemmArgs& a, hipStream_t st);

// ---- decode-step projections of the f32 parity mode on fragment-packed float32 operands (decode32.hip) -----------------
// Same arithmetic, bit for bit, as gemm_skinny_k<float> (the kernel the parity goldens were established with): 4 waves take
// the 16-wide k chunks round-robin, each chunk is four k-ordered v_mfma_f32_16x16x4_f32 steps, the 4 partial tiles are added
// ((w0 + w1) + w2) + w3.  Only WHERE the operands live changes (common.hpp pk32_off): one contiguous KiB per wave load.
struct Dec32Args {
  const float* Ap;              // activations, packed [ceil(M/16)][K/16][64][4]; with norm_w: the UN-normalised residual stream
  const float* Wp;              // weights, packed [N/16 (SILU_MUL: gate tiles then up tiles)][K/16][64][4]
  int M, N, K;                  // M = rows the buffers hold; live rows = *n_active
  const int32_t* n_active;      // device scalar or null (M)
  const float* X; int ldx;      // RMSNorm prologue: the same rows, row-major (the sum of squares is taken in gemm_skinny_k's order)
  const float* norm_w; float eps;   // [K] or null: A' = norm_w[k] * (A[m,k] * rstd[m]) applied to the fragments
  int epi;                      // EPI_STORE: C row-major | EPI_RES: C = res + acc (row-major) and Cp | EPI_SILU_MUL: Cp only
  float* C; int ldc;
  const float* res; int ldr;
  float* Cp; int kch_out;       // packed f32 copy of the output for the next projection (kch_out = its columns / 16), or null
  int n_cols;                   // EPI_STORE: columns of C that exist (the heads: N is padded to a multiple of 16); 0 = N
  // D32_EPI_QKV_ROPE (N = 2304, K = 768, q/k weight rows in the rope_row_perm order of engine.py): RoPE on q -> C (qkv buffer, natural
  // column order), RoPE on k -> KV cache, v -> KV cache, all float32, at (desc[row].b, desc[row].slot); rope_append_k's arithmetic
  const RowDesc* desc;
  const float* cos_t; const float* sin_t;
  float* kc; float* vc; int cmax;
  int force_mb;                 // tests only
  int w_nt;                     // set by the launcher
  int a_early;                  // set by the launcher (CTTS_D32_A_EARLY): first activation round requested before the RMSNorm prologue
  // FINAL-NORM fusion (decode heads, both modes): with `fnorm` the launch is `final RMSNorm -> hidden capture -> heads` in one kernel:
  // Ap = the UN-normalised residual stream (packed f32), norm_w = the final norm's gain, the row statistics are final_norm_k's
  // (gpt.hip) bit for bit, taken from the fragments; workgroups of weight tile 0 also write the normalised rows -- the step's hidden
  // states (gpt.py:430-436) -- to hid[desc[row].b][gen] with gen = desc[row].slot + 1 - (prompt_len ? prompt_len[b] : T)
  int fnorm;
  float* hid; int hid_cap;      // [slots, hid_cap, 768] or null
  int T; const int32_t* prompt_len;
};
enum { D32_EPI_QKV_ROPE = 100 };
hipError_t launch_gemm_dec32(const Dec32Args& a, hipStream_t st);
// prefill32.hip: the same arithmetic for prompt-sized M, register-blocked (one wave = 32 x 32 outputs x the 4 k-chunk classes), no LDS.
// rstd [M]: 1 / rms per row (launch_rows_rstd32) when a.norm_w is set.  epi: D32_EPI_QKV_ROPE | EPI_RES | EPI_SILU_MUL.
hipError_t launch_rows_rstd32(const float* X, int ldx, int M, float eps, float* rstd, hipStream_t st);
hipError_t launch_gemm_pre32(const Dec32Args& a, const float* rstd, hipStream_t st);
// decode32x.hip: the same four projections on SPLIT-bf16 operands (x = hi + lo bf16 planes, three bf16 MFMAs per product, f32 accumulation;
// RMSNorm gain folded into the weights, 1 / rms applied to the accumulator).  Planes are in decode.hip's fragment order (pk_off).
struct Dec32xArgs {
  const uint16_t* Ap; size_t a_plane;   // activations, hi plane [ceil(M/16)][K/32][64][8] bf16; the lo plane a_plane ELEMENTS behind it
  const uint16_t* Wp; size_t w_plane;   // weights likewise, [N/16 (SILU_MUL: gate tiles then up tiles)][K/32][64][8]
  int M, N, K;                          // M = rows the buffers hold; live rows = *n_active
  const int32_t* n_active;
  int rms; const float* X; int ldx; float eps;   // RMSNorm launches (K = 768): 1 / rms of the rows scales the accumulator -- from ssq_in, else from
                                                 // the rows themselves (row-major f32, gemm_skinny_k's arithmetic)
  const float* ssq_in;                  // RMSNorm launches: [rows][48] partial sums of squares of the rows (one per 16 columns), or null
  float* ssq_out;                       // RES: the same for the new residual rows, or null
  const float* rope_cs;                 // QKV_ROPE: [rows][64] cos[32] | sin[32] of each row's position (StepPrep.rope_cs), or null (cos_t / sin_t via desc)
  int epi;                              // D32_EPI_QKV_ROPE | EPI_RES | EPI_SILU_MUL
  float* C; int ldc;                    // RES: the new residual rows (row-major f32); QKV_ROPE: the qkv buffer
  const float* res; int ldr;
  uint16_t* Cp; size_t c_plane; int kch_out;   // the output as planes for the next projection (kch_out = its columns / 32), or null
  float* Cp32; int kch32_out;           // RES, optional: packed f32 copy (pk32_off, kch32_out = columns / 16): the heads' operand
  const RowDesc* desc; const float* cos_t; const float* sin_t; float* kc; float* vc; int cmax;   // QKV_ROPE, as Dec32Args
  int force_mb;                         // tests only
  int w_nt;                             // set by the launcher
};
hipError_t launch_gemm_dec32x(const Dec32xArgs& a, hipStream_t st);
const char* dec32_last_variant();   // "rms16" | "m16" | "generic": the kernel the calling thread's last launch picked (tests)

// ---- GPT step kernels -------------------------------------------------------------------------
struct GptRowMap {
  // query row m of a launch maps to (b, slot): prefill (q_per_b = T): b = m / T, slot = slot0 + m % T (slot0 > 0: a later
  // chunk of a prompt that is prefilled in pieces);
  // decode (q_per_b = 1): b = row_map ? row_map[m] : m, slot = len[b] - 1, and rows m >= *n_active do not exist.
  // Decode activations are COMPACT: the host packs the still-running utterances to the front (row_map, n_active;
  // refreshed at every finish poll), every per-utterance array (ids_buf, len, KV cache, hiddens, q ...) stays
  // indexed by the batch slot b.  Finished utterances thus cost nothing (the reference keeps stepping them until
  // the last one is done, gpt.py:512-518,592, but truncates their output at end_idx, so that work is unobservable).
  int q_per_b;
  const int32_t* len;       // [B] tokens present in ids_buf (prompt + generated)   (decode)
  const int32_t* kv_start;  // [B] left-pad slots (attention_mask == 0 there)
  const int32_t* row_map;   // [B] compact row -> batch slot, or null (identity)
  const int32_t* n_active;  // device scalar, or null (all rows)
  const uint8_t* finish;    // [B] or null (decode): rows that already sampled EOS.  The reference keeps stepping them
                            // until every row is done (gpt.py:512-518,592) but truncates their output at end_idx, so
                            // their per-step work is unobservable: the attention kernel skips their KV read.
  const RowDesc* desc;      // decode, optional: this step's row descriptors (replaces the row_map/len/kv_start/finish chain)
  // decode, perf mode, optional: remainder splitting of the (utterance, head) units over workgroups (see attention_k)
  float* sp_part;           // [sp_cus][ATT_SPLIT_MAX][66] partial (o[64], m, l) of the split units' pieces
  int32_t* sp_cnt;          // [sp_cus] arrival counters, zero between launches (the last arriver resets its counter)
  int sp_cus;               // compute units of the device (0: no splitting)
  int slot0;                // prefill: first prompt slot of the chunk this launch covers
  int desc_covers_all;      // decode: desc[] is valid for EVERY row of the grid (absent rows carry b = -1), so the attention
                            // kernel need not read *n_active first (one dependent load less in front of the KV stream)
#ifdef CTTS_PF_BUILD
  PfDesc pf;                // decode, perf mode: the gate/up weights of this layer, pulled towards L2 by a fifth wave per workgroup
#endif
  long long* dbg;           // probes only (tools/attn_phase_probe.py, env CTTS_ATT_DBG_PTR): [workgroups][8] phase stamps (100 MHz), or null
  // decode, perf mode, optional (launch_attention_oproj): o_proj + residual folded into the attention launch
  const uint16_t* wo_h;     // this layer's o_proj weight per head: [12][8 (k / 8)][768 columns][8] bf16 (engine.py pack_wo_heads)
  float* op_part;           // [rows][12][768] f32 partials of the (utterance, head) units
  int op_part_bytes;
  int32_t* op_cnt;          // [rows] arrival counters, zero between launches (the last arriver resets its counter)
  float* x32;               // [rows][768] f32 residual stream, updated in place by the row's last arriver
  uint16_t* xp;             // ... its bf16 copy in the fragment-packed order of decode.hip
  float* ssq;               // ... and its [rows][48] partial sums of squares
  // decode, perf mode, inside the fused QKV + attention launch (launch_qkv_attention): this layer's arrival words (head h at
  // [h * HO_STRIDE], decode_dev.hpp) that the QKV tiles bump, and the rows of the qkv buffer
  const int32_t* qf_flag;
  int qf_rows;
  int qf_kv_bytes;          // bytes of this layer's K (= V) cache: the units address it through a buffer descriptor with 32-bit offsets
  size_t x3_plane;          // decode, split-bf16 parity mode (out mode 4): elements between the hi and the lo plane of the packed output
};
#define ATT_SPLIT_MAX 8

// what the first kernel of a decode step additionally produces: the row descriptors, and (xb_packed) the bf16 copy of
// the embedding in the fragment-packed order of decode.hip instead of row-major
struct StepPrep {
  RowDesc* desc;            // [B] out, or null
  const int32_t* kv_start;  // [slots]
  const uint8_t* finish;    // [slots] or null
  int xb_packed;
  float* xp32;              // f32 parity mode: fragment-packed f32 copy of the new residual rows (pk32_off), or null
  // device-side compaction (optional): compact row m becomes the m-th utterance (ascending slot) whose finish flag is 0;
  // the kernel writes row_map_out[m] for the rows that exist and the live count to *n_active_out, which every later
  // kernel of the step reads -- finished utterances leave the step at once, without the host
  int32_t* row_map_out;     // [B] or null
  int32_t* n_active_out;    // device scalar or null
  const int32_t* order;     // [B] or null: visiting order of the compaction (permutation of the slots; host: descending context)
  float* rope_cs; const float* cos_t; const float* sin_t;   // [rows][64] out (cos[32] | sin[32] of the row's position) from the [max_pos][32] tables, or null
  int32_t* zero_p; int zero_n; int zero_stride;   // zero_n words, zero_stride ints apart, that the step's first kernel zeroes (arrival words
                                                  // of the fused QKV + attention launches), or null
  size_t xb_lo_plane;       // split-bf16 parity mode: with a packed `xb`, ALSO write the lo plane (x - bf16(x)) this many elements behind it; 0 = no
};
hipError_t launch_embed_codes(const float* emb_code /*[4,626,768]*/, const int64_t* ids_buf, int ids_row_stride /*Tcap*/,
                              const int32_t* len, float* x, uint16_t* xb /*null ok*/, float* ssq /*null ok*/, int B,
                              const int32_t* row_map, const int32_t* n_active, hipStream_t st, const StepPrep* prep = nullptr);
// parity mode prefill on packed operands: x [M,768] row-major -> the packed f32 order (pk32_off), and desc[m] of every prompt row
// (utterance = row_map ? row_map[m / q_per_b] : m / q_per_b, KV slot = slot0 + m % q_per_b, RoPE position, first visible key)
hipError_t launch_prefill_prep32(const float* x, float* xp32, RowDesc* desc, int q_per_b, int slot0, const int32_t* kv_start,
                                 const int32_t* row_map, int M, hipStream_t st);
hipError_t launch_rope_append(float* qkv /*[M,2304]*/, void* kcache, void* vcache, int kv_wt, int cmax,
                              const float* cos_tab, const float* sin_tab /*[max_pos,32]*/, GptRowMap rm, int M, hipStream_t st);
void attention_persist_override(int persist, int g, int d);   // tests / probes: < 0 (<= 0) leaves a field as it is
void attention_hpw_override(int hpw);   // heads of one utterance per decode-attention workgroup: 1 (default) | 2 | 3 | 4
hipError_t launch_attention(const float* qkv, const void* kcache, const void* vcache, int kv_wt, int cmax,
                            void* out /*[M,768] f32, or bf16 when out_bf16 (2: bf16 in the packed order of decode.hip)*/, int out_bf16,
                            GptRowMap rm, int M, hipStream_t st);
// decode, perf mode: attention + o_proj + residual in one launch (rm.wo_h .. rm.ssq set); replaces launch_attention(out_bf16 = 2) and
// the FEPI_RES launch of o_proj behind it
hipError_t launch_attention_oproj(const float* qkv, const void* kcache, const void* vcache, int cmax, GptRowMap rm, int M, hipStream_t st);
// decode, perf mode, M <= 64: QKV (RMSNorm scale + q/k/v_proj + RoPE + KV append) and attention as ONE launch -- the attention units
// request their old keys while the QKV tiles run and pick q / the newest key up through per-head arrival words (d.ho_flag == rm.qf_flag)
hipError_t launch_qkv_attention(const DecGemmArgs& d, const void* kcache, const void* vcache, int cmax, void* out_packed, GptRowMap rm, int M,
                                hipStream_t st);
hipError_t launch_final_norm(const float* x, int q_per_b, const float* w, float eps, float* hfin /*[B,768]*/,
                             float* hiddens /*[slots,max_new,768]*/, int max_new, const int32_t* len, int T, int B,
                             const int32_t* row_map, const int32_t* n_active, const int32_t* prompt_len, hipStream_t st,
                             float* hfin_packed = nullptr /* the same rows in the packed f32 order of decode32.hip, or null */,
                             const int32_t* last_row = nullptr /* [B]: row of utterance m's last prompt token (compact prompt pass), or null */);
// prompt pass over the valid prompt tokens only: gathers emb[b, kv_start[b] .. T) into consecutive rows of x, writes their descriptors
// and last_row[b]; the host knows the row count (sum of the attention mask)
hipError_t launch_prefill_compact(const float* emb, float* x, RowDesc* desc, int32_t* last_row, int B, int T, const int32_t* kv_start,
                                  hipStream_t st);

// The NEXT decode step's first kernel folded into sample_k's tail (multi-step graphs, device-side compaction): the workgroup that drew
// utterance b's tokens also writes that utterance's next input row -- embedding sum, its bf16 / split / packed copies, partial sums of
// squares, row descriptor, RoPE factors: everything embed_codes_k would -- at the SAME compact row m.  Compaction then lags: a row that
// finished keeps its (dead) compact row until the next embed_codes_k launch re-ranks the finish flags.  x == null: off.
struct EmbedNext {
  const float* emb_code; float* x; uint16_t* xb; float* ssq;
  StepPrep sp;
};

struct SampleArgs {
  const float* logits;      // [B, 4*626]
  int64_t* ids_buf;         // [B, Tcap, 4]
  int tcap, T;              // T = prompt length (history starts at slot T)
  int32_t* len;             // [B]  in/out (++)
  uint8_t* finish;          // [B]  in/out
  int32_t* end_idx;         // [B]  in/out
  const float* q;           // [nq, B*4, 626] Exp(1) draws; step uses slab (gen % nq)
  int nq;
  const float* temperature; // [4]
  const float* pow_table;   // [17] or null (no repetition penalty)
  float top_p_thr; int use_top_p;  // thr = float32(1 - top_p), rounded on the host
  int top_k;   int use_top_k;
  int min_new;
  int eos;
  int row_offset;           // global index of row 0 (multi-GPU shards keep the rows>=625 quirk global)
  int max_input_ids;        // 625
  const int32_t* stop_at;   // [B] or null: bench harness length forcing
  int B;
  const int32_t* row_map;   // decode: workgroup m samples utterance row_map[m] from logits row m; null = identity
  const int32_t* n_active;  // decode: device scalar; null = B
  const int32_t* prompt_len;  // [slots] per-utterance prompt length, or null (T for all)
  int q_rows;               // utterance slots in q (>= B)
  const int64_t* teacher;   // [slots, teacher_stride, 4] or null: teacher forcing (evaluation hook)
  int teacher_stride;
  int64_t* sampled;         // [slots, teacher_stride, 4] or null: the sampler's own draw of every step (before teacher forcing)
  const RowDesc* desc;      // decode with device-side compaction: this step's row descriptors (utterance + length in ONE load), or null
  int rng_device;           // 1: Exp(1) draws from the device generator (Philox4x32-10 keyed on rng_seed) instead of `q`
  int rng_per_step;         // device generator: 1 = a fresh draw every step (the reference's manual_seed=None), 0 = the same draw
                            // every step (manual_seed set: the reference re-seeds its generator at every step, gpt.py:504-507)
  const uint32_t* rng_nonce;            // [slots] or null: per-slot fourth counter word of the device generator (slot pools: bumped per admission)
  const unsigned long long* rng_seed;   // device scalar
  EmbedNext next;           // fold of the next step's embedding kernel (next.x == null: off)
  float* margin;            // [slots] or null: parity certificate, lowered to the step's smallest decision margin (include/chattts_amd.h)
  const int32_t* row_base;  // [B] or null: global index of sampling row 0 of utterance b (replaces row_offset + 4 b)
  long long* dbg;           // probes only (tools/sample_phase_probe.py, env CTTS_SAMPLE_DBG_PTR): [rows][8] phase stamps (100 MHz), or null
};
hipError_t launch_exp_draws(unsigned long long seed, int step, int row0, int rows, int V, float* out, hipStream_t st);
hipError_t launch_sample(const SampleArgs& a, hipStream_t st);
// refine-text mode: logits [B, V], q [nq, B, V], temperature[0]; no repetition penalty (see gpt.hip)
hipError_t launch_sample_text(const SampleArgs& a, int V, hipStream_t st);
hipError_t launch_embed_text(const float* emb_text, int n_text, const int64_t* ids_buf, int tcap, const int32_t* len, float* x,
                             uint16_t* xb, float* ssq, int B, const int32_t* row_map, const int32_t* n_active, hipStream_t st,
                             const StepPrep* prep = nullptr);

// ---- split-bf16 GEMM on pre-split, fragment-packed planes, LDS-DMA staged (codec_gemm.hip) -------
// plane layout of a [rows][K] matrix: [rows/32][K/16][hi|lo][lane = (k%16)/8*32 + row%32][k%8] bf16 (rows padded to 256)
enum { X3P_GELU_PACKED = 0, X3P_SCALE_RES = 1 };
struct X3pArgs {
  const uint16_t* Ap;    // activations, planes, rows padded to a multiple of 256
  const uint16_t* Wp;    // weights [N][K], planes
  int M, N, K;           // M valid rows; N % 256 == 0; K % 32 == 0
  int epi;
  const float* bias;     // [N]
  const float* gamma;    // [N]   (X3P_SCALE_RES)
  const float* res; int ldr;   // residual, f32 row-major (X3P_SCALE_RES)
  float* C; int ldc;     // X3P_SCALE_RES: C = res + gamma * (acc + bias), f32 row-major
  uint16_t* Cp;          // X3P_GELU_PACKED: gelu(acc + bias) as planes of a [rows][N] matrix (the next layer's A operand)
  long long* dbg;        // probe variant only (CTTS_X3P_VAR=3, tools/x3p_phase_probe.py): [n_workgroups][8] accumulated phase times
};
hipError_t launch_gemm_x3p(const X3pArgs& a, hipStream_t st);
// The same GEMM on ONE fp16 plane per operand ("h1p", gemm_mode 2): [rows/32][K/16][lane = (k%16)/8*32 + row%32][k%8] fp16, one
// v_mfma_f32_32x32x16_f16 per product instead of three bf16 ones, f32 accumulation.  K % 64 == 0.  Ap / Wp / Cp are that plane.
hipError_t launch_gemm_h1p(const X3pArgs& a, hipStream_t st);
// Round 6: one ConvNeXt MLP (pwconv1 -> GELU -> pwconv2 -> * gamma -> + residual, dvae.py:46-66) in ONE launch on the fp16 plane
// (mlp_fused_h1p_k, codec_gemm.hip): the `inter`-wide activation never leaves the CU.  Same MFMAs in the same k order per output
// element and the same epilogue arithmetic as launch_gemm_h1p(GELU_PACKED) followed by launch_gemm_h1p(SCALE_RES): bit-identical.
struct MlpArgs {
  const uint16_t* Ap;    // LayerNorm output, fp16 plane [rows/32][512/16][64][8], rows padded to a multiple of 256
  const uint16_t* W1p;   // pwconv1 [inter][512] as a plane
  const uint16_t* W2p;   // pwconv2 [512][inter] as a plane
  int M, inter;          // inter % 128 == 0, inter <= 2048
  const float* b1;       // [inter]
  const float* b2;       // [512]
  const float* gamma;    // [512]
  float* C;              // [M][512] f32 row-major residual stream, updated in place
  long long* dbg;        // probe only (CTTS_X3_DBG_PTR, tools/mlp_phase_probe.py): [n_workgroups][8] accumulated phase times
  int ipos_mode;         // A/B (CTTS_MLP_IPOS): where in a ring step a wave issues its refill: 0 by SIMD pair (default), 1 wave % 4, 2 all behind the barrier
};
hipError_t launch_mlp_fused_h1p(const MlpArgs& a, hipStream_t st);
// whether convnext_stack takes the one-launch kernel: CTTS_MLP_FUSED=1|2, default off (measured: profiles/r6M_mlp_ab.log)
bool mlp_fused_pays(int M);

// ---- codec kernels (channels-last [B, F, C]) ---------------------------------------------------
// yp != null (C = 512 only): the output goes out as the two bf16 planes gemm_x3p_k reads (plane_f16 = 0) or as the one fp16 plane
// gemm_h1p_k reads (plane_f16 = 1) instead of f32 rows
hipError_t launch_dwconv_ln(const float* x, const float* w /*[C,7]*/, const float* b, const float* ln_w, const float* ln_b,
                            float eps, int dil, float* y, int B, int F, int C, hipStream_t st, uint16_t* yp = nullptr, int plane_f16 = 0);
hipError_t launch_layernorm(const float* x, const float* w, const float* b, float eps, float* y, int rows, int C, hipStream_t st);
hipError_t launch_istft(const float* head /*[B,F,1026]*/, const float* window /*[1024]*/, const float* twiddle /*[512,2]*/,
                        float* frames /*[B,F,1024] scratch*/, float* wav /*[B,256(F-1)]*/, int B, int F, hipStream_t st);

// float32 waveform [rows][ld] (n samples per row) -> int16 PCM [rows][n] (+ optional keep mask [rows][ceil(n/8)]): tools/audio/np.py:7-11
hipError_t launch_float_to_int16(const float* wav, long long n, long long ld, int rows, int per_row, int product, float keep_thr,
                                 unsigned* peak, int16_t* pcm, uint8_t* keep, hipStream_t st);
hipError_t launch_copy16(const void* src, void* dst, size_t bytes, hipStream_t st);   // shader copy (dst may be pinned host memory)

// ---- full DVAE: mel front end + GFSQ (dvae.hip) ------------------------------------------------
// |STFT| of one waveform: center=True reflect padding, frame f = padded[256 f, 256 f + 1024) * window, 1024-point FFT,
// mag [F][516] (bins 0..512, then 3 zeros so that the mel projection's K is a multiple of 4)
hipError_t launch_stft_mag(const float* wav, int n, const float* window, const float* twiddle, float* mag, int F, hipStream_t st);
struct GfsqArgs {
  const float* in_w;   // [G][4][D]   project_in
  const float* in_b;   // [G][4]
  const float* out_w;  // [G][D][4]   project_out
  const float* out_b;  // [G][D]
  int levels[4];
  int G, R, D;         // D = 512 channels per group
  int bound_first;
};
hipError_t launch_gfsq_encode(const GfsqArgs& q, const float* feat /*[rows][G*D]*/, int32_t* codes /*[rows][G*R]*/, int rows, hipStream_t st);
hipError_t launch_gfsq_embed(const GfsqArgs& q, const int64_t* codes /*[rows][G*R]*/, float* feat /*[rows][G*D]*/, int rows, hipStream_t st);

// Makes the device that owns `stream` current for the calling thread while an entry point enqueues on it (and restores the
// previous one): kernels / graph capture / hipExtLaunch are issued against the CURRENT device, which need not be the engine's
// (Chat.load(device=cuda:1) without a torch.cuda.set_device).  The null stream carries no device: nothing is changed then.
struct CttsDeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit CttsDeviceGuard(void* stream) {
    hipDevice_t d;
    if (stream != nullptr && hipStreamGetDevice((hipStream_t)stream, &d) == hipSuccess && hipGetDevice(&prev) == hipSuccess && prev != (int)d)
      switched = hipSetDevice((int)d) == hipSuccess;
  }
  ~CttsDeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

int ctts_fail(const char* fmt, ...);   // sets the thread's last-error string, returns -1 (capi.hip)
