/* chattts_amd -- C ABI of the MI355X-native ChatTTS hot path (libchattts_amd.so).
 *
 * The reference (2noise/ChatTTS) is pure Python and has no FFI for this path; the boundary it
 * offers is the Python call seam that `use_vllm` / `experimental` already use to swap engines
 * (SURVEY.md 8b).  Each entry point below names the reference call it replaces.
 *
 * Conventions
 *   - every pointer named *_dev / documented "device" is a DEVICE pointer owned by the caller (torch
 *     tensors on the host side); the library never allocates or frees device memory, never
 *     synchronises the device, and enqueues everything on the hipStream_t it is given (passed as
 *     void*; NULL = the default stream), so calls are legal inside stream capture;
 *   - return value 0 = ok, negative = error; ctts_last_error() returns a thread-local message;
 *   - a handle is not re-entrant; use one handle per device / per concurrent generate() call.
 */
#ifndef CHATTTS_AMD_H
#define CHATTTS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTTS_F32 0
#define CTTS_BF16 1

typedef struct ctts_gpt ctts_gpt;     /* opaque: GPT engine (weights view + captured decode graph) */
typedef struct ctts_codec ctts_codec; /* opaque: DVAE decoder + Vocos */

const char* ctts_last_error(void);
int ctts_version(void);

/* ------------------------------------------------------------------------------------------------
 * GPT speech-token generator.
 * Replaces `GPT.generate` (ChatTTS/model/gpt.py:316-618): one prefill + N decode steps of
 *   LlamaModel.forward (gpt.py:419-427) -> final-norm hidden capture (:430-436) -> 4 weight-normed heads
 *   (:438-454, embed.py:27-35) -> temperature / repetition penalty / top-p / top-k / EOS mask /
 *   softmax / multinomial (:487-508, processors.py:18-58) -> finish / write-back (:512-577).
 * Weights arrive repacked by the host loader from the reference's safetensors layout:
 *   wqkv[l] = [q_proj; k_proj; v_proj] rows (2304 x 768), wgu[l] = [gate_proj; up_proj] (6144 x 768),
 *   heads = 4 folded weight-norm matrices stacked (2504 x 768, always f32), emb_code = 4 x 626 x 768 f32.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t n_layers;
  int32_t weight_dtype;            /* CTTS_F32 (parity mode) | CTTS_BF16 (perf mode) for the Llama linears */
  int32_t kv_dtype;                /* dtype of the KV cache */
  int32_t max_pos;                 /* rows of the RoPE tables */
  const void* const* wqkv;         /* host array [n_layers] of device pointers */
  const void* const* wo;           /* [768 x 768] */
  const void* const* wgu;          /* [6144 x 768] */
  const void* const* wd;           /* [768 x 3072] */
  const float* const* ln1;         /* input_layernorm.weight [768] */
  const float* const* ln2;         /* post_attention_layernorm.weight [768] */
  const float* norm;               /* final RMSNorm weight [768] */
  const float* emb_code;           /* [4,626,768] */
  const float* heads;              /* [2504,768] */
  const float* rope_cos;           /* [max_pos,32] float32, built by the host with the reference's own ops */
  const float* rope_sin;
  float rms_eps;
  /* refine-text mode (infer_text=True, gpt.py:406-407,439-440): optional, may be NULL / 0 */
  const float* emb_text;           /* [n_text,768] */
  const float* head_text;          /* [n_text,768] folded weight-norm text head */
  int32_t n_text;                  /* 21178 */
  /* optional (NULL: the decode step uses the row-major kernels): the same four matrices per layer in the fragment-packed order
   * of the weight dtype's decode kernel.  bf16 (csrc/decode.hip): [rows/16][K/32][lane = (k%32)/8*16 + row%16][k%8], one contiguous
   * KiB per (16-row tile, 32-wide k chunk); wqkv with the RoPE row permutation and the folded RMSNorm gain like `wqkv`.
   * f32 (csrc/decode32.hip, needs an f32 KV cache): [rows/16][K/16][lane = (k%16)/4*16 + row%16][k%4] of the plain matrices, wqkv
   * with the RoPE row permutation (no gain folding); same arithmetic, bit for bit, as the row-major f32 kernels. */
  const void* const* wqkv_pk;
  const void* const* wo_pk;
  const void* const* wgu_pk;
  const void* const* wd_pk;
  /* optional (NULL: row-major heads GEMM): `heads` / `head_text` zero-padded to a multiple of 16 rows, in the packed f32 order above
   * ([rows/16][768/16][64][4]); both modes (the heads are always f32); bit-identical logits */
  const float* heads_pk;
  const float* head_text_pk;
  /* optional (NULL: o_proj stays its own launch), perf mode only: o_proj.weight once more, sliced per attention head,
   * [12 heads][8 (k / 8)][768 output columns][8] bf16 = Wo[column][64 head + 8 (k / 8) + (k % 8)].  With it AND the environment
   * variable CTTS_ATT_OPROJ=1 the decode step folds o_proj + residual into the attention launch (csrc/gpt.hip attention_k<OPJ>; HF Llama
   * self_attn.o_proj, examples/onnx/modeling_llama.py:500,557): 83 launches per step instead of 103.  OPT-IN: measured slower than the
   * two launches (profiles/r4b_ab_oproj.log), so by default the pointer is ignored and o_proj stays its own launch. */
  const void* const* wo_hd;
  /* optional (NULL: the f32 MFMA kernels of csrc/decode32.hip), parity mode only, together with the *_pk copies above: the same four
   * matrices as SPLIT-fp16 planes for the decode step (csrc/decode32x.hip; round 5: bf16 planes) -- [2 planes: hi = fp16(w), lo' = fp16((w - hi) * 2^11)]
   * [N/16][K/32][64][8], 22 significant bits, the bf16 fragment order above per plane; wqkv_x3 / wgu_x3 carry the RMSNorm gain of ln1 / ln2 (w' = w * gain[k], folded BEFORE
   * the split) and the q / k row permutation of wqkv_pk.  Three fp16 MFMAs per product (hi*hi, and lo'*hi + hi*lo' on a second accumulator that enters with 2^-11) instead of f32 MFMA at a sixteenth of the rate; the
   * reference's token ids hold on every golden (tests/test_gpu_e2e.py).  The HOST chooses the arithmetic: no planes = exact f32 MFMA; planes =
   * split-fp16 decode steps, certified per call by ctts_gen_state.margin, with ctts_gen_state.proj_exact as the per-call exact fallback. */
  const void* const* wqkv_x3;
  const void* const* wo_x3;
  const void* const* wgu_x3;
  const void* const* wd_x3;
} ctts_gpt_weights;

/* One generate() call's device state (every array is caller-allocated, device memory). */
typedef struct {
  int32_t B, T, max_new;           /* rows, padded prompt length, max_new_token (gpt.py:323) */
  int64_t* ids_buf;                /* [B, T+max_new, 4]  inputs_ids_buf (gpt.py:372-379), prompt pre-filled */
  int32_t* len;                    /* [B] tokens present per row; host initialises to T */
  const int32_t* kv_start;         /* [B] number of left-pad slots (attention_mask == 0) */
  uint8_t* finish;                 /* [B] (gpt.py:346) host initialises 0 */
  int32_t* end_idx;                /* [B] (gpt.py:343) host initialises 0 */
  float* hiddens;                  /* [B, max_new, 768] per-step hidden states (gpt.py:435-436) */
  void* kcache;                    /* [n_layers, B, 12, T+max_new, 64] kv_dtype */
  void* vcache;
  const float* q;                  /* [nq, B*4, 626] Exp(1) draws of the CPU generator (gpt.py:501-508) */
  int32_t nq;                      /* 1 when manual_seed is set (same draw every step) */
  const float* temperature;        /* [4] (gpt.py:350-355) */
  const float* pow_table;          /* [17] penalty^f as torch computes it, or NULL (processors.py:29) */
  float top_p_thr;                 /* float32(1 - top_P) */
  int32_t use_top_p;
  int32_t top_k;
  int32_t use_top_k;
  int32_t min_new;                 /* min_new_token (gpt.py:494) */
  int32_t eos;                     /* 625 */
  int32_t row_offset;              /* global index of this shard's row 0 (b*4+k numbering, processors.py:24-27) */
  const int32_t* stop_at;          /* [B] or NULL: benchmark length forcing, not a reference feature */
  void* workspace;                 /* >= ctts_gpt_workspace_bytes(B, T) */
  size_t workspace_bytes;
  const int32_t* row_map;          /* [B] or NULL: compact decode row -> batch slot.  The host packs the utterances that are still
                                      running to the front and refreshes this (and n_active) whenever it polls `finish`; every
                                      array above stays indexed by the batch slot.  NULL = identity / all rows. */
  const int32_t* n_active;         /* device scalar or NULL: number of compact rows the decode step computes.  With row_map == NULL and
                                      n_active != NULL the library compacts ON THE DEVICE: the first kernel of every decode step ranks
                                      the utterances whose finish flag is 0 (ascending slot) and WRITES this scalar (it must be
                                      writable device memory), so finished utterances leave the step at once, without the host. */
  /* Slot-pool (continuous batching) extensions -- all optional, 0 / NULL = the plain generate() layout.  They let a
   * prefill of B freshly admitted utterances write into a larger pool of `kv_batch` utterance slots (row_map[m] = slot
   * of prefill row-group m), with every per-utterance array (ids_buf, len, kv_start, finish, end_idx, hiddens, q,
   * stop_at, prompt_len) allocated for the pool and indexed by slot. */
  int32_t cap;                     /* rows per slot in ids_buf and in the KV cache (0: T + max_new) */
  int32_t hid_cap;                 /* rows per slot in hiddens (0: max_new) */
  int32_t kv_batch;                /* utterance slots in kcache/vcache (0: B) */
  int32_t q_batch;                 /* utterance slots in q (0: B) */
  const int32_t* prompt_len;       /* [slots] padded prompt length of each slot (NULL: T for every row) */
  int32_t infer_text;              /* 1: refine-text mode -- text embedding/head, ONE sampling row per utterance (q is
                                      [nq, B, n_text], temperature[0]), the sampled id is written to all 4 slots
                                      (gpt.py:519-525); repetition penalty must be off */
  const int64_t* teacher_ids;      /* [slots, max_new, 4] or NULL: teacher forcing (evaluation hook, not a reference feature) -- the
                                      token WRITTEN at generation step i of utterance b is teacher_ids[b, i, :] instead of the
                                      sampled one; everything else (finish on EOS, lengths, hidden capture) is unchanged.  Used
                                      to bound the bf16 mode's drift against the reference's golden token stream. */
  int64_t* sampled_ids;            /* [slots, max_new (hid_cap), 4] or NULL: evaluation hook, the companion of teacher_ids -- the token
                                      the sampler itself drew at generation step i (after stop_at forcing, BEFORE teacher forcing
                                      replaced it).  With teacher_ids = the reference's stream this gives the teacher-forced token
                                      agreement rate of a numeric mode (bench.py `bf16_parity`). */
  const int32_t* order;            /* [B] or NULL: visiting order of the device-side compaction (a permutation of the utterance
                                      slots): compact row m of a decode step is the m-th utterance IN THIS ORDER whose finish flag
                                      is 0.  The host passes the utterances by descending context (the contexts of a batch differ
                                      only by the static valid prompt length), so the attention grid starts its longest
                                      (utterance, head) units first.  NULL = ascending slot.  No result depends on it. */
  /* Opt-in DEVICE generator for the Exp(1) draws of the multinomial (code mode only).  The reference draws them from
   * `torch.Generator(device=device)` (gpt.py:39,501-508): on its CPU path that is torch's CPU stream -- the parity contract, served
   * by `q` above -- and on a GPU device its device stream.  With rng_device = 1 `q` is not read: the sampling kernel draws
   * q = -log(u) itself from Philox4x32-10 keyed on rng_seed with counter (token / 4, global sampling row, step), so the default
   * unseeded path needs no per-step host draw / upload.  rng_per_step = 1: a fresh draw every step (manual_seed = None);
   * 0: the same draw every step (manual_seed set -- the reference re-seeds its generator at every step, gpt.py:504-507). */
  int32_t rng_device;
  int32_t rng_per_step;
  const uint64_t* rng_seed;        /* DEVICE scalar (read by the kernel at every step, so a captured graph serves every seed) */
  const uint32_t* rng_nonce;       /* [slots] or NULL, device generator only: a per-utterance-slot word that replaces the constant fourth Philox
                                      counter word.  A slot pool bumps it at every ADMISSION, so successive requests in one slot -- whose step
                                      index restarts at 0 -- do not replay the previous occupant's Exp(1) stream */
  /* PARITY CERTIFICATE (round 6).  [slots] float32 or NULL; the host initialises every entry to +inf.  At every step the sampling kernel
   * lowers margin[b] to the smallest distance -- in units of the TEMPERED logit, logit / temperature -- by which the step's outcome for
   * utterance b was decided: (1) log(r_best / r_second) of the multinomial's argmax(p / q) (gpt.py:497-508); (2) the value gap between the
   * last kept and the first dropped token at the cut the warpers make (processors.py:38-58, TopK / TopP prefix); (3) |log(cum / (1 - top_P))|
   * of the top-p test at the last kept and at the first top-p-dropped rank.  A perturbation of every tempered logit by less than
   * margin / 2 cannot change any sampled token of that utterance: the split-fp16 parity arithmetic (wqkv_x3 ...) is certified per call
   * against its measured logit error bound instead of by sample (chattts_amd/engine.py GptEngine.certify).  NULL: nothing is computed. */
  float* margin;
  /* [B] or NULL: global index of sampling row 0 of utterance b (b_global * 4 in code mode) -- replaces row_offset + 4 b when a shard holds a
   * NON-contiguous set of the caller's utterances (length-balanced data-parallel shards, chattts_amd/dist.py; the exact re-run of the
   * utterances a parity certificate flagged).  Keys the rows >= 625 repetition-penalty quirk (processors.py:24-27) and the device
   * generator's counter; `q` stays indexed by the local slot (the host uploads the selected rows of the CPU draw). */
  const int32_t* row_base;
  /* parity mode with BOTH decode copies loaded (wqkv_pk ... and wqkv_x3 ...): 1 = this call's decode steps run the f32 MFMA kernels
   * (csrc/decode32.hip) although the split-fp16 planes are there -- the exact fallback of a certificate that fired.  A captured graph
   * holds the choice it was built with. */
  int32_t proj_exact;
  /* "f32x3" mode, ctts_gpt_prefill only: the number of VALID prompt tokens of the batch (sum of the attention mask = sum of T - kv_start[b]),
   * known to the host.  > 0: the prompt pass runs over those rows only instead of over all B * T left-padded rows (the reference computes
   * the pad rows and never consumes them, gpt.py:234-241).  0: every row.  Same KV cache contents for the valid slots, same token. */
  int32_t prefill_valid_rows;
} ctts_gen_state;

int ctts_gpt_create(ctts_gpt** out, const ctts_gpt_weights* w);
void ctts_gpt_destroy(ctts_gpt* g);
size_t ctts_gpt_workspace_bytes(int32_t B, int32_t T);

/* step 0 of gpt.py:394: consumes emb[B,T,768] f32 (Embed.forward output, embed.py:52-79), fills the KV
 * cache, writes hiddens[:,0], samples token 0 into ids_buf[:,T], sets len = T+1.  In bf16 mode the four projections of
 * a layer run on LDS-tiled 128x128x64 MFMA tiles when B*T >= 256 rows (csrc/prefill.hip), on the decode kernels below that. */
int ctts_gpt_prefill(ctts_gpt* g, const ctts_gen_state* s, const float* emb, void* stream);
/* The same step 0 in pieces (long prompts, e.g. an `spk_smp` audio-code prompt of hundreds of tokens, core.py:435-453; interleaving
 * the prefill of newly admitted requests with running decode steps): chunk [t0, t0+tc) of the padded prompt of every row, emb_chunk
 * [B, tc, 768] f32.  Chunks must be issued in order; keys of earlier chunks are read from the KV cache.  Only the call with last != 0
 * (t0 + tc == T) runs the final norm / heads / sampling.  The workspace need only hold ctts_gpt_workspace_bytes(B, tc). */
int ctts_gpt_prefill_chunk(ctts_gpt* g, const ctts_gen_state* s, const float* emb_chunk, int32_t t0, int32_t tc, int32_t last, void* stream);
/* one iteration i > 0 of gpt.py:394-577, eager launches */
int ctts_gpt_decode_step(ctts_gpt* g, const ctts_gen_state* s, void* stream);
/* capture one decode step into a hipGraph (all per-step state is read from device memory, so the
 * same executable graph is replayed for every step), then replay it n times */
int ctts_gpt_graph_build(ctts_gpt* g, const ctts_gen_state* s, void* stream);
int ctts_gpt_graph_launch(ctts_gpt* g, int32_t n_steps, void* stream);
/* Round 5: the same captured step for a BOUND on the live rows.  The decode step's grids are sized for the batch B the state describes --
 * 12 attention workgroups and one row of every 16-row projection tile per utterance, finished or not (a finished row's workgroups leave
 * at their first load).  ctts_gpt_graph_build_rows captures the step once more with every grid sized for `rows` < B compact rows; the
 * kernels still read the device-side live count, so ANY bound >= it replays to the same bits as the plain graph.  A host that polls the
 * finish flags anyway (they only ever go up) launches the smallest bound it knows to hold: ctts_gpt_graph_launch_rows(g, n, rows, stream).
 * Needs ctts_gpt_graph_build first; build / destroy of the plain graph drop the bounded ones.  MEASURED: no gain on the C3 bench (the dead
 * workgroups were not what the step waits for, profiles/r5o_ab_graph_rows.log) -- the Python engine uses it only with CTTS_GRAPH_ROWS=1.  No reference counterpart (the reference steps
 * every row until the last one is done, gpt.py:512-518,592). */
int ctts_gpt_graph_build_rows(ctts_gpt* g, const ctts_gen_state* s, int32_t rows, void* stream);
int ctts_gpt_graph_launch_rows(ctts_gpt* g, int32_t n_steps, int32_t rows, void* stream);
void ctts_gpt_graph_destroy(ctts_gpt* g);

/* Per-kernel timing of one launch site (tag) of the eager decode step, for bench.py's roofline leg: the
 * launch goes through hipExtLaunchKernel with start/stop events (dispatch timestamps, as rocprofv3 reads them).
 * tags: 0 embed, 1 qkv, 2 rope_append, 3 attention, 4 o_proj, 5 gate_up, 6 down, 7 final_norm, 8 heads, 9 sample */
int ctts_gpt_profile_begin(ctts_gpt* g, int32_t tag, int32_t max_samples, int32_t stride /* time every stride-th launch */);
/* the individual durations (ms) of the launches timed since profile_begin, in launch order (sample j = the (j * stride)-th launch of the
 * tag); call before profile_end, which resets them.  bench.py fits duration = fixed + bytes / bandwidth over them. */
int ctts_gpt_profile_samples(ctts_gpt* g, float* ms_out, int32_t cap, int32_t* n_samples);
int ctts_gpt_profile_end(ctts_gpt* g, int32_t* n_samples, double* total_ms);

/* ------------------------------------------------------------------------------------------------
 * Acoustic decoder.  Replaces `self.decoder(batch_result)` + `self.vocos.decode(mel)` of
 * Chat._decode_to_wavs / _vocos_decode (ChatTTS/core.py:505-539), i.e. DVAE.forward decode branch
 * (dvae.py:276-297) and vocos.Vocos.decode.  Layout is channels-last: hid [B,T,768] (the per-token
 * hidden states, zero padded), mel [B,2T,100], wav [B,256(2T-1)].
 * Conv weights arrive repacked [Cout][tap][Cin]; depthwise kernels [7][512].
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  /* DVAE decoder (dvae.py:145-161,239) */
  const float* conv_in0_w; const float* conv_in0_b;   /* [128][3][384], [128] */
  const float* conv_in2_w; const float* conv_in2_b;   /* [512][3][128], [512] */
  int32_t n_dvae_blocks;
  const float* const* d_dw_w; const float* const* d_dw_b;    /* [7][512], [512] */
  const float* const* d_ln_w; const float* const* d_ln_b;
  const float* const* d_pw1_w; const float* const* d_pw1_b;  /* [2048][512] */
  const float* const* d_pw2_w; const float* const* d_pw2_b;  /* [512][2048] */
  const float* const* d_gamma;                               /* ConvNeXtBlock.weight [512] */
  const float* conv_out_w;                                   /* [384][512] */
  const float* out_conv_w;                                   /* [100][3][384] */
  const float* coef;                                         /* [100] */
  /* Vocos */
  const float* v_embed_w; const float* v_embed_b;            /* [512][7][100] */
  const float* v_norm_w; const float* v_norm_b;
  int32_t n_vocos_blocks;
  const float* const* v_dw_w; const float* const* v_dw_b;
  const float* const* v_ln_w; const float* const* v_ln_b;
  const float* const* v_pw1_w; const float* const* v_pw1_b;  /* [1536][512] */
  const float* const* v_pw2_w; const float* const* v_pw2_b;  /* [512][1536] */
  const float* const* v_gamma;
  const float* v_final_w; const float* v_final_b;
  const float* head_w; const float* head_b;                  /* [1026][512], [1026] */
  const float* window;                                       /* hann [1024] */
  const float* twiddle;                                      /* [512][2] cos/sin(2 pi k / 1024) */
  /* 0: every dense weight above is float32 [N][K] (f32-input MFMA tiles);
   * 1: every dense weight (conv_in*, *_pw1, *_pw2, conv_out, out_conv, v_embed, head) is instead a packed
   *    split-bf16 tensor [N][Kp/32][2][32] (per row and 32-wide k block: hi values, lo values; Kp = K rounded up to 32, zero padded), consumed by
   *    the bf16x3 tiles (3 bf16 MFMAs per product, f32-class accuracy).  Biases/LN/depthwise stay float32.
   * 2: as 1, but the ConvNeXt point-wise pairs (pwconv1 -> GELU -> pwconv2; 95 % of the decoder's flops) of batches from 12288
   *    frames run on ONE fp16 plane per operand (1 MFMA per product, f32 accumulation, activations saturated to +-65504 where they
   *    are rounded): the waveform stays within 1e-5 RMS of mode 1 (bar: 1e-4), the decoder takes about half the time.  The four
   *    *_x3p arrays below are then REQUIRED and hold fp16 planes [N/32][K/16][lane = (k%16)/8*32 + n%32][k%8]. */
  int32_t gemm_mode;
  /* gemm_mode 1, optional (NULL: the point-wise layers run on the tiles above at every size): the ConvNeXt pwconv1 / pwconv2
   * weights once more as PRE-SPLIT planes in MFMA fragment order, [N/32][K/16][hi|lo][lane = (k%16)/8*32 + n%32][k%8] bf16, for the
   * LDS-DMA staged kernel of csrc/codec_gemm.hip (used from 12288 frames; the depthwise-conv + LayerNorm kernel and the GELU
   * epilogue then write the activations as the same kind of planes) */
  const void* const* d_pw1_x3p; const void* const* d_pw2_x3p;
  const void* const* v_pw1_x3p; const void* const* v_pw2_x3p;
} ctts_codec_weights;

int ctts_codec_create(ctts_codec** out, const ctts_codec_weights* w);
void ctts_codec_destroy(ctts_codec* c);
size_t ctts_codec_workspace_bytes(int32_t B, int32_t F); /* F = mel frames = 2T */
/* ------------------------------------------------------------------------------------------------
 * Multi-GPU: ONE collective, the weight broadcast at load (SURVEY 8b `ctts_broadcast_weights`, 8e; the reference has no data-parallel
 * mode -- ChatTTS/core.py:458-468 batches inside one call only).  One process per GPU; utterances are sharded in contiguous row blocks
 * by the host (chattts_amd/dist.py shard_bounds) and never interact, so nothing else crosses GPUs.  librccl.so is dlopen'ed on the
 * first call (CTTS_RCCL_LIB overrides the name): single-GPU hosts need neither RCCL nor a communicator.  The library owns no weight
 * memory, so the broadcast is over the HOST'S list of device buffers (the repacked weights it is about to hand to ctts_gpt_create /
 * ctts_codec_create), in place, byte-typed, one ncclBroadcast per buffer on `stream` -- pack small tensors into a few flat buffers
 * first: a ring broadcast over xGMI is per-link bound (about 153 GB/s), per-call overhead dominates below ~1 MB.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { char internal[128]; } ctts_rccl_id;   /* = ncclUniqueId: made by ONE rank, carried to the others by the host's own means */
int ctts_rccl_unique_id(ctts_rccl_id* out);
int ctts_rccl_comm_create(void** comm /* ncclComm_t out */, int32_t world, const ctts_rccl_id* id, int32_t rank);   /* ncclCommInitRank on the current device */
void ctts_rccl_comm_destroy(void* comm);
int ctts_broadcast_weights(void* const* bufs, const size_t* bytes, int32_t n, void* comm /* ncclComm_t */, int32_t root, void* stream);

/* float32 waveform -> 16-bit PCM on the device: replaces `float_to_int16` (/root/reference/tools/audio/np.py:7-11) behind Chat.infer --
 * am = 32767 * 32768 // (int(ceil(max |x|)) * 32768); pcm = (x * am) truncated toward zero.  wav: [rows] rows of n samples, ld floats apart;
 * pcm: [rows][n] int16; per_row 0: ONE peak over the whole array (the function applied to a [B, n] block, examples/cmd/stream.py:44),
 * 1: a peak per row (one call per utterance: examples/web/funcs.py:206-209, tools/audio/pcm.py:29); product 0: the float64 product of the
 * reference's numba-jitted runtime, 1: the float32 product plain NumPy >= 2 forms from the same source line; keep_bits: NULL, or
 * [rows][ceil(n / 8)] bytes in np.packbits order, bit = |x| > keep_thr (the mask of Chat.infer's silence strip, core.py:262-265); peak:
 * [rows] uint32 device scratch.  An all-zero input gives zeros (the reference divides by zero).  Stream-ordered, two launches. */
int ctts_float_to_int16(const float* wav, int16_t* pcm, uint8_t* keep_bits, int32_t rows, int64_t n, int64_t ld, int32_t per_row,
                        int32_t product, float keep_thr, uint32_t* peak, void* stream);
/* Shader copy of `bytes` (a multiple of 16; both pointers 16-byte aligned) on `stream`.  `dst` may be PINNED HOST memory (mapped into
 * the device's address space): the float32 waveforms then reach the host -- the `.cpu().numpy()` that ends `Chat._decode_to_wavs`,
 * ChatTTS/core.py:508-510 -- as plain stores over PCIe, without the copy engines. */
int ctts_copy_bytes(void* dst, const void* src, size_t bytes, void* stream);
int ctts_dvae_decode(ctts_codec* c, const float* hid, float* mel, int32_t B, int32_t T, void* workspace, size_t ws_bytes, void* stream);
int ctts_vocos_decode(ctts_codec* c, const float* mel, float* wav, int32_t B, int32_t F, void* workspace, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Full DVAE (asset/DVAE.safetensors): audio -> 4 x T codes and codes -> mel through the GFSQ codebook.
 * Replaces `self.dvae(wav, "encode")` of `Chat.sample_audio_speaker` (ChatTTS/core.py:179-180 ->
 * DVAE.sample_audio, dvae.py:299-303, encode branch :265-274) and `self.dvae(batch_ids)` of
 * `Chat._decode_to_wavs(result.ids, use_decoder=False)` (core.py:518,535; decode branch dvae.py:276-297
 * with GFSQ._embed :87-97).  Dense weights are float32 [N][K] (f32-input MFMA tiles); convs repacked
 * [Cout][tap][Cin]; depthwise kernels [7][hidden].
 * ---------------------------------------------------------------------------------------------- */
typedef struct ctts_dvae ctts_dvae;

typedef struct {            /* DVAEDecoder(idim, odim, n_layer, bn_dim, hidden)  dvae.py:131-172 */
  int32_t idim, odim, hidden /* 256 | 512 */, bn_dim, n_blocks;
  const float* conv_in0_w; const float* conv_in0_b;   /* [bn][3][idim], [bn] */
  const float* conv_in2_w; const float* conv_in2_b;   /* [hidden][3][bn], [hidden] */
  const float* const* dw_w; const float* const* dw_b;
  const float* const* ln_w; const float* const* ln_b;
  const float* const* pw1_w; const float* const* pw1_b;   /* [4 hidden][hidden] */
  const float* const* pw2_w; const float* const* pw2_b;   /* [hidden][4 hidden] */
  const float* const* gamma;
  const float* conv_out_w;                                 /* [odim][hidden] */
} ctts_trunk_weights;

typedef struct {
  ctts_trunk_weights encoder;   /* 512 -> 1024, hidden 256 (config.py:33-39) */
  ctts_trunk_weights decoder;   /* 512 -> 512,  hidden 256 (config.py:40-46) */
  const float* ds0_w; const float* ds0_b;   /* downsample_conv.0: [512][3][100] */
  const float* ds1_w; const float* ds1_b;   /* downsample_conv.2 (k4, stride 2, pad 1) repacked as a k3 conv over frame
                                             * pairs: [512][3][1024] = {[0, W0], [W1, W2], [W3, 0]} */
  const float* out_conv_w;                  /* [100][3][512] */
  const float* coef;                        /* [100] */
  /* GroupedResidualFSQ(dim 1024, levels, num_quantizers R, groups G): per group project_in / project_out */
  const float* q_in_w; const float* q_in_b;     /* [G][4][D], [G][4] */
  const float* q_out_w; const float* q_out_b;   /* [G][D][4], [G][D] */
  int32_t levels[4];
  int32_t G, R, D;                          /* 2, 2, 512 */
  int32_t bound_first;                      /* 1: residual loop starts from bound(project_in(x)) (current library) */
  /* MelSpectrogram(n_fft 1024, hop 256, n_mels 100, center, power 1) buffers */
  const float* mel_window;                  /* [1024] */
  const float* mel_fb;                      /* [100][516]: fb^T, K padded 513 -> 516 with zeros */
  const float* twiddle;                     /* [512][2] cos/sin(2 pi k / 1024) */
} ctts_dvae_weights;

int ctts_dvae_create(ctts_dvae** out, const ctts_dvae_weights* w);
void ctts_dvae_destroy(ctts_dvae* c);
int32_t ctts_dvae_code_frames(int32_t n_samples);            /* T for a clip: F = 1 + n/256 mel frames, T = (F-2)/2 + 1 */
size_t ctts_dvae_encode_workspace_bytes(int32_t n_samples);
size_t ctts_dvae_decode_workspace_bytes(int32_t B, int32_t T);
/* wav [n_samples] f32 (24 kHz) -> codes [T][4] int32 (row t = the 4 code slots of frame t; DVAE.sample_audio
 * returns the transpose [4][T]) */
int ctts_dvae_encode(ctts_dvae* c, const float* wav, int32_t n_samples, int32_t* codes, void* workspace, size_t ws_bytes, void* stream);
/* codes [B][T][4] int64 (zero padded rows, core.py:525-533) -> mel [B][2T][100] */
int ctts_dvae_decode_codes(ctts_dvae* c, const int64_t* codes, float* mel, int32_t B, int32_t T, void* workspace, size_t ws_bytes,
                           void* stream);

/* ------------------------------------------------------------------------------------------------
 * Single-kernel entry points (unit parity tests call the kernels through these).
 * ---------------------------------------------------------------------------------------------- */
int ctts_k_gemm(int32_t tiled /* 0 skinny, 1 f32 tiles, 2 split-bf16 tiles */, const float* A, const void* W, float* C, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldc,
                int32_t wt, int32_t epi, const float* norm_w, float eps, const float* res, int32_t ldr, const float* bias,
                const float* gamma, int32_t taps, int32_t cin, int32_t frames, int32_t pad, int32_t dil, void* stream);
/* split-bf16 GEMM on pre-split planes in MFMA fragment order (csrc/codec_gemm.hip): Ap / Wp / Cp are
 * [rows/32][K/16][hi|lo][64][8] bf16 with rows padded to 256; epi 0: Cp = planes of gelu(A W^T + bias) (a [rows][N] matrix),
 * epi 1: C = res + gamma * (A W^T + bias), f32 [M][N].  N % 256 == 0, K % 32 == 0. */
int ctts_k_gemm_x3p(const uint16_t* Ap, const uint16_t* Wp, int32_t M, int32_t N, int32_t K, int32_t epi, const float* bias,
                    const float* gamma, const float* res, float* C, uint16_t* Cp, void* stream);
/* the same on ONE fp16 plane per operand (gemm_mode 2): Ap / Wp / Cp are [rows/32][K/16][64][8] fp16; K % 64 == 0 */
int ctts_k_gemm_h1p(const uint16_t* Ap, const uint16_t* Wp, int32_t M, int32_t N, int32_t K, int32_t epi, const float* bias,
                    const float* gamma, const float* res, float* C, uint16_t* Cp, void* stream);
/* a HIP stream confined to n_cus compute units (first_cu, first_cu + stride, ...): hipExtStreamCreateWithCUMask.  The host runs the acoustic
 * decoder of one batch on such a stream while the next batch is generated on the others (the overlapped DVAE / ISTFT side stream of BASELINE
 * config 4, at batch level: CodecEngine.decode_to_wavs_async); destroy with ctts_stream_destroy */
int ctts_stream_create_cu_mask(int32_t first_cu, int32_t n_cus, int32_t stride, int32_t complement /* 1: every CU but those */, void** stream);
int ctts_stream_destroy(void* stream);
/* one ConvNeXt MLP in ONE launch: C = C + gamma * (GELU(A W1^T + b1) W2^T + b2), A [M][512] / W1 [inter][512] / W2 [512][inter] as
 * fp16 planes (planes = 1), C [M][512] f32 in place; bit-identical to ctts_k_gemm_h1p(epi 0) followed by ctts_k_gemm_h1p(epi 1)
 * (ConvNeXtBlock.pwconv1 -> act -> pwconv2 -> gamma -> residual, ChatTTS/model/dvae.py:46-66) */
int ctts_k_mlp_fused(const uint16_t* Ap, const uint16_t* W1p, const uint16_t* W2p, int32_t M, int32_t inter, const float* b1,
                     const float* b2, const float* gamma, float* C, int32_t planes, void* stream);
/* perf-mode projection: bf16 activations/weights, optional per-row 1/rms from 48 partial sums of squares,
 * epi 0 = f32 store, 1 = residual add (+ bf16 copy + new partial sums), 2 = SiLU(gate)*up -> bf16 */
int ctts_k_gemm_fast(const uint16_t* A, int32_t lda, const uint16_t* W, int32_t M, int32_t N, int32_t K, const float* ssq_in, float eps,
                     int32_t epi, float* C32, int32_t ldc, uint16_t* Cb, int32_t ldcb, float* ssq_out, void* stream);
/* perf-mode fused RMSNorm-scale + QKV + RoPE + KV append (W rows permuted as engine.py `rope_row_perm` does) */
int ctts_k_qkv_rope(const uint16_t* A, const uint16_t* W, int32_t M, const float* ssq_in, float eps, float* qkv, uint16_t* kcache,
                    uint16_t* vcache, int32_t cmax, const float* cos_tab, const float* sin_tab, int32_t q_per_b, const int32_t* len,
                    const int32_t* kv_start, int32_t force_mb, void* stream);
/* decode-step projection on fragment-packed operands (csrc/decode.hip): Ap [ceil(M/16)][K/32][64][8] bf16, Wp packed likewise
 * (epi 2: gate tiles then up tiles); epi 1 = residual add (C32 in place, Cp packed bf16 copy with kch_out = N/32, ssq_out),
 * 2 = SiLU(gate)*up -> Cp packed (kch_out = N/32).  n_active: device scalar or NULL.  force_mb: rows/16 per workgroup (0 = default) */
int ctts_k_gemm_dec(const uint16_t* Ap, const uint16_t* Wp, int32_t M, int32_t N, int32_t K, const int32_t* n_active, const float* ssq_in,
                    float eps, int32_t epi, float* C32, int32_t ldc, uint16_t* Cp, int32_t kch_out, float* ssq_out, int32_t force_mb,
                    void* stream);
/* the same for the f32 parity mode (csrc/decode32.hip): Ap [ceil(M/16)][K/16][64][4] f32, Wp packed likewise; norm_w != NULL: RMSNorm
 * prologue (X = the same rows row-major, ldx) ; epi 0 = store C row-major, 1 = C = res + acc (row-major) and Cp (packed, kch_out = N/16),
 * 2 = SiLU(gate)*up -> Cp.  Bit-identical to ctts_k_gemm(tiled = 0, wt = f32) on the same values. */
int ctts_k_gemm_dec32(const float* Ap, const float* Wp, int32_t M, int32_t N, int32_t K, const int32_t* n_active, const float* X, int32_t ldx,
                      const float* norm_w, float eps, int32_t epi, float* C, int32_t ldc, const float* res, int32_t ldr, float* Cp,
                      int32_t kch_out, int32_t force_mb, int32_t n_cols /* epi 0: columns of C that exist (N padded to 16), 0 = N */,
                      void* stream);
/* The same projections on SPLIT-bf16 operands (csrc/decode32x.hip; ctts_gpt_weights.wqkv_x3 ...): Ap / Wp = hi planes in the bf16 fragment order
 * ([rows/16][K/32][64][8]), the lo planes a_plane / w_plane ELEMENTS behind them; three bf16 MFMAs per product (lo*hi + hi*lo + hi*hi), f32
 * accumulation.  X != NULL (K = 768): RMSNorm launch -- 1 / rms of the rows of X (row-major f32, gemm_skinny_k's arithmetic) scales the
 * accumulator, the gain is expected folded into W; ssq_in != NULL instead: [rows][48] partial sums of squares of the rows (one per 16 columns, what
 * the decode step's producers leave); ssq_out (epi 1): the same partials of the NEW rows.  force_mb: 0 = the launcher's choice, else rows per
 * workgroup / 16 (1 | 2 | 4), + 8 for eight instead of four waves.  epi 1 = C = res + acc (row-major f32), Cp = its planes (c_plane elements apart, kch_out =
 * N / 32) and optionally Cp32 = its packed f32 copy; epi 2 = SiLU(gate) * up -> Cp planes (W = gate tiles then up tiles).  Reference ops:
 * examples/onnx/modeling_llama.py:293,415-417,500. */
int ctts_k_gemm_dec32x(const uint16_t* Ap, int64_t a_plane, const uint16_t* Wp, int64_t w_plane, int32_t M, int32_t N, int32_t K,
                       const int32_t* n_active, const float* X, int32_t ldx, float eps, int32_t epi, float* C, int32_t ldc, const float* res,
                       int32_t ldr, uint16_t* Cp, int64_t c_plane, int32_t kch_out, float* Cp32, int32_t force_mb, const float* ssq_in,
                       float* ssq_out, void* stream);
/* Round 6: the LDS-tiled split-bf16 GEMM of the "f32x3" mode's PROMPT pass (csrc/prefill32x.hip): C = epi(rstd[row] * ((A diag(norm_w)) W^T)),
 * A [M, lda] and W [N (2N for epi 2: gate rows then up rows), K] float32 row-major, split hi | lo by the tile loader, three bf16 MFMAs per
 * product, f32 accumulation.  epi 0 = store, 1 = C = res + acc, 2 = silu(gate) * up.  norm_w / rstd: both or neither.  K % 32 == 0,
 * N % 128 == 0 (epi 2: % 64).  Reference ops: HF Llama projections, examples/onnx/modeling_llama.py:259-295,455-505. */
int ctts_k_gemm_pre_x3(const float* A, int32_t lda, const float* W, float* C, int32_t ldc, int32_t M, int32_t N, int32_t K, int32_t epi,
                       const float* norm_w, const float* rstd, const float* res, int32_t ldr, void* stream);
/* which decode32 kernel the calling thread's last ctts_k_gemm_dec32 / decode step picked: "rms16" | "m16" | "generic" (all bit-identical) */
const char* ctts_k_dec32_last_variant(void);
int ctts_k_rows_prep(const float* x32, uint16_t* xb, float* ssq, int32_t M, void* stream);
int ctts_k_rope_append(float* qkv, void* kcache, void* vcache, int32_t kv_dtype, int32_t cmax, const float* cos_tab, const float* sin_tab,
                       int32_t q_per_b, const int32_t* len, const int32_t* kv_start, int32_t M, void* stream);
int ctts_k_attention(const float* qkv, const void* kcache, const void* vcache, int32_t kv_dtype, int32_t cmax, float* out,
                     int32_t q_per_b, const int32_t* len, const int32_t* kv_start, int32_t M, void* stream);
/* prefill attention of the perf mode over a bf16 KV cache: M = B * q_per_b query rows (row m: utterance m / q_per_b, KV slot
 * slot0 + m % q_per_b -- slot0 > 0: a later chunk of a prompt prefilled in pieces), f32 output [M,768]; from q_per_b >= 128 the
 * flash-style MFMA kernel runs (csrc/gpt.hip attention_prefill_mfma_k), the per-row kernel below that */
int ctts_k_attention_prefill(const float* qkv, const uint16_t* kcache, const uint16_t* vcache, int32_t cmax, float* out, int32_t q_per_b,
                             int32_t slot0, const int32_t* kv_start, int32_t M, void* stream);
/* decode attention of the perf mode: one query row per utterance, bf16 KV cache [slots,12,cmax,64], output bf16 in the fragment-packed
 * order of csrc/decode.hip; desc [M][4] int32 = {utterance slot (-1: skip), KV slot of the query, RoPE position (unused here), first
 * visible key}; n_active: device scalar or NULL (M).  n_cu > 0 with part ([n_cu][8][66] f32) and cnt ([n_cu] int32, zeroed) turns on
 * remainder splitting of the (utterance, head) units over workgroups (csrc/gpt.hip attention_k); n_cu = 0: one workgroup per unit. */
int ctts_k_attention_dec(const float* qkv, const uint16_t* kcache, const uint16_t* vcache, int32_t cmax, uint16_t* out_packed,
                         const int32_t* desc, const int32_t* n_active, int32_t M, float* part, int32_t* cnt, int32_t n_cu, void* stream);
/* Decode attention of EITHER mode the way the decode step launches it (round 5): kv_dtype CTTS_BF16 = bf16 cache, bf16 output in the
 * fragment-packed order of csrc/decode.hip; CTTS_F32 = f32 cache, f32 output in the packed order of csrc/decode32.hip; 2 = f32 cache, output as
 * hi | lo bf16 planes in decode.hip's order, the lo plane ceil16(M) * 768 elements behind (the split-bf16 parity mode, csrc/decode32x.hip).  covers_all != 0:
 * desc is valid for all M rows (absent rows carry slot -1) and n_active is not read.  Runs the persistent grid (csrc/gpt.hip
 * attention_persist_k: <= one workgroup per CU walking the live (utterance, head) units) unless ctts_k_attention_cfg / CTTS_ATT_PERSIST=0
 * selected one workgroup per unit (attention_k); both give the same bits.  Reference op: examples/onnx/modeling_llama.py:455-475. */
int ctts_k_attention_dec2(const float* qkv, const void* kcache, const void* vcache, int32_t kv_dtype, int32_t cmax, void* out_packed,
                          const int32_t* desc, const int32_t* n_active, int32_t covers_all, int32_t M, void* stream);
/* tests / probes: decode attention launch shape of this process -- persist 0 | 1 (< 0: keep), workgroups of the persistent grid (<= 0:
 * keep; default = the device's CU count), KV blocks per wave ring 2 | 3 | 4 (<= 0: keep; default 4).  Never changes a result bit. */
int ctts_k_attention_cfg(int32_t persist, int32_t workgroups, int32_t ring);
/* round 5 A/B: `hpw` consecutive heads of one utterance per decode-attention workgroup (1 = the shipped grid of one workgroup per (utterance,
 * head); 2 | 3 | 4: the same 4-wave units, 12 / hpw workgroups of 4 hpw waves per utterance -- bit-identical output; env CTTS_ATT_HPW). */
int ctts_k_attention_heads_per_wg(int32_t hpw);
/* The same attention with o_proj + residual folded in (ctts_gpt_weights.wo_hd): wo_hd [12][8][768][8] bf16; part [ceil16(M)][12][768]
 * f32 scratch; cnt [M] int32, zero (left zero); x32 [M][768] f32 residual, updated in place; xp its bf16 copy in the fragment-packed
 * order; ssq [M][48] partial sums of squares of the new residual. */
int ctts_k_attention_oproj(const float* qkv, const uint16_t* kcache, const uint16_t* vcache, int32_t cmax, const uint16_t* wo_hd,
                           const int32_t* desc, const int32_t* n_active, int32_t M, float* part, int32_t* cnt, float* x32, uint16_t* xp,
                           float* ssq, void* stream);
/* what the library's per-call device guard does for `stream`: the device current before the call, the one that owns the stream, the one
 * current inside the guard, whether it switched; fails if the previous device is not restored */
int ctts_k_device_guard_probe(void* stream, int32_t* before, int32_t* stream_dev, int32_t* inside, int32_t* switched);
int ctts_k_embed_codes(const float* emb_code, const int64_t* ids_buf, int32_t tcap, const int32_t* len, float* x, int32_t B, void* stream);
int ctts_k_final_norm(const float* x, int32_t q_per_b, const float* w, float eps, float* hfin, float* hiddens, int32_t max_new,
                      const int32_t* len, int32_t T, int32_t B, void* stream);
int ctts_k_sample(const ctts_gen_state* s, const float* logits, void* stream);
/* the refine-text sampler alone (sample_text_k, gpt.py:439-440,477-525): logits [B, n_text], q [nq, B, n_text], temperature[0], ONE
 * sampling row per utterance, the token replicated over the 4 slots; no repetition penalty */
int ctts_k_sample_text(const ctts_gen_state* s, const float* logits, int32_t n_text, void* stream);
/* the device generator's Exp(1) draws (ctts_gen_state.rng_device) of sampling rows row0 .. row0+rows-1 at generation step `step`
 * as a [rows, V] float32 tensor -- distribution tests */
int ctts_k_exp_draws(uint64_t seed, int32_t step, int32_t row0, int32_t rows, int32_t V, float* out, void* stream);
int ctts_k_dwconv_ln(const float* x, const float* w, const float* b, const float* ln_w, const float* ln_b, float eps, int32_t dil,
                     float* y, int32_t B, int32_t F, void* stream);
int ctts_k_layernorm(const float* x, const float* w, const float* b, float eps, float* y, int32_t rows, void* stream);
int ctts_k_istft(const float* head, const float* window, const float* twiddle, float* frames, float* wav, int32_t B, int32_t F, void* stream);

#ifdef __cplusplus
}
#endif
#endif
