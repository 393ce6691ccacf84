// extern "C" boundary of libchattts_amd.so (declared in include/chattts_amd.h) and the launch
// sequences of the GPT step / DVAE / Vocos.  No device allocation, no device synchronisation.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <vector>

#include "../../include/chattts_amd.h"
#include "kernels.hpp"

static thread_local char g_err[512] = "";
int ctts_fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return -1;
}
#define fail ctts_fail
#define CK(expr)                                                                                   \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

extern "C" const char* ctts_last_error(void) { return g_err; }
extern "C" int ctts_version(void) { return 1; }

static const int HID = 768, INTER = 3072, NHEAD = 12, HDIM = 64, NVQ = 4, NAUDIO = 626, NTEXT_MAX = 21248;

// ------------------------------------------------------------------------------------------------
struct ctts_gpt {
  ctts_gpt_weights w;
  std::vector<const void*> wqkv, wo, wgu, wd;
  std::vector<const void*> wqkv_pk, wo_pk, wgu_pk, wd_pk;   // fragment-packed copies for the decode step, or empty
  std::vector<const void*> wo_hd;                           // o_proj per attention head (perf mode), or empty
  std::vector<const void*> wqkv_x3, wo_x3, wgu_x3, wd_x3;   // parity mode: hi | lo bf16 planes in fragment order (decode32x.hip), or empty
  bool dec_x3 = false;         // parity mode: the split-bf16 planes are loaded (decode32x.hip); ctts_gen_state.proj_exact picks decode32.hip per call
  bool qkv_att = false;        // perf mode decode, <= 64 rows: QKV + attention as ONE launch (gpt.hip qkv_attention_k); env CTTS_QKV_ATT=1, OFF by default
  bool att_oproj = false;      // perf mode decode: o_proj + residual folded into the attention launch -- OPT-IN, env CTTS_ATT_OPROJ=1 (default: separate launches)
  bool dec_packed = false;     // perf mode (bf16 weights): decode.hip
  bool dec_packed32 = false;   // parity mode (f32 weights): decode32.hip
  bool heads_packed = false;   // heads GEMM on packed f32 operands (decode32.hip), both modes
  int pf_mask = 0;             // env CTTS_PF: cross-kernel weight prefetch, bit mask (1 QKV->o_proj, 16 QKV->gate/up, 2 attention->gate/up)
  int temporal_layers = 0;     // env CTTS_W_TEMPORAL_LAYERS: decode weights of layers [0, N) loaded WITHOUT the non-temporal hint (A/B)
  bool pre32_packed = true;    // parity mode: prompt pass on the packed f32 kernels (env CTTS_PRE32_PACKED=0: row-major gemm_skinny_k)
  bool embed_fold = false;     // multi-step graphs: steps 2..n take their input rows from the previous step's sampling launch (env CTTS_EMBED_FOLD=1)
  bool pre_compact = true;     // "f32x3" mode: the prompt pass over the valid prompt tokens only (env CTTS_PRE_COMPACT=0: all B * T rows)
  bool pre_x3 = true;          // "f32x3" mode: prompt pass on the LDS-tiled split-bf16 GEMM (prefill32x.hip); env CTTS_PRE_X3=0: the f32 MFMA kernels
  bool fnorm_fuse = true;      // decode: final norm + hidden capture + heads in one launch (env CTTS_FNORM_FUSE=0: separate launches)
  std::vector<const float*> ln1, ln2;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  // optional second executable graph of `multi_steps` consecutive decode steps (env CTTS_GRAPH_STEPS > 1): one hipGraphLaunch per
  // multi_steps steps instead of one per step
  hipGraph_t graph_multi = nullptr;
  hipGraphExec_t exec_multi = nullptr;
  // round 5: the same step captured for a BOUND on the live rows (ctts_gpt_graph_build_rows): every grid of the step is sized for `rows`
  // compact rows instead of B -- the attention launch loses its dead workgroups (12 per finished utterance), the 16-row projections their
  // dead row tiles.  Nothing else changes: the kernels still read the live count, so a bound that is merely >= it gives the same bits.
  struct RowsGraph { hipGraph_t graph = nullptr, gm = nullptr; hipGraphExec_t exec = nullptr, exec_multi = nullptr; };
  std::map<int, RowsGraph> rows_graphs;
  int multi_steps = 1;
  // profiling (eager decode only)
  int prof_tag = -1;
  int prof_max = 0;
  bool skip_finished = true;  // env CTTS_SKIP_FINISHED=0 restores the reference's "finished rows keep stepping"
  int n_cu = 0;               // compute units of the device when attention remainder splitting is on (env CTTS_ATT_SPLIT=1); 0 = off
  int prof_stride = 1;   // time every prof_stride-th launch of the tag
  int prof_seen = 0;
  std::vector<hipEvent_t> ev0, ev1;
  int prof_n = 0;
};

static size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }
#define ATT_CUS_MAX 512   // upper bound of compute units the attention split state is sized for
#define QA_LAYERS_MAX 32  // layers the arrival words of the fused QKV + attention launches are sized for
#define QA_STRIDE HO_STRIDE   // kernels.hpp: ints between two arrival words
struct GptWs {
  float *x, *qkv, *ao, *act, *hfin, *logits, *ssq;
  uint16_t* xb;  // bf16 copy of the residual stream (perf mode); ao / act are reused as bf16 buffers there
  // decode step on fragment-packed operands (decode.hip): row tiles of 16 utterances
  uint16_t *xp, *aop, *actp;
  float *xp32, *aop32, *actp32;   // the same in float32 (parity mode, decode32.hip)
  float* hfinp;                   // final-norm rows in the packed f32 order (A operand of the packed heads GEMM, both modes)
  float* rstd;                    // parity mode prefill: 1 / rms of every prompt row (prefill32.hip)
  RowDesc* desc;
  float* att_part;    // attention remainder splitting: partials of the split units' pieces, and their arrival counters
  int32_t* att_cnt;
  int32_t* row_map;   // device-side compaction: this step's compact row -> utterance map (written by the step's first kernel)
  float* rope_cs;     // [Bp][64] cos[32] | sin[32] of every decode row's position, written by the step's first kernel (fused QKV + attention)
  int32_t* qa_flag;   // fused QKV + attention launches: arrival words [QA_LAYERS_MAX][12 heads], QA_STRIDE ints apart, zeroed by the step's first kernel
  float* op_part;     // attention + o_proj in one launch: [Bp][12][768] partials of the (utterance, head) units ...
  int32_t* op_cnt;    // ... and the rows' arrival counters, RIGHT BEHIND att_cnt: one memset zeroes both (cnt_bytes)
  size_t op_part_bytes;
  size_t bytes;
};
static size_t cnt_bytes(int B) { return ((size_t)ATT_CUS_MAX + ((size_t)B + 15) / 16 * 16) * sizeof(int32_t); }
static GptWs carve(void* base, int B, int T) {
  const size_t M = (size_t)B * (T > 1 ? T : 1);
  GptWs w;
  size_t off = 0;
  char* p = (char*)base;
  w.x = (float*)(p + off); off += align_up(M * HID * 4);
  w.qkv = (float*)(p + off); off += align_up(M * 3 * HID * 4);
  w.ao = (float*)(p + off); off += align_up(M * HID * 4);
  w.act = (float*)(p + off); off += align_up(M * INTER * 4);
  w.hfin = (float*)(p + off); off += align_up((size_t)B * HID * 4);
  w.logits = (float*)(p + off); off += align_up((size_t)B * NTEXT_MAX * 4);  // >= B*4*626; refine-text mode needs B*n_text
  w.ssq = (float*)(p + off); off += align_up(M * SSQ_PARTS * 4);
  w.xb = (uint16_t*)(p + off); off += align_up(M * HID * 2);
  const size_t Bp = ((size_t)B + 15) / 16 * 16;
  w.xp = (uint16_t*)(p + off); off += align_up(Bp * HID * 2);
  w.aop = (uint16_t*)(p + off); off += align_up(Bp * HID * 2);
  w.actp = (uint16_t*)(p + off); off += align_up(Bp * INTER * 2);
  // the packed f32 buffers and the row descriptors also serve the parity mode's PREFILL (M = B * T rows in 16-row tiles)
  const size_t Mp = (M + 15) / 16 * 16;
  w.xp32 = (float*)(p + off); off += align_up(Mp * HID * 4);
  w.aop32 = (float*)(p + off); off += align_up(Mp * HID * 4);
  w.actp32 = (float*)(p + off); off += align_up(Mp * INTER * 4);
  w.hfinp = (float*)(p + off); off += align_up(Bp * HID * 4);
  w.desc = (RowDesc*)(p + off); off += align_up(Mp * sizeof(RowDesc));
  w.rstd = (float*)(p + off); off += align_up(Mp * sizeof(float));
  w.row_map = (int32_t*)(p + off); off += align_up(Bp * sizeof(int32_t));
  w.att_part = (float*)(p + off); off += align_up((size_t)ATT_CUS_MAX * ATT_SPLIT_MAX * 66 * sizeof(float));
  w.att_cnt = (int32_t*)(p + off); off += align_up(cnt_bytes(B));
  w.op_cnt = w.att_cnt + ATT_CUS_MAX;
  w.op_part_bytes = Bp * NHEAD * HID * sizeof(float);
  w.op_part = (float*)(p + off); off += align_up(w.op_part_bytes);
  w.rope_cs = (float*)(p + off); off += align_up(Bp * 64 * sizeof(float));
  w.qa_flag = (int32_t*)(p + off); off += align_up((size_t)QA_LAYERS_MAX * NHEAD * QA_STRIDE * sizeof(int32_t));
  w.bytes = off;
  return w;
}

extern "C" size_t ctts_gpt_workspace_bytes(int32_t B, int32_t T) { return carve(nullptr, B, T).bytes; }

extern "C" int ctts_gpt_create(ctts_gpt** out, const ctts_gpt_weights* w) {
  if (!out || !w || w->n_layers <= 0) return fail("ctts_gpt_create: bad arguments");
  ctts_gpt* g = new ctts_gpt();
  g->w = *w;
  const int L = w->n_layers;
  g->wqkv.assign(w->wqkv, w->wqkv + L);
  g->wo.assign(w->wo, w->wo + L);
  g->wgu.assign(w->wgu, w->wgu + L);
  g->wd.assign(w->wd, w->wd + L);
  g->ln1.assign(w->ln1, w->ln1 + L);
  g->ln2.assign(w->ln2, w->ln2 + L);
  if (w->wqkv_pk && w->wo_pk && w->wgu_pk && w->wd_pk) {   // packed in the order of the weight dtype's decode kernel
    g->wqkv_pk.assign(w->wqkv_pk, w->wqkv_pk + L);
    g->wo_pk.assign(w->wo_pk, w->wo_pk + L);
    g->wgu_pk.assign(w->wgu_pk, w->wgu_pk + L);
    g->wd_pk.assign(w->wd_pk, w->wd_pk + L);
    const char* e = getenv("CTTS_DEC_PACKED");   // =0: decode on the row-major kernels (A/B)
    const bool on = !(e && atoi(e) == 0);
    g->dec_packed = on && w->weight_dtype == CTTS_BF16;
    g->dec_packed32 = on && w->weight_dtype != CTTS_BF16 && w->kv_dtype != CTTS_BF16;
  }
  { const char* e = getenv("CTTS_DEC_PACKED"); g->heads_packed = w->heads_pk != nullptr && !(e && atoi(e) == 0); }
  if (w->wqkv_x3 && w->wo_x3 && w->wgu_x3 && w->wd_x3 && w->weight_dtype != CTTS_BF16 && w->kv_dtype != CTTS_BF16) {   // (w*_pk: optional beside them)
    g->wqkv_x3.assign(w->wqkv_x3, w->wqkv_x3 + L);
    g->wo_x3.assign(w->wo_x3, w->wo_x3 + L);
    g->wgu_x3.assign(w->wgu_x3, w->wgu_x3 + L);
    g->wd_x3.assign(w->wd_x3, w->wd_x3 + L);
    g->dec_x3 = true;   // the host's choice (no planes = exact f32 MFMA); per call: ctts_gen_state.proj_exact
  }
  if (w->wo_hd && g->dec_packed) {
    g->wo_hd.assign(w->wo_hd, w->wo_hd + L);
    // OFF by default: measured on the C3 bench the fused launch costs what the two launches cost (13.9 us vs 9.07 + 4.87) and the step
    // gets 26 us SLOWER (profiles/r4b_ab_oproj.log): the 12 -> 1 hand-off through memory is a chain of dependent round trips
    const char* e = getenv("CTTS_ATT_OPROJ");
    g->att_oproj = e && atoi(e) == 1;
  }
  { const char* e = getenv("CTTS_SKIP_FINISHED"); if (e && atoi(e) == 0) g->skip_finished = false; }
  { const char* e = getenv("CTTS_FNORM_FUSE"); if (e && atoi(e) == 0) g->fnorm_fuse = false; }
  { const char* e = getenv("CTTS_W_TEMPORAL_LAYERS"); if (e) g->temporal_layers = atoi(e); }
  { const char* e = getenv("CTTS_PF"); if (e) g->pf_mask = atoi(e); }
  // OFF by default: measured on the C3 bench the fused launch is worth +0.1 ... +0.8 % (13.9 us against 5.3 + 9.1 us; more prefetch ahead of
  // the wait LOSES 4 %, none loses 4 %: profiles/r4f_ab_qkv_att_pre.log, r4g_ab_qkv_att_variants.log) -- not enough to make a launch that
  // spins on device memory the default
  { const char* e = getenv("CTTS_QKV_ATT"); g->qkv_att = e && atoi(e) == 1; }
  { const char* e = getenv("CTTS_PRE32_PACKED"); if (e && atoi(e) == 0) g->pre32_packed = false; }
  { const char* e = getenv("CTTS_PRE_X3"); if (e && atoi(e) == 0) g->pre_x3 = false; }
  { const char* e = getenv("CTTS_PRE_COMPACT"); if (e && atoi(e) == 0) g->pre_compact = false; }
  { const char* e = getenv("CTTS_EMBED_FOLD"); if (e) g->embed_fold = atoi(e) != 0; }
  {
    int dev = 0, cus = 0;
    // OFF by default: measured on the C3 bench it does not pay (attention 9.3 -> 9.8 us per launch, 1296 -> 1280 audio-s/s,
    // profiles/r2e_ab_split.log) and it makes a row's bf16 result depend on how many rows share the step
    const char* e = getenv("CTTS_ATT_SPLIT");
    if (e && atoi(e) == 1 && hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0 && cus <= ATT_CUS_MAX)
      g->n_cu = cus;
  }
  *out = g;
  return 0;
}

extern "C" void ctts_gpt_graph_destroy(ctts_gpt* g) {
  if (!g) return;
  if (g->exec) { (void)hipGraphExecDestroy(g->exec); g->exec = nullptr; }
  if (g->graph) { (void)hipGraphDestroy(g->graph); g->graph = nullptr; }
  if (g->exec_multi) { (void)hipGraphExecDestroy(g->exec_multi); g->exec_multi = nullptr; }
  if (g->graph_multi) { (void)hipGraphDestroy(g->graph_multi); g->graph_multi = nullptr; }
  for (auto& kv : g->rows_graphs) {
    if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    if (kv.second.exec_multi) (void)hipGraphExecDestroy(kv.second.exec_multi);
    if (kv.second.gm) (void)hipGraphDestroy(kv.second.gm);
  }
  g->rows_graphs.clear();
}

extern "C" void ctts_gpt_destroy(ctts_gpt* g) {
  if (!g) return;
  ctts_gpt_graph_destroy(g);
  for (auto e : g->ev0) (void)hipEventDestroy(e);
  for (auto e : g->ev1) (void)hipEventDestroy(e);
  delete g;
}

thread_local hipEvent_t ctts_prof_start = nullptr, ctts_prof_stop = nullptr;

// arms the start/stop events for the single kernel launch that follows (see CTTS_LAUNCH in kernels.hpp)
struct Prof {
  ctts_gpt* g; bool on;
  Prof(ctts_gpt* g_, int tag, hipStream_t, bool allow) : g(g_), on(false) {
    if (allow && g->prof_tag == tag && g->prof_n < g->prof_max && (g->prof_seen++ % g->prof_stride) == 0) {
      on = true;
      ctts_prof_start = g->ev0[g->prof_n];
      ctts_prof_stop = g->ev1[g->prof_n];
    }
  }
  ~Prof() {
    if (on) { g->prof_n++; ctts_prof_start = nullptr; ctts_prof_stop = nullptr; }
  }
};

static SampleArgs make_sample_args(const ctts_gen_state* s, const float* logits) {
  SampleArgs a;
  a.logits = logits; a.ids_buf = s->ids_buf; a.tcap = s->cap ? s->cap : s->T + s->max_new; a.T = s->T; a.len = s->len; a.finish = s->finish;
  a.end_idx = s->end_idx; a.q = s->q; a.nq = s->nq; a.temperature = s->temperature; a.pow_table = s->pow_table;
  a.top_p_thr = s->top_p_thr; a.use_top_p = s->use_top_p; a.top_k = s->top_k; a.use_top_k = s->use_top_k;
  a.min_new = s->min_new; a.eos = s->eos; a.row_offset = s->row_offset; a.max_input_ids = NAUDIO - 1; a.stop_at = s->stop_at;
  a.B = s->B; a.row_map = nullptr; a.n_active = nullptr; a.prompt_len = s->prompt_len; a.q_rows = s->q_batch ? s->q_batch : s->B;
  a.teacher = s->teacher_ids; a.teacher_stride = s->hid_cap ? s->hid_cap : s->max_new; a.sampled = s->sampled_ids;
  a.dbg = nullptr;
  { const char* e = getenv("CTTS_SAMPLE_DBG_PTR"); if (e) a.dbg = (long long*)strtoull(e, nullptr, 0); }   // probes only
  a.desc = nullptr; a.rng_device = s->rng_device; a.rng_per_step = s->rng_per_step; a.rng_seed = reinterpret_cast<const unsigned long long*>(s->rng_seed);
  a.rng_nonce = s->rng_nonce;
  a.margin = s->margin; a.row_base = s->row_base;
  memset(&a.next, 0, sizeof(a.next));
  return a;
}

// ws_T: prompt slots the workspace must hold for THIS call (a prompt chunk needs ctts_gpt_workspace_bytes(B, tc) only; the decode
// step carves for one row per utterance but keeps the prefill's carve layout, so it is checked against the geometry the caller
// allocated for: s->T unless `decode_ws_T` says otherwise)
static int check_state(const ctts_gpt* g, const ctts_gen_state* s, int ws_T = 0) {
  if (!g || !s) return fail("null engine/state");
  if (s->B <= 0 || s->T <= 0 || s->max_new <= 0) return fail("bad B/T/max_new");
  const int cap_ = s->cap ? s->cap : s->T + s->max_new;
  if (cap_ > g->w.max_pos) return fail("slot capacity T + max_new (%d) exceeds the RoPE table (%d)", cap_, g->w.max_pos);
  if (s->cap && s->T + 1 > s->cap) return fail("prompt does not fit the slot capacity");
  if (s->workspace_bytes < ctts_gpt_workspace_bytes(s->B, ws_T > 0 ? ws_T : s->T)) return fail("workspace too small");
  if (!s->rng_device && (s->nq <= 0 || !s->q)) return fail("q draws missing");
  if (s->rng_device && !s->rng_seed) return fail("rng_device needs the rng_seed device scalar");
  if (s->rng_device && s->infer_text) return fail("the device generator serves the code mode only (refine-text samples from `q`)");
  if (g->w.weight_dtype == CTTS_BF16 && g->w.kv_dtype != CTTS_BF16) return fail("perf mode needs a bf16 KV cache");
  if (s->infer_text) {
    if (!g->w.emb_text || !g->w.head_text || g->w.n_text <= 0 || g->w.n_text > NTEXT_MAX) return fail("text head/embedding not loaded");
    if (s->pow_table) return fail("refine-text mode does not support a repetition penalty (the reference's own processor mis-broadcasts there)");
    if (s->eos < 0 || s->eos >= g->w.n_text) return fail("bad eos for text mode");
  }
  return 0;
}

// Device-side compaction of the decode batch: the caller passes a writable `n_active` scalar and NO row_map; the first kernel
// of every step then ranks the utterances whose finish flag is 0 (ascending slot), writes the map and the live count, and the
// step computes exactly those rows.  (With a host row_map -- slot pools -- the host stays in charge of both.)
static bool dev_compact(const ctts_gpt* g, const ctts_gen_state* s) {
  return s->row_map == nullptr && s->n_active != nullptr && g->skip_finished;
}

// the 20-layer body + heads + sampling over M = B * q_per_b rows
static bool prefill_compacts(const ctts_gpt* g, const ctts_gen_state* s) {   // the prompt pass of this call runs over the valid tokens only
  return g->w.weight_dtype != CTTS_BF16 && g->dec_x3 && !s->proj_exact && g->pre_x3 && g->pre_compact && s->row_map == nullptr &&
         s->prefill_valid_rows > 0 && s->prefill_valid_rows < s->B * s->T;
}

static int run_step(ctts_gpt* g, const ctts_gen_state* s, int q_per_b, hipStream_t st, bool prof_ok, int slot0 = 0, bool heads = true,
                    int ws_T = 0, int rows = 0, int pre_rows = 0, const EmbedNext* next = nullptr) {
  const GptWs ws = carve(s->workspace, s->B, ws_T ? ws_T : s->T);
  // `rows` (decode only): a bound on the live compact rows this step can have -- grids and M are sized for it instead of B (buffer
  // geometry, the KV cache's batch stride above all, stays B's)
  // pre_rows (whole-prompt pass, "f32x3"): the compact rows of the valid prompt tokens (prefill_compact_k wrote ws.x, ws.desc, ws.row_map = last_row)
  const int B = s->B, Bh = (q_per_b == 1 && rows > 0 && rows < s->B) ? rows : s->B, M = q_per_b == 1 ? Bh : (pre_rows > 0 ? pre_rows : B * q_per_b);
  const int cmax = s->cap ? s->cap : s->T + s->max_new;
  const int wt = g->w.weight_dtype, kt = g->w.kv_dtype;
  const size_t kv_layer = (size_t)(s->kv_batch ? s->kv_batch : B) * NHEAD * cmax * HDIM * (kt == CTTS_BF16 ? 2 : 4);
  const bool dec = q_per_b == 1;
  // decode: compact row -> slot (see GptRowMap) -- the host's map, or the one the step's first kernel derives from the finish
  // flags (device-side compaction); prefill: row group -> slot of a pool (or null)
  const int32_t* rmap = (dec && dev_compact(g, s)) ? ws.row_map : s->row_map;
  // CTTS_SKIP_FINISHED=0 with a caller that left the compaction to the device (row_map == NULL): nobody writes *n_active then,
  // so it must not be read -- every row steps, like the reference (gpt.py:512-518,592)
  const bool no_compact = dec && !g->skip_finished && s->row_map == nullptr;
  const int32_t* nact = (dec && !no_compact) ? s->n_active : nullptr;
  GptRowMap rm{q_per_b, s->len, s->kv_start, rmap, nact, (dec && g->skip_finished) ? s->finish : nullptr, nullptr, nullptr, nullptr, 0, slot0, 0};
  const bool fast = wt == CTTS_BF16;  // perf mode: bf16 activations, RMSNorm gain folded into wqkv / wgu by the loader
  const bool packed = fast && dec && g->dec_packed;   // decode step on fragment-packed operands (decode.hip)
  if (dec || pre_rows > 0) rm.desc = ws.desc;          // written by the embedding kernel at the head of the step / by prefill_compact_k
  if (dec && dev_compact(g, s)) rm.desc_covers_all = 1; // ... for every row of the grid (absent rows: b = -1)
  if (packed && g->n_cu > 0) { rm.sp_part = ws.att_part; rm.sp_cnt = ws.att_cnt; rm.sp_cus = g->n_cu; }
  if (packed) { const char* e = getenv("CTTS_ATT_DBG_PTR"); if (e) rm.dbg = (long long*)strtoull(e, nullptr, 0); }   // probes only
  // decode on packed operands: final RMSNorm + hidden capture + heads are ONE launch (decode32.hip gemm_dec32_fnorm16_k; the residual
  // stream reaches it in the packed f32 order: parity mode keeps it that way anyway, perf mode has the last down_proj write it)
  // (the fused launch reads the last down_proj's packed f32 rows: written by the x3 step and by the packed f32 step, not by the row-major one)
  const bool packed32_ = !fast && dec && ((g->dec_x3 && !s->proj_exact) || g->dec_packed32);
  const bool fuse_fnorm = dec && heads && g->fnorm_fuse && g->heads_packed && (packed || packed32_) &&
                          (s->infer_text ? g->w.head_text_pk : g->w.heads_pk) != nullptr;
  for (int l = 0; packed && l < g->w.n_layers; ++l) {
    void* kc = (char*)s->kcache + kv_layer * l;
    void* vc = (char*)s->vcache + kv_layer * l;
    DecGemmArgs d;
    memset(&d, 0, sizeof(d));
    d.M = M; d.eps = g->w.rms_eps; d.n_active = nact;
    d.force_nt = l < g->temporal_layers ? 1 : 0;   // A/B knob: the first N layers' weights with plain loads (candidates for the Infinity Cache)
#ifdef CTTS_PF_BUILD
    // cross-kernel weight prefetch (common.hpp): QKV -> this layer's o_proj; attention -> gate/up; gate/up -> down and the next layer's
    // QKV (last layer: the heads).  CTTS_PF is a bit mask (default 0 = off: every combination measured a net loss,
    // profiles/r3s_ab_prefetch.log): 1 QKV -> o_proj, 16 QKV -> gate/up, 2 attention -> gate/up
    const PfDesc pf_none{nullptr, 0, 0};
    const PfDesc pf_o{(const char*)g->wo_pk[l], 16u * HID * 2u, HID / 16u};
    const PfDesc pf_gu{(const char*)g->wgu_pk[l], 16u * HID * 2u, 2u * INTER / 16u};
    const PfDesc pf_d{(const char*)g->wd_pk[l], 16u * INTER * 2u, HID / 16u};
    const PfDesc pf_next = l + 1 < g->w.n_layers ? PfDesc{(const char*)g->wqkv_pk[l + 1], 16u * HID * 2u, 3u * HID / 16u}
                           : (fuse_fnorm && !s->infer_text) ? PfDesc{(const char*)g->w.heads_pk, 16u * HID * 4u, (NVQ * NAUDIO + 15u) / 16u} : pf_none;
    d.pf[0] = (g->pf_mask & 1) ? pf_o : pf_none; d.pf[1] = (g->pf_mask & 16) ? pf_gu : pf_none;
    rm.pf = (g->pf_mask & 2) ? pf_gu : pf_none;
#endif
    // RMSNorm scale + QKV + RoPE + KV append
    d.Ap = ws.xp; d.Wp = (const uint16_t*)g->wqkv_pk[l]; d.N = 3 * HID; d.K = HID; d.ssq_in = ws.ssq; d.epi = FEPI_QKV_ROPE;
    d.C32 = ws.qkv; d.ldc = 3 * HID; d.desc = ws.desc; d.cos_t = g->w.rope_cos; d.sin_t = g->w.rope_sin;
    d.kc = (uint16_t*)kc; d.vc = (uint16_t*)vc; d.cmax = cmax;
    const bool fuse_qa = g->qkv_att && M <= 64 && kv_layer < ((size_t)1 << 31) && g->w.n_layers <= QA_LAYERS_MAX && g->n_cu == 0 && g->pf_mask == 0 && !(g->att_oproj) && rm.dbg == nullptr;
    bool need_o = true;   // o_proj + residual as its own launch
    if (fuse_qa) {   // QKV + attention: one launch (gpt.hip qkv_attention_k)
      d.ho_flag = ws.qa_flag + (size_t)l * NHEAD * QA_STRIDE; rm.qf_flag = d.ho_flag; d.rope_cs = ws.rope_cs; rm.qf_kv_bytes = (int)kv_layer;
      Prof p(g, 3, st, prof_ok);
      CK(launch_qkv_attention(d, kc, vc, cmax, ws.aop, rm, M, st));
    } else {
      { Prof p(g, 1, st, prof_ok); CK(launch_gemm_dec(d, st)); }
      if (g->att_oproj && g->n_cu == 0) {   // attention + o_proj + residual: one launch (gpt.hip attention_k<OPJ>)
        rm.wo_h = (const uint16_t*)g->wo_hd[l]; rm.op_part = ws.op_part; rm.op_part_bytes = (int)ws.op_part_bytes; rm.op_cnt = ws.op_cnt;
        rm.x32 = ws.x; rm.xp = ws.xp; rm.ssq = ws.ssq;
        Prof p(g, 3, st, prof_ok);
        CK(launch_attention_oproj(ws.qkv, kc, vc, cmax, rm, M, st));
        need_o = false;
      } else {
        Prof p(g, 3, st, prof_ok);
        CK(launch_attention(ws.qkv, kc, vc, kt, cmax, ws.aop, 2, rm, M, st));
      }
    }
    if (need_o) {
      d.Ap = ws.aop; d.Wp = (const uint16_t*)g->wo_pk[l]; d.N = HID; d.ssq_in = nullptr; d.epi = FEPI_RES; d.C32 = ws.x; d.ldc = HID;
      d.Cp = ws.xp; d.kch_out = HID / 32; d.ssq_out = ws.ssq; d.ho_flag = nullptr;
      Prof p(g, 4, st, prof_ok);
      CK(launch_gemm_dec(d, st));
    }
    d.Ap = ws.xp; d.Wp = (const uint16_t*)g->wgu_pk[l]; d.N = INTER; d.ssq_in = ws.ssq; d.epi = FEPI_SILU; d.C32 = nullptr;
    d.Cp = ws.actp; d.kch_out = INTER / 32; d.ssq_out = nullptr;
#ifdef CTTS_PF_BUILD
    d.pf[0] = pf_none; d.pf[1] = pf_none; (void)pf_d; (void)pf_next;   // (gate/up has no auxiliary wave: see decode.hip)
#endif
    { Prof p(g, 5, st, prof_ok); CK(launch_gemm_dec(d, st)); }
    d.Ap = ws.actp; d.Wp = (const uint16_t*)g->wd_pk[l]; d.N = HID; d.K = INTER; d.ssq_in = nullptr; d.epi = FEPI_RES; d.C32 = ws.x;
    d.ldc = HID; d.Cp = ws.xp; d.kch_out = HID / 32; d.ssq_out = ws.ssq;
    d.Cp32 = (fuse_fnorm && l == g->w.n_layers - 1) ? ws.xp32 : nullptr;   // operand of the fused final-norm + heads launch
    { Prof p(g, 6, st, prof_ok); CK(launch_gemm_dec(d, st)); }
  }
  for (int l = 0; fast && !packed && l < g->w.n_layers; ++l) {
    void* kc = (char*)s->kcache + kv_layer * l;
    void* vc = (char*)s->vcache + kv_layer * l;
    uint16_t* aob = (uint16_t*)ws.ao;
    uint16_t* actb = (uint16_t*)ws.act;
    FastGemmArgs f;
    memset(&f, 0, sizeof(f));
    f.M = M; f.eps = g->w.rms_eps; f.n_active = nact; f.row_map = rmap; f.slot0 = slot0;
    // RMSNorm + QKV + RoPE + KV append in one launch (q/k weight rows are permuted by the loader)
    f.A = ws.xb; f.lda = HID; f.W = (const uint16_t*)g->wqkv[l]; f.N = 3 * HID; f.K = HID; f.ssq_in = ws.ssq; f.epi = FEPI_QKV_ROPE;
    f.C32 = ws.qkv; f.ldc = 3 * HID;
    f.q_per_b = q_per_b; f.len = s->len; f.kv_start = s->kv_start; f.cos_t = g->w.rope_cos; f.sin_t = g->w.rope_sin;
    f.kc = (uint16_t*)kc; f.vc = (uint16_t*)vc; f.cmax = cmax;
    { Prof p(g, 1, st, prof_ok); CK(launch_gemm_fast(f, st)); }
    { Prof p(g, 3, st, prof_ok); CK(launch_attention(ws.qkv, kc, vc, kt, cmax, aob, 1, rm, M, st)); }
    f.A = aob; f.W = (const uint16_t*)g->wo[l]; f.N = HID; f.K = HID; f.ssq_in = nullptr; f.epi = FEPI_RES; f.C32 = ws.x; f.ldc = HID;
    f.Cb = ws.xb; f.ldcb = HID; f.ssq_out = ws.ssq;
    { Prof p(g, 4, st, prof_ok); CK(launch_gemm_fast(f, st)); }
    f.A = ws.xb; f.W = (const uint16_t*)g->wgu[l]; f.N = INTER; f.K = HID; f.ssq_in = ws.ssq; f.epi = FEPI_SILU; f.C32 = nullptr;
    f.Cb = actb; f.ldcb = INTER; f.ssq_out = nullptr;
    { Prof p(g, 5, st, prof_ok); CK(launch_gemm_fast(f, st)); }
    f.A = actb; f.lda = INTER; f.W = (const uint16_t*)g->wd[l]; f.N = HID; f.K = INTER; f.ssq_in = nullptr; f.epi = FEPI_RES;
    f.C32 = ws.x; f.ldc = HID; f.Cb = ws.xb; f.ldcb = HID; f.ssq_out = ws.ssq;
    { Prof p(g, 6, st, prof_ok); CK(launch_gemm_fast(f, st)); }
  }
  // parity mode decode step on SPLIT-bf16 operands (decode32x.hip): hi | lo planes in the buffers the f32 kernels use for their packed
  // operands (same bytes: 2 planes x 2 B), the heads' packed f32 operand in ws.hfinp
  const bool x3 = !fast && dec && g->dec_x3 && !s->proj_exact;
  const size_t Bp16 = ((size_t)M + 15) / 16 * 16;
  for (int l = 0; x3 && l < g->w.n_layers; ++l) {
    void* kc = (char*)s->kcache + kv_layer * l;
    void* vc = (char*)s->vcache + kv_layer * l;
    uint16_t* xpx = reinterpret_cast<uint16_t*>(ws.xp32);
    uint16_t* aopx = reinterpret_cast<uint16_t*>(ws.aop32);
    uint16_t* actx = reinterpret_cast<uint16_t*>(ws.actp32);
    Dec32xArgs d;
    memset(&d, 0, sizeof(d));
    d.M = M; d.eps = g->w.rms_eps; d.n_active = nact;
    // RMSNorm (gain folded into the weights, 1 / rms on the accumulator) + QKV + RoPE + KV append
    d.Ap = xpx; d.a_plane = Bp16 * HID; d.Wp = (const uint16_t*)g->wqkv_x3[l]; d.w_plane = (size_t)3 * HID * HID; d.N = 3 * HID; d.K = HID;
    d.rms = 1; d.X = ws.x; d.ldx = HID; d.ssq_in = ws.ssq; d.rope_cs = ws.rope_cs; d.epi = D32_EPI_QKV_ROPE; d.C = ws.qkv; d.ldc = 3 * HID; d.desc = ws.desc;
    d.cos_t = g->w.rope_cos; d.sin_t = g->w.rope_sin; d.kc = (float*)kc; d.vc = (float*)vc; d.cmax = cmax;
    { Prof p(g, 1, st, prof_ok); CK(launch_gemm_dec32x(d, st)); }
    rm.x3_plane = Bp16 * HID;
    { Prof p(g, 3, st, prof_ok); CK(launch_attention(ws.qkv, kc, vc, kt, cmax, aopx, 4, rm, M, st)); }
    // o_proj + residual
    d.Ap = aopx; d.a_plane = Bp16 * HID; d.Wp = (const uint16_t*)g->wo_x3[l]; d.w_plane = (size_t)HID * HID; d.N = HID; d.rms = 0; d.X = nullptr;
    d.ssq_in = nullptr; d.ssq_out = ws.ssq;
    d.epi = EPI_RES; d.C = ws.x; d.ldc = HID; d.res = ws.x; d.ldr = HID; d.Cp = xpx; d.c_plane = Bp16 * HID; d.kch_out = HID / 32;
    { Prof p(g, 4, st, prof_ok); CK(launch_gemm_dec32x(d, st)); }
    // RMSNorm + gate/up + SiLU*up
    d.Ap = xpx; d.a_plane = Bp16 * HID; d.Wp = (const uint16_t*)g->wgu_x3[l]; d.w_plane = (size_t)2 * INTER * HID; d.N = INTER; d.rms = 1; d.X = ws.x;
    d.ssq_in = ws.ssq; d.ssq_out = nullptr;
    d.epi = EPI_SILU_MUL; d.C = nullptr; d.res = nullptr; d.Cp = actx; d.c_plane = Bp16 * INTER; d.kch_out = INTER / 32;
    { Prof p(g, 5, st, prof_ok); CK(launch_gemm_dec32x(d, st)); }
    // down_proj + residual (last layer: also the packed f32 rows the fused final-norm + heads launch reads)
    d.Ap = actx; d.a_plane = Bp16 * INTER; d.Wp = (const uint16_t*)g->wd_x3[l]; d.w_plane = (size_t)HID * INTER; d.N = HID; d.K = INTER; d.rms = 0;
    d.ssq_in = nullptr; d.ssq_out = ws.ssq;
    d.X = nullptr; d.epi = EPI_RES; d.C = ws.x; d.ldc = HID; d.res = ws.x; d.ldr = HID; d.Cp = xpx; d.c_plane = Bp16 * HID; d.kch_out = HID / 32;
    d.Cp32 = (fuse_fnorm && l == g->w.n_layers - 1) ? ws.hfinp : nullptr; d.kch32_out = HID / 16;
    { Prof p(g, 6, st, prof_ok); CK(launch_gemm_dec32x(d, st)); }
  }
  const bool packed32 = !fast && dec && g->dec_packed32 && !x3;   // parity mode decode step on fragment-packed f32 operands (decode32.hip)
  for (int l = 0; packed32 && l < g->w.n_layers; ++l) {
    void* kc = (char*)s->kcache + kv_layer * l;
    void* vc = (char*)s->vcache + kv_layer * l;
    Dec32Args d;
    memset(&d, 0, sizeof(d));
    d.M = M; d.eps = g->w.rms_eps; d.n_active = nact;
    // RMSNorm + QKV
    d.Ap = ws.xp32; d.Wp = (const float*)g->wqkv_pk[l]; d.N = 3 * HID; d.K = HID; d.X = ws.x; d.ldx = HID; d.norm_w = g->ln1[l];
    // ... + RoPE + KV append in the epilogue (q / k weight rows of the packed copy are permuted by the loader)
    d.epi = D32_EPI_QKV_ROPE; d.C = ws.qkv; d.ldc = 3 * HID; d.desc = ws.desc; d.cos_t = g->w.rope_cos; d.sin_t = g->w.rope_sin;
    d.kc = (float*)kc; d.vc = (float*)vc; d.cmax = cmax;
    { Prof p(g, 1, st, prof_ok); CK(launch_gemm_dec32(d, st)); }
    { Prof p(g, 3, st, prof_ok); CK(launch_attention(ws.qkv, kc, vc, kt, cmax, ws.aop32, 3, rm, M, st)); }
    // o_proj + residual
    d.Ap = ws.aop32; d.Wp = (const float*)g->wo_pk[l]; d.N = HID; d.X = nullptr; d.norm_w = nullptr; d.epi = EPI_RES; d.C = ws.x; d.ldc = HID;
    d.res = ws.x; d.ldr = HID; d.Cp = ws.xp32; d.kch_out = HID / 16;
    { Prof p(g, 4, st, prof_ok); CK(launch_gemm_dec32(d, st)); }
    // RMSNorm + gate/up + SiLU*up
    d.Ap = ws.xp32; d.Wp = (const float*)g->wgu_pk[l]; d.N = INTER; d.X = ws.x; d.norm_w = g->ln2[l]; d.epi = EPI_SILU_MUL; d.C = nullptr;
    d.res = nullptr; d.Cp = ws.actp32; d.kch_out = INTER / 16;
    { Prof p(g, 5, st, prof_ok); CK(launch_gemm_dec32(d, st)); }
    // down_proj + residual
    d.Ap = ws.actp32; d.Wp = (const float*)g->wd_pk[l]; d.N = HID; d.K = INTER; d.X = nullptr; d.norm_w = nullptr; d.epi = EPI_RES;
    d.C = ws.x; d.ldc = HID; d.res = ws.x; d.ldr = HID; d.Cp = ws.xp32; d.kch_out = HID / 16;
    { Prof p(g, 6, st, prof_ok); CK(launch_gemm_dec32(d, st)); }
  }
  // parity mode PREFILL on packed operands (round 3, prefill32.hip): the prompt rows are packed once, every projection is a
  // register-blocked launch over fragment-order operands (one wave = 32 x 32 outputs x the 4 k-chunk classes; same chunk order per
  // class, same ((s0+s1)+s2)+s3, 1/rms per row from wave_row_rstd like gemm_skinny_k), RoPE + KV append run in the QKV epilogue from
  // per-row descriptors, attention writes its output packed.  Bit-identical to the row-major prefill (CTTS_PRE32_PACKED=0), which
  // the goldens were made with.  (The decode kernels' generic 64-row body was tried first: no faster than row-major at this M.)
  // "f32x3" mode (round 6): the prompt rows go through the row-major path below with its four projections on the LDS-tiled split-bf16
  // GEMM (prefill32x.hip) -- 3 bf16 MFMAs per product instead of f32 MFMA at a sixteenth of the rate; an exact call keeps the f32 kernels
  const bool pre_x3 = !fast && !dec && g->dec_x3 && !s->proj_exact && g->pre_x3;
  const bool pre32 = !fast && !dec && g->dec_packed32 && g->pre32_packed && !pre_x3;
  if (pre32) {
    CK(launch_prefill_prep32(ws.x, ws.xp32, ws.desc, q_per_b, slot0, s->kv_start, s->row_map, M, st));
  }
  for (int l = 0; pre32 && l < g->w.n_layers; ++l) {
    void* kc = (char*)s->kcache + kv_layer * l;
    void* vc = (char*)s->vcache + kv_layer * l;
    Dec32Args d;
    memset(&d, 0, sizeof(d));
    d.M = M; d.eps = g->w.rms_eps; d.n_active = nullptr;
    CK(launch_rows_rstd32(ws.x, HID, M, g->w.rms_eps, ws.rstd, st));
    d.Ap = ws.xp32; d.Wp = (const float*)g->wqkv_pk[l]; d.N = 3 * HID; d.K = HID; d.norm_w = g->ln1[l];
    d.epi = D32_EPI_QKV_ROPE; d.C = ws.qkv; d.ldc = 3 * HID; d.desc = ws.desc; d.cos_t = g->w.rope_cos; d.sin_t = g->w.rope_sin;
    d.kc = (float*)kc; d.vc = (float*)vc; d.cmax = cmax;
    CK(launch_gemm_pre32(d, ws.rstd, st));
    CK(launch_attention(ws.qkv, kc, vc, kt, cmax, ws.aop32, 3, rm, M, st));
    d.Ap = ws.aop32; d.Wp = (const float*)g->wo_pk[l]; d.N = HID; d.norm_w = nullptr; d.epi = EPI_RES; d.C = ws.x; d.ldc = HID;
    d.res = ws.x; d.ldr = HID; d.Cp = ws.xp32; d.kch_out = HID / 16;
    CK(launch_gemm_pre32(d, nullptr, st));
    CK(launch_rows_rstd32(ws.x, HID, M, g->w.rms_eps, ws.rstd, st));
    d.Ap = ws.xp32; d.Wp = (const float*)g->wgu_pk[l]; d.N = INTER; d.norm_w = g->ln2[l]; d.epi = EPI_SILU_MUL; d.C = nullptr;
    d.res = nullptr; d.Cp = ws.actp32; d.kch_out = INTER / 16;
    CK(launch_gemm_pre32(d, ws.rstd, st));
    d.Ap = ws.actp32; d.Wp = (const float*)g->wd_pk[l]; d.N = HID; d.K = INTER; d.norm_w = nullptr; d.epi = EPI_RES;
    d.C = ws.x; d.ldc = HID; d.res = ws.x; d.ldr = HID; d.Cp = ws.xp32; d.kch_out = HID / 16;
    CK(launch_gemm_pre32(d, nullptr, st));
  }
  for (int l = 0; !fast && !packed32 && !pre32 && !x3 && l < g->w.n_layers; ++l) {
    void* kc = (char*)s->kcache + kv_layer * l;
    void* vc = (char*)s->vcache + kv_layer * l;
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.taps = 1;
    a.n_active = nact;
    // RMSNorm + QKV
    a.A = ws.x; a.lda = HID; a.W = g->wqkv[l]; a.C = ws.qkv; a.ldc = 3 * HID; a.M = M; a.N = 3 * HID; a.K = HID; a.wt = wt;
    a.epi = EPI_STORE; a.norm_w = g->ln1[l]; a.eps = g->w.rms_eps;
    if (pre_x3) { CK(launch_rows_rstd32(ws.x, HID, M, g->w.rms_eps, ws.rstd, st)); CK(launch_gemm_pre_x3(a, ws.rstd, st)); }
    else { Prof p(g, 1, st, prof_ok); CK(launch_gemm_skinny(a, st)); }
    { Prof p(g, 2, st, prof_ok); CK(launch_rope_append(ws.qkv, kc, vc, kt, cmax, g->w.rope_cos, g->w.rope_sin, rm, M, st)); }
    { Prof p(g, 3, st, prof_ok); CK(launch_attention(ws.qkv, kc, vc, kt, cmax, ws.ao, 0, rm, M, st)); }
    // o_proj + residual
    a.A = ws.ao; a.lda = HID; a.W = g->wo[l]; a.C = ws.x; a.ldc = HID; a.N = HID; a.K = HID; a.epi = EPI_RES; a.norm_w = nullptr;
    a.res = ws.x; a.ldr = HID;
    if (pre_x3) CK(launch_gemm_pre_x3(a, nullptr, st));
    else { Prof p(g, 4, st, prof_ok); CK(launch_gemm_skinny(a, st)); }
    // RMSNorm + gate/up + SiLU*up
    a.A = ws.x; a.W = g->wgu[l]; a.C = ws.act; a.ldc = INTER; a.N = INTER; a.K = HID; a.epi = EPI_SILU_MUL; a.norm_w = g->ln2[l];
    a.res = nullptr;
    if (pre_x3) { CK(launch_rows_rstd32(ws.x, HID, M, g->w.rms_eps, ws.rstd, st)); CK(launch_gemm_pre_x3(a, ws.rstd, st)); }
    else { Prof p(g, 5, st, prof_ok); CK(launch_gemm_skinny(a, st)); }
    // down_proj + residual
    a.A = ws.act; a.lda = INTER; a.W = g->wd[l]; a.C = ws.x; a.ldc = HID; a.N = HID; a.K = INTER; a.epi = EPI_RES; a.norm_w = nullptr;
    a.res = ws.x; a.ldr = HID;
    if (pre_x3) CK(launch_gemm_pre_x3(a, nullptr, st));
    else { Prof p(g, 6, st, prof_ok); CK(launch_gemm_skinny(a, st)); }
  }
  if (!heads) return 0;   // a prompt chunk that is not the last one: its K/V rows are in the cache, nothing is sampled
  if (fuse_fnorm) {
    const int nlog = s->infer_text ? g->w.n_text : NVQ * NAUDIO;
    Dec32Args d;
    memset(&d, 0, sizeof(d));
    d.Ap = x3 ? ws.hfinp : ws.xp32; d.Wp = s->infer_text ? g->w.head_text_pk : g->w.heads_pk; d.M = Bh; d.N = (nlog + 15) / 16 * 16; d.K = HID; d.n_active = nact;
    d.epi = EPI_STORE; d.C = ws.logits; d.ldc = nlog; d.n_cols = nlog;
    d.fnorm = 1; d.norm_w = g->w.norm; d.eps = g->w.rms_eps; d.desc = ws.desc; d.hid = s->hiddens; d.hid_cap = s->hid_cap ? s->hid_cap : s->max_new;
    d.T = s->T; d.prompt_len = s->prompt_len;
    Prof p(g, 8, st, prof_ok);
    CK(launch_gemm_dec32(d, st));
  } else {
  { Prof p(g, 7, st, prof_ok);
    CK(launch_final_norm(ws.x, q_per_b, g->w.norm, g->w.rms_eps, ws.hfin, s->hiddens, s->hid_cap ? s->hid_cap : s->max_new, s->len, s->T, Bh, rmap,
                         nact, s->prompt_len, st, ws.hfinp, pre_rows > 0 ? ws.row_map : nullptr)); }
  {
    const int nlog = s->infer_text ? g->w.n_text : NVQ * NAUDIO;   // gpt.py:439-440 text head | :441-454 four code heads
    const float* hpk = s->infer_text ? g->w.head_text_pk : g->w.heads_pk;
    if (g->heads_packed && hpk != nullptr) {   // same arithmetic as the row-major kernel below, operands in fragment order
      Dec32Args d;
      memset(&d, 0, sizeof(d));
      d.Ap = ws.hfinp; d.Wp = hpk; d.M = Bh; d.N = (nlog + 15) / 16 * 16; d.K = HID; d.n_active = nact; d.epi = EPI_STORE;
      d.C = ws.logits; d.ldc = nlog; d.n_cols = nlog;
      Prof p(g, 8, st, prof_ok);
      CK(launch_gemm_dec32(d, st));
    } else {
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.taps = 1;
    a.A = ws.hfin; a.lda = HID; a.W = s->infer_text ? g->w.head_text : g->w.heads; a.C = ws.logits; a.ldc = nlog; a.M = Bh; a.N = nlog;
    a.K = HID; a.wt = WT_F32; a.epi = EPI_STORE; a.n_active = nact;
    Prof p(g, 8, st, prof_ok);
    CK(launch_gemm_skinny(a, st));
    }
  }
  }
  {
    Prof p(g, 9, st, prof_ok);
    SampleArgs sa = make_sample_args(s, ws.logits);
    sa.B = Bh;   // (the grid; q_rows keeps the batch geometry)
    sa.row_map = rmap; sa.n_active = nact;
    if (dec && dev_compact(g, s)) sa.desc = ws.desc;   // one load instead of the n_active -> row_map -> len chain
    if (next != nullptr && sa.desc != nullptr && !s->infer_text) sa.next = *next;   // the next step's first kernel, folded into this launch
    if (s->infer_text) CK(launch_sample_text(sa, g->w.n_text, st));
    else CK(launch_sample(sa, st));
  }
  return 0;
}

extern "C" int ctts_gpt_prefill(ctts_gpt* g, const ctts_gen_state* s, const float* emb, void* stream) {
  if (check_state(g, s)) return -1;
  CttsDeviceGuard dg(stream);
  hipStream_t st = (hipStream_t)stream;
  const GptWs ws = carve(s->workspace, s->B, s->T);
  CK(hipMemsetAsync(ws.att_cnt, 0, cnt_bytes(s->B), st));   // arrival counters of the attention split
  if (prefill_compacts(g, s)) {
    // the prompt pass over the valid prompt tokens only: their embeddings gathered into consecutive rows, descriptors per row
    CK(launch_prefill_compact(emb, ws.x, ws.desc, ws.row_map, s->B, s->T, s->kv_start, st));
    if (run_step(g, s, s->T, st, false, 0, true, 0, 0, s->prefill_valid_rows)) return -1;
    CK(hipMemsetAsync(carve(s->workspace, s->B, 1).att_cnt, 0, cnt_bytes(s->B), st));
    return 0;
  }
  CK(hipMemcpyAsync(ws.x, emb, (size_t)s->B * s->T * HID * 4, hipMemcpyDeviceToDevice, st));
  if (g->w.weight_dtype == CTTS_BF16) CK(launch_rows_prep(ws.x, ws.xb, ws.ssq, s->B * s->T, st));
  if (run_step(g, s, s->T, st, false)) return -1;
  // arrival counters of the attention split at the place the DECODE steps carve them (one row per utterance), once the
  // prompt-sized buffers above are dead
  CK(hipMemsetAsync(carve(s->workspace, s->B, 1).att_cnt, 0, cnt_bytes(s->B), st));
  return 0;
}

extern "C" int ctts_gpt_prefill_chunk(ctts_gpt* g, const ctts_gen_state* s, const float* emb_chunk, int32_t t0, int32_t tc, int32_t last,
                                      void* stream) {
  if (check_state(g, s, tc > 0 ? tc : 1)) return -1;
  if (t0 < 0 || tc <= 0 || t0 + tc > s->T || (last && t0 + tc != s->T)) return fail("ctts_gpt_prefill_chunk: bad chunk [%d, %d) of a %d-slot prompt", t0, t0 + tc, s->T);
  if (s->workspace_bytes < ctts_gpt_workspace_bytes(s->B, tc)) return fail("workspace too small for the chunk");
  CttsDeviceGuard dg(stream);
  hipStream_t st = (hipStream_t)stream;
  const GptWs ws = carve(s->workspace, s->B, tc);
  CK(hipMemcpyAsync(ws.x, emb_chunk, (size_t)s->B * tc * HID * 4, hipMemcpyDeviceToDevice, st));
  if (g->w.weight_dtype == CTTS_BF16) CK(launch_rows_prep(ws.x, ws.xb, ws.ssq, s->B * tc, st));
  if (run_step(g, s, tc, st, false, t0, last != 0, tc)) return -1;
  // the decode steps carve the workspace for one row per utterance: zero THEIR attention-split arrival counters once the
  // chunk-sized buffers above are dead
  if (last) CK(hipMemsetAsync(carve(s->workspace, s->B, 1).att_cnt, 0, cnt_bytes(s->B), st));
  return 0;
}

// The decode step carves the workspace for ONE row per utterance (ctts_gpt_workspace_bytes(B, 1)) whatever the prompt length was: a
// caller that prefills in chunks of tc slots needs max(bytes(B, tc), bytes(B, 1)), not bytes(B, T).  Nothing in the workspace
// survives from the prefill into the decode steps (all generation state lives in ctts_gen_state's own arrays).
// skip_embed / fold_next (multi-step graphs, env CTTS_EMBED_FOLD=1): step i > 0 of the group finds its input rows written by step i - 1's
// sampling launch (EmbedNext), step i < n - 1 writes them for step i + 1
static int decode_body(ctts_gpt* g, const ctts_gen_state* s, hipStream_t st, bool prof_ok, int rows = 0, bool skip_embed = false,
                       bool fold_next = false) {
  const GptWs ws = carve(s->workspace, s->B, 1);
  EmbedNext nx;
  memset(&nx, 0, sizeof(nx));
  { Prof p(g, 0, st, prof_ok); const bool fast = g->w.weight_dtype == CTTS_BF16;
    const bool packed = fast && g->dec_packed;
    const bool dc = dev_compact(g, s);
    const bool x3 = !fast && g->dec_x3 && !s->proj_exact;   // split-bf16 parity mode: the residual rows as hi | lo planes (decode32x.hip)
    StepPrep sp{ws.desc, s->kv_start, g->skip_finished ? s->finish : nullptr, (packed || x3) ? 1 : 0, (!fast && g->dec_packed32 && !x3) ? ws.xp32 : nullptr,
                dc ? ws.row_map : nullptr,
                dc ? const_cast<int32_t*>(s->n_active) : nullptr, dc ? s->order : nullptr,
                ((packed && g->qkv_att && g->w.n_layers <= QA_LAYERS_MAX) || x3) ? ws.rope_cs : nullptr, g->w.rope_cos, g->w.rope_sin,
                // the arrival words are sized for QA_LAYERS_MAX layers: the same predicate as `fuse_qa` in run_step (a deeper model
                // never takes the fused launch and must not zero past the carve either)
                (packed && g->qkv_att && g->w.n_layers <= QA_LAYERS_MAX) ? ws.qa_flag : nullptr, g->w.n_layers * NHEAD, QA_STRIDE};
    const int32_t* nact0 = (!g->skip_finished && s->row_map == nullptr) ? nullptr : s->n_active;
    uint16_t* xb = fast ? (packed ? ws.xp : ws.xb) : x3 ? reinterpret_cast<uint16_t*>(ws.xp32) : nullptr;
    if (x3) sp.xb_lo_plane = ((size_t)((rows > 0 && rows < s->B) ? rows : s->B) + 15) / 16 * 16 * HID;   // = run_step's plane stride for this bound
    const bool foldable = dc && !s->infer_text && sp.zero_p == nullptr;
    if (fold_next && foldable) { nx.emb_code = g->w.emb_code; nx.x = ws.x; nx.xb = xb; nx.ssq = (fast || x3) ? ws.ssq : nullptr; nx.sp = sp; }
    if (skip_embed && foldable) { /* written by the previous step's sampling launch */ }
    else if (s->infer_text)
      CK(launch_embed_text(g->w.emb_text, g->w.n_text, s->ids_buf, s->cap ? s->cap : s->T + s->max_new, s->len, ws.x, xb,
                           (fast || x3) ? ws.ssq : nullptr, s->B, s->row_map, nact0, st, &sp));
    else
      CK(launch_embed_codes(g->w.emb_code, s->ids_buf, s->cap ? s->cap : s->T + s->max_new, s->len, ws.x, xb, (fast || x3) ? ws.ssq : nullptr, s->B,
                            s->row_map, nact0, st, &sp)); }
  return run_step(g, s, 1, st, prof_ok, 0, true, 1, rows, 0, nx.x != nullptr ? &nx : nullptr);
}

extern "C" int ctts_gpt_decode_step(ctts_gpt* g, const ctts_gen_state* s, void* stream) {
  if (check_state(g, s, 1)) return -1;
  CttsDeviceGuard dg(stream);
  return decode_body(g, s, (hipStream_t)stream, true);
}

// captures decode_body (rows bound `rows`, 0 = the whole batch) as a 1-step graph and as a graph of g->multi_steps steps
static int capture_graphs(ctts_gpt* g, const ctts_gen_state* s, hipStream_t st, int rows, hipGraph_t* graph, hipGraphExec_t* exec, hipGraph_t* gm,
                          hipGraphExec_t* exec_multi) {
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  const int rc = decode_body(g, s, st, false, rows);
  hipGraph_t gr = nullptr;
  hipError_t e = hipStreamEndCapture(st, &gr);
  if (rc != 0) { if (gr) (void)hipGraphDestroy(gr); return -1; }
  if (e != hipSuccess) return fail("hipStreamEndCapture: %s", hipGetErrorString(e));
  *graph = gr;
  CK(hipGraphInstantiate(exec, gr, nullptr, nullptr, 0));
  if (g->multi_steps > 1) {
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    int rc2 = 0;
    for (int i = 0; i < g->multi_steps && rc2 == 0; ++i)
      rc2 = decode_body(g, s, st, false, rows, g->embed_fold && i > 0, g->embed_fold && i + 1 < g->multi_steps);
    hipGraph_t g2 = nullptr;
    hipError_t e3 = hipStreamEndCapture(st, &g2);
    if (rc2 != 0) { if (g2) (void)hipGraphDestroy(g2); return -1; }
    if (e3 != hipSuccess) return fail("hipStreamEndCapture (multi-step graph): %s", hipGetErrorString(e3));
    *gm = g2;
    CK(hipGraphInstantiate(exec_multi, g2, nullptr, nullptr, 0));
  }
  return 0;
}

extern "C" int ctts_gpt_graph_build(ctts_gpt* g, const ctts_gen_state* s, void* stream) {
  if (check_state(g, s, 1)) return -1;
  CttsDeviceGuard dg(stream);
  ctts_gpt_graph_destroy(g);
  hipStream_t st = (hipStream_t)stream;
  if (st == nullptr) return fail("graph capture needs a non-default stream");
  {  // the decode workspace may never have seen a prefill (slot pools prefill into their own): zero the arrival counters of the
     // attention split once, stream-ordered before anything the graph will run
    const GptWs ws = carve(s->workspace, s->B, 1);
    CK(hipMemsetAsync(ws.att_cnt, 0, cnt_bytes(s->B), st));
  }
  // default 8 steps per launch of the multi-step graph: +0.5 % on the C3 bench over one hipGraphLaunch per step
  // (profiles/r3b_ab_fnorm_graphsteps.log: 1325 -> 1331-1334 audio-s/s with 8, 1328-1331 with 16)
  { const char* e2 = getenv("CTTS_GRAPH_STEPS"); g->multi_steps = e2 ? atoi(e2) : 8; }
  if (g->multi_steps < 1 || g->multi_steps > 64) g->multi_steps = 1;
  return capture_graphs(g, s, st, 0, &g->graph, &g->exec, &g->graph_multi, &g->exec_multi);
}

// The same step for a bound on the live rows (see ctts_gpt::rows_graphs).  Needs ctts_gpt_graph_build of the same state first (it owns the
// geometry; ctts_gpt_graph_build / _destroy drop every bounded graph too).  rows >= B: nothing to build (the plain graph is that graph).
extern "C" int ctts_gpt_graph_build_rows(ctts_gpt* g, const ctts_gen_state* s, int32_t rows, void* stream) {
  if (check_state(g, s, 1)) return -1;
  if (!g->exec) return fail("ctts_gpt_graph_build_rows: build the plain graph first");
  if (rows <= 0) return fail("ctts_gpt_graph_build_rows: bad bound");
  if (rows >= s->B || g->rows_graphs.count(rows)) return 0;
  CttsDeviceGuard dg(stream);
  hipStream_t st = (hipStream_t)stream;
  if (st == nullptr) return fail("graph capture needs a non-default stream");
  ctts_gpt::RowsGraph rg;
  if (capture_graphs(g, s, st, rows, &rg.graph, &rg.exec, &rg.gm, &rg.exec_multi)) return -1;
  g->rows_graphs[rows] = rg;
  return 0;
}

extern "C" int ctts_gpt_graph_launch(ctts_gpt* g, int32_t n_steps, void* stream) {
  if (!g || !g->exec) return fail("no captured graph");
  CttsDeviceGuard dg(stream);
  int left = n_steps;
  while (g->exec_multi && left >= g->multi_steps) { CK(hipGraphLaunch(g->exec_multi, (hipStream_t)stream)); left -= g->multi_steps; }
  for (int i = 0; i < left; ++i) CK(hipGraphLaunch(g->exec, (hipStream_t)stream));
  return 0;
}

// n_steps replays of the step captured for the bound `rows` (the caller guarantees live rows <= rows for all of them; rows >= B or a
// bound that was never built: the plain graph)
extern "C" int ctts_gpt_graph_launch_rows(ctts_gpt* g, int32_t n_steps, int32_t rows, void* stream) {
  if (!g || !g->exec) return fail("no captured graph");
  auto it = g->rows_graphs.find(rows);
  if (rows <= 0 || it == g->rows_graphs.end()) return ctts_gpt_graph_launch(g, n_steps, stream);
  CttsDeviceGuard dg(stream);
  int left = n_steps;
  while (it->second.exec_multi && left >= g->multi_steps) { CK(hipGraphLaunch(it->second.exec_multi, (hipStream_t)stream)); left -= g->multi_steps; }
  for (int i = 0; i < left; ++i) CK(hipGraphLaunch(it->second.exec, (hipStream_t)stream));
  return 0;
}

extern "C" int ctts_gpt_profile_begin(ctts_gpt* g, int32_t tag, int32_t max_samples, int32_t stride) {
  if (!g || max_samples <= 0 || stride <= 0) return fail("bad profile args");
  while ((int)g->ev0.size() < max_samples) {
    hipEvent_t a, b;
    // hipEventDisableSystemFence: a default event makes the dispatch that signals it end with a system-scope release (L2
    // write-back towards the host) -- time that rocprofv3's own completion signals do not add to a kernel
    static int sysfence = -1;
    if (sysfence < 0) { const char* e = getenv("CTTS_PROF_SYSFENCE"); sysfence = (e && atoi(e) == 1) ? 1 : 0; }
    const unsigned fl = sysfence ? hipEventDefault : hipEventDisableSystemFence;
    CK(hipEventCreateWithFlags(&a, fl));
    CK(hipEventCreateWithFlags(&b, fl));
    g->ev0.push_back(a);
    g->ev1.push_back(b);
  }
  g->prof_tag = tag; g->prof_max = max_samples; g->prof_n = 0; g->prof_stride = stride; g->prof_seen = 0;
  return 0;
}

extern "C" int ctts_gpt_profile_samples(ctts_gpt* g, float* ms_out, int32_t cap, int32_t* n_samples) {
  if (!g || !ms_out || cap < 0) return fail("bad profile_samples args");
  const int n = g->prof_n < cap ? g->prof_n : cap;
  for (int i = 0; i < n; ++i) {
    CK(hipEventSynchronize(g->ev1[i]));
    CK(hipEventElapsedTime(&ms_out[i], g->ev0[i], g->ev1[i]));
  }
  if (n_samples) *n_samples = n;
  return 0;
}

extern "C" int ctts_gpt_profile_end(ctts_gpt* g, int32_t* n_samples, double* total_ms) {
  if (!g) return fail("null engine");
  double tot = 0.0;
  for (int i = 0; i < g->prof_n; ++i) {
    CK(hipEventSynchronize(g->ev1[i]));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, g->ev0[i], g->ev1[i]));
    tot += ms;
  }
  if (n_samples) *n_samples = g->prof_n;
  if (total_ms) *total_ms = tot;
  g->prof_tag = -1; g->prof_n = 0;
  return 0;
}

// ------------------------------------------------------------------------------------------------
struct ctts_codec {
  ctts_codec_weights w;
  std::vector<const float*> d[9], v[9];
  std::vector<const void*> dx[2], vx[2];   // pwconv1 / pwconv2 as pre-split fragment-order planes (codec_gemm.hip), or empty
  int x3p_min_rows = 1024;                 // frames from which the point-wise layers take the LDS-DMA kernel (env CTTS_X3P_MIN_ROWS, 0 = never).
                                           // Round 4: 12288 -> 1024 -- the first streamed window of a batch (64 x 144 = 9216 frames; 16 x 144 = 2304)
                                           // was on the register-staged tiles: TTFS p50 55.3 -> 50.4 ms at batch 64, 44.4 -> 40.1 ms at batch 16,
                                           // C5 total 252.7 -> 239.4 ms (profiles/r4o_ab_x3p_min_rows.log; 256 measures the same as 1024)
};

extern "C" int ctts_codec_create(ctts_codec** out, const ctts_codec_weights* w) {
  if (!out || !w) return fail("ctts_codec_create: bad arguments");
  ctts_codec* c = new ctts_codec();
  c->w = *w;
  const float* const* dsrc[9] = {w->d_dw_w, w->d_dw_b, w->d_ln_w, w->d_ln_b, w->d_pw1_w, w->d_pw1_b, w->d_pw2_w, w->d_pw2_b, w->d_gamma};
  const float* const* vsrc[9] = {w->v_dw_w, w->v_dw_b, w->v_ln_w, w->v_ln_b, w->v_pw1_w, w->v_pw1_b, w->v_pw2_w, w->v_pw2_b, w->v_gamma};
  for (int i = 0; i < 9; ++i) {
    c->d[i].assign(dsrc[i], dsrc[i] + w->n_dvae_blocks);
    c->v[i].assign(vsrc[i], vsrc[i] + w->n_vocos_blocks);
  }
  if (w->gemm_mode != 0 && w->gemm_mode != 1 && w->gemm_mode != 2) { delete c; return fail("ctts_codec_create: gemm_mode must be 0, 1 or 2"); }
  if (w->gemm_mode == 2 && !(w->d_pw1_x3p && w->d_pw2_x3p && w->v_pw1_x3p && w->v_pw2_x3p)) {
    delete c;
    return fail("ctts_codec_create: gemm_mode 2 needs the fp16 planes of the point-wise weights");
  }
  if (w->gemm_mode >= 1 && w->d_pw1_x3p && w->d_pw2_x3p && w->v_pw1_x3p && w->v_pw2_x3p) {
    c->dx[0].assign(w->d_pw1_x3p, w->d_pw1_x3p + w->n_dvae_blocks);
    c->dx[1].assign(w->d_pw2_x3p, w->d_pw2_x3p + w->n_dvae_blocks);
    c->vx[0].assign(w->v_pw1_x3p, w->v_pw1_x3p + w->n_vocos_blocks);
    c->vx[1].assign(w->v_pw2_x3p, w->v_pw2_x3p + w->n_vocos_blocks);
  }
  { const char* e = getenv("CTTS_X3P_MIN_ROWS"); if (e) c->x3p_min_rows = atoi(e); }
  *out = c;
  return 0;
}
extern "C" void ctts_codec_destroy(ctts_codec* c) { delete c; }

struct CodecWs {
  float *a, *b, *big, *mid, *frames;
  size_t bytes;
};
static CodecWs carve_codec(void* base, int B, int F) {
  const size_t R = ((size_t)B * F + 255) / 256 * 256;   // whole 256-row tiles: the packed planes of codec_gemm.hip are padded to them
  CodecWs w;
  size_t off = 0;
  char* p = (char*)base;
  w.a = (float*)(p + off); off += align_up(R * 512 * 4);
  w.b = (float*)(p + off); off += align_up(R * 512 * 4);
  w.big = (float*)(p + off); off += align_up(R * 2048 * 4);
  w.mid = (float*)(p + off); off += align_up(R * 384 * 4);
  w.frames = (float*)(p + off); off += align_up(R * 1024 * 4);
  w.bytes = off;
  return w;
}
extern "C" size_t ctts_codec_workspace_bytes(int32_t B, int32_t F) { return carve_codec(nullptr, B, F).bytes; }

static GemmArgs lin(const float* A, int lda, const float* W, float* C, int ldc, int M, int N, int K, int epi) {
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = lda; a.W = W; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.wt = WT_F32; a.epi = epi; a.taps = 1;
  return a;
}
// dense layer of the decoder: f32 MFMA tiles, or split-bf16 tiles when the weights were packed [N][Kp/32][2][32] bf16
static hipError_t dense(const ctts_codec* c, const GemmArgs& a, hipStream_t st);
static GemmArgs conv(const float* X, int cin, const float* W, float* C, int cout, int B, int F, int taps, int pad, int epi) {
  GemmArgs a = lin(X, cin, W, C, cout, B * F, cout, taps * cin, epi);
  a.taps = taps; a.cin = cin; a.frames = F; a.pad = pad; a.dil = 1;
  return a;
}

static int convnext_stack(const ctts_codec* c, int n, const std::vector<const float*>* p, const std::vector<const void*>* px, int inter, int dil,
                          CodecWs& ws, int B, int F, hipStream_t st) {
  const int R = B * F;
  const bool f16 = c->w.gemm_mode == 2;   // one fp16 plane per operand (gemm_h1p_k) instead of hi | lo bf16 planes (gemm_x3p_k)
  const bool x3p = c->w.gemm_mode >= 1 && !px[0].empty() && c->x3p_min_rows > 0 && R >= c->x3p_min_rows && inter % 256 == 0;
  for (int i = 0; x3p && i < n; ++i) {
    // depthwise conv + LayerNorm -> bf16 planes; pwconv1 + GELU -> bf16 planes; pwconv2 * gamma + residual -> f32 rows
    uint16_t* bp = reinterpret_cast<uint16_t*>(ws.b);     // [R256][512] as hi / lo planes: the bytes of the f32 buffer
    uint16_t* bigp = reinterpret_cast<uint16_t*>(ws.big);
    CK(launch_dwconv_ln(ws.a, p[0][i], p[1][i], p[2][i], p[3][i], 1e-6f, dil, nullptr, B, F, 512, st, bp, f16 ? 1 : 0));
    if (f16 && mlp_fused_pays(R)) {   // round 6: the pair in one launch, the inter-wide activation stays on the CU (bit-identical)
      MlpArgs m;
      memset(&m, 0, sizeof(m));
      m.Ap = bp; m.W1p = (const uint16_t*)px[0][i]; m.W2p = (const uint16_t*)px[1][i]; m.M = R; m.inter = inter;
      m.b1 = p[5][i]; m.b2 = p[7][i]; m.gamma = p[8][i]; m.C = ws.a;
      CK(launch_mlp_fused_h1p(m, st));
      continue;
    }
    X3pArgs g;
    memset(&g, 0, sizeof(g));
    g.Ap = bp; g.Wp = (const uint16_t*)px[0][i]; g.M = R; g.N = inter; g.K = 512; g.epi = X3P_GELU_PACKED; g.bias = p[5][i]; g.Cp = bigp;
    CK(f16 ? launch_gemm_h1p(g, st) : launch_gemm_x3p(g, st));
    g.Ap = bigp; g.Wp = (const uint16_t*)px[1][i]; g.N = 512; g.K = inter; g.epi = X3P_SCALE_RES; g.bias = p[7][i]; g.gamma = p[8][i];
    g.res = ws.a; g.ldr = 512; g.C = ws.a; g.ldc = 512; g.Cp = nullptr;
    CK(f16 ? launch_gemm_h1p(g, st) : launch_gemm_x3p(g, st));
  }
  for (int i = 0; !x3p && i < n; ++i) {
    CK(launch_dwconv_ln(ws.a, p[0][i], p[1][i], p[2][i], p[3][i], 1e-6f, dil, ws.b, B, F, 512, st));
    GemmArgs g1 = lin(ws.b, 512, p[4][i], ws.big, inter, R, inter, 512, EPI_BIAS_GELU);
    g1.bias = p[5][i];
    CK(dense(c, g1, st));
    GemmArgs g2 = lin(ws.big, inter, p[6][i], ws.a, 512, R, 512, inter, EPI_BIAS_SCALE_RES);
    g2.bias = p[7][i]; g2.gamma = p[8][i]; g2.res = ws.a; g2.ldr = 512;
    CK(dense(c, g2, st));
  }
  return 0;
}

static hipError_t dense(const ctts_codec* c, const GemmArgs& a, hipStream_t st) {
  return c->w.gemm_mode >= 1 ? launch_gemm_tiled_bf16x3(a, st) : launch_gemm_tiled(a, st);   // mode 2: only the ConvNeXt point-wise pairs are fp16
}

extern "C" int ctts_dvae_decode(ctts_codec* c, const float* hid, float* mel, int32_t B, int32_t T, void* workspace, size_t ws_bytes,
                                void* stream) {
  if (!c || B <= 0 || T <= 0) return fail("ctts_dvae_decode: bad arguments");
  CttsDeviceGuard dg(stream);
  const int F = 2 * T;
  if (ws_bytes < ctts_codec_workspace_bytes(B, F)) return fail("codec workspace too small");
  hipStream_t st = (hipStream_t)stream;
  CodecWs ws = carve_codec(workspace, B, F);
  // dvae.py:281-287: [B,T,768] viewed as [B,2T,384] channels-last
  GemmArgs c0 = conv(hid, 384, c->w.conv_in0_w, ws.big, 128, B, F, 3, 1, EPI_BIAS_GELU);
  c0.bias = c->w.conv_in0_b;
  CK(dense(c, c0, st));
  GemmArgs c2 = conv(ws.big, 128, c->w.conv_in2_w, ws.a, 512, B, F, 3, 1, EPI_BIAS);
  c2.bias = c->w.conv_in2_b;
  CK(dense(c, c2, st));
  if (convnext_stack(c, c->w.n_dvae_blocks, c->d, c->dx, 2048, 2, ws, B, F, st)) return -1;
  CK(dense(c, lin(ws.a, 512, c->w.conv_out_w, ws.mid, 384, B * F, 384, 512, EPI_STORE), st));
  GemmArgs oc = conv(ws.mid, 384, c->w.out_conv_w, mel, 100, B, F, 3, 1, EPI_SCALE);
  oc.gamma = c->w.coef;
  CK(dense(c, oc, st));
  return 0;
}

extern "C" int ctts_vocos_decode(ctts_codec* c, const float* mel, float* wav, int32_t B, int32_t F, void* workspace, size_t ws_bytes,
                                 void* stream) {
  if (!c || B <= 0 || F < 2) return fail("ctts_vocos_decode: bad arguments");
  CttsDeviceGuard dg(stream);
  if (ws_bytes < ctts_codec_workspace_bytes(B, F)) return fail("codec workspace too small");
  hipStream_t st = (hipStream_t)stream;
  CodecWs ws = carve_codec(workspace, B, F);
  GemmArgs e = conv(mel, 100, c->w.v_embed_w, ws.b, 512, B, F, 7, 3, EPI_BIAS);
  e.bias = c->w.v_embed_b;
  CK(dense(c, e, st));
  CK(launch_layernorm(ws.b, c->w.v_norm_w, c->w.v_norm_b, 1e-6f, ws.a, B * F, 512, st));
  if (convnext_stack(c, c->w.n_vocos_blocks, c->v, c->vx, 1536, 1, ws, B, F, st)) return -1;
  CK(launch_layernorm(ws.a, c->w.v_final_w, c->w.v_final_b, 1e-6f, ws.b, B * F, 512, st));
  GemmArgs h = lin(ws.b, 512, c->w.head_w, ws.big, 1026, B * F, 1026, 512, EPI_BIAS);
  h.bias = c->w.head_b;
  CK(dense(c, h, st));
  CK(launch_istft(ws.big, c->w.window, c->w.twiddle, ws.frames, wav, B, F, st));
  return 0;
}

// ------------------------------------------------------------------------------------------------
// single-kernel entry points
// ------------------------------------------------------------------------------------------------
extern "C" int ctts_k_gemm(int32_t tiled, const float* A, const void* W, float* C, int32_t M, int32_t N, int32_t K, int32_t lda,
                           int32_t ldc, int32_t wt, int32_t epi, const float* norm_w, float eps, const float* res, int32_t ldr,
                           const float* bias, const float* gamma, int32_t taps, int32_t cin, int32_t frames, int32_t pad, int32_t dil,
                           void* stream) {
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.W = W; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldc; a.wt = wt; a.epi = epi; a.norm_w = norm_w; a.eps = eps;
  a.res = res; a.ldr = ldr; a.bias = bias; a.gamma = gamma; a.taps = taps > 0 ? taps : 1; a.cin = cin; a.frames = frames; a.pad = pad;
  a.dil = dil;
  { const char* e = getenv("CTTS_X3_DBG_PTR"); if (e) a.dbg = (long long*)strtoull(e, nullptr, 0); }   // probe builds only
  CK(tiled == 2 ? launch_gemm_tiled_bf16x3(a, (hipStream_t)stream) : tiled ? launch_gemm_tiled(a, (hipStream_t)stream) : launch_gemm_skinny(a, (hipStream_t)stream));
  return 0;
}
extern "C" int ctts_k_gemm_x3p(const uint16_t* Ap, const uint16_t* Wp, int32_t M, int32_t N, int32_t K, int32_t epi, const float* bias,
                               const float* gamma, const float* res, float* C, uint16_t* Cp, void* stream) {
  X3pArgs g;
  memset(&g, 0, sizeof(g));
  g.Ap = Ap; g.Wp = Wp; g.M = M; g.N = N; g.K = K; g.epi = epi; g.bias = bias; g.gamma = gamma; g.res = res; g.ldr = N; g.C = C; g.ldc = N; g.Cp = Cp;
  { const char* e = getenv("CTTS_X3_DBG_PTR"); if (e) g.dbg = (long long*)strtoull(e, nullptr, 0); }   // probe variant only
  CK(launch_gemm_x3p(g, (hipStream_t)stream));
  return 0;
}
extern "C" int ctts_k_gemm_h1p(const uint16_t* Ap, const uint16_t* Wp, int32_t M, int32_t N, int32_t K, int32_t epi, const float* bias,
                               const float* gamma, const float* res, float* C, uint16_t* Cp, void* stream) {
  X3pArgs g;
  memset(&g, 0, sizeof(g));
  g.Ap = Ap; g.Wp = Wp; g.M = M; g.N = N; g.K = K; g.epi = epi; g.bias = bias; g.gamma = gamma; g.res = res; g.ldr = N; g.C = C; g.ldc = N; g.Cp = Cp;
  { const char* e = getenv("CTTS_X3_DBG_PTR"); if (e) g.dbg = (long long*)strtoull(e, nullptr, 0); }   // probe variant only (CTTS_H1P_PROBE=1)
  CK(launch_gemm_h1p(g, (hipStream_t)stream));
  return 0;
}
// ------------------------------------------------------------------------------------------------
// A HIP stream confined to a subset of the CUs (hipExtStreamCreateWithCUMask): the acoustic decoder of batch i on a side stream of a few CUs
// while batch i + 1 is generated on the rest (CodecEngine.decode_to_wavs_async, CTTS_CODEC_CUS).  CUs first, first + stride, ... (n of them).
// ------------------------------------------------------------------------------------------------
extern "C" int ctts_stream_create_cu_mask(int32_t first_cu, int32_t n_cus, int32_t stride, int32_t complement, void** stream) {
  if (!stream || n_cus <= 0 || first_cu < 0 || stride <= 0) return fail("ctts_stream_create_cu_mask: bad arguments");
  int dev = 0, ncu = 0;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
  int set = 0;
  for (int i = 0; i < n_cus; ++i) {
    const int cu = first_cu + i * stride;
    if (cu >= ncu) break;
    mask[cu >> 5] |= 1u << (cu & 31);
    ++set;
  }
  if (complement) {   // every CU of the device EXCEPT those: the stream of the work the confined stream must not disturb
    for (int cu = 0; cu < ncu; ++cu) mask[cu >> 5] ^= 1u << (cu & 31);
    set = ncu - set;
  }
  if (set <= 0) return fail("ctts_stream_create_cu_mask: no CU selected");
  hipStream_t st = nullptr;
  CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
  *stream = (void*)st;
  return 0;
}
extern "C" int ctts_stream_destroy(void* stream) {
  if (stream) CK(hipStreamDestroy((hipStream_t)stream));
  return 0;
}

extern "C" int ctts_k_mlp_fused(const uint16_t* Ap, const uint16_t* W1p, const uint16_t* W2p, int32_t M, int32_t inter, const float* b1,
                                const float* b2, const float* gamma, float* C, int32_t planes, void* stream) {
  if (planes != 1) return fail("ctts_k_mlp_fused: planes must be 1 (fp16 plane)");
  MlpArgs m;
  memset(&m, 0, sizeof(m));
  m.Ap = Ap; m.W1p = W1p; m.W2p = W2p; m.M = M; m.inter = inter; m.b1 = b1; m.b2 = b2; m.gamma = gamma; m.C = C;
  { const char* e = getenv("CTTS_X3_DBG_PTR"); if (e) m.dbg = (long long*)strtoull(e, nullptr, 0); }   // probe only
  CK(launch_mlp_fused_h1p(m, (hipStream_t)stream));
  return 0;
}
extern "C" int ctts_k_gemm_fast(const uint16_t* A, int32_t lda, const uint16_t* W, int32_t M, int32_t N, int32_t K, const float* ssq_in,
                                float eps, int32_t epi, float* C32, int32_t ldc, uint16_t* Cb, int32_t ldcb, float* ssq_out, void* stream) {
  FastGemmArgs f;
  memset(&f, 0, sizeof(f));
  f.A = A; f.lda = lda; f.W = W; f.M = M; f.N = N; f.K = K; f.ssq_in = ssq_in; f.eps = eps; f.epi = epi; f.C32 = C32; f.ldc = ldc;
  f.Cb = Cb; f.ldcb = ldcb; f.ssq_out = ssq_out;
  { const char* e = getenv("CTTS_GEMM_DBG_PTR"); if (e) f.dbg = (long long*)strtoull(e, nullptr, 0); }
  CK(launch_gemm_fast(f, (hipStream_t)stream));
  return 0;
}
extern "C" int ctts_k_qkv_rope(const uint16_t* A, const uint16_t* W, int32_t M, const float* ssq_in, float eps, float* qkv, uint16_t* kcache,
                               uint16_t* vcache, int32_t cmax, const float* cos_tab, const float* sin_tab, int32_t q_per_b,
                               const int32_t* len, const int32_t* kv_start, int32_t force_mb, void* stream) {
  FastGemmArgs f;
  memset(&f, 0, sizeof(f));
  f.A = A; f.lda = HID; f.W = W; f.M = M; f.N = 3 * HID; f.K = HID; f.ssq_in = ssq_in; f.eps = eps; f.epi = FEPI_QKV_ROPE; f.C32 = qkv;
  f.ldc = 3 * HID; f.q_per_b = q_per_b; f.len = len; f.kv_start = kv_start; f.cos_t = cos_tab; f.sin_t = sin_tab; f.kc = kcache;
  f.vc = vcache; f.cmax = cmax; f.force_mb = force_mb;
  CK(launch_gemm_fast(f, (hipStream_t)stream));
  return 0;
}
extern "C" int ctts_k_gemm_dec(const uint16_t* Ap, const uint16_t* Wp, int32_t M, int32_t N, int32_t K, const int32_t* n_active,
                               const float* ssq_in, float eps, int32_t epi, float* C32, int32_t ldc, uint16_t* Cp, int32_t kch_out,
                               float* ssq_out, int32_t force_mb, void* stream) {
  if (epi == FEPI_QKV_ROPE) return fail("ctts_k_gemm_dec: the fused QKV epilogue is reached through the decode step only");
  DecGemmArgs d;
  memset(&d, 0, sizeof(d));
  d.Ap = Ap; d.Wp = Wp; d.M = M; d.N = N; d.K = K; d.n_active = n_active; d.ssq_in = ssq_in; d.eps = eps; d.epi = epi; d.C32 = C32;
  d.ldc = ldc; d.Cp = Cp; d.kch_out = kch_out; d.ssq_out = ssq_out; d.force_mb = force_mb;
  { const char* e = getenv("CTTS_GEMM_DBG_PTR"); if (e) d.dbg = (long long*)strtoull(e, nullptr, 0); }   // probes only
  CK(launch_gemm_dec(d, (hipStream_t)stream));
  return 0;
}
extern "C" int ctts_k_gemm_dec32(const float* Ap, const float* Wp, int32_t M, int32_t N, int32_t K, const int32_t* n_active, const float* X,
                                 int32_t ldx, const float* norm_w, float eps, int32_t epi, float* C, int32_t ldc, const float* res, int32_t ldr,
                                 float* Cp, int32_t kch_out, int32_t force_mb, int32_t n_cols, void* stream) {
  Dec32Args d;
  memset(&d, 0, sizeof(d));
  d.n_cols = n_cols;
  d.Ap = Ap; d.Wp = Wp; d.M = M; d.N = N; d.K = K; d.n_active = n_active; d.X = X; d.ldx = ldx; d.norm_w = norm_w; d.eps = eps; d.epi = epi;
  d.C = C; d.ldc = ldc; d.res = res; d.ldr = ldr; d.Cp = Cp; d.kch_out = kch_out; d.force_mb = force_mb;
  CK(launch_gemm_dec32(d, (hipStream_t)stream));
  return 0;
}
extern "C" int ctts_k_gemm_dec32x(const uint16_t* Ap, int64_t a_plane, const uint16_t* Wp, int64_t w_plane, int32_t M, int32_t N, int32_t K,
                                  const int32_t* n_active, const float* X, int32_t ldx, float eps, int32_t epi, float* C, int32_t ldc,
                                  const float* res, int32_t ldr, uint16_t* Cp, int64_t c_plane, int32_t kch_out, float* Cp32, int32_t force_mb,
                                  const float* ssq_in, float* ssq_out, void* stream) {
  Dec32xArgs d;
  memset(&d, 0, sizeof(d));
  d.ssq_in = ssq_in; d.ssq_out = ssq_out;
  d.Ap = Ap; d.a_plane = (size_t)a_plane; d.Wp = Wp; d.w_plane = (size_t)w_plane; d.M = M; d.N = N; d.K = K; d.n_active = n_active;
  d.rms = X != nullptr || ssq_in != nullptr; d.X = X; d.ldx = ldx; d.eps = eps; d.epi = epi; d.C = C; d.ldc = ldc; d.res = res; d.ldr = ldr;
  d.Cp = Cp; d.c_plane = (size_t)c_plane; d.kch_out = kch_out; d.Cp32 = Cp32; d.kch32_out = N / 16; d.force_mb = force_mb;
  CK(launch_gemm_dec32x(d, (hipStream_t)stream));
  return 0;
}
extern "C" const char* ctts_k_dec32_last_variant(void) { return dec32_last_variant(); }
extern "C" int ctts_k_rows_prep(const float* x32, uint16_t* xb, float* ssq, int32_t M, void* stream) {
  CK(launch_rows_prep(x32, xb, ssq, M, (hipStream_t)stream));
  return 0;
}
extern "C" int ctts_k_rope_append(float* qkv, void* kcache, void* vcache, int32_t kv_dtype, int32_t cmax, const float* cos_tab,
                                  const float* sin_tab, int32_t q_per_b, const int32_t* len, const int32_t* kv_start, int32_t M,
                                  void* stream) {
  GptRowMap rm{q_per_b, len, kv_start, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
  CK(launch_rope_append(qkv, kcache, vcache, kv_dtype, cmax, cos_tab, sin_tab, rm, M, (hipStream_t)stream));
  return 0;
}
extern "C" int ctts_k_attention(const float* qkv, const void* kcache, const void* vcache, int32_t kv_dtype, int32_t cmax, float* out,
                                int32_t q_per_b, const int32_t* len, const int32_t* kv_start, int32_t M, void* stream) {
  GptRowMap rm{q_per_b, len, kv_start, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
  CK(launch_attention(qkv, kcache, vcache, kv_dtype, cmax, out, 0, rm, M, (hipStream_t)stream));
  return 0;
}
extern "C" int ctts_k_attention_prefill(const float* qkv, const uint16_t* kcache, const uint16_t* vcache, int32_t cmax, float* out, int32_t q_per_b,
                                        int32_t slot0, const int32_t* kv_start, int32_t M, void* stream) {
  GptRowMap rm{q_per_b, nullptr, kv_start, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, slot0, 0};
  CK(launch_attention(qkv, kcache, vcache, WT_BF16, cmax, out, 0, rm, M, (hipStream_t)stream));
  return 0;
}
extern "C" int ctts_k_attention_dec(const float* qkv, const uint16_t* kcache, const uint16_t* vcache, int32_t cmax, uint16_t* out_packed,
                                    const int32_t* desc, const int32_t* n_active, int32_t M, float* part, int32_t* cnt, int32_t n_cu,
                                    void* stream) {
  if (n_cu < 0 || n_cu > ATT_CUS_MAX) return fail("ctts_k_attention_dec: bad n_cu");
  GptRowMap rm{1, nullptr, nullptr, nullptr, n_active, nullptr, reinterpret_cast<const RowDesc*>(desc), part, cnt, n_cu, 0, 0};
  CK(launch_attention(qkv, kcache, vcache, WT_BF16, cmax, out_packed, 2, rm, M, (hipStream_t)stream));
  return 0;
}
// decode attention of either mode on fragment-packed output, straight to launch_attention (kv_dtype CTTS_BF16: bf16 cache, packed bf16 output;
// CTTS_F32: f32 cache, packed f32 output); `covers_all`: descriptors are valid for all M rows (absent rows carry b = -1) and n_active is not read
extern "C" int ctts_k_attention_dec2(const float* qkv, const void* kcache, const void* vcache, int32_t kv_dtype, int32_t cmax, void* out_packed,
                                     const int32_t* desc, const int32_t* n_active, int32_t covers_all, int32_t M, void* stream) {
  if (!desc || M <= 0 || kv_dtype < 0 || kv_dtype > 2) return fail("ctts_k_attention_dec2: bad arguments");
  GptRowMap rm{1, nullptr, nullptr, nullptr, n_active, nullptr, reinterpret_cast<const RowDesc*>(desc), nullptr, nullptr, 0, 0, covers_all};
  if (kv_dtype == 2) rm.x3_plane = ((size_t)M + 15) / 16 * 16 * HID;   // f32 cache, output as hi | lo bf16 planes (the split-bf16 parity mode)
  CK(launch_attention(qkv, kcache, vcache, kv_dtype == CTTS_BF16 ? WT_BF16 : WT_F32, cmax, out_packed, kv_dtype == CTTS_BF16 ? 2 : kv_dtype == 2 ? 4 : 3, rm, M,
                      (hipStream_t)stream));
  return 0;
}
extern "C" int ctts_k_attention_cfg(int32_t persist, int32_t workgroups, int32_t ring) {
  if (ring > 0 && ring != 2 && ring != 3 && ring != 4) return fail("ctts_k_attention_cfg: ring depth must be 2, 3 or 4");
  attention_persist_override(persist, workgroups, ring);
  return 0;
}
extern "C" int ctts_k_attention_heads_per_wg(int32_t hpw) {
  if (hpw != 1 && hpw != 2 && hpw != 3 && hpw != 4) return fail("ctts_k_attention_heads_per_wg: 1, 2, 3 or 4");
  attention_hpw_override(hpw);
  return 0;
}
extern "C" int ctts_k_attention_oproj(const float* qkv, const uint16_t* kcache, const uint16_t* vcache, int32_t cmax, const uint16_t* wo_hd,
                                      const int32_t* desc, const int32_t* n_active, int32_t M, float* part, int32_t* cnt, float* x32,
                                      uint16_t* xp, float* ssq, void* stream) {
  if (!wo_hd || !desc || !part || !cnt || !x32 || !xp || !ssq || M <= 0) return fail("ctts_k_attention_oproj: bad arguments");
  GptRowMap rm{1, nullptr, nullptr, nullptr, n_active, nullptr, reinterpret_cast<const RowDesc*>(desc), nullptr, nullptr, 0, 0, 0};
  rm.wo_h = wo_hd; rm.op_part = part; rm.op_part_bytes = (int)((size_t)((M + 15) / 16 * 16) * NHEAD * HID * sizeof(float)); rm.op_cnt = cnt;
  rm.x32 = x32; rm.xp = xp; rm.ssq = ssq;
  CK(launch_attention_oproj(qkv, kcache, vcache, cmax, rm, M, (hipStream_t)stream));
  return 0;
}
extern "C" int ctts_k_embed_codes(const float* emb_code, const int64_t* ids_buf, int32_t tcap, const int32_t* len, float* x, int32_t B,
                                  void* stream) {
  CK(launch_embed_codes(emb_code, ids_buf, tcap, len, x, nullptr, nullptr, B, nullptr, nullptr, (hipStream_t)stream));
  return 0;
}
extern "C" int ctts_k_final_norm(const float* x, int32_t q_per_b, const float* w, float eps, float* hfin, float* hiddens,
                                 int32_t max_new, const int32_t* len, int32_t T, int32_t B, void* stream) {
  CK(launch_final_norm(x, q_per_b, w, eps, hfin, hiddens, max_new, len, T, B, nullptr, nullptr, nullptr, (hipStream_t)stream));
  return 0;
}
extern "C" int ctts_k_gemm_pre_x3(const float* A, int32_t lda, const float* W, float* C, int32_t ldc, int32_t M, int32_t N, int32_t K, int32_t epi,
                                  const float* norm_w, const float* rstd, const float* res, int32_t ldr, void* stream) {
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.taps = 1; a.A = A; a.lda = lda; a.W = W; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.wt = WT_F32; a.epi = epi;
  a.norm_w = norm_w; a.res = res; a.ldr = ldr;
  CK(launch_gemm_pre_x3(a, rstd, (hipStream_t)stream));
  return 0;
}
extern "C" int ctts_k_sample(const ctts_gen_state* s, const float* logits, void* stream) {
  if (!s || !logits) return fail("ctts_k_sample: bad arguments");
  CK(launch_sample(make_sample_args(s, logits), (hipStream_t)stream));
  return 0;
}
extern "C" int ctts_k_sample_text(const ctts_gen_state* s, const float* logits, int32_t n_text, void* stream) {
  if (!s || !logits || n_text <= 0 || n_text > NTEXT_MAX) return fail("ctts_k_sample_text: bad arguments");
  CK(launch_sample_text(make_sample_args(s, logits), n_text, (hipStream_t)stream));
  return 0;
}
// what CttsDeviceGuard (kernels.hpp) does for `stream` on the calling thread: the device current before, the device that owns the stream,
// the device current INSIDE the guard, and whether it had to switch (tests: the guard's path runs on a 1-GPU box too)
extern "C" int ctts_k_device_guard_probe(void* stream, int32_t* before, int32_t* stream_dev, int32_t* inside, int32_t* switched) {
  if (!before || !stream_dev || !inside || !switched) return fail("ctts_k_device_guard_probe: null argument");
  int b = -1, in = -1;
  CK(hipGetDevice(&b));
  *stream_dev = -1;
  if (stream != nullptr) { hipDevice_t d; CK(hipStreamGetDevice((hipStream_t)stream, &d)); *stream_dev = (int32_t)d; }
  {
    CttsDeviceGuard dg(stream);
    CK(hipGetDevice(&in));
    *switched = dg.switched ? 1 : 0;
  }
  int after = -1;
  CK(hipGetDevice(&after));
  if (after != b) return fail("CttsDeviceGuard did not restore the current device (%d -> %d)", b, after);
  *before = b; *inside = in;
  return 0;
}
extern "C" int ctts_float_to_int16(const float* wav, int16_t* pcm, uint8_t* keep_bits, int32_t rows, int64_t n, int64_t ld, int32_t per_row,
                                   int32_t product, float keep_thr, uint32_t* peak, void* stream) {
  if (!wav || !pcm || !peak || rows < 0 || n < 0 || ld < n || (product != 0 && product != 1)) return fail("ctts_float_to_int16: bad arguments");
  CttsDeviceGuard dg(stream);
  CK(launch_float_to_int16(wav, n, ld, rows, per_row != 0, product, keep_thr, peak, pcm, keep_bits, (hipStream_t)stream));
  return 0;
}
extern "C" int ctts_copy_bytes(void* dst, const void* src, size_t bytes, void* stream) {
  if (!dst || !src || (bytes & 15) || (((uintptr_t)dst | (uintptr_t)src) & 15)) return fail("ctts_copy_bytes: pointers and size must be 16-byte aligned");
  CttsDeviceGuard dg(stream);
  CK(launch_copy16(src, dst, bytes, (hipStream_t)stream));
  return 0;
}
extern "C" int ctts_k_exp_draws(uint64_t seed, int32_t step, int32_t row0, int32_t rows, int32_t V, float* out, void* stream) {
  if (!out || rows <= 0 || V <= 0) return fail("ctts_k_exp_draws: bad arguments");
  CK(launch_exp_draws(seed, step, row0, rows, V, out, (hipStream_t)stream));
  return 0;
}
extern "C" int ctts_k_dwconv_ln(const float* x, const float* w, const float* b, const float* ln_w, const float* ln_b, float eps,
                                int32_t dil, float* y, int32_t B, int32_t F, void* stream) {
  CK(launch_dwconv_ln(x, w, b, ln_w, ln_b, eps, dil, y, B, F, 512, (hipStream_t)stream));
  return 0;
}
extern "C" int ctts_k_layernorm(const float* x, const float* w, const float* b, float eps, float* y, int32_t rows, void* stream) {
  CK(launch_layernorm(x, w, b, eps, y, rows, 512, (hipStream_t)stream));
  return 0;
}
extern "C" int ctts_k_istft(const float* head, const float* window, const float* twiddle, float* frames, float* wav, int32_t B,
                            int32_t F, void* stream) {
  CK(launch_istft(head, window, twiddle, frames, wav, B, F, (hipStream_t)stream));
  return 0;
}


// ------------------------------------------------------------------------------------------------
// The one collective of the path: the weights, broadcast from one rank to the others at load -- RCCL over xGMI.
// librccl.so is loaded lazily, on the first call, so a single-GPU host needs neither the library nor a communicator.  The engines hold
// no weight memory of their own (the host allocates every buffer and hands pointers to ctts_*_create), so what is broadcast is the
// host's list of device buffers, in place, BEFORE the engines are created on the receiving ranks.
// ------------------------------------------------------------------------------------------------
namespace {
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, ctts_rccl_id, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool tried = false;
};
Rccl g_rccl;
int rccl_load() {
  if (g_rccl.lib) return 0;
  if (g_rccl.tried) return fail("librccl.so could not be loaded");
  g_rccl.tried = true;
  const char* names[] = {getenv("CTTS_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* n : names) {
    if (!n || !*n) continue;
    g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.lib) break;
  }
  if (!g_rccl.lib) return fail("librccl.so not found (dlopen: %s); set CTTS_RCCL_LIB", dlerror());
  g_rccl.GetUniqueId = (int (*)(void*))dlsym(g_rccl.lib, "ncclGetUniqueId");
  g_rccl.CommInitRank = (int (*)(void**, int, ctts_rccl_id, int))dlsym(g_rccl.lib, "ncclCommInitRank");
  g_rccl.CommDestroy = (int (*)(void*))dlsym(g_rccl.lib, "ncclCommDestroy");
  g_rccl.Broadcast = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(g_rccl.lib, "ncclBroadcast");
  g_rccl.GetErrorString = (const char* (*)(int))dlsym(g_rccl.lib, "ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.Broadcast) {
    dlclose(g_rccl.lib); g_rccl.lib = nullptr;
    return fail("librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclBroadcast");
  }
  return 0;
}
int rccl_fail(const char* what, int rc) { return fail("%s: %s (%d)", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error", rc); }
}  // namespace

extern "C" int ctts_rccl_unique_id(ctts_rccl_id* out) {
  if (!out) return fail("ctts_rccl_unique_id: null argument");
  if (rccl_load()) return -1;
  const int rc = g_rccl.GetUniqueId(out);
  return rc == 0 ? 0 : rccl_fail("ncclGetUniqueId", rc);
}
extern "C" int ctts_rccl_comm_create(void** comm, int32_t world, const ctts_rccl_id* id, int32_t rank) {
  if (!comm || !id || world <= 0 || rank < 0 || rank >= world) return fail("ctts_rccl_comm_create: bad arguments");
  if (rccl_load()) return -1;
  const int rc = g_rccl.CommInitRank(comm, world, *id, rank);
  return rc == 0 ? 0 : rccl_fail("ncclCommInitRank", rc);
}
extern "C" void ctts_rccl_comm_destroy(void* comm) {
  if (comm && g_rccl.lib) (void)g_rccl.CommDestroy(comm);
}
extern "C" int ctts_broadcast_weights(void* const* bufs, const size_t* bytes, int32_t n, void* comm, int32_t root, void* stream) {
  if (n < 0 || (n > 0 && (!bufs || !bytes)) || !comm) return fail("ctts_broadcast_weights: bad arguments");
  if (rccl_load()) return -1;
  CttsDeviceGuard dg(stream);
  for (int i = 0; i < n; ++i) {
    if (bytes[i] == 0) continue;
    if (!bufs[i]) return fail("ctts_broadcast_weights: buffer %d is null", i);
    const int rc = g_rccl.Broadcast(bufs[i], bufs[i], bytes[i], /* ncclUint8 */ 1, root, comm, (hipStream_t)stream);   // in place, byte-typed
    if (rc != 0) return rccl_fail("ncclBroadcast", rc);
  }
  return 0;
}
