// Prices ONE dependency edge of a decode-step-like chain of kernels on MI355X under different dispatch mechanisms
// ("overlapped dependent launches").  Every stage: each workgroup reads 24 KB of cold "weights", then the
// value a workgroup of the PREVIOUS stage (on another XCD) wrote, writes its own.  Chain of 104 stages x R repetitions.
//
//   E   hipGraph of a stream-captured chain (edges between the nodes)                          -- today's decode step
//   N   hipGraph whose 104 kernel nodes have NO edges; the dependency is a device counter       -- does the runtime overlap them?
//   X   eager hipExtLaunchKernel(hipExtAnyOrderLaunch) + counters                               -- is the flag honoured on gfx950?
//   Q1  own HSA queue, AQL packets with barrier = 1 and agent-scope acquire / release fences    -- control: equals E?
//   Q2  own HSA queue, barrier = 1, NO fences in the packet header, sc1 (write-through) payload -- what the fences cost
//   Q0  own HSA queue, barrier = 0 (packets may overlap), counters + sc1 payload                -- the overlapped chain
//   Q0L the same with the weights requested AFTER the wait (separates the boundary from the prefetch credit)
//
// Counter protocol (MI355X guide, Guideline 16 in its counter form): payload stores sc1 -> every wave s_waitcnt vmcnt(0) ->
// __syncthreads -> lane 0: relaxed agent atomic add on the stage's per-XCD shard; the consumer's wave 0 polls the 8 shards relaxed
// (s_sleep between polls, BOUNDED by a 2 ms timeout that raises an error word instead of hanging), then sc1 loads.
// Every mode checks the final values (stage count) -- a stale hand-off shows up as a wrong sum.
//
// build:  hipcc --offload-arch=gfx950 -O3 --genco --no-gpu-bundle-output tools/overlap_probe.hip -o tools/overlap_probe.hsaco
//         hipcc --offload-arch=gfx950 -O3 tools/overlap_probe.hip -lhsa-runtime64 -o tools/overlap_probe.bin
// run:    tools/overlap_probe.bin tools/overlap_probe.hsaco [dev]      (dev: the own queue's kernarg segments live in device memory)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

struct StageArgs {
  const uint4* w;       // this stage's weights: n_wg * 24 KB
  const unsigned* in;   // predecessor's output
  unsigned* out;
  unsigned* cnt_prev;   // 8 shards, 64 bytes apart, or null: no wait
  unsigned* cnt_mine;   // or null: no arrive
  unsigned target;      // arrivals (summed over the shards) that mean "predecessor done"
  unsigned n_elem;      // elements of in / out (n_wg * 256)
  unsigned* err;        // timeout word
  int flags;            // 1: sc1 payload loads / stores; 2: weights requested after the wait
};

__device__ __forceinline__ uint4 ld_nt(const uint4* p) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}

#ifndef __HIP_DEVICE_COMPILE__
#define MY_XCC() 0u
#else
__device__ __forceinline__ unsigned xcc_id_() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 7u;
}
#define MY_XCC() xcc_id_()
#endif

extern "C" __global__ __launch_bounds__(256) void k_stage(StageArgs a) {
  const int tid = threadIdx.x, wg = blockIdx.x, lane = tid & 63;
  uint4 wv[6];
  const uint4* wp = a.w + (size_t)wg * 1536 + tid;
  const bool late = (a.flags & 2) != 0;
  if (!late) {
#pragma unroll
    for (int j = 0; j < 6; ++j) wv[j] = ld_nt(wp + j * 256);
  }
  if (a.cnt_prev != nullptr) {
    if (tid < 64) {
      const long long t0 = wall_clock64();
      for (;;) {
        unsigned v = lane < 8 ? __hip_atomic_load(a.cnt_prev + lane * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        const unsigned total = __builtin_amdgcn_readfirstlane(v);
        if (total >= a.target) break;
        if (wall_clock64() - t0 > 200000ll) {   // 2 ms at 100 MHz: a dependency that never arrives must not hang the GPU
          if (lane == 0) atomicExch(a.err, 1u);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
  }
  if (late) {
#pragma unroll
    for (int j = 0; j < 6; ++j) wv[j] = ld_nt(wp + j * 256);
  }
  const unsigned i = (unsigned)wg * 256u + (unsigned)tid;
  const unsigned src = (i + 19u * 256u) % a.n_elem;   // a workgroup 19 further on: another XCD
  const unsigned x = (a.flags & 1) ? __hip_atomic_load(a.in + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.in[src];
  unsigned s = 0;
#pragma unroll
  for (int j = 0; j < 6; ++j) s ^= wv[j].x ^ wv[j].y ^ wv[j].z ^ wv[j].w;   // the weights are zeros: s == 0
  const unsigned y = x + 1u + s;
  if (a.flags & 1) __hip_atomic_store(a.out + i, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else a.out[i] = y;
  if (a.cnt_mine != nullptr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(a.cnt_mine + MY_XCC() * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

#ifndef __HIP_DEVICE_COMPILE__
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define HK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m_ = nullptr; hsa_status_string(s_, &m_); printf("%s: %s (line %d)\n", #x, m_ ? m_ : "?", __LINE__); return 1; } } while (0)

static const int CHAIN = 104, REPS = 20, MAXWG = 768;
static const size_t STAGE_W = (size_t)MAXWG * 24576;   // weight bytes per stage slot

struct Bufs {
  uint4* w; unsigned *b0, *b1, *cnt, *err;   // cnt: [CHAIN + 1][8 shards][16 words]
};
static unsigned* cnt_of(const Bufs& b, int stage) { return b.cnt + (size_t)(stage + 1) * 128; }   // stage -1 = slot 0

static StageArgs stage_args(const Bufs& b, int i, int rep, int n_wg, bool counters, int flags) {
  StageArgs a;
  a.w = (const uint4*)((const char*)b.w + (size_t)i * STAGE_W);
  a.in = (i & 1) ? b.b1 : b.b0;
  a.out = (i & 1) ? b.b0 : b.b1;
  // monotonic counters: stage i of repetition r has seen (r + 1) * n_wg arrivals when it is done; stage 0 waits for the LAST
  // stage of the previous repetition (slot CHAIN - 1), nothing in repetition 0
  a.cnt_prev = nullptr; a.cnt_mine = nullptr; a.target = 0;
  if (counters) {
    a.cnt_mine = cnt_of(b, i);
    if (i > 0) { a.cnt_prev = cnt_of(b, i - 1); a.target = (unsigned)(rep + 1) * n_wg; }
    else if (rep > 0) { a.cnt_prev = cnt_of(b, CHAIN - 1); a.target = (unsigned)rep * n_wg; }
  }
  a.n_elem = (unsigned)n_wg * 256u;
  a.err = b.err;
  a.flags = flags;
  return a;
}

static int reset(const Bufs& b) {
  CK(hipMemset(b.b0, 0, (size_t)MAXWG * 256 * 4));
  CK(hipMemset(b.b1, 0, (size_t)MAXWG * 256 * 4));
  CK(hipMemset(b.cnt, 0, (size_t)(CHAIN + 1) * 128 * 4));
  CK(hipMemset(b.err, 0, 4));
  CK(hipDeviceSynchronize());
  return 0;
}
static int check(const Bufs& b, int n_wg, unsigned expect, const char* name, double us) {
  std::vector<unsigned> h((size_t)n_wg * 256);
  unsigned err = 0;
  CK(hipMemcpy(h.data(), b.b0, h.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&err, b.err, 4, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (unsigned v : h) bad += v != expect;
  printf("  %-4s n_wg %3d: %6.3f us per stage   values %s (%zu of %zu wrong, expect %u, first %u)%s\n", name, n_wg, us, bad ? "WRONG" : "ok", bad,
         h.size(), expect, h[0], err ? "   TIMEOUT raised" : "");
  fflush(stdout);
  return 0;
}

// ---- HSA side --------------------------------------------------------------------------------------------------------------
static hsa_agent_t g_gpu; static bool g_have_gpu = false;
static hsa_status_t agent_cb(hsa_agent_t ag, void*) {
  hsa_device_type_t t;
  if (hsa_agent_get_info(ag, HSA_AGENT_INFO_DEVICE, &t) == HSA_STATUS_SUCCESS && t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = ag; g_have_gpu = true; }
  return HSA_STATUS_SUCCESS;
}

struct HsaKernel { uint64_t object; uint32_t kernarg, group, priv; };

static int hsa_load(const char* path, HsaKernel* k, hsa_executable_t* exe_out) {
  FILE* f = fopen(path, "rb");
  if (!f) { printf("cannot open %s\n", path); return 1; }
  fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
  static std::vector<char> blob; blob.resize(n);
  if (fread(blob.data(), 1, n, f) != (size_t)n) { fclose(f); return 1; }
  fclose(f);
  hsa_code_object_reader_t rd;
  HK(hsa_code_object_reader_create_from_memory(blob.data(), blob.size(), &rd));
  hsa_executable_t exe;
  HK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
  HK(hsa_executable_load_agent_code_object(exe, g_gpu, rd, nullptr, nullptr));
  HK(hsa_executable_freeze(exe, nullptr));
  hsa_executable_symbol_t sym;
  HK(hsa_executable_get_symbol_by_name(exe, "k_stage.kd", &g_gpu, &sym));
  HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k->object));
  HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k->kernarg));
  HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k->group));
  HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k->priv));
  *exe_out = exe;
  return 0;
}

// One timed run on the HSA queue: REPS * CHAIN dispatch packets + one barrier packet that carries the completion signal.
static char* g_kdev = nullptr;   // kernarg pool in DEVICE memory (argv[2] == "dev"): the host pool is then only the staging copy
static int run_hsa(hsa_queue_t* q, const HsaKernel& k, char* kernarg_pool, size_t kernarg_stride, hsa_signal_t done, const Bufs& b, int n_wg,
                   int barrier_bit, int fence_scope, bool counters, int flags, const char* name) {
  if (reset(b)) return 1;
  const uint32_t mask = q->size - 1;
  uint64_t widx = hsa_queue_load_write_index_relaxed(q);
  const uint64_t first = widx;
  hsa_signal_store_relaxed(done, 1);
  // pass 0: one untimed warm-up repetition is part of the same submission; timing covers everything, divided by all stages
  const int total = REPS * CHAIN;
  if ((uint64_t)total + 2 > q->size) { printf("queue too small\n"); return 1; }
  for (int n = 0; n < total; ++n) {
    const int rep = n / CHAIN, i = n % CHAIN;
    char* ka = kernarg_pool + (size_t)n * kernarg_stride;
    memset(ka, 0, kernarg_stride);
    const StageArgs a = stage_args(b, i, rep, n_wg, counters, flags);
    memcpy(ka, &a, sizeof(a));
    // code-object-v5 implicit arguments behind the explicit ones (8-byte aligned): block counts (3 x u32), group sizes (3 x u16)
    const size_t hid = (sizeof(StageArgs) + 7) & ~(size_t)7;
    if (hid + 24 <= kernarg_stride) {
      uint32_t bc[3] = {(uint32_t)n_wg, 1, 1}; uint16_t gs[3] = {256, 1, 1};
      memcpy(ka + hid, bc, 12); memcpy(ka + hid + 12, gs, 6);
    }
    hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)q->base_address + (widx & mask);
    hsa_kernel_dispatch_packet_t body;
    memset(&body, 0, sizeof(body));
    body.workgroup_size_x = 256; body.workgroup_size_y = 1; body.workgroup_size_z = 1;
    body.grid_size_x = (uint32_t)n_wg * 256u; body.grid_size_y = 1; body.grid_size_z = 1;
    body.private_segment_size = k.priv; body.group_segment_size = k.group;
    body.kernel_object = k.object; body.kernarg_address = g_kdev ? (void*)(g_kdev + (size_t)n * kernarg_stride) : (void*)ka;
    body.completion_signal.handle = 0;
    const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (barrier_bit << HSA_PACKET_HEADER_BARRIER) |
                                       (fence_scope << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (fence_scope << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
    const uint16_t setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
    // body first (everything but the first 4 bytes), then the header word with release semantics
    memcpy((char*)p + 4, (char*)&body + 4, sizeof(body) - 4);
    __atomic_store_n((uint32_t*)p, (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
    ++widx;
  }
  {   // barrier-AND packet: waits for every earlier packet (barrier bit), system-scope release, completion signal
    hsa_barrier_and_packet_t* p = (hsa_barrier_and_packet_t*)q->base_address + (widx & mask);
    hsa_barrier_and_packet_t body;
    memset(&body, 0, sizeof(body));
    body.completion_signal = done;
    const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                       (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
    memcpy((char*)p + 4, (char*)&body + 4, sizeof(body) - 4);
    __atomic_store_n((uint32_t*)p, (uint32_t)header, __ATOMIC_RELEASE);
    ++widx;
  }
  if (g_kdev) { CK(hipMemcpy(g_kdev, kernarg_pool, (size_t)total * kernarg_stride, hipMemcpyHostToDevice)); CK(hipDeviceSynchronize()); }
  hsa_queue_store_write_index_screlease(q, widx);
  const auto t0 = std::chrono::steady_clock::now();
  hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)(widx - 1));
  const hsa_signal_value_t v = hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 5000000000ull /* 5 s */, HSA_WAIT_STATE_ACTIVE);
  const auto t1 = std::chrono::steady_clock::now();
  if (v >= 1) { printf("  %-4s n_wg %3d: completion signal never fired (first packet %llu)\n", name, n_wg, (unsigned long long)first); return 1; }
  const double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / total;
  return check(b, n_wg, (unsigned)total, name, us);
}

int main(int argc, char** argv) {
  const char* hsaco = argc > 1 ? argv[1] : "tools/overlap_probe.hsaco";
  hipStream_t st; CK(hipStreamCreate(&st));
  Bufs b;
  CK(hipMalloc(&b.w, (size_t)CHAIN * STAGE_W));
  CK(hipMemset(b.w, 0, (size_t)CHAIN * STAGE_W));
  CK(hipMalloc(&b.b0, (size_t)MAXWG * 256 * 4)); CK(hipMalloc(&b.b1, (size_t)MAXWG * 256 * 4));
  CK(hipMalloc(&b.cnt, (size_t)(CHAIN + 1) * 128 * 4)); CK(hipMalloc(&b.err, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grids[] = {64, 192, 540, 768};

  // ---- E: captured chain, edges --------------------------------------------------------------------------------------------
  printf("E: hipGraph of a captured chain (edges), plain payload\n");
  for (int n_wg : grids) {
    if (reset(b)) return 1;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < CHAIN; ++i) { StageArgs a = stage_args(b, i, 0, n_wg, false, 0); hipLaunchKernelGGL(k_stage, dim3(n_wg), dim3(256), 0, st, a); }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < REPS; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    check(b, n_wg, REPS * CHAIN, "E", ms * 1e3 / (REPS * CHAIN));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }

  // ---- N: graph whose kernel nodes have no edges, counters reset by a memset in front of every launch ------------------------
  printf("N: hipGraph, kernel nodes WITHOUT edges + counters (sc1 payload)\n");
  for (int n_wg : grids) {
    if (reset(b)) return 1;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipGraphCreate(&g, 0));
    std::vector<StageArgs> args(CHAIN);
    std::vector<void*> argp(CHAIN);
    for (int i = 0; i < CHAIN; ++i) {
      args[i] = stage_args(b, i, 0, n_wg, true, 1);
      if (i == 0) { args[i].cnt_prev = nullptr; }
      argp[i] = &args[i];
      hipKernelNodeParams kp;
      memset(&kp, 0, sizeof(kp));
      kp.func = (void*)k_stage; kp.gridDim = dim3(n_wg); kp.blockDim = dim3(256); kp.sharedMemBytes = 0; kp.kernelParams = &argp[i]; kp.extra = nullptr;
      hipGraphNode_t node;
      CK(hipGraphAddKernelNode(&node, g, nullptr, 0, &kp));
    }
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < REPS; ++r) {
      CK(hipMemsetAsync(b.cnt, 0, (size_t)(CHAIN + 1) * 128 * 4, st));
      CK(hipGraphLaunch(ge, st));
    }
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    check(b, n_wg, REPS * CHAIN, "N", ms * 1e3 / (REPS * CHAIN));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }

  // ---- X: eager any-order launches + counters --------------------------------------------------------------------------------
  printf("X: eager hipExtLaunchKernel(hipExtAnyOrderLaunch) + counters (host-bound below ~3 us)\n");
  for (int n_wg : grids) {
    if (reset(b)) return 1;
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < REPS; ++r)
      for (int i = 0; i < CHAIN; ++i) {
        StageArgs a = stage_args(b, i, r, n_wg, true, 1);
        void* ap[1] = {&a};
        CK(hipExtLaunchKernel((const void*)k_stage, dim3(n_wg), dim3(256), ap, 0, st, nullptr, nullptr, hipExtAnyOrderLaunch));
      }
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    check(b, n_wg, REPS * CHAIN, "X", ms * 1e3 / (REPS * CHAIN));
  }

  // ---- own HSA queue --------------------------------------------------------------------------------------------------------
  HK(hsa_init());
  HK(hsa_iterate_agents(agent_cb, nullptr));
  if (!g_have_gpu) { printf("no HSA GPU agent\n"); return 1; }
  HsaKernel k; hsa_executable_t exe;
  if (hsa_load(hsaco, &k, &exe)) return 1;
  printf("HSA: kernel object 0x%llx, kernarg %u B (explicit %zu), group %u, private %u\n", (unsigned long long)k.object, k.kernarg, sizeof(StageArgs), k.group, k.priv);
  hsa_queue_t* q = nullptr;
  HK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
  hsa_signal_t done;
  HK(hsa_signal_create(1, 0, nullptr, &done));
  const size_t stride = ((size_t)k.kernarg + 255) & ~(size_t)255;
  char* kpool = nullptr;
  CK(hipHostMalloc((void**)&kpool, stride * REPS * CHAIN, hipHostMallocDefault));
  if (argc > 2 && !strcmp(argv[2], "dev")) { CK(hipMalloc((void**)&g_kdev, stride * REPS * CHAIN)); printf("kernarg segments in DEVICE memory\n"); }
  else printf("kernarg segments in pinned HOST memory\n");
  struct { const char* name; int barrier, fence; bool counters; int flags; const char* what; } modes[] = {
    {"Q1", 1, HSA_FENCE_SCOPE_AGENT, false, 0, "barrier = 1, agent acquire/release fences, plain payload"},
    {"Q1s", 1, HSA_FENCE_SCOPE_AGENT, false, 1, "barrier = 1, agent fences, sc1 payload"},
    {"Q2", 1, HSA_FENCE_SCOPE_NONE, false, 1, "barrier = 1, NO fences, sc1 payload"},
    {"Q2c", 1, HSA_FENCE_SCOPE_NONE, true, 1, "barrier = 1, NO fences, sc1 payload + counters (what the protocol itself costs)"},
    {"Q0", 0, HSA_FENCE_SCOPE_NONE, true, 1, "barrier = 0, counters, sc1 payload, weights requested BEFORE the wait"},
    {"Q0L", 0, HSA_FENCE_SCOPE_NONE, true, 3, "barrier = 0, counters, sc1 payload, weights requested AFTER the wait"},
    {"Q0a", 0, HSA_FENCE_SCOPE_AGENT, true, 1, "barrier = 0, counters, agent fences in the header"},
  };
  for (const auto& m : modes) {
    printf("%s: own HSA queue, %s\n", m.name, m.what);
    for (int n_wg : grids)
      if (run_hsa(q, k, kpool, stride, done, b, n_wg, m.barrier, m.fence, m.counters, m.flags, m.name)) { printf("  (mode aborted)\n"); break; }
  }
  return 0;
}
#endif
