// Micro-benchmark for the "one persistent decode-step kernel" question (batch-1 / C2 section): what does one
// step-wide synchronisation cost INSIDE a launch on this MI355X, next to the dependent kernel boundary of the hipGraph chain
// it would replace?  Same body in both forms (every workgroup reads a value another workgroup wrote in the previous phase and
// writes one back), 104 phases = the 104 launches of one decode step:
//   (a) 104 dependent launches of 256 x 256 threads, replayed from a hipGraph        (what the engine does today)
//   (b) ONE launch of 256 workgroups (one per CU) walking 104 phases separated by a grid barrier on one monotonic counter
//   (c) the same with an XCD-hierarchical barrier (per-XCD arrival counter -> top counter -> per-XCD generation word)
// Every spin is bounded: a barrier that does not complete within SPIN_LIMIT ticks sets an error word and every later wait bails.
//   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int NWG = 256, NT = 256, NX = 8, N = NWG * NT;
constexpr long long SPIN_LIMIT = 200000000;   // s_memtime ticks: 0.1 s at 2 GHz, 2 s if the counter runs at 100 MHz

struct Sync { unsigned ctr; unsigned pad0[31]; unsigned top; unsigned pad1[31]; unsigned xc[NX][32]; unsigned gen[NX][32]; unsigned err; };

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ bool wait_ge(const unsigned* p, unsigned target, Sync* s) {
  const long long t0 = __builtin_readcyclecounter();
  while (ld_relaxed(p) < target) {
    __builtin_amdgcn_s_sleep(1);
    if (ld_relaxed(&s->err)) return false;
    if ((long long)__builtin_readcyclecounter() - t0 > SPIN_LIMIT) { __hip_atomic_store(&s->err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
  }
  return true;
}

// (b) one counter: lane 0 releases, arrives, polls, acquires
__device__ __forceinline__ void barrier_counter(Sync* s, unsigned phase) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(&s->ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    wait_ge(&s->ctr, phase * NWG, s);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// (c) hierarchical: workgroup w sits on XCD w % 8 (round-robin dispatch).  The last arriver of an XCD is its leader for the
// phase: release fence, arrive at the top counter, wait for all 8 leaders, acquire, publish the XCD's generation word.
__device__ __forceinline__ void barrier_xcd(Sync* s, unsigned phase) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const int x = blockIdx.x % NX;
    const unsigned per = NWG / NX;
    const unsigned old = __hip_atomic_fetch_add(&s->xc[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == phase * per - 1) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(&s->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      wait_ge(&s->top, phase * NX, s);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(&s->gen[x][0], phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      wait_ge(&s->gen[x][0], phase, s);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  }
  __syncthreads();
}

__device__ __forceinline__ void hop(const float* in, float* out, int i) { out[i] = in[(i + 4096 + 257) & (N - 1)] + 1.0f; }

__global__ void k_hop(const float* __restrict__ in, float* __restrict__ out) { hop(in, out, blockIdx.x * NT + threadIdx.x); }

template <int KIND>
__global__ __launch_bounds__(NT) void k_persistent(float* b0, float* b1, Sync* s, int phases, unsigned base) {
  const int i = blockIdx.x * NT + threadIdx.x;
  for (int p = 0; p < phases; ++p) {
    hop((p & 1) ? b1 : b0, (p & 1) ? b0 : b1, i);
    if (KIND == 0) barrier_counter(s, base + p + 1); else barrier_xcd(s, base + p + 1);
  }
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  float *b0, *b1; Sync* s;
  CK(hipMalloc(&b0, N * 4)); CK(hipMalloc(&b1, N * 4)); CK(hipMalloc(&s, sizeof(Sync)));
  CK(hipMemset(b0, 0, N * 4)); CK(hipMemset(b1, 0, N * 4)); CK(hipMemset(s, 0, sizeof(Sync)));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int PH = 104, REPS = 20;
  float ms;
  // (a) launch chain in a graph
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int p = 0; p < PH; ++p) hipLaunchKernelGGL(k_hop, dim3(NWG), dim3(NT), 0, st, (p & 1) ? b1 : b0, (p & 1) ? b0 : b1);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(a, st));
  for (int r = 0; r < REPS; ++r) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
  printf("(a) hipGraph chain of %d dependent launches  : %.2f us per boundary (%.1f us per step)\n", PH, ms * 1e3 / (REPS * PH), ms * 1e3 / REPS);
  // check the chain's arithmetic once so the persistent forms can be compared with it
  CK(hipMemset(b0, 0, N * 4)); CK(hipMemset(b1, 0, N * 4));
  CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
  float want; CK(hipMemcpy(&want, b0 + 12345, 4, hipMemcpyDeviceToHost));
  for (int kind = 0; kind < 2; ++kind) {
    CK(hipMemset(b0, 0, N * 4)); CK(hipMemset(b1, 0, N * 4)); CK(hipMemset(s, 0, sizeof(Sync)));
    unsigned base = 0;
    auto launch = [&]() {
      if (kind == 0) hipLaunchKernelGGL(k_persistent<0>, dim3(NWG), dim3(NT), 0, st, b0, b1, s, PH, base);
      else           hipLaunchKernelGGL(k_persistent<1>, dim3(NWG), dim3(NT), 0, st, b0, b1, s, PH, base);
      base += PH;
    };
    launch(); CK(hipStreamSynchronize(st));
    float got; unsigned err;
    CK(hipMemcpy(&got, b0 + 12345, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&err, &s->err, 4, hipMemcpyDeviceToHost));
    if (err) { printf("(%c) barrier timed out -- skipped\n", 'b' + kind); continue; }
    for (int w = 0; w < 2; ++w) launch();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st));
    for (int r = 0; r < REPS; ++r) launch();
    CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    CK(hipMemcpy(&err, &s->err, 4, hipMemcpyDeviceToHost));
    printf("(%c) ONE launch, %d phases, %s grid barrier: %.2f us per phase (%.1f us per step)  first-launch value %s%s\n", 'b' + kind, PH,
           kind == 0 ? "single-counter   " : "XCD-hierarchical ", ms * 1e3 / (REPS * PH), ms * 1e3 / REPS, got == want ? "== chain" : "!= chain",
           err ? "  [TIMED OUT]" : "");
  }
  return 0;
}
