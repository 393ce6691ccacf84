// Micro-benchmark (round 3): would a KV prefetch one layer ahead pay?  Three questions, one 28 MB "attention-like" read
// (540 workgroups x 4 waves, 54 KB contiguous per workgroup, all loads in flight at once):
//   (1) how long does it take COLD (HBM) vs after the same bytes were read a moment ago and ≈19 MB of other traffic went by
//       (what the Infinity Cache / the XCD L2s still hold across kernel boundaries), with the same and with a shifted
//       workgroup -> segment mapping (same XCD L2 vs another XCD's);
//   (2) what does a CONCURRENT reader on a second stream cost a chain of dependent small kernels (the decode step's shape);
//   (3) how fast a few resident workgroups can pull the 28 MB (the prefetcher's own rate).
//   hipcc --offload-arch=gfx950 -O3 tools/mall_probe.hip -o /tmp/mall_probe && /tmp/mall_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// workgroup b reads segment (b + shift) % n of `seg` bytes; 256 threads x 16 B = 4 KiB per round, all rounds issued back to back
__global__ void __launch_bounds__(256) read_seg(const char* __restrict__ base, int seg, int n, int shift, unsigned* __restrict__ sink) {
  const int s = (blockIdx.x + shift) % n;
  const u32x4* p = reinterpret_cast<const u32x4*>(base + (size_t)s * seg) + threadIdx.x;
  const int rounds = seg / 4096;
  u32x4 acc = {0, 0, 0, 0};
#pragma unroll 14
  for (int r = 0; r < rounds; ++r) { const u32x4 v = p[r * 256]; acc ^= v; }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) sink[0] = 1;
}

// `loops` passes over the n segments by a SMALL resident grid (the prefetcher's shape): workgroup b takes segments b, b+G, ...
__global__ void __launch_bounds__(256) read_loop(const char* __restrict__ base, int seg, int n, int loops, unsigned* __restrict__ sink) {
  u32x4 acc = {0, 0, 0, 0};
  const int rounds = seg / 4096;
  for (int l = 0; l < loops; ++l)
    for (int s = blockIdx.x; s < n; s += gridDim.x) {
      const u32x4* p = reinterpret_cast<const u32x4*>(base + (size_t)s * seg) + threadIdx.x;
#pragma unroll 14
      for (int r = 0; r < rounds; ++r) { const u32x4 v = p[r * 256]; acc ^= v; }
    }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) sink[0] = 1;
}

__global__ void k_hop1(const float* __restrict__ in, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  out[i] = in[(i + 4096) & 16383] + 1.0f;
}

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
  hipStream_t st, side; CK(hipStreamCreate(&st)); CK(hipStreamCreate(&side));
  const int seg = 14 * 4096, n = 540;                      // 57,344 B x 540 = 30.97 MB (C3's average attention launch: 28.4 MB)
  const size_t a_bytes = (size_t)seg * n;
  const size_t flush_bytes = (size_t)768 << 20, w_bytes = (size_t)19 << 20;
  char *A, *F, *W; unsigned* sink;
  CK(hipMalloc(&A, a_bytes)); CK(hipMalloc(&F, flush_bytes)); CK(hipMalloc(&W, w_bytes)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(A, 1, a_bytes)); CK(hipMemset(F, 2, flush_bytes)); CK(hipMemset(W, 3, w_bytes)); CK(hipMemset(sink, 0, 64));
  hipEvent_t e0, e1; CK(hipEventCreateWithFlags(&e0, hipEventDisableSystemFence)); CK(hipEventCreateWithFlags(&e1, hipEventDisableSystemFence));
  const int fl_n = (int)(flush_bytes / seg), w_n = (int)(w_bytes / seg);

  auto flush = [&]() { hipLaunchKernelGGL(read_seg, dim3(fl_n), dim3(256), 0, st, F, seg, fl_n, 0, sink); };
  auto other = [&]() { hipLaunchKernelGGL(read_seg, dim3(w_n), dim3(256), 0, st, W, seg, w_n, 0, sink); };
  auto warm = [&](int shift) { hipLaunchKernelGGL(read_seg, dim3(n), dim3(256), 0, st, A, seg, n, shift, sink); };
  auto timed = [&](int shift, double* us) -> int {
    hipExtLaunchKernelGGL(read_seg, dim3(n), dim3(256), 0, st, e0, e1, 0, (const char*)A, seg, n, shift, sink);
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); *us = ms * 1e3; return 0;
  };
  struct Case { const char* name; int warm_shift; bool with_other; int timed_shift; };
  const Case cases[] = {
      {"cold (768 MB of other reads before)", -1, false, 0},
      {"read just before, same mapping", 0, false, 0},
      {"read just before, mapping shifted by 3 (other XCD)", 0, false, 3},
      {"read before + 19 MB of other reads, same mapping", 0, true, 0},
      {"read before + 19 MB of other reads, shifted by 3", 0, true, 3},
      {"read before + 3 x 19 MB of other reads, same mapping", 0, true, 100},
  };
  printf("(1) %d workgroups x %d B = %.2f MB per launch\n", n, seg, a_bytes / 1e6);
  for (const Case& c : cases) {
    std::vector<double> t;
    for (int rep = 0; rep < 15; ++rep) {
      flush();
      if (c.warm_shift >= 0) warm(c.warm_shift);
      if (c.with_other) { other(); if (c.timed_shift == 100) { other(); other(); } }
      double us; if (timed(c.timed_shift == 100 ? 0 : c.timed_shift, &us)) return 1;
      t.push_back(us);
    }
    const double m = median(t);
    printf("  %-56s %6.2f us  = %5.2f TB/s   (min %.2f, max %.2f)\n", c.name, m, a_bytes / m / 1e6, *std::min_element(t.begin(), t.end()),
           *std::max_element(t.begin(), t.end()));
  }

  // (3) the prefetcher's own rate: G resident workgroups walking the 540 segments
  printf("(3) small resident grid reading the same %.2f MB (cold)\n", a_bytes / 1e6);
  for (int G : {32, 64, 128, 256}) {
    std::vector<double> t;
    for (int rep = 0; rep < 7; ++rep) {
      flush();
      hipExtLaunchKernelGGL(read_loop, dim3(G), dim3(256), 0, st, e0, e1, 0, (const char*)A, seg, n, 1, sink);
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms * 1e3);
    }
    const double m = median(t);
    printf("  %3d workgroups: %7.2f us = %5.2f TB/s\n", G, m, a_bytes / m / 1e6);
  }

  // (4) the decode projections' weight fetch in miniature: G workgroups x S bytes, every load in flight at once, latency-bound.
  //     cold (HBM) vs "warm in the Infinity Cache only" (read before, then 64 MB of other reads through every L2) vs straight after.
  printf("(4) small latency-bound reads (the shape of a decode projection's weight fetch)\n");
  {
    char* B2; const size_t b2 = (size_t)64 << 20; CK(hipMalloc(&B2, b2)); CK(hipMemset(B2, 5, b2));
    const int b2n = (int)(b2 / seg);
    struct Shape { const char* name; int G; int S; };
    const Shape shapes[] = {{"QKV     3.5 MB", 144, 6 * 4096}, {"o_proj  1.2 MB", 48, 6 * 4096}, {"gate/up 9.4 MB", 256, 9 * 4096}, {"down    4.7 MB", 192, 6 * 4096}};
    for (const Shape& sh : shapes) {
      double med[3];
      for (int mode = 0; mode < 3; ++mode) {   // 0 cold, 1 Infinity-Cache-warm, 2 straight after the same read
        std::vector<double> t;
        for (int rep = 0; rep < 15; ++rep) {
          flush();
          if (mode >= 1) hipLaunchKernelGGL(read_seg, dim3(sh.G), dim3(256), 0, st, (const char*)A, sh.S, sh.G, 0, sink);
          if (mode == 1) hipLaunchKernelGGL(read_seg, dim3(b2n), dim3(256), 0, st, (const char*)B2, seg, b2n, 0, sink);
          hipExtLaunchKernelGGL(read_seg, dim3(sh.G), dim3(256), 0, st, e0, e1, 0, (const char*)A, sh.S, sh.G, mode == 1 ? 3 : 0, sink);
          CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms * 1e3);
        }
        med[mode] = median(t);
      }
      printf("  %s (%3d workgroups x %5d B): cold %.2f us | Infinity-Cache-warm %.2f us | straight after %.2f us\n", sh.name, sh.G, sh.S, med[0], med[1], med[2]);
    }
  }

  // (5) what ONE workgroup per CU can pull from the L2s against three per CU (the parity mode's gate/up launch is 768 workgroups of
  //     144 KB = 3 per CU; a 3-column-tile variant would be 256 workgroups of 336 KB): the same 110 MB in total out of a 19 MB
  //     (L2 / Infinity Cache resident after the first pass) set, 14 x 16 B per lane in flight per workgroup.
  printf("(5) 110 MB out of a 19 MB set: workgroups x bytes each (every CU re-reads slices other CUs read too)\n");
  {
    struct Shape { int G; int S; };
    const Shape shapes[] = {{768, 36 * 4096}, {512, 54 * 4096}, {256, 108 * 4096}, {256, 82 * 4096}};
    const int wn = (int)(w_bytes / 4096);
    for (const Shape& sh : shapes) {
      std::vector<double> t;
      for (int rep = 0; rep < 12; ++rep) {
        // segment index modulo the set: consecutive workgroups start 144 KB apart and wrap around the 19 MB
        hipExtLaunchKernelGGL(read_seg, dim3(sh.G), dim3(256), 0, st, e0, e1, 0, (const char*)W, sh.S, (int)(w_bytes / sh.S), 0, sink);
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep >= 2) t.push_back(ms * 1e3);
      }
      const double m = median(t);
      printf("  %4d workgroups x %6d B = %6.1f MB: %6.2f us = %5.2f TB/s  (%.0f GB/s per CU)\n", sh.G, sh.S, (double)sh.G * sh.S / 1e6, m,
             (double)sh.G * sh.S / m / 1e6, (double)sh.G * sh.S / m / 1e3 / 256.0);
    }
    (void)wn;
  }

  // (2) a dependent chain with and without a concurrent reader on another stream
  float *b0, *b1; CK(hipMalloc(&b0, 16384 * 4)); CK(hipMalloc(&b1, 16384 * 4)); CK(hipMemset(b0, 0, 16384 * 4)); CK(hipMemset(b1, 0, 16384 * 4));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < 104; ++i) hipLaunchKernelGGL(k_hop1, dim3(64), dim3(256), 0, st, (i & 1) ? b1 : b0, (i & 1) ? b0 : b1);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  printf("(2) chain of 104 dependent 64-workgroup kernels (graph), 20 replays\n");
  for (int G : {0, 32, 64, 128}) {
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, st));
    CK(hipDeviceSynchronize());
    // the reader runs for >= the chain's 20 x 104 x ~1.8 us = 3.8 ms: loops x 31 MB at the rate measured in (3)
    if (G) hipLaunchKernelGGL(read_loop, dim3(G), dim3(256), 0, side, (const char*)A, seg, n, G >= 128 ? 600 : G >= 64 ? 320 : 160, sink);
    CK(hipEventRecord(a, st));
    for (int rep = 0; rep < 20; ++rep) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(b, st));
    CK(hipEventSynchronize(b));
    const bool side_busy = G && hipStreamQuery(side) == hipErrorNotReady;
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipDeviceSynchronize());
    printf("  concurrent reader of %3d workgroups: %.3f us per chain kernel%s\n", G, ms * 1e3 / (20.0 * 104),
           G ? (side_busy ? "  (reader still running when the chain ended)" : "  (reader ended before the chain!)") : "");
  }
  return 0;
}
