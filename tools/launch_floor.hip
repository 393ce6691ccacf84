// Micro-benchmark: per-kernel cost of a dependent chain of small kernels on MI355X, eager vs hipGraph.
//   hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty() {}
// each block reads a value the previous kernel wrote (cross-XCD dependent load) and writes one back
__global__ void k_hop1(const float* __restrict__ in, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  out[i] = in[(i + 4096) & 16383] + 1.0f;
}
// two dependent hops: index load -> data load
__global__ void k_hop2(const int* __restrict__ idx, const float* __restrict__ in, float* __restrict__ out, int* __restrict__ idx_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = idx[i];
  out[i] = in[j & 16383] + 1.0f;
  idx_out[i] = (j + 4099) & 16383;
}

template <typename F>
static int run(const char* name, F launch, int chain, hipStream_t st) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  // eager
  for (int w = 0; w < 2; ++w) for (int i = 0; i < chain; ++i) launch(i, st);
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(a, st));
  for (int rep = 0; rep < 20; ++rep) for (int i = 0; i < chain; ++i) launch(i, st);
  CK(hipEventRecord(b, st));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double eager = ms * 1e3 / (20.0 * chain);
  // graph
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < chain; ++i) launch(i, st);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(a, st));
  for (int rep = 0; rep < 20; ++rep) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(b, st));
  CK(hipEventSynchronize(b));
  CK(hipEventElapsedTime(&ms, a, b));
  printf("%-34s chain %4d: eager %.2f us/kernel, graph %.2f us/kernel\n", name, chain, eager, ms * 1e3 / (20.0 * chain));
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return 0;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  float *b0, *b1; int *i0, *i1;
  CK(hipMalloc(&b0, 16384 * 4)); CK(hipMalloc(&b1, 16384 * 4)); CK(hipMalloc(&i0, 16384 * 4)); CK(hipMalloc(&i1, 16384 * 4));
  CK(hipMemset(b0, 0, 16384 * 4)); CK(hipMemset(b1, 0, 16384 * 4)); CK(hipMemset(i0, 0, 16384 * 4)); CK(hipMemset(i1, 0, 16384 * 4));
  for (int grid : {1, 64, 256, 768}) {
    printf("grid %d x 256 threads\n", grid);
    run("empty", [&](int, hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, s); }, 104, st);
    if (grid * 256 <= 16384 * 16) {
      const int g2 = grid > 64 ? 64 : grid;
      run("1 dependent hop", [&](int i, hipStream_t s) { hipLaunchKernelGGL(k_hop1, dim3(g2), dim3(256), 0, s, (i & 1) ? b1 : b0, (i & 1) ? b0 : b1); }, 104, st);
      run("2 dependent hops", [&](int i, hipStream_t s) { hipLaunchKernelGGL(k_hop2, dim3(g2), dim3(256), 0, s, (i & 1) ? i1 : i0, (i & 1) ? b1 : b0, (i & 1) ? b0 : b1, (i & 1) ? i0 : i1); }, 104, st);
    }
  }
  return 0;
}
