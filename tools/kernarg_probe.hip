// Micro-benchmark: does the SIZE of a kernel's by-value argument struct change the cost of a short dependent kernel on MI355X?
// (round 3: growing DecGemmArgs by 32 bytes made the 5 us o_proj launch 1 us slower.)  A chain of 104 graph-captured launches; each
// kernel reads a pointer from the FIRST or the LAST 8 bytes of an N-byte struct, loads what the previous kernel wrote, writes one value.
//   hipcc --offload-arch=gfx950 -O3 tools/kernarg_probe.hip -o /tmp/kernarg_probe && /tmp/kernarg_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int N> struct Args { const float* first; char pad[N - 24]; float* out; const float* last; };
template <> struct Args<24> { const float* first; float* out; const float* last; };

template <int N, bool LAST>
__global__ void k_hop(Args<N> a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const float* in = LAST ? a.last : a.first;
  a.out[i] = in[(i + 4096) & 16383] + 1.0f;
}

template <int N, bool LAST>
static int run(float* b0, float* b1, hipStream_t st, int grid) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < 104; ++i) {
    Args<N> a; memset(&a, 0, sizeof(a));
    a.first = a.last = (i & 1) ? b1 : b0; a.out = (i & 1) ? b0 : b1;
    hipLaunchKernelGGL((k_hop<N, LAST>), dim3(grid), dim3(256), 0, st, a);
  }
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int rep = 0; rep < 30; ++rep) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("args %4zu bytes, pointer read from the %s field, grid %3d: %.3f us per kernel\n", sizeof(Args<N>), LAST ? "LAST " : "FIRST", grid, ms * 1e3 / (30.0 * 104));
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return 0;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  float *b0, *b1;
  CK(hipMalloc(&b0, 65536 * 4)); CK(hipMalloc(&b1, 65536 * 4)); CK(hipMemset(b0, 0, 65536 * 4)); CK(hipMemset(b1, 0, 65536 * 4));
  for (int grid : {64, 192}) {
#define R(N) if (run<N, false>(b0, b1, st, grid)) return 1; if (run<N, true>(b0, b1, st, grid)) return 1;
    R(24) R(64) R(128) R(192) R(224) R(256) R(288) R(320) R(512) R(1024)
  }
  return 0;
}
