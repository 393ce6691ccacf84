// Micro-benchmark (round 6): what bounds the k loop of the acoustic decoder's 256 x 256 MFMA tiles (csrc/codec_gemm.hip)?  The phase probe
// of round 3 put a 64-wide k stage at 1.85 us against 0.86 us of MFMA work per SIMD and named the L2 -> LDS fill: 64 KiB per CU and stage =
// 35 GB/s per CU = 9 TB/s over the chip.  This probe runs ONLY the operand movement of gemm_h1p_k at the bench's roofline shape
// (M 65,536, N 2048, K 512, fp16 fragment planes, 2048 tiles in the kernel's XCD-aware order, 512 threads, one 32 KiB ring slot per
// 32-wide k block) in variants, to tell a hardware rate from a scheduling loss:
//   0  LDS-DMA (global_load_lds_dwordx4), 3 slots in flight, no barrier, nothing reads the LDS      -> the raw fill rate
//   1  as 0 with the kernel's protocol: stages of two slots, s_waitcnt vmcnt(0) + s_barrier per stage
//   2  global_load_dwordx4 -> VGPR -> ds_write_b128 (register staged), 2 slots in flight, no barrier
//   3  as 0 plus the kernel's fragment reads (6 ds_read_b128 per wave and 16-wide k block), no MFMA
//   4  as 0, two workgroups of 256 threads per CU on 128 x 256 tiles (24 KiB slots)
//   5  as 0 reading ONLY the weight panel (2 MiB, L2 resident): is the A panel's first touch (HBM / MALL) what limits?
//   hipcc --offload-arch=gfx950 -O3 tools/fill_probe.hip -o /tmp/fill_probe && /tmp/fill_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int FRAG = 512;            // fp16 elements of one 32 x 16 fragment (1 KiB)

template <int VAR>
__global__ __launch_bounds__(512, 2) void fill_k(const uint16_t* __restrict__ Ap, const uint16_t* __restrict__ Wp, int M, int N, int K, unsigned* __restrict__ sink) {
  constexpr int SLOT = 32 * FRAG, NSLOT = 4;
  __shared__ __attribute__((aligned(16))) uint16_t lds[NSLOT * SLOT];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nx = N / 256, ny = M / 256, T = nx * ny, per = (T + 7) / 8;
  const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || t >= T) return;
  const int m0 = (t / nx) * 256, n0 = (t % nx) * 256, kb16 = K >> 4, nq = K >> 5;
  const uint16_t* ag = (VAR == 5 ? Wp + ((size_t)(((t * 8) % (N / 32)) + wave) % (N / 32) * kb16) * FRAG : Ap + ((size_t)((m0 >> 5) + wave) * kb16) * FRAG) + lane * 8;
  const uint16_t* wg = Wp + ((size_t)((n0 >> 5) + wave) * kb16) * FRAG + lane * 8;
  unsigned acc = 0;
  auto issue = [&](int q) {
    uint16_t* la = lds + (q % NSLOT) * SLOT + wave * 2 * FRAG;
    uint16_t* lw = la + 16 * FRAG;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ag + ((size_t)q * 2 + h) * FRAG), (__attribute__((address_space(3))) void*)(la + h * FRAG), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wg + ((size_t)q * 2 + h) * FRAG), (__attribute__((address_space(3))) void*)(lw + h * FRAG), 16, 0, 0);
    }
  };
  if (VAR == 0 || VAR == 3 || VAR == 5) {
    issue(0); if (nq > 1) issue(1); if (nq > 2) issue(2);
    for (int q = 0; q < nq; ++q) {
      if (q + 2 < nq) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (q + 1 < nq) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (VAR == 3) {
        __builtin_amdgcn_s_barrier();     // the reads below touch other waves' pieces
        const int wm = wave & 3, wn = wave >> 2;
        const uint16_t* la = lds + (q % NSLOT) * SLOT + lane * 8;
        const uint16_t* lw = la + 16 * FRAG;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int i = 0; i < 2; ++i) { const u32x4 v = *reinterpret_cast<const u32x4*>(la + ((wm * 2 + i) * 2 + h) * FRAG); acc ^= v.x ^ v.w; }
#pragma unroll
          for (int j = 0; j < 4; ++j) { const u32x4 v = *reinterpret_cast<const u32x4*>(lw + ((wn * 4 + j) * 2 + h) * FRAG); acc ^= v.y ^ v.z; }
        }
      }
      if (q + 3 < nq) issue(q + 3);
    }
  } else if (VAR == 1) {
    const int np = K >> 6;
    issue(0); issue(1);
    for (int p = 0; p < np; ++p) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (p + 1 < np) { issue(2 * p + 2); issue(2 * p + 3); }
    }
  } else if (VAR == 2) {
    u32x4 r[2][4];
    auto ld = [&](int q, u32x4* v) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        v[h * 2 + 0] = *reinterpret_cast<const u32x4*>(ag + ((size_t)q * 2 + h) * FRAG);
        v[h * 2 + 1] = *reinterpret_cast<const u32x4*>(wg + ((size_t)q * 2 + h) * FRAG);
      }
    };
    auto stw = [&](int q, const u32x4* v) {
      uint16_t* la = lds + (q % NSLOT) * SLOT + wave * 2 * FRAG + lane * 8;
      uint16_t* lw = la + 16 * FRAG;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        *reinterpret_cast<u32x4*>(la + h * FRAG) = v[h * 2 + 0];
        *reinterpret_cast<u32x4*>(lw + h * FRAG) = v[h * 2 + 1];
      }
    };
    ld(0, r[0]); if (nq > 1) ld(1, r[1]);
    for (int q = 0; q < nq; q += 2) {
      stw(q, r[0]); if (q + 2 < nq) ld(q + 2, r[0]);
      if (q + 1 < nq) { stw(q + 1, r[1]); if (q + 3 < nq) ld(q + 3, r[1]); }
    }
  }
  __syncthreads();
  acc ^= lds[(tid * 8) & (NSLOT * SLOT - 1)];
  if (acc == 0x9e3779b9u) sink[0] = 1;
}

// variant 4: 256 threads, 128 x 256 tile, 24 KiB slots (4 A + 8 W row tiles x 2 fragments), ring of 3 = 72 KiB -> two workgroups per CU
__global__ __launch_bounds__(256, 2) void fill2_k(const uint16_t* __restrict__ Ap, const uint16_t* __restrict__ Wp, int M, int N, int K, unsigned* __restrict__ sink) {
  constexpr int SLOT = 24 * FRAG, NSLOT = 3;
  __shared__ __attribute__((aligned(16))) uint16_t lds[NSLOT * SLOT];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nx = N / 256, ny = M / 128, T = nx * ny, per = (T + 7) / 8;
  const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || t >= T) return;
  const int m0 = (t / nx) * 128, n0 = (t % nx) * 256, kb16 = K >> 4, nq = K >> 5;
  const uint16_t* ag = Ap + ((size_t)((m0 >> 5) + wave) * kb16) * FRAG + lane * 8;
  const uint16_t* wg0 = Wp + ((size_t)((n0 >> 5) + wave) * kb16) * FRAG + lane * 8;
  const uint16_t* wg1 = Wp + ((size_t)((n0 >> 5) + 4 + wave) * kb16) * FRAG + lane * 8;
  auto issue = [&](int q) {   // 6 pieces per wave
    uint16_t* la = lds + (q % NSLOT) * SLOT + wave * 2 * FRAG;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ag + ((size_t)q * 2 + h) * FRAG), (__attribute__((address_space(3))) void*)(la + h * FRAG), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wg0 + ((size_t)q * 2 + h) * FRAG), (__attribute__((address_space(3))) void*)(la + (8 + h) * FRAG), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wg1 + ((size_t)q * 2 + h) * FRAG), (__attribute__((address_space(3))) void*)(la + (16 + h) * FRAG), 16, 0, 0);
    }
  };
  issue(0); if (nq > 1) issue(1);
  for (int q = 0; q < nq; ++q) {
    if (q + 1 < nq) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (q + 2 < nq) issue(q + 2);
  }
  __syncthreads();
  if (lds[(tid * 8) % (NSLOT * SLOT)] == 0x9e37 && lds[3] == 0x79b9) sink[0] = 1;
}

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
  const int M = 65536, N = 2048, K = 512;
  uint16_t *A, *W; unsigned* sink;
  CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(A, 1, (size_t)M * K * 2)); CK(hipMemset(W, 2, (size_t)N * K * 2)); CK(hipMemset(sink, 0, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int tiles = (N / 256) * (M / 256);
  const double fill_bytes = (double)tiles * (K / 32) * 32768.0;
  const char* what[6] = {"LDS-DMA, 3 slots in flight, no barrier", "LDS-DMA, the kernel's stage protocol (vmcnt(0) + s_barrier per 64 of k)", "global_load -> VGPR -> ds_write_b128, no barrier",
                         "LDS-DMA + barrier + the fragment reads (no MFMA)", "LDS-DMA, 2 workgroups x 256 threads per CU, 128 x 256 tiles", "LDS-DMA, weight panel only (L2 resident)"};
  for (int v = 0; v < 6; ++v) {
    std::vector<double> ts;
    for (int rep = 0; rep < 12; ++rep) {
      CK(hipEventRecord(e0, 0));
      const dim3 g(((tiles + 7) / 8) * 8);
      if (v == 0) hipLaunchKernelGGL(fill_k<0>, g, dim3(512), 0, 0, A, W, M, N, K, sink);
      else if (v == 1) hipLaunchKernelGGL(fill_k<1>, g, dim3(512), 0, 0, A, W, M, N, K, sink);
      else if (v == 2) hipLaunchKernelGGL(fill_k<2>, g, dim3(512), 0, 0, A, W, M, N, K, sink);
      else if (v == 3) hipLaunchKernelGGL(fill_k<3>, g, dim3(512), 0, 0, A, W, M, N, K, sink);
      else if (v == 5) hipLaunchKernelGGL(fill_k<5>, g, dim3(512), 0, 0, A, W, M, N, K, sink);
      else hipLaunchKernelGGL(fill2_k, dim3(((2 * tiles + 7) / 8) * 8), dim3(256), 0, 0, A, W, M, N, K, sink);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 2) ts.push_back(ms * 1e3);
    }
    const double us = median(ts), bytes = v == 4 ? fill_bytes * 1.5 : fill_bytes;
    printf("variant %d  %-78s %8.1f us  fill %7.1f MB -> %6.2f TB/s over the chip = %5.1f GB/s per CU  (gemm_h1p_k: 268 us with the MFMAs and the epilogue)\n",
           v, what[v], us, bytes / 1e6, bytes / us / 1e6, bytes / us / 1e3 / 256);
  }
  return 0;
}
