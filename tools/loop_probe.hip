// Micro-benchmark (round 6), the companion of tools/fill_probe.hip: the k loop of gemm_h1p_k (csrc/codec_gemm.hip) WITHOUT its epilogue, at
// the bench's roofline shape (M 65,536, N 2048, K 512; fp16 fragment planes), to find out where the loop's 1.85 us per 64-wide stage go
// when the operand movement alone runs at 0.9 us per stage (fill_probe variant 1) and the MFMAs alone need 0.86:
//   0  the shipped loop: 512 threads, 256 x 256, stages of 64, vmcnt(0) + s_barrier per stage, reads one 16-wide block ahead
//   1  as 0 without the fragment reads (MFMAs on whatever the registers hold)           -> is it the LDS reads?
//   2  as 0 without the MFMAs (reads only)                                              -> is it the MFMAs?
//   3  as 0 without the DMA (operands read from LDS that is never refilled; no vmcnt)   -> the loop with free operands
//   4  256 threads, 128 x 256 tile, ring of 3 x 24 KiB, one barrier per 32 of k, TWO workgroups per CU (their barriers are independent)
//   5  as 4 with 64 x 256 wave tiles -> no: kept out (accumulators)
//   hipcc --offload-arch=gfx950 -O3 tools/loop_probe.hip -o /tmp/loop_probe && /tmp/loop_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int FRAG = 512;
#define SB() __builtin_amdgcn_sched_barrier(0)
#define GLL(g, l) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g), (__attribute__((address_space(3))) void*)(l), 16, 0, 0)

template <int VAR>
__global__ __launch_bounds__(512, 2) void loop_k(const uint16_t* __restrict__ Ap, const uint16_t* __restrict__ Wp, int M, int N, int K, float* __restrict__ sink) {
  constexpr int SLOT = 32 * FRAG, NSLOT = 4;
  constexpr bool RD = VAR != 1, MM = VAR != 2, DMA = VAR != 3;
  __shared__ __attribute__((aligned(16))) uint16_t lds[NSLOT * SLOT];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, wm = wave & 3, wn = wave >> 2;
  const int nx = N / 256, ny = M / 256, T = nx * ny, per = (T + 7) / 8;
  const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || t >= T) return;
  const int m0 = (t / nx) * 256, n0 = (t % nx) * 256, kb16 = K >> 4, np = K >> 6;
  const uint16_t* ag = Ap + ((size_t)((m0 >> 5) + wave) * kb16) * FRAG + lane * 8;
  const uint16_t* wg = Wp + ((size_t)((n0 >> 5) + wave) * kb16) * FRAG + lane * 8;
  auto issue = [&](int q) {
    if (!DMA) return;
    uint16_t* la = lds + (q % NSLOT) * SLOT + wave * 2 * FRAG;
    uint16_t* lw = la + 16 * FRAG;
#pragma unroll
    for (int h = 0; h < 2; ++h) { GLL(ag + ((size_t)q * 2 + h) * FRAG, la + h * FRAG); GLL(wg + ((size_t)q * 2 + h) * FRAG, lw + h * FRAG); }
  };
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  f16x8 fa0[2], fw0[4], fa1[2], fw1[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) { fa0[i] = (f16x8)(_Float16)(lane * 0.001f); fa1[i] = fa0[i]; }
#pragma unroll
  for (int j = 0; j < 4; ++j) { fw0[j] = (f16x8)(_Float16)(lane * 0.002f); fw1[j] = fw0[j]; }
  auto rd = [&](f16x8* fa, f16x8* fw, int u) {
    if (!RD) return;
    const uint16_t* la = lds + ((u >> 1) % NSLOT) * SLOT + lane * 8;
    const uint16_t* lw = la + 16 * FRAG;
    const int h = u & 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const f16x8*>(la + ((wm * 2 + i) * 2 + h) * FRAG);
#pragma unroll
    for (int j = 0; j < 4; ++j) fw[j] = *reinterpret_cast<const f16x8*>(lw + ((wn * 4 + j) * 2 + h) * FRAG);
  };
  auto mm_a = [&](const f16x8* fa, const f16x8* fw) { if (MM) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0], fw[0], acc[0][0], 0, 0, 0); else acc[0][0][0] += (float)fa[0][0] + (float)fw[0][1]; };
  auto mm_b = [&](const f16x8* fa, const f16x8* fw) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (i + j > 0) { if (MM) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fw[j], acc[i][j], 0, 0, 0); else acc[i][j][0] += (float)fa[i][2] + (float)fw[j][3]; }
  };
  issue(0); issue(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (np > 1) { issue(2); issue(3); }
  rd(fa0, fw0, 0);
  for (int p = 0; p < np; ++p) {
    const int u = 4 * p;
    SB(); mm_a(fa0, fw0); SB(); rd(fa1, fw1, u + 1); SB(); mm_b(fa0, fw0);
    SB(); mm_a(fa1, fw1); SB(); rd(fa0, fw0, u + 2); SB(); mm_b(fa1, fw1);
    SB(); mm_a(fa0, fw0); SB(); rd(fa1, fw1, u + 3); SB(); mm_b(fa0, fw0);
    SB(); mm_a(fa1, fw1); SB();
    if (p + 1 < np) {
      if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (p + 2 < np) { issue(2 * p + 4); issue(2 * p + 5); }
      rd(fa0, fw0, u + 4);
      SB();
    }
    mm_b(fa1, fw1);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 1.2345f) sink[0] = s;
}

// variant 4: 256 threads (4 waves in a row: wave w = rows 32 w .. of the 128, all 256 columns? no -- 2 x 2 waves of 64 x 128), 128 x 256
// tile; slot = 32 of k = (4 A + 8 W row tiles) x 2 fragments = 24 KiB; ring of 3 (72 KiB) -> two workgroups per CU; one barrier per slot,
// two slots in flight behind the one being multiplied
__device__ __forceinline__ float gelu_fast_p(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f); p = fmaf(p, t, -0.284496736f); p = fmaf(p, t, 0.254829592f);
  const float h = 0.5f * (p * t) * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
  return x * (x < 0.f ? h : 1.0f - h);
}
__device__ __forceinline__ uint32_t pk(float a, float b) { return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)a) | ((uint32_t)__builtin_bit_cast(uint16_t, (_Float16)b) << 16); }
// EP: 0 no epilogue | 1 GELU math only | 2 LDS transpose + pack + stores (no GELU) | 3 16 direct 16-byte stores per lane (no LDS, no GELU) | 4 all of it (the shipped epilogue)
template <int EP>
__global__ __launch_bounds__(256, 2) void loop2_k(const uint16_t* __restrict__ Ap, const uint16_t* __restrict__ Wp, int M, int N, int K, float* __restrict__ sink, uint16_t* __restrict__ Cp, const float* __restrict__ biasp, int stagger_ticks) {
  constexpr int SLOT = 24 * FRAG, NSLOT = 3;
  __shared__ __attribute__((aligned(16))) uint16_t lds[NSLOT * SLOT];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, wm = wave & 1, wn = wave >> 1;
  const int nx = N / 256, ny = M / 128, T = nx * ny, per = (T + 7) / 8;
  const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || t >= T) return;
  if (stagger_ticks > 0) {   // first generation only: the second resident workgroup of every CU starts half a tile late (100 MHz ticks)
    const int w = blockIdx.x >> 3;
    if (w >= 32 && w < 64) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < stagger_ticks) __builtin_amdgcn_s_sleep(32); }
  }
  const int m0 = (t / nx) * 128, n0 = (t % nx) * 256, kb16 = K >> 4, nq = K >> 5;
  const uint16_t* ag = Ap + ((size_t)((m0 >> 5) + wave) * kb16) * FRAG + lane * 8;
  const uint16_t* wg0 = Wp + ((size_t)((n0 >> 5) + wave) * kb16) * FRAG + lane * 8;
  const uint16_t* wg1 = Wp + ((size_t)((n0 >> 5) + 4 + wave) * kb16) * FRAG + lane * 8;
  auto issue = [&](int q) {   // 6 pieces per wave: A row tile `wave`, W row tiles `wave` and `4 + wave`
    uint16_t* l = lds + (q % NSLOT) * SLOT + wave * 2 * FRAG;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      GLL(ag + ((size_t)q * 2 + h) * FRAG, l + h * FRAG);
      GLL(wg0 + ((size_t)q * 2 + h) * FRAG, l + (8 + h) * FRAG);
      GLL(wg1 + ((size_t)q * 2 + h) * FRAG, l + (16 + h) * FRAG);
    }
  };
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  f16x8 fa0[2], fw0[4], fa1[2], fw1[4];
  auto rd = [&](f16x8* fa, f16x8* fw, int u) {   // 16-wide k block u: slot u / 2, half u % 2
    const uint16_t* l = lds + ((u >> 1) % NSLOT) * SLOT + lane * 8;
    const int h = u & 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const f16x8*>(l + ((wm * 2 + i) * 2 + h) * FRAG);
#pragma unroll
    for (int j = 0; j < 4; ++j) fw[j] = *reinterpret_cast<const f16x8*>(l + (8 + (wn * 4 + j) * 2 + h) * FRAG);
  };
  auto mm_a = [&](const f16x8* fa, const f16x8* fw) { acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0], fw[0], acc[0][0], 0, 0, 0); };
  auto mm_b = [&](const f16x8* fa, const f16x8* fw) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (i + j > 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fw[j], acc[i][j], 0, 0, 0);
  };
  issue(0); if (nq > 1) issue(1);
  if (nq > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (nq > 2) issue(2);
  rd(fa0, fw0, 0);
  for (int q = 0; q < nq; ++q) {
    SB(); mm_a(fa0, fw0); SB(); rd(fa1, fw1, 2 * q + 1); SB(); mm_b(fa0, fw0);
    SB(); mm_a(fa1, fw1); SB();
    if (q + 1 < nq) {
      // slot q + 1 must have landed; slot q + 2 may stay in flight
      if (q + 2 < nq) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();          // everybody has read slot q (their fragments are in registers): refill it with q + 3
      if (q + 3 < nq) issue(q + 3);
      rd(fa0, fw0, 2 * q + 2);
      SB();
    }
    mm_b(fa1, fw1);
  }
  __syncthreads();
  const int nb16 = N >> 4, mb = m0 + wm * 64, nb = n0 + wn * 128;
  if (EP == 1) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += gelu_fast_p(acc[i][j][r] + 0.1f);
    if (s == 1.2345f) sink[0] = s;
  } else if (EP == 2 || EP == 4) {
    float* scr = reinterpret_cast<float*>(lds) + wave * (32 * 36);
    float b4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b4[j] = biasp[nb + j * 32 + (lane & 31)];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cb = nb + j * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          scr[rr * 36 + (lane & 31)] = EP == 4 ? gelu_fast_p(acc[i][j][r] + b4[j]) : acc[i][j][r] + b4[j];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int item = lane + 64 * it, rr = item >> 2, cg = item & 3;
          const float4 v0 = *reinterpret_cast<const float4*>(scr + rr * 36 + cg * 8);
          const float4 v1 = *reinterpret_cast<const float4*>(scr + rr * 36 + cg * 8 + 4);
          const int row = mb + i * 32 + rr, k = cb + cg * 8;
          const size_t o = (((size_t)(row >> 5) * nb16 + (k >> 4)) * 64 + (((k & 15) >> 3) << 5) + (row & 31)) * 8 + (k & 7);
          *reinterpret_cast<uint4*>(Cp + o) = make_uint4(pk(v0.x, v0.y), pk(v0.z, v0.w), pk(v1.x, v1.y), pk(v1.z, v1.w));
        }
        __builtin_amdgcn_wave_barrier();
      }
  } else if (EP == 3) {
    // the same 16 KiB per wave, straight from the accumulators: lane-linear 16-byte stores, 1 KiB per store instruction
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const size_t o = ((((size_t)((mb >> 5) + i) * nb16 + ((nb + j * 32) >> 4) + h) * 64) + lane) * 8;
          *reinterpret_cast<uint4*>(Cp + o) = make_uint4(pk(acc[i][j][8 * h], acc[i][j][8 * h + 1]), pk(acc[i][j][8 * h + 2], acc[i][j][8 * h + 3]),
                                                          pk(acc[i][j][8 * h + 4], acc[i][j][8 * h + 5]), pk(acc[i][j][8 * h + 6], acc[i][j][8 * h + 7]));
        }
  } else {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 1.2345f) sink[0] = s;
  }
}

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
  const int M = 65536, N = 2048, K = 512;
  uint16_t *A, *W, *Cp; float *sink, *biasp;
  CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&Cp, (size_t)M * N * 2)); CK(hipMalloc(&biasp, N * 4)); CK(hipMemset(biasp, 0, N * 4));
  CK(hipMemset(A, 0x3c, (size_t)M * K * 2)); CK(hipMemset(W, 0x2c, (size_t)N * K * 2));   // fp16 1.06 and 0.065: real toggling in the MFMA CK(hipMemset(sink, 0, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int tiles = (N / 256) * (M / 256);
  const char* what[12] = {"shipped loop (512 threads, 256 x 256, stage 64)", "... without the fragment reads", "... without the MFMAs", "... without the DMA",
                         "256 threads, 128 x 256, ring 3 x 24 KiB, 2 workgroups per CU", "  + GELU arithmetic only", "  + LDS transpose, fp16 pack, stores (no GELU)",
                         "  + 16 direct 16-byte stores per lane (no LDS, no GELU)", "  + the whole GELU epilogue", "  + whole epilogue, second resident workgroup 6 us late",
                         "  + whole epilogue, second resident workgroup 12 us late", "  + whole epilogue, second resident workgroup 18 us late"};
  for (int v = 0; v < 12; ++v) {
    std::vector<double> ts;
    for (int rep = 0; rep < 12; ++rep) {
      CK(hipEventRecord(e0, 0));
      const dim3 g(((tiles + 7) / 8) * 8);
      if (v == 0) hipLaunchKernelGGL(loop_k<0>, g, dim3(512), 0, 0, A, W, M, N, K, sink);
      else if (v == 1) hipLaunchKernelGGL(loop_k<1>, g, dim3(512), 0, 0, A, W, M, N, K, sink);
      else if (v == 2) hipLaunchKernelGGL(loop_k<2>, g, dim3(512), 0, 0, A, W, M, N, K, sink);
      else if (v == 3) hipLaunchKernelGGL(loop_k<3>, g, dim3(512), 0, 0, A, W, M, N, K, sink);
      else if (v == 4) hipLaunchKernelGGL(loop2_k<0>, dim3(((2 * tiles + 7) / 8) * 8), dim3(256), 0, 0, A, W, M, N, K, sink, Cp, biasp, 0);
      else if (v == 5) hipLaunchKernelGGL(loop2_k<1>, dim3(((2 * tiles + 7) / 8) * 8), dim3(256), 0, 0, A, W, M, N, K, sink, Cp, biasp, 0);
      else if (v == 6) hipLaunchKernelGGL(loop2_k<2>, dim3(((2 * tiles + 7) / 8) * 8), dim3(256), 0, 0, A, W, M, N, K, sink, Cp, biasp, 0);
      else if (v == 7) hipLaunchKernelGGL(loop2_k<3>, dim3(((2 * tiles + 7) / 8) * 8), dim3(256), 0, 0, A, W, M, N, K, sink, Cp, biasp, 0);
      else if (v == 8) hipLaunchKernelGGL(loop2_k<4>, dim3(((2 * tiles + 7) / 8) * 8), dim3(256), 0, 0, A, W, M, N, K, sink, Cp, biasp, 0);
      else hipLaunchKernelGGL(loop2_k<4>, dim3(((2 * tiles + 7) / 8) * 8), dim3(256), 0, 0, A, W, M, N, K, sink, Cp, biasp, 600 * (v - 8));
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 2) ts.push_back(ms * 1e3);
    }
    const double us = median(ts);
    printf("variant %d  %-66s %8.1f us = %6.1f TFLOP/s = %.3f of 2.5 PF   (MFMA floor 55 us; gemm_h1p_k with its epilogue: 268 us)\n", v, what[v], us,
           2.0 * M * N * K / us * 1e-6, 2.0 * M * N * K / us * 1e-6 / 2500.0);
  }
  return 0;
}
