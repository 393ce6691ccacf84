// Acoustic-decoder kernels for gfx950 that are not GEMMs (channels-last [B, F, C] layout):
//   dwconv_ln : ConvNeXt depthwise Conv1d(k7, dilation d, zero pad 3d) + LayerNorm(C = 512 | 256, eps 1e-6)
//               (/root/reference/ChatTTS/model/dvae.py:26-35,49-52 and vocos.modules.ConvNeXtBlock)
//   layernorm : LayerNorm(512) (VocosBackbone.norm / final_layer_norm)
//   istft     : ISTFTHead tail -- mag = min(exp(.), 1e2), S = mag (cos p + i sin p), per-frame
//               irfft(1024) x hann window, overlap-add, / window envelope, trim 512 each side
//               (/root/reference/examples/onnx/exporter.py:395-404 + torch.istft(center=True)).
// All three are HBM-bound streaming kernels: one wave per frame, 8 contiguous channels per lane
// (two 16-byte loads), wave-shuffle reductions for the LayerNorm statistics.
#include "common.hpp"
#include "kernels.hpp"

// channels = 64 lanes x CPL: 512 (CPL 8: both ConvNeXt stacks of the decoder path) or 256 (CPL 4: the full DVAE's trunks)
// planes != null: the normalised row goes out as hi / lo bf16 planes in the fragment order of codec_gemm.hip (a lane's 8 channels are
// one 16-byte slot of each plane) instead of f32
template <int CPL, bool F16 = false>
__device__ __forceinline__ void ln_finish(float* y, const float* __restrict__ lw, const float* __restrict__ lb, float eps, int c0,
                                          float* __restrict__ dst, uint16_t* __restrict__ planes = nullptr, int row = 0) {
  constexpr int C = 64 * CPL;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) s += y[i];
  const float mean = wave_sum(s) * (1.0f / C);
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) { const float d = y[i] - mean; v = fmaf(d, d, v); }   // explicit: the same bits in every kernel this is inlined into
  const float rstd = 1.0f / sqrtf(wave_sum(v) * (1.0f / C) + eps);
#pragma unroll
  for (int q = 0; q < CPL / 4; ++q) {
    const float4 w = *reinterpret_cast<const float4*>(lw + c0 + 4 * q), b = *reinterpret_cast<const float4*>(lb + c0 + 4 * q);
    float4 o;
    o.x = fmaf((y[4 * q + 0] - mean) * rstd, w.x, b.x); o.y = fmaf((y[4 * q + 1] - mean) * rstd, w.y, b.y);
    o.z = fmaf((y[4 * q + 2] - mean) * rstd, w.z, b.z); o.w = fmaf((y[4 * q + 3] - mean) * rstd, w.w, b.w);
    if (planes == nullptr) *reinterpret_cast<float4*>(dst + c0 + 4 * q) = o;
    else { y[4 * q] = o.x; y[4 * q + 1] = o.y; y[4 * q + 2] = o.z; y[4 * q + 3] = o.w; }
  }
  if constexpr (CPL == 8) {
   if (planes != nullptr && row >= 0) {   // row < 0 (ln_pack): the caller packs and stores
    constexpr int KB16 = C / 16;
    if constexpr (F16) {   // one fp16 plane (codec_gemm.hip: gemm_h1p_k): a lane's 8 channels are one 16-byte slot of it
      const size_t o = (((size_t)(row >> 5) * KB16 + (c0 >> 4)) * 64 + (((c0 & 15) >> 3) << 5) + (row & 31)) * 8;
      *reinterpret_cast<uint4*>(planes + o) = make_uint4(pack_f16x2(y[0], y[1]), pack_f16x2(y[2], y[3]), pack_f16x2(y[4], y[5]), pack_f16x2(y[6], y[7]));
      return;
    }
    uint32_t h[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      h[q] = pack_bf16x2(y[2 * q], y[2 * q + 1]);
      l[q] = pack_bf16x2(y[2 * q] - __uint_as_float(h[q] << 16), y[2 * q + 1] - __uint_as_float(h[q] & 0xffff0000u));
    }
    const size_t o = ((((size_t)(row >> 5) * KB16 + (c0 >> 4)) * 2) * 64 + (((c0 & 15) >> 3) << 5) + (row & 31)) * 8;
    *reinterpret_cast<uint4*>(planes + o) = make_uint4(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<uint4*>(planes + o + 512) = make_uint4(l[0], l[1], l[2], l[3]);
   }
  }
}

// w is the depthwise kernel transposed on the host to [7][C] so that a lane's CPL channels are contiguous
template <int CPL, bool F16 = false>
__global__ __launch_bounds__(256) void dwconv_ln_k(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                   const float* __restrict__ lw, const float* __restrict__ lb, float eps, int dil,
                                                   float* __restrict__ y, int F, int rows, uint16_t* __restrict__ yp) {
  constexpr int C = 64 * CPL;
  // XCD-aware frame order: workgroup L runs on XCD L % 8 (observed placement, speed only) and each XCD has its own L2.  The 7
  // taps of a frame are the rows f - 3 dil .. f + 3 dil; with the plain order every XCD touches every 8th group of 4 frames and
  // pulls all their neighbours from HBM itself (PMC: 2.2x the algorithmic bytes).  Here each XCD gets ONE contiguous run of
  // frames, so the neighbouring rows are hits in its own L2.  The grid is padded to a multiple of 8 workgroups.
  const int per = gridDim.x >> 3;
  const int blk = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const int row = blk * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63, c0 = lane * CPL;
  const int bi = row / F, f = row - bi * F;
  float acc[CPL];
#pragma unroll
  for (int q = 0; q < CPL / 4; ++q) {
    const float4 b0 = *reinterpret_cast<const float4*>(b + c0 + 4 * q);
    acc[4 * q] = b0.x; acc[4 * q + 1] = b0.y; acc[4 * q + 2] = b0.z; acc[4 * q + 3] = b0.w;
  }
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int fs = f + (j - 3) * dil;
    if (fs >= 0 && fs < F) {
      const float* xp = x + ((size_t)bi * F + fs) * C + c0;
#pragma unroll
      for (int q = 0; q < CPL / 4; ++q) {
        const float4 x0 = *reinterpret_cast<const float4*>(xp + 4 * q);
        const float4 w0 = *reinterpret_cast<const float4*>(w + j * C + c0 + 4 * q);
        acc[4 * q] = fmaf(w0.x, x0.x, acc[4 * q]); acc[4 * q + 1] = fmaf(w0.y, x0.y, acc[4 * q + 1]);
        acc[4 * q + 2] = fmaf(w0.z, x0.z, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(w0.w, x0.w, acc[4 * q + 3]);
      }
    }
  }
  ln_finish<CPL, F16>(acc, lw, lb, eps, c0, y + (size_t)row * C, yp, row);
}

// Large batches (round 3): the kernel above gives every frame its own wave, so each input row is requested by the 7 waves whose taps
// touch it -- 0.94 GB of L2 reads per launch at 65,536 frames for 134 MB of input, 130 us where HBM would need 40.  Here a wave walks
// RUN frames f0, f0 + dil, f0 + 2 dil, ... of one utterance (a run of RUN * dil consecutive frames is shared by `dil` waves, one per
// phase) and keeps the 7 rows of its current frame in a register ring of 9 (two rows requested ahead): every row is loaded once per
// wave that needs it (+ 6 / RUN halo), the depthwise weights (56 registers) and the bias once per wave instead of once per frame.
// Same arithmetic per frame as dwconv_ln_k: acc = bias, taps 0..6 in order (a tap outside [0, F) multiplies a zero row).
template <int CPL, bool F16>
__global__ __launch_bounds__(256) void dwconv_ln_run_k(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                       const float* __restrict__ lw, const float* __restrict__ lb, float eps, int dil,
                                                       float* __restrict__ y, int F, int B, int runs, uint16_t* __restrict__ yp) {
  constexpr int C = 64 * CPL;
  constexpr int RUN = 36, NSL = 9;   // RUN % NSL == 0: the ring indices below are compile-time constants
  const int per = gridDim.x >> 3;    // XCD-aware order: one contiguous range of (utterance, run, phase) per XCD (see dwconv_ln_k)
  const int blk = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const int gw = blk * 4 + (threadIdx.x >> 6);
  const int wpu = runs * dil;        // waves per utterance
  if (gw >= B * wpu) return;
  const int bi = gw / wpu, rem = gw - bi * wpu;
  const int run = rem / dil, ph = rem - run * dil;
  const int f0 = run * (RUN * dil) + ph;
  if (f0 >= F) return;
  const int lane = threadIdx.x & 63, c0 = lane * CPL;
  float wt[7][CPL], bias[CPL];
#pragma unroll
  for (int q = 0; q < CPL / 4; ++q) {
    const float4 b0 = *reinterpret_cast<const float4*>(b + c0 + 4 * q);
    bias[4 * q] = b0.x; bias[4 * q + 1] = b0.y; bias[4 * q + 2] = b0.z; bias[4 * q + 3] = b0.w;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const float4 w0 = *reinterpret_cast<const float4*>(w + j * C + c0 + 4 * q);
      wt[j][4 * q] = w0.x; wt[j][4 * q + 1] = w0.y; wt[j][4 * q + 2] = w0.z; wt[j][4 * q + 3] = w0.w;
    }
  }
  const float* xb = x + (size_t)bi * F * C + c0;
  float s[NSL][CPL];
  auto ldrow = [&](float* dst, int t_rel) {   // row f0 + dil * t_rel, zeros outside the utterance (Conv1d zero padding)
    const int r = f0 + dil * t_rel;
    const bool ok = r >= 0 && r < F;
    const float* xp = xb + (size_t)min(max(r, 0), F - 1) * C;
#pragma unroll
    for (int q = 0; q < CPL / 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(xp + 4 * q);
      dst[4 * q] = ok ? v.x : 0.f; dst[4 * q + 1] = ok ? v.y : 0.f; dst[4 * q + 2] = ok ? v.z : 0.f; dst[4 * q + 3] = ok ? v.w : 0.f;
    }
  };
  // slot (t + k) % NSL holds row t - 3 + k (k = 0..8): the 7 taps of frame t are k = 0..6, k = 7, 8 are on their way
#pragma unroll
  for (int k = 0; k < NSL - 1; ++k) ldrow(s[k], k - 3);
  for (int t0 = 0; t0 < RUN; t0 += NSL) {
#pragma unroll
    for (int u = 0; u < NSL; ++u) {
      const int t = t0 + u, f = f0 + dil * t;
      if (f >= F) return;                     // wave-uniform
      ldrow(s[(u + NSL - 1) % NSL], t + 5);   // row t + 5 = k 8 of this frame
      float acc[CPL];
#pragma unroll
      for (int i = 0; i < CPL; ++i) acc[i] = bias[i];
#pragma unroll
      for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int i = 0; i < CPL; ++i) acc[i] = fmaf(wt[j][i], s[(u + j) % NSL][i], acc[i]);
      const int row = bi * F + f;
      ln_finish<CPL, F16>(acc, lw, lb, eps, c0, y + (size_t)row * C, yp, row);
    }
  }
}

// Round 6 (third session): the planes' WRITES.  dwconv_ln_run_k hands every normalised row to the GEMMs' fragment order as 64 isolated
// 16-byte stores per plane (a lane's 8 channels of one row; the other rows of a 128-byte line come from other iterations or other waves,
// microseconds apart).  Per-dispatch PMC (profiles/r6V_pmc_dwconv_per_dispatch.txt): 415 MB written per Vocos launch (dilation 1) and
// 231 MB per DVAE launch (dilation 2) for 134 MB of planes -- partial lines leave the L2 before their neighbours arrive.  Here a wave walks
// RUN CONSECUTIVE frames of one utterance (any dilation: the register ring holds rows t - 3 DIL .. t + 3 DIL + prefetch), parks the packed
// planes of FOUR consecutive rows in a wave-private LDS tile and writes them transposed: lanes 4 q .. 4 q + 3 hold rows r .. r + 3 of
// channel block q, so every store instruction covers 64 contiguous bytes per lane quad -- whole 64-byte requests.  Same arithmetic per
// frame (acc = bias, taps 0..6 in order, ln_finish's statistics): the planes are bit-identical to dwconv_ln_run_k's.
template <int CPL, bool F16>
__device__ __forceinline__ void ln_pack(float* y, const float* __restrict__ lw, const float* __restrict__ lb, float eps, int c0, uint4& hi, uint4& lo) {
  ln_finish<CPL, F16>(y, lw, lb, eps, c0, nullptr, reinterpret_cast<uint16_t*>(1), -1);   // planes != null, row < 0: normalise in place, no store
  if constexpr (F16) {
    hi = make_uint4(pack_f16x2(y[0], y[1]), pack_f16x2(y[2], y[3]), pack_f16x2(y[4], y[5]), pack_f16x2(y[6], y[7]));
    lo = hi;
  } else {
    uint32_t h[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      h[q] = pack_bf16x2(y[2 * q], y[2 * q + 1]);
      l[q] = pack_bf16x2(y[2 * q] - __uint_as_float(h[q] << 16), y[2 * q + 1] - __uint_as_float(h[q] & 0xffff0000u));
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
  }
}

template <bool F16, int DIL>
__global__ __launch_bounds__(256) void dwconv_ln_seq_k(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                       const float* __restrict__ lw, const float* __restrict__ lb, float eps, int F, int B, int runs,
                                                       uint16_t* __restrict__ yp) {
  constexpr int CPL = 8, C = 512, KB16 = C / 16;
  constexpr int NSL = DIL == 1 ? 9 : 16, RUN = DIL == 1 ? 36 : 48;   // ring slots (6 DIL + 1 taps' span + 2 | 3 rows ahead); RUN % NSL == 0, RUN % 4 == 0
  constexpr int NP = F16 ? 1 : 2, RS = 65;                           // planes; LDS row stride in 16-byte units (65: the transposed reads hit distinct banks)
  __shared__ uint4 stg[4][NP][4][RS];
  const int per = gridDim.x >> 3;    // XCD-aware order: one contiguous range of (utterance, run) per XCD (see dwconv_ln_k)
  const int blk = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const int wave = threadIdx.x >> 6, gw = blk * 4 + wave;
  if (gw >= B * runs) return;
  const int bi = gw / runs, run = gw - bi * runs;
  const int f0 = run * RUN;
  if (f0 >= F) return;
  const int lane = threadIdx.x & 63, c0 = lane * CPL;
  float wt[7][CPL], bias[CPL];
#pragma unroll
  for (int q = 0; q < CPL / 4; ++q) {
    const float4 b0 = *reinterpret_cast<const float4*>(b + c0 + 4 * q);
    bias[4 * q] = b0.x; bias[4 * q + 1] = b0.y; bias[4 * q + 2] = b0.z; bias[4 * q + 3] = b0.w;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const float4 w0 = *reinterpret_cast<const float4*>(w + j * C + c0 + 4 * q);
      wt[j][4 * q] = w0.x; wt[j][4 * q + 1] = w0.y; wt[j][4 * q + 2] = w0.z; wt[j][4 * q + 3] = w0.w;
    }
  }
  const float* xb = x + (size_t)bi * F * C + c0;
  float s[NSL][CPL];
  auto ldrow = [&](float* dst, int t_rel) {   // row f0 + t_rel, zeros outside the utterance (Conv1d zero padding)
    const int r = f0 + t_rel;
    const bool ok = r >= 0 && r < F;
    const float* xp = xb + (size_t)min(max(r, 0), F - 1) * C;
#pragma unroll
    for (int q = 0; q < CPL / 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(xp + 4 * q);
      dst[4 * q] = ok ? v.x : 0.f; dst[4 * q + 1] = ok ? v.y : 0.f; dst[4 * q + 2] = ok ? v.z : 0.f; dst[4 * q + 3] = ok ? v.w : 0.f;
    }
  };
  // the four parked rows go out transposed: lane 4 q + j writes row j of channel block 16 i + q (i = 0..3)
  const int fj = lane & 3, fq = lane >> 2;
  auto flush = [&](int row0, int n) {   // rows row0 .. row0 + n - 1 (global row index) are parked in slots 0 .. n - 1
    if (fj < n) {
      const int row = row0 + fj;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int bk = 16 * i + fq;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const uint4 v = stg[wave][p][fj][bk];
          const size_t o = F16 ? (((size_t)(row >> 5) * KB16 + (bk >> 1)) * 64 + ((bk & 1) << 5) + (row & 31)) * 8
                               : ((((size_t)(row >> 5) * KB16 + (bk >> 1)) * 2 + p) * 64 + ((bk & 1) << 5) + (row & 31)) * 8;
          *reinterpret_cast<uint4*>(yp + o) = v;
        }
      }
    }
  };
  // slot (t + k) % NSL holds row t - 3 DIL + k: the 7 taps of frame t are k = 0, DIL, .., 6 DIL; the slots behind them are on their way
#pragma unroll
  for (int k = 0; k < NSL - 1; ++k) ldrow(s[k], k - 3 * DIL);
  for (int t0 = 0; t0 < RUN; t0 += NSL) {
#pragma unroll
    for (int u = 0; u < NSL; ++u) {
      const int t = t0 + u, f = f0 + t;
      if (f >= F) {                             // wave-uniform: the utterance ends inside this run
        if (t & 3) flush(bi * F + f - (t & 3), t & 3);
        return;
      }
      ldrow(s[(u + NSL - 1) % NSL], t - 3 * DIL + NSL - 1);
      float acc[CPL];
#pragma unroll
      for (int i = 0; i < CPL; ++i) acc[i] = bias[i];
#pragma unroll
      for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int i = 0; i < CPL; ++i) acc[i] = fmaf(wt[j][i], s[(u + j * DIL) % NSL][i], acc[i]);
      uint4 hi, lo;
      ln_pack<CPL, F16>(acc, lw, lb, eps, c0, hi, lo);
      stg[wave][0][t & 3][lane] = hi;
      if (!F16) stg[wave][NP - 1][t & 3][lane] = lo;
      if ((t & 3) == 3) flush(bi * F + f - 3, 4);
    }
  }
}

hipError_t launch_dwconv_ln(const float* x, const float* w, const float* b, const float* ln_w, const float* ln_b, float eps, int dil,
                            float* y, int B, int F, int C, hipStream_t st, uint16_t* yp, int plane_f16) {
  const int rows = B * F;
  const int nblk = ((rows + 3) / 4 + 7) / 8 * 8;   // multiple of 8: one contiguous run of frames per XCD (see the kernel)
  if (yp != nullptr && C != 512) return hipErrorInvalidValue;
  static int run_min = -1;   // CTTS_DWCONV_RUN_MIN_ROWS: frames from which the sliding-window kernel is used (0 = never); below it one wave per frame
  if (run_min < 0) { const char* e = getenv("CTTS_DWCONV_RUN_MIN_ROWS"); run_min = e ? atoi(e) : 12288; }
  if (C == 512 && run_min > 0 && rows >= run_min && dil >= 1 && dil <= 4) {
    // CTTS_DWCONV_SEQ (A/B, the bit-identity test; read at every launch): 0 = dwconv_ln_run_k's isolated 16-byte plane stores everywhere; 1 (default) =
    // dwconv_ln_seq_k at dilation 1 (Vocos: 415 -> 141 MB written, 148 -> 92 us per launch at 65,536 frames); 2 = also at dilation 2 (DVAE: 231 -> 169 MB
    // written but 107 -> 110 us: the 16-row register ring and its halo cost what the writes save; profiles/r6W_dwconv_ab.txt)
    const char* es = getenv("CTTS_DWCONV_SEQ");
    const int seq = es ? atoi(es) : 1;
    if (yp != nullptr && ((seq >= 1 && dil == 1) || (seq >= 2 && dil == 2))) {
      const int run = dil == 1 ? 36 : 48, nruns = (F + run - 1) / run;
      const int nbs = ((B * nruns + 3) / 4 + 7) / 8 * 8;
      if (dil == 1) {
        if (plane_f16) hipLaunchKernelGGL((dwconv_ln_seq_k<true, 1>), dim3(nbs), dim3(256), 0, st, x, w, b, ln_w, ln_b, eps, F, B, nruns, yp);
        else hipLaunchKernelGGL((dwconv_ln_seq_k<false, 1>), dim3(nbs), dim3(256), 0, st, x, w, b, ln_w, ln_b, eps, F, B, nruns, yp);
      } else {
        if (plane_f16) hipLaunchKernelGGL((dwconv_ln_seq_k<true, 2>), dim3(nbs), dim3(256), 0, st, x, w, b, ln_w, ln_b, eps, F, B, nruns, yp);
        else hipLaunchKernelGGL((dwconv_ln_seq_k<false, 2>), dim3(nbs), dim3(256), 0, st, x, w, b, ln_w, ln_b, eps, F, B, nruns, yp);
      }
      return hipGetLastError();
    }
    const int runs = (F + 36 * dil - 1) / (36 * dil);
    const int waves = B * runs * dil;
    const int nb = ((waves + 3) / 4 + 7) / 8 * 8;
    if (yp != nullptr && plane_f16) hipLaunchKernelGGL((dwconv_ln_run_k<8, true>), dim3(nb), dim3(256), 0, st, x, w, b, ln_w, ln_b, eps, dil, y, F, B, runs, yp);
    else hipLaunchKernelGGL((dwconv_ln_run_k<8, false>), dim3(nb), dim3(256), 0, st, x, w, b, ln_w, ln_b, eps, dil, y, F, B, runs, yp);
    return hipGetLastError();
  }
  if (C == 512 && yp != nullptr && plane_f16) hipLaunchKernelGGL((dwconv_ln_k<8, true>), dim3(nblk), dim3(256), 0, st, x, w, b, ln_w, ln_b, eps, dil, y, F, rows, yp);
  else if (C == 512) hipLaunchKernelGGL(dwconv_ln_k<8>, dim3(nblk), dim3(256), 0, st, x, w, b, ln_w, ln_b, eps, dil, y, F, rows, yp);
  else if (C == 256) hipLaunchKernelGGL(dwconv_ln_k<4>, dim3(nblk), dim3(256), 0, st, x, w, b, ln_w, ln_b, eps, dil, y, F, rows, (uint16_t*)nullptr);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void layernorm_k(const float* __restrict__ x, const float* __restrict__ lw, const float* __restrict__ lb,
                                                   float eps, float* __restrict__ y, int rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63, c0 = lane * 8;
  const float* xp = x + (size_t)row * 512 + c0;
  const float4 x0 = *reinterpret_cast<const float4*>(xp), x1 = *reinterpret_cast<const float4*>(xp + 4);
  float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
  ln_finish<8>(v, lw, lb, eps, c0, y + (size_t)row * 512);
}

hipError_t launch_layernorm(const float* x, const float* w, const float* b, float eps, float* y, int rows, int C, hipStream_t st) {
  if (C != 512) return hipErrorInvalidValue;
  hipLaunchKernelGGL(layernorm_k, dim3((rows + 3) / 4), dim3(256), 0, st, x, w, b, eps, y, rows);
  return hipGetLastError();
}

// ---- ISTFT -----------------------------------------------------------------------------------
#define NFFT 1024
#define NBIN 513
#define HOP 256

// one workgroup per frame: build the Hermitian-extended spectrum in LDS, 1024-point inverse complex
// radix-2 DIT FFT (10 stages, 512 butterflies per stage, 2 per thread), keep the real part, scale by
// window / N.  twiddle[k] = (cos, sin)(2 pi k / 1024), k < 512, computed in double on the host.
__global__ __launch_bounds__(256) void istft_frames_k(const float* __restrict__ head, const float* __restrict__ window,
                                                      const float2* __restrict__ tw, float* __restrict__ frames) {
  __shared__ float re[NFFT], im[NFFT];
  const int fr = blockIdx.x, t = threadIdx.x;
  const float* hp = head + (size_t)fr * (2 * NBIN);
  for (int k = t; k < NBIN; k += 256) {
    const float mag = fminf(expf(hp[k]), 100.0f);
    const float ph = hp[NBIN + k];
    float xr = mag * cosf(ph), xi = mag * sinf(ph);
    if (k == 0 || k == NFFT / 2) xi = 0.f;  // c2r ignores the imaginary part of DC and Nyquist
    const int r0 = __brev((unsigned)k) >> 22;  // 10-bit bit reversal
    re[r0] = xr; im[r0] = xi;
    if (k > 0 && k < NFFT / 2) {
      const int r1 = __brev((unsigned)(NFFT - k)) >> 22;
      re[r1] = xr; im[r1] = -xi;
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int s = 1; s <= 10; ++s) {
    const int half = 1 << (s - 1);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int bt = t + 256 * u;
      const int pos = bt & (half - 1);
      const int i = ((bt >> (s - 1)) << s) + pos, j = i + half;
      const float2 w = tw[pos << (10 - s)];  // e^{+2 pi i pos / 2^s}
      const float ur = re[i], ui = im[i];
      const float vr = re[j] * w.x - im[j] * w.y, vi = re[j] * w.y + im[j] * w.x;
      re[i] = ur + vr; im[i] = ui + vi;
      re[j] = ur - vr; im[j] = ui - vi;
    }
    __syncthreads();
  }
  float* fp = frames + (size_t)fr * NFFT;
  for (int n = t; n < NFFT; n += 256) fp[n] = re[n] * (1.0f / NFFT) * window[n];
}

// overlap-add + envelope division + center trim:  wav[b, n], n in [0, HOP*(F-1))
__global__ __launch_bounds__(256) void istft_ola_k(const float* __restrict__ frames, const float* __restrict__ window,
                                                   float* __restrict__ wav, int F, int wlen) {
  const int b = blockIdx.y;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= wlen) return;
  const int tt = n + NFFT / 2;
  int f_hi = tt / HOP; if (f_hi > F - 1) f_hi = F - 1;
  int f_lo = (tt - (NFFT - 1) + HOP - 1) / HOP; if (f_lo < 0) f_lo = 0;
  float y = 0.f, env = 0.f;
  for (int f = f_lo; f <= f_hi; ++f) {
    const int o = tt - f * HOP;
    y += frames[((size_t)b * F + f) * NFFT + o];
    const float w = window[o];
    env += w * w;
  }
  wav[(size_t)b * wlen + n] = y / env;
}

hipError_t launch_istft(const float* head, const float* window, const float* twiddle, float* frames, float* wav, int B, int F,
                        hipStream_t st) {
  if (F < 2) return hipErrorInvalidValue;
  hipLaunchKernelGGL(istft_frames_k, dim3(B * F), dim3(256), 0, st, head, window, (const float2*)twiddle, frames);
  const int wlen = HOP * (F - 1);
  hipLaunchKernelGGL(istft_ola_k, dim3((wlen + 255) / 256, B), dim3(256), 0, st, frames, window, wav, F, wlen);
  return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// Shader copy: n bytes (a multiple of 16, both pointers 16-byte aligned) src -> dst.  The destination may be PINNED HOST memory (it
// is mapped into the device's address space): the result tensors of the path then reach the host without the copy engines, as
// plain stores over PCIe.  (`Chat._decode_to_wavs` ends with `.cpu().numpy()`, core.py:508-510.)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy16_k(const u128* __restrict__ src, u128* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
hipError_t launch_copy16(const void* src, void* dst, size_t bytes, hipStream_t st) {
  const size_t n16 = bytes / 16;
  if (n16 == 0) return hipSuccess;
  const size_t blocks = (n16 + 255) / 256;
  hipLaunchKernelGGL(copy16_k, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, st, (const u128*)src, (u128*)dst, n16);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// float32 waveform -> 16-bit PCM on the device (round 5; SURVEY 8f-3, /root/reference/tools/audio/np.py:7-11 `float_to_int16`):
//     am = 32767 * 32768 // (int(ceil(max |x|)) * 32768);   out = (x * am).astype(int16)      (truncation toward zero)
// The peak is taken over the whole [rows, n] array (how examples/cmd/stream.py:44 applies it to a streamed block) or per row (one call
// per utterance: examples/web/funcs.py:206-209, tools/audio/pcm.py:29).  The product is formed the way the reference's RUNTIME forms it:
// the function is numba-jitted, and numba types `float32[:] * int64` as float64 -- exact for |am| < 2^15 -- (`product` 0); `product` 1
// is what plain NumPy >= 2 does with the same source line (python int = weak scalar: a float32 product, rounded BEFORE the truncation).
// Optionally also writes one bit per sample, |x| > keep_thr: the mask of `wav[np.abs(wav) > 1e-5]` (core.py:262-265), so that the
// silence strip of Chat.infer can run on the int16 samples -- the waveform then leaves the GPU at 2 bytes + 1 bit per sample, not 4 bytes.
// An all-zero input gives zeros (the reference divides by zero there).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void absmax_rows_k(const float* __restrict__ x, long long n, long long ld, int per_row, unsigned* __restrict__ peak) {
  const int row = blockIdx.y;
  const float* xr = x + (size_t)row * ld;
  unsigned m = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    m = max(m, __float_as_uint(xr[i]) & 0x7fffffffu);   // |x| as an unsigned: the IEEE order of non-negative floats
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0 && m != 0) atomicMax(peak + (per_row ? row : 0), m);
}
__global__ __launch_bounds__(256) void pcm16_k(const float* __restrict__ x, long long n, long long ld, int per_row, int product,
                                               const unsigned* __restrict__ peak, float keep_thr, int16_t* __restrict__ out,
                                               uint8_t* __restrict__ keep) {
  const int row = blockIdx.y;
  const float pk = __uint_as_float(peak[per_row ? row : 0]);
  // a non-finite sample makes the peak Inf / NaN (the bit pattern of |x| orders above every finite value): the cast would be undefined
  // behaviour -- such a row gets scale 0, i.e. zeros (the reference's numpy path yields garbage there)
  const long long c = (pk < 3.0e38f) ? (long long)ceilf(pk) : 0;
  const long long am = c > 0 ? (32767ll * 32768ll) / (c * 32768ll) : 0;
  const float* xr = x + (size_t)row * ld;
  int16_t* orow = out + (size_t)row * n;
  const long long nb = (n + 7) >> 3;                     // groups of 8 samples = 16 bytes of PCM = one byte of mask
  for (long long gi = (long long)blockIdx.x * 256 + threadIdx.x; gi < nb; gi += (long long)gridDim.x * 256) {
    unsigned bits = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const long long i = gi * 8 + e;
      if (i < n) {
        const float v = xr[i];
        const int q = product == 0 ? (int)((double)v * (double)am) : (int)(v * (float)am);
        orow[i] = (int16_t)q;
        if (fabsf(v) > keep_thr) bits |= 0x80u >> e;      // np.packbits order: first sample in the top bit
      }
    }
    if (keep != nullptr) keep[(size_t)row * nb + gi] = (uint8_t)bits;
  }
}
hipError_t launch_float_to_int16(const float* wav, long long n, long long ld, int rows, int per_row, int product, float keep_thr,
                                 unsigned* peak, int16_t* pcm, uint8_t* keep, hipStream_t st) {
  if (rows <= 0 || n <= 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(peak, 0, sizeof(unsigned) * (per_row ? rows : 1), st);
  if (e != hipSuccess) return e;
  const long long per_blk = 256ll * 16;
  const unsigned gx = (unsigned)min((n + per_blk - 1) / per_blk, 1024ll);
  hipLaunchKernelGGL(absmax_rows_k, dim3(gx, rows), dim3(256), 0, st, wav, n, ld, per_row, peak);
  const long long nb = (n + 7) >> 3;
  const unsigned gx2 = (unsigned)min((nb + 255) / 256, 2048ll);
  hipLaunchKernelGGL(pcm16_k, dim3(gx2, rows), dim3(256), 0, st, wav, n, ld, per_row, product, (const unsigned*)peak, keep_thr, pcm, keep);
  return hipGetLastError();
}
