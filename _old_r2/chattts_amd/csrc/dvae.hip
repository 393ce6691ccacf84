// Full DVAE for gfx950: audio -> codes (`DVAE.forward(mode="encode")`, dvae.py:265-274) and
// codes -> mel through the GFSQ codebook (`use_decoder=False`, dvae.py:276-297).  Channels-last [B, F, C] like the
// decoder path; every dense layer runs on the f32-input MFMA tiles of gemm.hip (this path runs once per reference
// clip / per utterance, it is not on the decode roofline -- accuracy first), the ConvNeXt depthwise + LayerNorm on
// codec.hip's kernel.  Kernels of this file:
//   stft_mag     : torch.stft(center, reflect, onesided).abs() of one clip -- one workgroup per frame, 1024-point
//                  radix-2 FFT in LDS (the forward twin of codec.hip's istft_frames_k)
//   gfsq_encode  : GroupedResidualFSQ indices (restated algorithm, see oracle/dvae_np.py) -- one wave per frame
//   gfsq_embed   : indices -> features (`get_output_from_indices`)
#include <math.h>
#include <string.h>
#include <vector>

#include "../../include/chattts_amd.h"
#include "common.hpp"
#include "kernels.hpp"

#define NFFT 1024
#define NBIN 513
#define MAGLD 516
#define HOP 256

__global__ __launch_bounds__(256) void stft_mag_k(const float* __restrict__ wav, int n, const float* __restrict__ window,
                                                  const float2* __restrict__ tw, float* __restrict__ mag) {
  __shared__ float re[NFFT], im[NFFT];
  const int fr = blockIdx.x, t = threadIdx.x;
  for (int k = t; k < NFFT; k += 256) {
    int i = fr * HOP + k - NFFT / 2;          // index into the unpadded clip
    if (i < 0) i = -i;                        // reflect (no edge repeat), pad = 512 < n
    if (i >= n) i = 2 * (n - 1) - i;
    const int r = __brev((unsigned)k) >> 22;  // 10-bit bit reversal
    re[r] = wav[i] * window[k];
    im[r] = 0.f;
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
      const float2 w = tw[pos << (10 - s)];   // (cos, sin)(2 pi pos / 2^s); forward transform uses e^{-i.}
      const float ur = re[i], ui = im[i];
      const float vr = re[j] * w.x + im[j] * w.y, vi = im[j] * w.x - re[j] * w.y;
      re[i] = ur + vr; im[i] = ui + vi;
      re[j] = ur - vr; im[j] = ui - vi;
    }
    __syncthreads();
  }
  float* mp = mag + (size_t)fr * MAGLD;
  for (int k = t; k < MAGLD; k += 256) mp[k] = k < NBIN ? sqrtf(re[k] * re[k] + im[k] * im[k]) : 0.f;
}

hipError_t launch_stft_mag(const float* wav, int n, const float* window, const float* twiddle, float* mag, int F, hipStream_t st) {
  if (n <= NFFT / 2 || F != 1 + n / HOP) return hipErrorInvalidValue;
  hipLaunchKernelGGL(stft_mag_k, dim3(F), dim3(256), 0, st, wav, n, window, (const float2*)twiddle, mag);
  return hipGetLastError();
}

// ---- GFSQ ------------------------------------------------------------------------------------------------------
// FSQ of one 4-vector: bound(z) = tanh(z + shift) * half_l - offset (half_l = (L-1)(1+1e-3)/2; offset, shift = 0 for
// odd L), q = rint(.), code = q / (L/2), index = sum((q + L/2) * basis).
__device__ __forceinline__ float fsq_bound(float z, int L) {
  const float half_l = (float)(L - 1) * 1.001f * 0.5f;
  const float offset = (L & 1) ? 0.f : 0.5f;
  const float shift = (L & 1) ? 0.f : atanhf(offset / half_l);
  return tanhf(z + shift) * half_l - offset;
}

__global__ __launch_bounds__(256) void gfsq_encode_k(GfsqArgs q, const float* __restrict__ feat, int32_t* __restrict__ codes, int rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int D = q.D;
  for (int g = 0; g < q.G; ++g) {
    const float* x = feat + (size_t)row * q.G * D + (size_t)g * D;
    float z[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const float* w = q.in_w + ((size_t)g * 4 + d) * D;
      float s = 0.f;
      for (int c = lane * 4; c < D; c += 256) {
        const float4 xv = *reinterpret_cast<const float4*>(x + c), wv = *reinterpret_cast<const float4*>(w + c);
        s += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
      }
      z[d] = wave_sum(s) + q.in_b[g * 4 + d];
    }
    if (lane == 0) {
      float res[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) res[d] = q.bound_first ? fsq_bound(z[d], q.levels[d]) : z[d];
      for (int r = 0; r < q.R; ++r) {
        int idx = 0, basis = 1;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const int L = q.levels[d];
          const float scale = powf((float)(L - 1), (float)-r);     // (L-1)^-r, exact for the powers of two used here
          const float hw = (float)(L / 2);
          const float qv = rintf(fsq_bound(res[d] / scale, L));     // round half to even (torch.round)
          const float code = qv / hw;
          res[d] -= code * scale;
          idx += (int)(code * hw + hw) * basis;
          basis *= L;
        }
        codes[(size_t)row * q.G * q.R + g * q.R + r] = idx;
      }
    }
  }
}

__global__ __launch_bounds__(256) void gfsq_embed_k(GfsqArgs q, const int64_t* __restrict__ codes, float* __restrict__ feat, int rows) {
  const int row = blockIdx.x;
  const int D = q.D;
  for (int g = 0; g < q.G; ++g) {
    float zq[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < q.R; ++r) {
      long long idx = codes[(size_t)row * q.G * q.R + g * q.R + r];
      int basis = 1;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int L = q.levels[d];
        const int lv = (int)((idx / basis) % L);
        const float hw = (float)(L / 2);
        zq[d] += ((float)lv - hw) / hw * powf((float)(L - 1), (float)-r);
        basis *= L;
      }
    }
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
      const float4 w = *reinterpret_cast<const float4*>(q.out_w + ((size_t)g * D + c) * 4);
      feat[(size_t)row * q.G * D + (size_t)g * D + c] = (((zq[0] * w.x + zq[1] * w.y) + zq[2] * w.z) + zq[3] * w.w) + q.out_b[g * D + c];
    }
  }
}

hipError_t launch_gfsq_encode(const GfsqArgs& q, const float* feat, int32_t* codes, int rows, hipStream_t st) {
  if (q.D % 4 != 0 || q.G < 1 || q.R < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gfsq_encode_k, dim3((rows + 3) / 4), dim3(256), 0, st, q, feat, codes, rows);
  return hipGetLastError();
}
hipError_t launch_gfsq_embed(const GfsqArgs& q, const int64_t* codes, float* feat, int rows, hipStream_t st) {
  hipLaunchKernelGGL(gfsq_embed_k, dim3(rows), dim3(256), 0, st, q, codes, feat, rows);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// extern "C": ctts_dvae_* (include/chattts_amd.h)
// ---------------------------------------------------------------------------------------------------------------
#define CK(expr)                                                                                   \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) return ctts_fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

struct ctts_dvae {
  ctts_dvae_weights w;
  std::vector<const float*> e[9], d[9];
  GfsqArgs q;
};

static void copy_blocks(std::vector<const float*>* dst, const ctts_trunk_weights& t) {
  const float* const* src[9] = {t.dw_w, t.dw_b, t.ln_w, t.ln_b, t.pw1_w, t.pw1_b, t.pw2_w, t.pw2_b, t.gamma};
  for (int i = 0; i < 9; ++i) dst[i].assign(src[i], src[i] + t.n_blocks);
}

extern "C" int ctts_dvae_create(ctts_dvae** out, const ctts_dvae_weights* w) {
  if (!out || !w) return ctts_fail("ctts_dvae_create: bad arguments");
  for (const ctts_trunk_weights* t : {&w->encoder, &w->decoder})
    if ((t->hidden != 256 && t->hidden != 512) || t->idim % 4 || t->bn_dim % 4 || t->odim % 4 || t->n_blocks < 0)
      return ctts_fail("ctts_dvae_create: unsupported trunk dims");
  if (w->encoder.idim != 512 || w->decoder.idim != 512 || w->decoder.odim != 512 || w->G * w->D != w->encoder.odim || w->G * w->D != 1024)
    return ctts_fail("ctts_dvae_create: dims do not match DVAE(dim=512, vq dim 1024)");
  ctts_dvae* c = new ctts_dvae();
  c->w = *w;
  copy_blocks(c->e, w->encoder);
  copy_blocks(c->d, w->decoder);
  GfsqArgs& q = c->q;
  q.in_w = w->q_in_w; q.in_b = w->q_in_b; q.out_w = w->q_out_w; q.out_b = w->q_out_b;
  for (int i = 0; i < 4; ++i) q.levels[i] = w->levels[i];
  q.G = w->G; q.R = w->R; q.D = w->D; q.bound_first = w->bound_first;
  *out = c;
  return 0;
}
extern "C" void ctts_dvae_destroy(ctts_dvae* c) { delete c; }

static size_t al(size_t v) { return (v + 255) & ~(size_t)255; }
struct DvaeWs {
  float *mag, *mel, *x0, *x1, *h, *a, *b, *big, *feat;
  size_t bytes;
};
// R = frames the trunk runs on (encode: T = code frames; decode: B * 2T), F = mel frames of the clip (encode only)
static DvaeWs carve_dvae(void* base, size_t R, size_t F) {
  DvaeWs w;
  size_t off = 0;
  char* p = (char*)base;
  auto take = [&](size_t bytes) { float* r = (float*)(p + off); off += al(bytes); return r; };
  const size_t Fe = (F + 1) & ~(size_t)1;
  w.mag = take(F * MAGLD * 4);
  w.mel = take(F * 100 * 4);
  w.x0 = take(Fe * 512 * 4);
  w.x1 = take(R * 512 * 4);
  w.h = take(R * 128 * 4);
  w.a = take(R * 512 * 4);
  w.b = take(R * 512 * 4);
  w.big = take(R * 2048 * 4);
  w.feat = take(R * 1024 * 4);
  w.bytes = off;
  return w;
}
extern "C" size_t ctts_dvae_encode_workspace_bytes(int32_t n_samples) {
  const size_t F = 1 + (size_t)n_samples / HOP;
  return carve_dvae(nullptr, F / 2 + 1, F).bytes;
}
extern "C" size_t ctts_dvae_decode_workspace_bytes(int32_t B, int32_t T) { return carve_dvae(nullptr, (size_t)B * 2 * T, 0).bytes; }
extern "C" int32_t ctts_dvae_code_frames(int32_t n_samples) {
  const int F = 1 + n_samples / HOP;
  return F < 2 ? 0 : (F - 2) / 2 + 1;
}

static GemmArgs lin(const float* A, int lda, const float* W, float* C, int ldc, int M, int N, int K, int epi) {
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = lda; a.W = W; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.wt = WT_F32; a.epi = epi; a.taps = 1;
  return a;
}
static GemmArgs conv(const float* X, int cin, const float* W, float* C, int cout, int rows, int F, int taps, int pad, int epi) {
  GemmArgs a = lin(X, cin, W, C, cout, rows, cout, taps * cin, epi);
  a.taps = taps; a.cin = cin; a.frames = F; a.pad = pad; a.dil = 1;
  return a;
}

// DVAEDecoder.forward (dvae.py:163-172): x [B*F][idim] -> out [B*F][odim]; a/b [R][hidden], big [R][4 hidden], h [R][bn]
static int trunk(const ctts_trunk_weights& t, const std::vector<const float*>* p, const float* x, float* out, DvaeWs& ws, int B, int F,
                 hipStream_t st) {
  const int R = B * F, H = t.hidden;
  GemmArgs c0 = conv(x, t.idim, t.conv_in0_w, ws.h, t.bn_dim, R, F, 3, 1, EPI_BIAS_GELU);
  c0.bias = t.conv_in0_b;
  CK(launch_gemm_tiled(c0, st));
  GemmArgs c2 = conv(ws.h, t.bn_dim, t.conv_in2_w, ws.a, H, R, F, 3, 1, EPI_BIAS);
  c2.bias = t.conv_in2_b;
  CK(launch_gemm_tiled(c2, st));
  for (int i = 0; i < t.n_blocks; ++i) {
    CK(launch_dwconv_ln(ws.a, p[0][i], p[1][i], p[2][i], p[3][i], 1e-6f, 2, ws.b, B, F, H, st));
    GemmArgs g1 = lin(ws.b, H, p[4][i], ws.big, 4 * H, R, 4 * H, H, EPI_BIAS_GELU);
    g1.bias = p[5][i];
    CK(launch_gemm_tiled(g1, st));
    GemmArgs g2 = lin(ws.big, 4 * H, p[6][i], ws.a, H, R, H, 4 * H, EPI_BIAS_SCALE_RES);
    g2.bias = p[7][i]; g2.gamma = p[8][i]; g2.res = ws.a; g2.ldr = H;
    CK(launch_gemm_tiled(g2, st));
  }
  CK(launch_gemm_tiled(lin(ws.a, H, t.conv_out_w, out, t.odim, R, t.odim, H, EPI_STORE), st));
  return 0;
}

extern "C" int ctts_dvae_encode(ctts_dvae* c, const float* wav, int32_t n_samples, int32_t* codes, void* workspace, size_t ws_bytes,
                                void* stream) {
  if (!c || !wav || !codes) return ctts_fail("ctts_dvae_encode: bad arguments");
  if (n_samples <= NFFT / 2) return ctts_fail("ctts_dvae_encode: clip shorter than the STFT's reflect padding (%d <= 512 samples)", n_samples);
  const int F = 1 + n_samples / HOP, T = ctts_dvae_code_frames(n_samples);
  if (T < 1) return ctts_fail("ctts_dvae_encode: clip too short");
  if (ws_bytes < ctts_dvae_encode_workspace_bytes(n_samples)) return ctts_fail("dvae workspace too small");
  CttsDeviceGuard dg(stream);
  hipStream_t st = (hipStream_t)stream;
  DvaeWs ws = carve_dvae(workspace, F / 2 + 1, F);
  const ctts_dvae_weights& w = c->w;
  // mel front end (dvae.py:175-206) and the division by coef (:267-269)
  CK(launch_stft_mag(wav, n_samples, w.mel_window, w.twiddle, ws.mag, F, st));
  GemmArgs m = lin(ws.mag, MAGLD, w.mel_fb, ws.mel, 100, F, 100, MAGLD, EPI_LOG_DIV);
  m.gamma = w.coef;
  CK(launch_gemm_tiled(m, st));
  // downsample_conv (dvae.py:229-235).  The stride-2 k4 conv reads frame pairs: [F][512] viewed as [F/2][1024] makes it
  // a stride-1 k3 conv with the host-repacked weight [512][3][1024]; an odd F gets one zero frame appended.
  const int Fe = (F + 1) & ~1;
  if (Fe != F) CK(hipMemsetAsync(ws.x0 + (size_t)F * 512, 0, 512 * sizeof(float), st));
  GemmArgs d0 = conv(ws.mel, 100, w.ds0_w, ws.x0, 512, F, F, 3, 1, EPI_BIAS_GELU);
  d0.bias = w.ds0_b;
  CK(launch_gemm_tiled(d0, st));
  GemmArgs d1 = conv(ws.x0, 1024, w.ds1_w, ws.x1, 512, T, Fe / 2, 3, 1, EPI_BIAS_GELU);
  d1.bias = w.ds1_b;
  CK(launch_gemm_tiled(d1, st));
  if (trunk(w.encoder, c->e, ws.x1, ws.feat, ws, 1, T, st)) return -1;
  CK(launch_gfsq_encode(c->q, ws.feat, codes, T, st));
  return 0;
}

extern "C" int ctts_dvae_decode_codes(ctts_dvae* c, const int64_t* codes, float* mel, int32_t B, int32_t T, void* workspace,
                                      size_t ws_bytes, void* stream) {
  if (!c || !codes || !mel || B <= 0 || T <= 0) return ctts_fail("ctts_dvae_decode_codes: bad arguments");
  if (ws_bytes < ctts_dvae_decode_workspace_bytes(B, T)) return ctts_fail("dvae workspace too small");
  CttsDeviceGuard dg(stream);
  hipStream_t st = (hipStream_t)stream;
  DvaeWs ws = carve_dvae(workspace, (size_t)B * 2 * T, 0);
  // feat [B*T][1024] IS [B*2T][512] in channels-last (dvae.py:281-287, see oracle/dvae_np.py)
  CK(launch_gfsq_embed(c->q, codes, ws.feat, B * T, st));
  if (trunk(c->w.decoder, c->d, ws.feat, ws.x1, ws, B, 2 * T, st)) return -1;
  GemmArgs oc = conv(ws.x1, 512, c->w.out_conv_w, mel, 100, B * 2 * T, 2 * T, 3, 1, EPI_SCALE);
  oc.gamma = c->w.coef;
  CK(launch_gemm_tiled(oc, st));
  return 0;
}
