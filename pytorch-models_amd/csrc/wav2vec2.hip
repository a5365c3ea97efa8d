// wav2vec2.hip - the pieces of Wav2Vec2 / Data2VecAudio / SEW that are not a plain GEMM, LayerNorm or attention:
//   * layer 0 of the feature encoder: Conv1d(1, C0, k, stride) on the raw waveform, its norm and GELU, written
//     TIME-major in bf16 (reference: pytorch_models/audio/wav2vec2.py:19-39,80: conv -> LayerNorm1d | InstanceNorm1d
//     | Identity -> GELU);
//   * the regrouping that turns the grouped positional conv (wav2vec2.py:70-74, data2vec_audio.py:23-30) into G
//     K-contiguous GEMMs on linear_bf16.hip.
//
// Layout decision: every activation of the feature encoder is (clip, time, channel).  With K ordered (tap, channel)
// the window of a Conv1d(C, C', k, stride s) at output step t is the k*C CONTIGUOUS values starting at row t*s, so
// layers 1..6 are pm_linear_bf16_ex calls (row stride s*C, K = k*C) with GELU in the epilogue, and the channel
// LayerNorm of the non-legacy stem is the row-wise pm_layernorm_ex - the transpose(1, 2) pairs of the reference never
// happen.  Layer 0 has K = 10: no GEMM, a lane keeps the taps of its 8 channels in registers and a wave emits one
// 1 KiB output row per step; HBM-bound on the output (2*C0 bytes per step; 32.8 MB per 10 s clip at C0 = 512).
//
// InstanceNorm1d (legacy stem: statistics over TIME per clip and channel) needs a global reduction before the first
// output can be written.  Neither a stored conv output nor a second conv pass is needed for it: the conv is linear in
// the waveform, so per clip  sum_t v[c, t] = sum_j w[c, j] S[j]  and  sum_t v[c, t]^2 = sum_{j, j'} w[c, j] w[c, j'] R[j][j']
// with S[j] = sum_t x[s t + j] and R[j][j'] = sum_t x[s t + j] x[s t + j'] - 10 + 55 numbers per clip, whatever C0 is.
// One pass over the waveform (640 KB per 10 s clip) accumulates them per 1024-step chunk in a fixed order (no atomics:
// bit-reproducible), a finalize kernel folds the chunks in fp64 and evaluates the two forms per channel, centred
// (variance = w^T (R/T - s s^T) w), into (mean, rstd); the normalising pass then computes the conv once.  (The first
// version recomputed the conv for the statistics: 189 us + 37 us at 32 x 10 s; this one takes ~15 us.)
#include "common.h"

namespace {

constexpr int TCH = 256;  // time steps per workgroup of the conv pass

enum { MODE_NONE = 0, MODE_LAYERNORM = 1, MODE_INSTANCE = 2 };
constexpr int ACH = 1024;      // time steps per workgroup of the autocorrelation pass
constexpr int APART = 80;      // floats per (clip, chunk) partial: KT sums + KT (KT + 1) / 2 products, padded

template <int KT, int MODE>
__global__ __launch_bounds__(256) void w2v_stem0_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ stats,
                                                        float eps, bf16* __restrict__ out, int64_t L, int T0, int C0,
                                                        int stride) {
  // The kernel is VALU-bound before it is HBM-bound (80 FMAs + 8 GELUs per lane and step against a 16-byte store), so
  // all per-channel arithmetic is on channel PAIRS: v_pk_fma_f32 for the taps, the packed polynomial GELU of common.h.
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int c0 = lane * 8;
  const bool on = c0 < C0;  // C0 % 8 == 0: a lane's 8 channels are all inside or all outside
  f32x2 wr[KT][4], bs[4], g[4], bt[4], mu[4], rs[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = on ? c0 + i : 0;
#pragma unroll
    for (int j = 0; j < KT; ++j) wr[j][i >> 1][i & 1] = on ? w[c * KT + j] : 0.f;
    bs[i >> 1][i & 1] = (on && bias) ? bias[c] : 0.f;
    g[i >> 1][i & 1] = (on && gamma) ? gamma[c] : 1.f;
    bt[i >> 1][i & 1] = (on && beta) ? beta[c] : 0.f;
    if constexpr (MODE == MODE_INSTANCE) {
      mu[i >> 1][i & 1] = stats[((int64_t)b * C0 + c) * 2];
      rs[i >> 1][i & 1] = stats[((int64_t)b * C0 + c) * 2 + 1];
    }
  }
  if constexpr (MODE == MODE_INSTANCE) {  // (v - mu) * rs * g + bt = v * (rs g) + (bt - mu rs g)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      g[i] = rs[i] * g[i];
      bt[i] = bt[i] - mu[i] * g[i];
    }
  }
  const f32x2 zero2 = {0.f, 0.f};
  const float* xb = x + (int64_t)b * L;
  const int t_end = min((chunk + 1) * TCH, T0);
  const float inv_c = 1.0f / (float)C0;
  // wave-uniform addresses: the taps arrive by scalar loads, requested one step ahead of their use
  float xn[KT];
  {
    const float* xs = xb + (int64_t)min(chunk * TCH + wave, T0 - 1) * stride;
#pragma unroll
    for (int j = 0; j < KT; ++j) xn[j] = xs[j];
  }
  for (int t = chunk * TCH + wave; t < t_end; t += 4) {
    float xc[KT];
#pragma unroll
    for (int j = 0; j < KT; ++j) xc[j] = xn[j];
    {
      const float* xs = xb + (int64_t)min(t + 4, T0 - 1) * stride;
#pragma unroll
      for (int j = 0; j < KT; ++j) xn[j] = xs[j];
    }
    f32x2 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = bs[i];
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      const float xv = xc[j];
      const f32x2 xv2 = {xv, xv};
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = __builtin_elementwise_fma(wr[j][i], xv2, v[i]);
    }
    if constexpr (MODE == MODE_LAYERNORM) {  // over the C0 channels of this step (LayerNorm1d, wav2vec2.py:14-16)
      const f32x2 s2 = (v[0] + v[1]) + (v[2] + v[3]);
      const float mean = wave_sum(on ? s2[0] + s2[1] : 0.f) * inv_c;
      const f32x2 mean2 = {mean, mean};
      f32x2 q2 = zero2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] -= mean2;
        q2 = __builtin_elementwise_fma(v[i], v[i], q2);
      }
      const float rstd = rsqrtf(wave_sum(on ? q2[0] + q2[1] : 0.f) * inv_c + eps);
      const f32x2 rstd2 = {rstd, rstd};
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = __builtin_elementwise_fma(v[i] * rstd2, g[i], bt[i]);
    }
    if constexpr (MODE == MODE_INSTANCE) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = __builtin_elementwise_fma(v[i], g[i], bt[i]);
    }
    if (on) {
      bf16x8 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x2 a = gelu_poly2(v[i]);
        o[2 * i] = (bf16)a[0];
        o[2 * i + 1] = (bf16)a[1];
      }
      *(bf16x8*)(out + ((int64_t)b * T0 + t) * C0 + c0) = o;
    }
  }
}

// (S, R) of one ACH-step chunk of one clip: thread-private sums over its steps, then a fixed-order wave / workgroup fold
template <int KT>
__global__ __launch_bounds__(256) void w2v_autocorr_kernel(const float* __restrict__ x, float* __restrict__ partials, int64_t L,
                                                           int T0, int stride, int nchunk) {
  constexpr int NP = KT + KT * (KT + 1) / 2;
  __shared__ float red[4][NP];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const float* xb = x + (int64_t)b * L;
  float acc[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) acc[p] = 0.f;
  const int t_end = min((chunk + 1) * ACH, T0);
  for (int t = chunk * ACH + threadIdx.x; t < t_end; t += 256) {
    float xv[KT];
#pragma unroll
    for (int j = 0; j < KT; ++j) xv[j] = xb[(int64_t)t * stride + j];
    int p = KT;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      acc[j] += xv[j];
#pragma unroll
      for (int j2 = j; j2 < KT; ++j2) {
        acc[p] = fmaf(xv[j], xv[j2], acc[p]);
        ++p;
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const float v = wave_sum(acc[p]);
    if (lane == 0) red[wave][p] = v;
  }
  __syncthreads();
  if (threadIdx.x < NP) {
    const int p = threadIdx.x;
    partials[((int64_t)b * nchunk + chunk) * APART + p] = ((red[0][p] + red[1][p]) + red[2][p]) + red[3][p];
  }
}

// one workgroup per clip: fold the chunks in fp64, then per channel mean = w.S / T0 (+ bias) and the centred variance
template <int KT>
__global__ __launch_bounds__(256) void w2v_stats_finalize_kernel(const float* __restrict__ partials, const float* __restrict__ w,
                                                                 const float* __restrict__ bias, float* __restrict__ stats, int C0,
                                                                 int nchunk, int T0, float eps) {
  constexpr int NP = KT + KT * (KT + 1) / 2;
  __shared__ double tot[NP];
  const int b = blockIdx.x;
  if (threadIdx.x < NP) {
    double s = 0.0;
    for (int k = 0; k < nchunk; ++k) s += (double)partials[((int64_t)b * nchunk + k) * APART + threadIdx.x];
    tot[threadIdx.x] = s / T0;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C0; c += 256) {
    double wv[KT], mean = 0.0, var = 0.0;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      wv[j] = (double)w[c * KT + j];
      mean += wv[j] * tot[j];
    }
    int p = KT;
#pragma unroll
    for (int j = 0; j < KT; ++j)
#pragma unroll
      for (int j2 = j; j2 < KT; ++j2) {
        const double cov = tot[p++] - tot[j] * tot[j2];  // E[x_j x_j'] - E[x_j] E[x_j']
        var += (j2 == j ? 1.0 : 2.0) * wv[j] * wv[j2] * cov;
      }
    var = fmax(var, 0.0);  // biased, like InstanceNorm1d
    stats[((int64_t)b * C0 + c) * 2] = (float)(mean + (bias ? (double)bias[c] : 0.0));
    stats[((int64_t)b * C0 + c) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// out[b][g][pl + t][c] = x[b][t][g*cg + c] (bf16), zero for the pl leading / pr trailing rows and for c in [cg, cgp)
template <bool XF32>
__global__ __launch_bounds__(256) void group_windows_kernel(const void* __restrict__ x, int64_t ldx, bf16* __restrict__ out,
                                                            int T, int G, int cg, int cgp, int pl, int Tp, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one thread per 8 output channels
  if (i >= total) return;
  const int per_row = cgp >> 3;
  const int ch = (int)(i % per_row);
  int64_t r = i / per_row;
  const int tp = (int)(r % Tp);
  r /= Tp;
  const int g = (int)(r % G);
  const int64_t b = r / G;
  const int t = tp - pl;
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16)0.f;
  if (t >= 0 && t < T) {
    const int64_t src = (b * T + t) * ldx + (int64_t)g * cg;
    if ((cg & 7) == 0) {
      if constexpr (XF32) {
        const f32x4 a = *(const f32x4*)((const float*)x + src + ch * 8), c = *(const f32x4*)((const float*)x + src + ch * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = (bf16)a[e]; o[4 + e] = (bf16)c[e]; }
      } else {
        o = *(const bf16x8*)((const bf16*)x + src + ch * 8);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = ch * 8 + e;
        if (c < cg) o[e] = XF32 ? (bf16)((const float*)x)[src + c] : ((const bf16*)x)[src + c];
      }
    }
  }
  *(bf16x8*)(out + i * 8) = o;
}

// out[b][t][:] = (x[b][2t][:] + x[b][2t+1][:]) / 2 in fp32, rounded once (F.avg_pool1d(x, 2) over time, sew.py:33)
__global__ __launch_bounds__(256) void avgpool_time2_kernel(const bf16* __restrict__ x, bf16* __restrict__ out, int T, int To, int d8,
                                                            int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int ch = (int)(i % d8);
  const int64_t r = i / d8;
  const int t = (int)(r % To);
  const int64_t b = r / To;
  const bf16* p = x + ((b * T + 2 * t) * d8 + ch) * 8;
  const bf16x8 a = *(const bf16x8*)p, c = *(const bf16x8*)(p + (int64_t)d8 * 8);
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16)(0.5f * ((float)a[e] + (float)c[e]));
  *(bf16x8*)(out + i * 8) = o;
}

template <int MODE>
int launch_stem0(int k, dim3 grid, hipStream_t st, const float* x, const float* w, const float* bias, const float* gamma,
                 const float* beta, const float* stats, float eps, bf16* out, int64_t L, int T0, int C0, int stride) {
  if (k != 10) return PM_EUNSUPPORTED;  // every model of the family opens with Conv1d(1, C0, 10, 5)
  hipLaunchKernelGGL((w2v_stem0_kernel<10, MODE>), grid, dim3(256), 0, st, x, w, bias, gamma, beta, stats, eps, out, L, T0, C0,
                     stride);
  return PM_OK;
}

}  // namespace

extern "C" int64_t pm_w2v_stem0_scratch_floats(int64_t B, int64_t T0) { return B * ((T0 + ACH - 1) / ACH) * APART; }

extern "C" int pm_w2v_stem0(const float* x, const float* w, const float* bias, int norm, const float* gamma, const float* beta,
                            float eps, float* partials, float* stats, void* out, int64_t B, int64_t L, int64_t C0, int64_t k,
                            int64_t stride, void* stream) {
  if (!x || !w || !out || B < 0 || L <= 0 || C0 <= 0 || k <= 0 || stride <= 0) return PM_EINVAL;
  if (norm < PM_W2V_NORM_NONE || norm > PM_W2V_NORM_INSTANCE) return PM_EINVAL;
  if (norm != PM_W2V_NORM_NONE && (!gamma || !beta)) return PM_EINVAL;
  if (norm == PM_W2V_NORM_INSTANCE && (!partials || !stats)) return PM_EINVAL;
  if (B == 0) return PM_OK;
  if (L < k) return PM_EINVAL;
  if (C0 % 8 || C0 > 512 || k != 10) return PM_EUNSUPPORTED;
  if ((uintptr_t)out & 15) return PM_EALIGN;
  const int64_t T0 = (L - k) / stride + 1;
  const int64_t nchunk = (T0 + TCH - 1) / TCH;
  if (T0 > 0x7fffffff / 8 || B > 65535) return PM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)nchunk, (unsigned)B);
  int rc;
  if (norm == PM_W2V_NORM_INSTANCE) {
    const int na = (int)((T0 + ACH - 1) / ACH);
    hipLaunchKernelGGL((w2v_autocorr_kernel<10>), dim3((unsigned)na, (unsigned)B), dim3(256), 0, st, x, partials, L, (int)T0,
                       (int)stride, na);
    hipLaunchKernelGGL((w2v_stats_finalize_kernel<10>), dim3((unsigned)B), dim3(256), 0, st, partials, w, bias, stats, (int)C0, na,
                       (int)T0, eps);
    rc = launch_stem0<MODE_INSTANCE>((int)k, grid, st, x, w, bias, gamma, beta, stats, eps, (bf16*)out, L, (int)T0, (int)C0,
                                     (int)stride);
  } else if (norm == PM_W2V_NORM_LAYER) {
    rc = launch_stem0<MODE_LAYERNORM>((int)k, grid, st, x, w, bias, gamma, beta, nullptr, eps, (bf16*)out, L, (int)T0, (int)C0,
                                      (int)stride);
  } else {
    rc = launch_stem0<MODE_NONE>((int)k, grid, st, x, w, bias, nullptr, nullptr, nullptr, eps, (bf16*)out, L, (int)T0, (int)C0,
                                 (int)stride);
  }
  if (rc != PM_OK) return rc;
  PM_CHECK_LAUNCH();
  return PM_OK;
}

extern "C" int pm_group_windows(const void* x, int64_t ldx, int x_dtype, void* out, int64_t B, int64_t T, int64_t G, int64_t cg,
                                int64_t cgp, int64_t pad_left, int64_t pad_right, void* stream) {
  if (!x || !out || B < 0 || T <= 0 || G <= 0 || cg <= 0 || pad_left < 0 || pad_right < 0) return PM_EINVAL;
  if (x_dtype != PM_BF16 && x_dtype != PM_F32) return PM_EINVAL;
  if (cgp < cg || cgp % 8 || ldx < G * cg) return PM_EINVAL;
  if (B == 0) return PM_OK;
  if ((uintptr_t)out & 15) return PM_EALIGN;
  if (cg % 8 == 0 && ((ldx % 8) || ((uintptr_t)x & 15))) return PM_EALIGN;
  const int64_t Tp = T + pad_left + pad_right;
  const int64_t total = B * G * Tp * (cgp / 8);
  if (Tp > 0x7fffffff || (total + 255) / 256 > 0x7fffffff) return PM_EINVAL;
  dim3 grid((unsigned)((total + 255) / 256));
  if (x_dtype == PM_F32)
    hipLaunchKernelGGL(group_windows_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, (bf16*)out, (int)T, (int)G,
                       (int)cg, (int)cgp, (int)pad_left, (int)Tp, total);
  else
    hipLaunchKernelGGL(group_windows_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, (bf16*)out, (int)T, (int)G,
                       (int)cg, (int)cgp, (int)pad_left, (int)Tp, total);
  PM_CHECK_LAUNCH();
  return PM_OK;
}

extern "C" int pm_avgpool_time2(const void* x, void* out, int64_t B, int64_t T, int64_t d, void* stream) {
  if (!x || !out || B < 0 || T < 2 || d <= 0) return PM_EINVAL;
  if (d % 8) return PM_EUNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)out) & 15) return PM_EALIGN;
  if (B == 0) return PM_OK;
  const int64_t To = T / 2, total = B * To * (d / 8);
  if (T > 0x7fffffff || (total + 255) / 256 > 0x7fffffff) return PM_EINVAL;
  hipLaunchKernelGGL(avgpool_time2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)x, (bf16*)out, (int)T, (int)To, (int)(d / 8), total);
  PM_CHECK_LAUNCH();
  return PM_OK;
}
