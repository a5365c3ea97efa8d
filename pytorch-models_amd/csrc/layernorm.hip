// layernorm.hip - nn.LayerNorm over the last dim, one wavefront per row.
// (reference: pytorch_models/transformer.py:87,90,93; image/vit.py:69; audio2text/whisper.py:27,45)
//
// Roofline: HBM-bound, algorithmic bytes = M*d*(sizeof(x) + sizeof(y)).
// A 64-lane wave owns one row: 16-byte vector loads (8 bf16 / 4 f32 per lane per step), the row stays
// in registers, mean and the CENTRED second moment are reduced across the wave in fp32 (two passes
// over registers, like the reference's mean / biased variance), then the affine is applied and the
// row is written back with 16-byte (f32) or 8/16-byte (bf16) stores.
#include "common.h"

namespace {

constexpr int MAX_CHUNKS = 8;  // 8 chunks x 64 lanes x 8 elements = d <= 4096

template <bool XF32>
__device__ __forceinline__ void load8(const void* x, int64_t off, float (&v)[8]) {
  if constexpr (XF32) {
    const f32x4 a = *(const f32x4*)((const float*)x + off);
    const f32x4 b = *(const f32x4*)((const float*)x + off + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
  } else {
    const bf16x8 a = *(const bf16x8*)((const bf16*)x + off);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
  }
}

template <bool YF32>
__device__ __forceinline__ void store8(void* y, int64_t off, const float (&v)[8]) {
  if constexpr (YF32) {
    *(f32x4*)((float*)y + off) = f32x4{v[0], v[1], v[2], v[3]};
    *(f32x4*)((float*)y + off + 4) = f32x4{v[4], v[5], v[6], v[7]};
  } else {
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (bf16)v[i];
    *(bf16x8*)((bf16*)y + off) = o;
  }
}

template <bool XF32, bool YF32, int NCH>
__global__ __launch_bounds__(256) void layernorm_kernel(const void* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, void* __restrict__ y, int64_t ldy, int64_t M, int d,
                                                        int act, const void* __restrict__ resid, int64_t ldr, int resid_f32, int rms) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nchunk = d >> 3;
  float v[NCH][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lane + c * 64;
    if (ch < nchunk) {
      load8<XF32>(x, row * ldx + ch * 8, v[c]);
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[c][i];
    }
  }
  const float inv_d = 1.0f / (float)d;
  const float mean = rms ? 0.f : wave_sum(s) * inv_d;  // rms: no centring (T5's LayerNorm, text/t5.py:15-25)
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (lane + c * 64 < nchunk) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        v[c][i] -= mean;
        q = fmaf(v[c][i], v[c][i], q);
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) * inv_d + eps);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lane + c * 64;
    if (ch < nchunk) {
      float g[8], b[8];
      if (gamma) {  // elementwise_affine=False (data2vec_audio.py:27) passes null
        load8<true>(gamma, ch * 8, g);
        if (beta) {
          load8<true>(beta, ch * 8, b);
#pragma unroll
          for (int i = 0; i < 8; ++i) v[c][i] = fmaf(v[c][i] * rstd, g[i], b[i]);
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[c][i] = v[c][i] * rstd * g[i];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[c][i] *= rstd;
      }
      if (act == PM_ACT_GELU) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[c][i] = apply_act<PM_ACT_GELU, true>(v[c][i]);
      }
      if (resid) {
        if (resid_f32) load8<true>(resid, row * ldr + ch * 8, g);
        else load8<false>(resid, row * ldr + ch * 8, g);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[c][i] += g[i];
      }
      store8<YF32>(y, row * ldy + ch * 8, v[c]);
    }
  }
}

template <bool XF32, bool YF32>
void launch(int nch, dim3 grid, hipStream_t st, const void* x, int64_t ldx, const float* g, const float* b, float eps,
            void* y, int64_t ldy, int64_t M, int d, int act, const void* resid, int64_t ldr, int resid_f32, int rms) {
#define PM_LN(N)                                                                                                       \
  hipLaunchKernelGGL((layernorm_kernel<XF32, YF32, N>), grid, dim3(256), 0, st, x, ldx, g, b, eps, y, ldy, M, d, act, \
                     resid, ldr, resid_f32, rms);                                                                      \
  break
  switch (nch) {
    case 1: PM_LN(1);
    case 2: PM_LN(2);
    case 3: PM_LN(3);
    case 4: PM_LN(4);
    default: PM_LN(MAX_CHUNKS);
  }
#undef PM_LN
}

}  // namespace

// y = act(LayerNorm(x)) + resid; gamma / beta both null = no affine; act in {NONE, GELU}; resid null = none.
static int layernorm_impl(const void* x, int64_t ldx, int x_dtype, const float* gamma, const float* beta, float eps, int act,
                          const void* resid, int64_t ldr, int resid_dtype, void* y, int64_t ldy, int y_dtype, int64_t M,
                          int64_t d, int rms, void* stream) {
  if (!x || !y || M < 0 || d <= 0 || (!rms && (gamma == nullptr) != (beta == nullptr)) || (rms && (!gamma || beta))) return PM_EINVAL;
  if ((x_dtype != PM_BF16 && x_dtype != PM_F32) || (y_dtype != PM_BF16 && y_dtype != PM_F32)) return PM_EINVAL;
  if (act != PM_ACT_NONE && act != PM_ACT_GELU) return PM_EUNSUPPORTED;
  if (resid && resid_dtype != PM_BF16 && resid_dtype != PM_F32) return PM_EINVAL;
  if (M == 0) return PM_OK;
  if (d % 8 != 0 || d > MAX_CHUNKS * 512) return PM_EUNSUPPORTED;
  if (ldx < d || ldy < d || (resid && ldr < d)) return PM_EINVAL;
  if (ldx % 8 || ldy % 8 || (resid && ldr % 8)) return PM_EALIGN;
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)resid) & 15) return PM_EALIGN;
  const int64_t nblk = (M + 3) / 4;
  if (nblk > 0x7fffffff) return PM_EINVAL;
  const int nch = (int)((d / 8 + 63) / 64);
  dim3 grid((unsigned)nblk);
  hipStream_t st = (hipStream_t)stream;
  const bool xf = x_dtype == PM_F32, yf = y_dtype == PM_F32;
  const int rf = resid_dtype == PM_F32;
  if (xf && yf) launch<true, true>(nch, grid, st, x, ldx, gamma, beta, eps, y, ldy, M, (int)d, act, resid, ldr, rf, rms);
  else if (xf) launch<true, false>(nch, grid, st, x, ldx, gamma, beta, eps, y, ldy, M, (int)d, act, resid, ldr, rf, rms);
  else if (yf) launch<false, true>(nch, grid, st, x, ldx, gamma, beta, eps, y, ldy, M, (int)d, act, resid, ldr, rf, rms);
  else launch<false, false>(nch, grid, st, x, ldx, gamma, beta, eps, y, ldy, M, (int)d, act, resid, ldr, rf, rms);
  PM_CHECK_LAUNCH();
  return PM_OK;
}

extern "C" int pm_layernorm_ex(const void* x, int64_t ldx, int x_dtype, const float* gamma, const float* beta, float eps,
                               int act, const void* resid, int64_t ldr, int resid_dtype, void* y, int64_t ldy, int y_dtype,
                               int64_t M, int64_t d, void* stream) {
  return layernorm_impl(x, ldx, x_dtype, gamma, beta, eps, act, resid, ldr, resid_dtype, y, ldy, y_dtype, M, d, 0, stream);
}

extern "C" int pm_layernorm(const void* x, int64_t ldx, int x_dtype, const float* gamma, const float* beta, float eps,
                            void* y, int64_t ldy, int y_dtype, int64_t M, int64_t d, void* stream) {
  if (!gamma || !beta) return PM_EINVAL;
  return layernorm_impl(x, ldx, x_dtype, gamma, beta, eps, PM_ACT_NONE, nullptr, 0, PM_BF16, y, ldy, y_dtype, M, d, 0, stream);
}

// y = x * rsqrt(mean(x^2) + eps) * gamma: LayerNorm without centring and without bias (text/t5.py:15-25)
extern "C" int pm_rmsnorm(const void* x, int64_t ldx, int x_dtype, const float* gamma, float eps, void* y, int64_t ldy,
                          int y_dtype, int64_t M, int64_t d, void* stream) {
  return layernorm_impl(x, ldx, x_dtype, gamma, nullptr, eps, PM_ACT_NONE, nullptr, 0, PM_BF16, y, ldy, y_dtype, M, d, 1, stream);
}

namespace {
// out[m, f] = gelu_tanh(h[m, f]) * h[m, F + f]: the gate of GEGLU (text/t5.py:29-38) over a packed [w; v] projection
__global__ __launch_bounds__(256) void geglu_kernel(const bf16* __restrict__ h, int64_t ldh, bf16* __restrict__ out, int64_t ldo,
                                                    int F8, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int64_t m = i / F8;
  const int c = (int)(i - m * F8);
  const bf16x8 a = *(const bf16x8*)(h + m * ldh + c * 8), b = *(const bf16x8*)(h + m * ldh + (int64_t)F8 * 8 + c * 8);
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16)(apply_act<PM_ACT_GELU_TANH, true>((float)a[e]) * (float)b[e]);
  *(bf16x8*)(out + m * ldo + c * 8) = o;
}
}  // namespace

extern "C" int pm_geglu(const void* h, int64_t ldh, void* out, int64_t ldo, int64_t M, int64_t F, void* stream) {
  if (!h || !out || M < 0 || F <= 0 || ldh < 2 * F || ldo < F) return PM_EINVAL;
  if (F % 8) return PM_EUNSUPPORTED;
  if (ldh % 8 || ldo % 8 || (((uintptr_t)h | (uintptr_t)out) & 15)) return PM_EALIGN;
  if (M == 0) return PM_OK;
  const int64_t total = M * (F / 8);
  if ((total + 255) / 256 > 0x7fffffff) return PM_EINVAL;
  hipLaunchKernelGGL(geglu_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)h, ldh,
                     (bf16*)out, ldo, (int)(F / 8), total);
  PM_CHECK_LAUNCH();
  return PM_OK;
}
