// attention_generic.hip - scaled dot-product attention for head dims other than 64 (multiples of 4 up to 128, multiples of 2 up to 64), e.g. the
// 80-wide heads of ViT-H (pytorch_models/image/vit.py:106-113: H = (32, 1280, 16)), 16 / 32-wide heads of small MHAs, the
// 16 .. 60-wide heads of MobileViT's encoders (pytorch_models/image/mobile_vit.py:62,106-108: d = 64 .. 240 over 4 heads: 20 and
// 30 are multiples of 2 only).
// (reference: F.scaled_dot_product_attention at pytorch_models/transformer.py:52, same addressing as attention_bf16.hip)
//
// Correctness-first and off the benchmark path (every BASELINE config has head_dim 64): fp32 VALU arithmetic, a
// workgroup handles 8 queries of one (batch, head), scores for all keys parked in LDS (Lk <= 2048), exact softmax.
#include <cstdlib>

#include "common.h"

// attention_f32.hip
int pm_attention_f32_hd64_launch(const float* q, int64_t qsb, int64_t qst, const float* k, int64_t ksb, int64_t kst, const float* v,
                                 int64_t vsb, int64_t vst, float* o, int64_t osb, int64_t ost, int64_t B, int64_t H, int64_t Lq,
                                 int64_t Lk, hipStream_t st);

namespace {

constexpr int GQ = 8, GMAXK = 2048, GMAXD = 128;

// T = bf16 (the head dims the MFMA kernel does not cover) or float (modules whose parameters are fp32, and the exact
// Whisper pipeline: the same fp32 arithmetic on fp32 operands, nothing rounded on the way in or out)
template <typename T>
__device__ __forceinline__ void ld8(const T* p, float (&o)[8]) {
  if constexpr (sizeof(T) == 2) {
    const bf16x8 v = *(const bf16x8*)p;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
  } else {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = a[e]; o[4 + e] = b[e]; }
  }
}

template <typename T>
__device__ __forceinline__ void ld4(const T* p, float (&o)[8]) {  // four values (head dims that are multiples of 4 only)
  if constexpr (sizeof(T) == 2) {
    const bf16x4 v = *(const bf16x4*)p;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (float)v[e];
  } else {
    const f32x4 a = *(const f32x4*)p;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = a[e];
  }
#pragma unroll
  for (int e = 4; e < 8; ++e) o[e] = 0.f;
}

template <typename T>
__device__ __forceinline__ void ld2(const T* p, float (&o)[8]) {  // two values (head dims that are multiples of 2 only)
  if constexpr (sizeof(T) == 2) {
    const bf16x2 v = *(const bf16x2*)p;
    o[0] = (float)v[0];
    o[1] = (float)v[1];
  } else {
    const f32x2 a = *(const f32x2*)p;
    o[0] = a[0];
    o[1] = a[1];
  }
#pragma unroll
  for (int e = 2; e < 8; ++e) o[e] = 0.f;
}

template <typename T>
__global__ __launch_bounds__(256) void attn_generic_kernel(const T* __restrict__ Q, int64_t qsb, int64_t qst,
                                                           const T* __restrict__ K, int64_t ksb, int64_t kst,
                                                           const T* __restrict__ V, int64_t vsb, int64_t vst,
                                                           T* __restrict__ O, int64_t osb, int64_t ost, int H, int Lq,
                                                           int Lk, int hd, int nqb, int causal,
                                                           const float* __restrict__ bias, int64_t bsb, int64_t bsh,
                                                           int64_t bsq, float scale, int step) {
  __shared__ float sc[GQ * GMAXK];
  __shared__ float qs[GQ * GMAXD];
  __shared__ float inv_sum[GQ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qb = blockIdx.x % nqb, bh = blockIdx.x / nqb;
  const int h = bh % H, b = bh / H;
  const int q0 = qb * GQ;
  const T* Qp = Q + (int64_t)b * qsb + (int64_t)h * hd;
  const T* Kp = K + (int64_t)b * ksb + (int64_t)h * hd;
  const T* Vp = V + (int64_t)b * vsb + (int64_t)h * hd;

  for (int i = tid; i < GQ * hd; i += 256) {
    const int qq = i / hd, dd = i - qq * hd;
    const int qi = q0 + qq < Lq ? q0 + qq : Lq - 1;
    qs[qq * GMAXD + dd] = (float)Qp[(int64_t)qi * qst + dd];
  }
  __syncthreads();
  // ---- scores: one key per thread and pass, dotted with the 8 queries
  for (int key = tid; key < Lk; key += 256) {
    float acc[GQ];
#pragma unroll
    for (int qq = 0; qq < GQ; ++qq) acc[qq] = 0.f;
    const T* kr = Kp + (int64_t)key * kst;
    if (step == 8) {
      for (int d0 = 0; d0 < hd; d0 += 8) {
        float kv[8];
        ld8(kr + d0, kv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float kf = kv[e];
#pragma unroll
          for (int qq = 0; qq < GQ; ++qq) acc[qq] = fmaf(qs[qq * GMAXD + d0 + e], kf, acc[qq]);
        }
      }
    } else if (step == 2) {
      for (int d0 = 0; d0 < hd; d0 += 2) {
        float kv[8];
        ld2(kr + d0, kv);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float kf = kv[e];
#pragma unroll
          for (int qq = 0; qq < GQ; ++qq) acc[qq] = fmaf(qs[qq * GMAXD + d0 + e], kf, acc[qq]);
        }
      }
    } else {  // head dims 4 (mod 8), or rows aligned to 4 elements only: the same sums in the same order, 4 values per load
      for (int d0 = 0; d0 < hd; d0 += 4) {
        float kv[8];
        ld4(kr + d0, kv);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float kf = kv[e];
#pragma unroll
          for (int qq = 0; qq < GQ; ++qq) acc[qq] = fmaf(qs[qq * GMAXD + d0 + e], kf, acc[qq]);
        }
      }
    }
#pragma unroll
    for (int qq = 0; qq < GQ; ++qq) {
      const int qi = q0 + qq;
      float s = acc[qq] * scale;
      if (bias && qi < Lq) s += bias[(int64_t)b * bsb + (int64_t)h * bsh + (int64_t)qi * bsq + key];
      if (causal && key > qi) s = -INFINITY;
      sc[qq * GMAXK + key] = s;
    }
  }
  __syncthreads();
  // ---- softmax: wave w owns queries 2w, 2w+1
  for (int qq = wave * 2; qq < wave * 2 + 2; ++qq) {
    float mx = -INFINITY;
    for (int k = lane; k < Lk; k += 64) mx = fmaxf(mx, sc[qq * GMAXK + k]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int k = lane; k < Lk; k += 64) {
      const float p = expf(sc[qq * GMAXK + k] - mx);
      sc[qq * GMAXK + k] = p;
      sum += p;
    }
    sum = wave_sum(sum);
    if (lane == 0) inv_sum[qq] = 1.0f / sum;
  }
  __syncthreads();
  // ---- P.V: thread -> (query tid / 32, 4 dims (tid % 32) * 4); head dims that are multiples of 2 only: 2 dims per thread
  const int qq = tid >> 5;
  if (step == 2) {
    const int d0 = (tid & 31) * 2;
    if (d0 < hd) {
      float o[2] = {0.f, 0.f};
      for (int key = 0; key < Lk; ++key) {
        const float p = sc[qq * GMAXK + key];
        float vv[8];
        ld2(Vp + (int64_t)key * vst + d0, vv);
        o[0] = fmaf(p, vv[0], o[0]);
        o[1] = fmaf(p, vv[1], o[1]);
      }
      const int qi = q0 + qq;
      if (qi < Lq) {
        T* op = O + (int64_t)b * osb + (int64_t)qi * ost + (int64_t)h * hd + d0;
        if constexpr (sizeof(T) == 2) *(bf16x2*)op = bf16x2{(bf16)(o[0] * inv_sum[qq]), (bf16)(o[1] * inv_sum[qq])};
        else *(f32x2*)op = f32x2{o[0] * inv_sum[qq], o[1] * inv_sum[qq]};
      }
    }
    return;
  }
  const int d0 = (tid & 31) * 4;
  if (d0 < hd) {
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    for (int key = 0; key < Lk; ++key) {
      const float p = sc[qq * GMAXK + key];
      float vv[4];
      if constexpr (sizeof(T) == 2) {
        const bf16x4 v4 = *(const bf16x4*)(Vp + (int64_t)key * vst + d0);
#pragma unroll
        for (int e = 0; e < 4; ++e) vv[e] = (float)v4[e];
      } else {
        const f32x4 v4 = *(const f32x4*)(Vp + (int64_t)key * vst + d0);
#pragma unroll
        for (int e = 0; e < 4; ++e) vv[e] = v4[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaf(p, vv[e], o[e]);
    }
    const int qi = q0 + qq;
    if (qi < Lq) {
      T* op = O + (int64_t)b * osb + (int64_t)qi * ost + (int64_t)h * hd + d0;
      if constexpr (sizeof(T) == 2) {
        bf16x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = (bf16)(o[e] * inv_sum[qq]);
        *(bf16x4*)op = ov;
      } else {
        *(f32x4*)op = f32x4{o[0] * inv_sum[qq], o[1] * inv_sum[qq], o[2] * inv_sum[qq], o[3] * inv_sum[qq]};
      }
    }
  }
}

}  // namespace

extern "C" int pm_attention_generic_bf16(const void* q, int64_t q_stride_b, int64_t q_stride_t, const void* k,
                                         int64_t k_stride_b, int64_t k_stride_t, const void* v, int64_t v_stride_b,
                                         int64_t v_stride_t, void* o, int64_t o_stride_b, int64_t o_stride_t, int64_t B,
                                         int64_t H, int64_t Lq, int64_t Lk, int64_t head_dim, int causal, const float* bias,
                                         int64_t bias_stride_b, int64_t bias_stride_h, int64_t bias_stride_q, void* stream) {
  if (!q || !k || !v || !o || B < 0 || H <= 0 || Lq < 0 || Lk <= 0 || head_dim <= 0) return PM_EINVAL;
  if (B == 0 || Lq == 0) return PM_OK;
  if (head_dim % 2 || head_dim > GMAXD || Lk > GMAXK || (head_dim % 4 && head_dim > 64)) return PM_EUNSUPPORTED;
  const int unit = head_dim % 4 ? 2 : 4;  // elements per access of the narrowest path this head_dim needs
  if ((q_stride_t | k_stride_t | v_stride_t | q_stride_b | k_stride_b | v_stride_b | o_stride_t | o_stride_b) % unit) return PM_EALIGN;
  if (((uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & (unit * 2 - 1)) return PM_EALIGN;
  // 16-byte key loads where rows allow it (head_dim and every stride a multiple of 8 elements, 16-byte bases), 8-byte ones otherwise
  const int step = unit == 2 ? 2
                   : (head_dim % 8 == 0 && !((q_stride_t | k_stride_t | v_stride_t | q_stride_b | k_stride_b | v_stride_b) % 8) &&
                      !(((uintptr_t)k | (uintptr_t)v) & 15)) ? 8 : 4;
  if (bias && bias_stride_q < Lk) return PM_EINVAL;
  const int nqb = (int)((Lq + GQ - 1) / GQ);
  const int64_t nblk = B * H * nqb;
  if (nblk > 0x7fffffff) return PM_EINVAL;
  hipLaunchKernelGGL(attn_generic_kernel<bf16>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const bf16*)q, q_stride_b,
                     q_stride_t, (const bf16*)k, k_stride_b, k_stride_t, (const bf16*)v, v_stride_b, v_stride_t, (bf16*)o,
                     o_stride_b, o_stride_t, (int)H, (int)Lq, (int)Lk, (int)head_dim, nqb, causal, bias, bias_stride_b,
                     bias_stride_h, bias_stride_q, 1.0f / sqrtf((float)head_dim), step);
  PM_CHECK_LAUNCH();
  return PM_OK;
}

/* The same kernel on fp32 operands (strides in ELEMENTS): attention of modules whose parameters are fp32 and of the exact
 * Whisper pipeline.  q, k, v, o f32 with unit last stride; head_dim % 4 == 0 (<= 128) or % 2 == 0 (<= 64), Lk <= 2048; strides multiples of 4 (2). */
extern "C" int pm_attention_generic_f32(const float* q, int64_t q_stride_b, int64_t q_stride_t, const float* k,
                                        int64_t k_stride_b, int64_t k_stride_t, const float* v, int64_t v_stride_b,
                                        int64_t v_stride_t, float* o, int64_t o_stride_b, int64_t o_stride_t, int64_t B,
                                        int64_t H, int64_t Lq, int64_t Lk, int64_t head_dim, int causal, const float* bias,
                                        int64_t bias_stride_b, int64_t bias_stride_h, int64_t bias_stride_q, void* stream) {
  if (!q || !k || !v || !o || B < 0 || H <= 0 || Lq < 0 || Lk <= 0 || head_dim <= 0) return PM_EINVAL;
  if (B == 0 || Lq == 0) return PM_OK;
  if (head_dim % 2 || head_dim > GMAXD || Lk > GMAXK || (head_dim % 4 && head_dim > 64)) return PM_EUNSUPPORTED;
  const int unit32 = head_dim % 4 ? 2 : 4;
  if ((q_stride_t | k_stride_t | v_stride_t | q_stride_b | k_stride_b | v_stride_b | o_stride_t | o_stride_b) % unit32) return PM_EALIGN;
  if (((uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & (unit32 * 4 - 1)) return PM_EALIGN;
  if (bias && bias_stride_q < Lk) return PM_EINVAL;
  // head_dim 64 without mask / bias (encoder layers, ViT): the f32-input matrix pipe (attention_f32.hip), same fp32 arithmetic
  static const bool f32_mfma = [] { const char* e = getenv("PM_ATTN_F32_MFMA"); return !e || atoi(e) != 0; }();
  if (f32_mfma && head_dim == 64 && !causal && !bias && !(((uintptr_t)q) & 15) && q_stride_t % 4 == 0 && q_stride_b % 4 == 0) {
    const int rc = pm_attention_f32_hd64_launch(q, q_stride_b, q_stride_t, k, k_stride_b, k_stride_t, v, v_stride_b, v_stride_t, o,
                                                o_stride_b, o_stride_t, B, H, Lq, Lk, (hipStream_t)stream);
    if (rc != PM_OK) return rc;
    PM_CHECK_LAUNCH();
    return PM_OK;
  }
  const int nqb = (int)((Lq + GQ - 1) / GQ);
  const int64_t nblk = B * H * nqb;
  if (nblk > 0x7fffffff) return PM_EINVAL;
  hipLaunchKernelGGL(attn_generic_kernel<float>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, q, q_stride_b,
                     q_stride_t, k, k_stride_b, k_stride_t, v, v_stride_b, v_stride_t, o, o_stride_b, o_stride_t, (int)H,
                     (int)Lq, (int)Lk, (int)head_dim, nqb, causal, bias, bias_stride_b, bias_stride_h, bias_stride_q,
                     1.0f / sqrtf((float)head_dim), head_dim % 4 ? 2 : head_dim % 8 == 0 ? 8 : 4);
  PM_CHECK_LAUNCH();
  return PM_OK;
}
