// conv2d.hip - small 2-D convolutions on NHWC bf16: the k x k (3 x 3) dense, strided and depthwise convolutions of MobileViT's
// MobileNetV2 blocks, with the eval-mode BatchNorm folded into weight and bias on the host and SiLU in the epilogue; and the
// global average pool that ends the network.
// (reference: conv_norm_act / MBConv / MobileViTBlock of pytorch_models/image/mobile_vit.py:10-69; nn.AdaptiveAvgPool2d(1) at
//  mobile_vit.py:100.  1 x 1 convolutions are GEMMs over the NHWC rows and run on pm_linear_bf16.)
//
// Off the benchmark path and tiny (3 .. 320 channels, <= 128 x 128 pixels): a direct convolution, one output value per thread,
// fp32 accumulation in (kh, kw, ci) order, 16-byte operand loads where the channel count allows.  Threads of a wave walk the
// output channels of one pixel: the pixel's window is read once per wave (same addresses), the weights stream.
#include "common.h"

namespace {

template <int ACT>
__global__ __launch_bounds__(256) void conv2d_nhwc_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                          const float* __restrict__ bias, const bf16* __restrict__ resid,
                                                          bf16* __restrict__ y, int N, int H, int W, int Cin, int Ho, int Wo,
                                                          int Cout, int kh, int kw, int stride, int pad, int cin_g, int cout_g) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)N * Ho * Wo * Cout;
  if (idx >= total) return;
  const int co = (int)(idx % Cout);
  const int64_t pix = idx / Cout;
  const int wo = (int)(pix % Wo);
  const int ho = (int)((pix / Wo) % Ho);
  const int n = (int)(pix / ((int64_t)Wo * Ho));
  const int g = co / cout_g;
  const bf16* wr = w + (int64_t)co * kh * kw * cin_g;
  const bool vec = (cin_g % 8 == 0) && (Cin % 8 == 0);
  float acc = 0.f;
  for (int i = 0; i < kh; ++i) {
    const int hi = ho * stride - pad + i;
    if (hi < 0 || hi >= H) continue;
    for (int j = 0; j < kw; ++j) {
      const int wi = wo * stride - pad + j;
      if (wi < 0 || wi >= W) continue;
      const bf16* xp = x + (((int64_t)n * H + hi) * W + wi) * Cin + g * cin_g;
      const bf16* wp = wr + (i * kw + j) * cin_g;
      if (vec) {
        for (int c = 0; c < cin_g; c += 8) {
          const bf16x8 xv = *(const bf16x8*)(xp + c), wv = *(const bf16x8*)(wp + c);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc = fmaf((float)xv[e], (float)wv[e], acc);
        }
      } else {
        for (int c = 0; c < cin_g; ++c) acc = fmaf((float)xp[c], (float)wp[c], acc);
      }
    }
  }
  float v = acc + (bias ? bias[co] : 0.f);
  v = apply_act<ACT, true>(v);
  if (resid) v += (float)resid[idx];
  y[idx] = (bf16)v;
}

// mean over the HW rows of each sample: x (N, HW, C) bf16 -> y (N, C) bf16; one thread per (n, c), fp32 sum in row order
__global__ __launch_bounds__(256) void mean_rows_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int N, int HW, int C) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * C) return;
  const int n = idx / C, c = idx - n * C;
  const bf16* p = x + (int64_t)n * HW * C + c;
  float s = 0.f;
  for (int r = 0; r < HW; ++r) s += (float)p[(int64_t)r * C];
  y[idx] = (bf16)(s / (float)HW);
}

}  // namespace

extern "C" int pm_conv2d_nhwc_bf16(const void* x, int64_t N, int64_t H, int64_t W, int64_t Cin, const void* w, const float* bias,
                                   const void* resid, void* y, int64_t Cout, int64_t kh, int64_t kw, int64_t stride, int64_t pad,
                                   int64_t groups, int act, void* stream) {
  if (!x || !w || !y || N < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0 ||
      groups <= 0)
    return PM_EINVAL;
  if (Cin % groups || Cout % groups) return PM_EINVAL;
  if (N == 0) return PM_OK;
  const int64_t Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return PM_EINVAL;
  const int64_t cin_g = Cin / groups, cout_g = Cout / groups;
  if (cin_g % 8 == 0 && Cin % 8 == 0 && (((uintptr_t)x | (uintptr_t)w) & 15)) return PM_EALIGN;
  const int64_t total = N * Ho * Wo * Cout;
  if (total > (int64_t)0x7fffffff * 256 || H > (1 << 20) || W > (1 << 20) || Cin > (1 << 20) || Cout > (1 << 20)) return PM_EINVAL;
  const dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
#define PM_CGO(A)                                                                                                        \
  hipLaunchKernelGGL((conv2d_nhwc_kernel<A>), grid, dim3(256), 0, st, (const bf16*)x, (const bf16*)w, bias, (const bf16*)resid, \
                     (bf16*)y, (int)N, (int)H, (int)W, (int)Cin, (int)Ho, (int)Wo, (int)Cout, (int)kh, (int)kw, (int)stride,   \
                     (int)pad, (int)cin_g, (int)cout_g)
  switch (act) {
    case PM_ACT_NONE: PM_CGO(PM_ACT_NONE); break;
    case PM_ACT_SILU: PM_CGO(PM_ACT_SILU); break;
    case PM_ACT_RELU: PM_CGO(PM_ACT_RELU); break;
    default: return PM_EUNSUPPORTED;
  }
#undef PM_CGO
  PM_CHECK_LAUNCH();
  return PM_OK;
}

extern "C" int pm_mean_rows_bf16(const void* x, void* y, int64_t N, int64_t HW, int64_t C, void* stream) {
  if (!x || !y || N < 0 || HW <= 0 || C <= 0) return PM_EINVAL;
  if (N == 0) return PM_OK;
  if (N * C > 0x7fffffff) return PM_EINVAL;
  hipLaunchKernelGGL(mean_rows_kernel, dim3((unsigned)((N * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x,
                     (bf16*)y, (int)N, (int)HW, (int)C);
  PM_CHECK_LAUNCH();
  return PM_OK;
}
