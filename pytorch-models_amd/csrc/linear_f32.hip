// linear_f32.hip - nn.Linear in fp32 for modules whose parameters are fp32 (the reference's default dtype) and for the
// "exact" Whisper pipeline: y[M, N] = act(x[M, K] w[N, K]^T + bias[N]) + resid[M, N], everything fp32.
// (reference: the same nn.Linear sites as linear_bf16.hip - pytorch_models/transformer.py:28-31,47-53,59-66 - plus the
// convolutions taken as GEMMs over windows: image/vit.py:64, audio2text/whisper.py:16-21.)
//
// The matrix pipe has an f32-input MFMA (v_mfma_f32_16x16x4_f32): exact fp32 products, fp32 accumulation, bit for bit a
// k-ordered fmaf chain, at the fp32 vector rate (157 TFLOP/s peak: 1 / 16 of bf16).  That is what the reference's
// 2e-5 / 5e-5 tolerances need; it is NOT the throughput path - a model that wants speed is cast to bf16.
// Tile 128 x 128 x 16, four waves as 2 x 2 (64 x 64 each = 4 x 4 MFMA tiles, 64 accumulator VGPRs), operands staged
// through registers into LDS tiles with 80-byte rows (16 floats + 4 of padding: the fragment read - lane (row = l & 15,
// k = l >> 4) - then spreads over the banks up to a 2-way conflict), double-buffered.  The weight tile is the MFMA A operand, so a
// lane ends up with 4 consecutive features of one row: 16-byte epilogue stores.
#include "common.h"

namespace {

constexpr int FBM = 128, FBN = 128, FBK = 16, FPITCH = 20;  // floats per LDS row
constexpr int FTILE = FBM * FPITCH;                          // floats per operand tile

template <int ACT>
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ X, int64_t ldx, int x_rows_per_batch,
                                                         int64_t x_batch_stride, const float* __restrict__ W, int64_t ldw,
                                                         const float* __restrict__ bias, const float* __restrict__ resid,
                                                         int64_t ldr, int resid_period, float* __restrict__ Y, int64_t ldy,
                                                         int M, int N, int K, int tiles_n) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * FTILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int m0 = tm * FBM, n0 = tn * FBN;
  const int nk = (K + FBK - 1) / FBK;

  // staging: thread -> (row = tid >> 2 (+ 64), 4 floats at k = (tid & 3) * 4) of each operand tile
  const int srow = tid >> 2, sk = (tid & 3) * 4;
  const float* xp[2];
  const float* wp[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int gm = m0 + srow + i * 64;
    gm = gm < M ? gm : M - 1;
    if (x_rows_per_batch > 0) {
      const int bb = gm / x_rows_per_batch;
      xp[i] = X + (int64_t)bb * x_batch_stride + (int64_t)(gm - bb * x_rows_per_batch) * ldx + sk;
    } else {
      xp[i] = X + (int64_t)gm * ldx + sk;
    }
    int gn = n0 + srow + i * 64;
    gn = gn < N ? gn : N - 1;
    wp[i] = W + (int64_t)gn * ldw + sk;
  }
  f32x4 xr[2], wr[2];
  auto fetch = [&](int kt) {
    const int k = kt * FBK + sk;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (k + 3 < K) {
        xr[i] = *(const f32x4*)(xp[i] + kt * FBK);
        wr[i] = *(const f32x4*)(wp[i] + kt * FBK);
      } else {  // K tail: element-wise, zero beyond K
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xr[i][e] = k + e < K ? xp[i][kt * FBK + e] : 0.f;
          wr[i][e] = k + e < K ? wp[i][kt * FBK + e] : 0.f;
        }
      }
    }
  };
  auto put = [&](int buf) {
    float* xs = smem + buf * 2 * FTILE;
    float* ws = xs + FTILE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *(f32x4*)(xs + (srow + i * 64) * FPITCH + sk) = xr[i];
      *(f32x4*)(ws + (srow + i * 64) * FPITCH + sk) = wr[i];
    }
  };

  f32x4 acc[4][4];  // [feature subtile j][row subtile i]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  fetch(0);
  put(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) fetch(kt + 1);  // global loads of the next K step fly during this step's MFMAs
    const float* xs = smem + (kt & 1) * 2 * FTILE;
    const float* ws = xs + FTILE;
#pragma unroll
    for (int ss = 0; ss < FBK / 4; ++ss) {
      float a[4], b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = ws[(wn * 64 + j * 16 + fr) * FPITCH + ss * 4 + fq];
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = xs[(wm * 64 + i * 16 + fr) * FPITCH + ss * 4 + fq];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[i], acc[j][i], 0, 0, 0);
    }
    if (kt + 1 < nk) put((kt + 1) & 1);
    __syncthreads();
  }

  // D[row = feature 4 fq + r][col = token fr]
  const bool vec = (N % 4 == 0) && (ldy % 4 == 0) && (!resid || ldr % 4 == 0) && !((uintptr_t)Y & 15) &&
                   !(resid && ((uintptr_t)resid & 15)) && !(bias && ((uintptr_t)bias & 15));
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wn * 64 + j * 16 + fq * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 64 + i * 16 + fr;
      if (m >= M || n >= N) continue;
      const int64_t rrow = resid_period ? m % resid_period : m;
      f32x4 v = acc[j][i];
      if (vec) {
        if (bias) v += *(const f32x4*)(bias + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = apply_act<ACT, true>(v[r]);
        if (resid) v += *(const f32x4*)(resid + rrow * ldr + n);
        *(f32x4*)(Y + (int64_t)m * ldy + n) = v;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r >= N) continue;
          float e = apply_act<ACT, true>(v[r] + (bias ? bias[n + r] : 0.f));
          if (resid) e += resid[rrow * ldr + n + r];
          Y[(int64_t)m * ldy + n + r] = e;
        }
      }
    }
  }
}

}  // namespace

extern "C" int pm_linear_f32(const float* x, int64_t ldx, int64_t x_rows_per_batch, int64_t x_batch_stride, const float* w,
                             int64_t ldw, const float* bias, const float* resid, int64_t ldr, int64_t resid_period, float* y,
                             int64_t ldy, int64_t M, int64_t N, int64_t K, int act, void* stream) {
  if (!x || !w || !y || M < 0 || N <= 0 || K <= 0) return PM_EINVAL;
  if (M == 0) return PM_OK;
  if (ldx < 0 || ldw < K || ldy < N || (resid && ldr < N) || x_rows_per_batch < 0 || resid_period < 0) return PM_EINVAL;
  if (ldx % 4 || ldw % 4 || x_batch_stride % 4 || (((uintptr_t)x | (uintptr_t)w) & 15)) return PM_EALIGN;  // 16-byte row chunks
  if (M > (1 << 30) || N > (1 << 30) || K > (1 << 30) || resid_period > (1 << 30)) return PM_EINVAL;
  const int tiles_m = (int)((M + FBM - 1) / FBM), tiles_n = (int)((N + FBN - 1) / FBN);
  const int64_t nblk = (int64_t)tiles_m * tiles_n;
  if (nblk > 0x7fffffff) return PM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
#define PM_F32GO(A)                                                                                                       \
  hipLaunchKernelGGL((linear_f32_kernel<A>), dim3((unsigned)nblk), dim3(256), 0, st, x, ldx, (int)x_rows_per_batch,          \
                     x_batch_stride, w, ldw, bias, resid, ldr, (int)resid_period, y, ldy, (int)M, (int)N, (int)K, tiles_n)
  switch (act) {
    case PM_ACT_NONE: PM_F32GO(PM_ACT_NONE); break;
    case PM_ACT_GELU: PM_F32GO(PM_ACT_GELU); break;
    case PM_ACT_GELU_TANH: PM_F32GO(PM_ACT_GELU_TANH); break;
    case PM_ACT_RELU: PM_F32GO(PM_ACT_RELU); break;
    case PM_ACT_SILU: PM_F32GO(PM_ACT_SILU); break;
    default: return PM_EINVAL;
  }
#undef PM_F32GO
  PM_CHECK_LAUNCH();
  return PM_OK;
}
