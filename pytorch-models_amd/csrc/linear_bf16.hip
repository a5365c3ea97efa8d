// linear_bf16.hip - y = act(x w^T + bias) + resid on MFMA, the nn.Linear of the transformer blocks
// (reference: pytorch_models/transformer.py:28-31,47-53,59-66 and the residual adds at :98-100,124-125).
//
// Roofline: MFMA-bound (2*M*N*K flop against (M*K + N*K + M*N)*2 bytes).
//
// Tile: 128 (tokens) x 128 (features) x 64 (K) per 256-thread workgroup, 4 waves as 2 x 2, each wave
// 64 x 64 = 4 x 4 MFMA 16x16x32 bf16 tiles.  Operands are staged HBM -> LDS with 16-byte
// global_load_lds into a double buffer (2 x 32 KiB -> 2 workgroups per CU, so one workgroup's
// epilogue overlaps the other's main loop); the XOR swizzle is applied on the SOURCE address because
// the LDS side of global_load_lds is lane-linear.  The MFMA is issued with the WEIGHT tile as the A
// operand and the token tile as B, so an accumulator lane holds 4 consecutive features of one token:
// the epilogue packs them into one 8-byte (bf16) or 16-byte (f32) store.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;  // 16 KiB per operand tile

// Stage a 128-row x 64-col bf16 tile of G into `tile`: each wave copies 32 rows with 4 instructions of
// 8 rows x 128 B.  off[i] is the element offset of (this lane's row, this lane's swizzled 16-byte chunk).
__device__ __forceinline__ void stage_tile(const bf16* __restrict__ G, const int64_t (&off)[4], int k0, char* tile,
                                           int wave) {
#pragma unroll
  for (int i = 0; i < 4; ++i) glds16(G + off[i] + k0, tile + (wave * 32 + i * 8) * 128);
}

__device__ __forceinline__ bf16x8 read_frag(const char* tile, int row, int chunk) {
  return *(const bf16x8*)(tile + row * 128 + swz_pos(row, chunk) * 16);
}

template <int ACT, bool YF32>
__global__ __launch_bounds__(256, 2) void linear_bf16_kernel(
    const bf16* __restrict__ X, int64_t ldx, const bf16* __restrict__ W, int64_t ldw, const float* __restrict__ bias,
    const void* resid, int64_t ldr, int resid_f32, int resid_period, void* Y, int64_t ldy, int M, int N, int K,
    int tiles_n, int x_rows_per_batch, int64_t x_batch_stride, int vec_ok) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int wm = wave >> 1, wn = wave & 1;


  f32x4 acc[4][4];  // [feature subtile j][token subtile i]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // row offsets are K-loop invariants: out-of-range rows are clamped (and masked at the store); rows of x may be
  // addressed in two levels (batch, row-in-batch) so that e.g. a strided conv window walks a padded buffer
  int64_t xoff[4], woff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rt = wave * 32 + i * 8 + (lane >> 3);
    const int chunk = swz_pos(rt, lane & 7);  // involution: position p holds chunk p ^ f(row)
    int gm = m0 + rt;
    gm = gm < M ? gm : M - 1;
    int gn = n0 + rt;
    gn = gn < N ? gn : N - 1;
    if (x_rows_per_batch > 0) {
      const int bb = gm / x_rows_per_batch;
      xoff[i] = (int64_t)bb * x_batch_stride + (int64_t)(gm - bb * x_rows_per_batch) * ldx + chunk * 8;
    } else {
      xoff[i] = (int64_t)gm * ldx + chunk * 8;
    }
    woff[i] = (int64_t)gn * ldw + chunk * 8;
  }
  const int nk = K / BK;
  stage_tile(X, xoff, 0, smem, wave);
  stage_tile(W, woff, 0, smem + TILE_BYTES, wave);
  wait_vmcnt0();
  __syncthreads();

  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    char* xcur = smem + (kt & 1) * 2 * TILE_BYTES;  // buffer b: X tile at 2b, W tile at 2b + 1
    char* wcur = xcur + TILE_BYTES;
    if (kt + 1 < nk) {
      char* xnxt = smem + ((kt + 1) & 1) * 2 * TILE_BYTES;
      stage_tile(X, xoff, (kt + 1) * BK, xnxt, wave);
      stage_tile(W, woff, (kt + 1) * BK, xnxt + TILE_BYTES, wave);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[4], b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = read_frag(wcur, wn * 64 + j * 16 + fr, s * 4 + fq);
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = read_frag(xcur, wm * 64 + i * 16 + fr, s * 4 + fq);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], b[i], acc[j][i], 0, 0, 0);
    }
    wait_vmcnt0();
    __syncthreads();
  }

  // epilogue: D[row = feature 4*fq + r][col = token fr]
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wn * 64 + j * 16 + fq * 4;
    if (n >= N) continue;
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias) {
      if (vec_ok) {
        bv = *(const f32x4*)(bias + n);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = n + r < N ? bias[n + r] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 64 + i * 16 + fr;
      if (m >= M) continue;
      f32x4 v = acc[j][i] + bv;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = apply_act<ACT, YF32>(v[r]);
      if (vec_ok) {  // N, ldy, ldr multiples of 4: a lane's 4 features are one aligned vector
        if (resid) {
          const int64_t ro = (int64_t)(resid_period ? m % resid_period : m) * ldr + n;
          if (resid_f32) {
            v += *(const f32x4*)((const float*)resid + ro);
          } else {
            const bf16x4 rv = *(const bf16x4*)((const bf16*)resid + ro);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
          }
        }
        if constexpr (YF32) {
          *(f32x4*)((float*)Y + (int64_t)m * ldy + n) = v;
        } else {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16)v[r];
          *(bf16x4*)((bf16*)Y + (int64_t)m * ldy + n) = o;
        }
      } else {  // ragged N (e.g. a 51865-row vocabulary): element-wise
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r >= N) continue;
          float e = v[r];
          if (resid) {
            const int64_t ro = (int64_t)(resid_period ? m % resid_period : m) * ldr + n + r;
            e += resid_f32 ? ((const float*)resid)[ro] : (float)((const bf16*)resid)[ro];
          }
          if constexpr (YF32) ((float*)Y)[(int64_t)m * ldy + n + r] = e;
          else ((bf16*)Y)[(int64_t)m * ldy + n + r] = (bf16)e;
        }
      }
    }
  }
}

template <bool YF32>
int launch_act(int act, dim3 grid, hipStream_t st, const bf16* X, int64_t ldx, const bf16* W, int64_t ldw,
               const float* bias, const void* resid, int64_t ldr, int resid_f32, int resid_period, void* Y, int64_t ldy,
               int M, int N, int K, int tiles_n, int xrpb, int64_t xbs, int vec_ok) {
#define PM_GO(A)                                                                                                    \
  hipLaunchKernelGGL((linear_bf16_kernel<A, YF32>), grid, dim3(256), 0, st, X, ldx, W, ldw, bias, resid, ldr, resid_f32, \
                     resid_period, Y, ldy, M, N, K, tiles_n, xrpb, xbs, vec_ok);                                    \
  break
  switch (act) {
    case PM_ACT_NONE: PM_GO(PM_ACT_NONE);
    case PM_ACT_GELU: PM_GO(PM_ACT_GELU);
    case PM_ACT_GELU_TANH: PM_GO(PM_ACT_GELU_TANH);
    case PM_ACT_RELU: PM_GO(PM_ACT_RELU);
    case PM_ACT_SILU: PM_GO(PM_ACT_SILU);
    default: return PM_EINVAL;
  }
#undef PM_GO
  return PM_OK;
}

}  // namespace

static int linear_impl(const void* x, int64_t ldx, int64_t x_rows_per_batch, int64_t x_batch_stride, const void* w,
                       int64_t ldw, const float* bias, const void* resid, int64_t ldr, int resid_dtype,
                       int64_t resid_period, void* y, int64_t ldy, int y_dtype, int64_t M, int64_t N, int64_t K, int act,
                       void* stream) {
  if (!x || !w || !y || M < 0 || N <= 0 || K <= 0) return PM_EINVAL;
  if (M == 0) return PM_OK;
  if (y_dtype != PM_BF16 && y_dtype != PM_F32) return PM_EINVAL;
  if (resid && resid_dtype != PM_BF16 && resid_dtype != PM_F32) return PM_EINVAL;
  if (K % BK != 0) return PM_EUNSUPPORTED;
  if (ldx < 0 || ldw < K || ldy < N || (resid && ldr < N) || x_rows_per_batch < 0 || resid_period < 0) return PM_EINVAL;
  if (ldx % 8 || ldw % 8 || x_batch_stride % 8) return PM_EALIGN;
  if (((uintptr_t)x | (uintptr_t)w) & 15) return PM_EALIGN;
  const int vec_ok = (N % 4 == 0) && (ldy % 4 == 0) && (!resid || ldr % 4 == 0) &&
                     !((uintptr_t)y & (y_dtype == PM_F32 ? 15 : 7)) && !(bias && ((uintptr_t)bias & 15)) &&
                     !(resid && ((uintptr_t)resid & (resid_dtype == PM_F32 ? 15 : 7)));
  if (M > (1 << 30) || N > (1 << 30) || K > (1 << 30) || resid_period > (1 << 30)) return PM_EINVAL;
  const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (int)((N + BN - 1) / BN);
  const int64_t nblk = (int64_t)tiles_m * tiles_n;
  if (nblk > 0x7fffffff) return PM_EINVAL;
  dim3 grid((unsigned)nblk);
  hipStream_t st = (hipStream_t)stream;
  int rc = (y_dtype == PM_F32)
               ? launch_act<true>(act, grid, st, (const bf16*)x, ldx, (const bf16*)w, ldw, bias, resid, ldr,
                                  resid_dtype == PM_F32, (int)resid_period, y, ldy, (int)M, (int)N, (int)K, tiles_n,
                                  (int)x_rows_per_batch, x_batch_stride, vec_ok)
               : launch_act<false>(act, grid, st, (const bf16*)x, ldx, (const bf16*)w, ldw, bias, resid, ldr,
                                   resid_dtype == PM_F32, (int)resid_period, y, ldy, (int)M, (int)N, (int)K, tiles_n,
                                   (int)x_rows_per_batch, x_batch_stride, vec_ok);
  if (rc != PM_OK) return rc;
  PM_CHECK_LAUNCH();
  return PM_OK;
}

extern "C" int pm_linear_bf16(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias,
                              const void* resid, int64_t ldr, int resid_dtype, void* y, int64_t ldy, int y_dtype,
                              int64_t M, int64_t N, int64_t K, int act, void* stream) {
  if (ldx < K) return PM_EINVAL;
  return linear_impl(x, ldx, 0, 0, w, ldw, bias, resid, ldr, resid_dtype, 0, y, ldy, y_dtype, M, N, K, act, stream);
}

extern "C" int pm_linear_bf16_ex(const void* x, int64_t ldx, int64_t x_rows_per_batch, int64_t x_batch_stride,
                                 const void* w, int64_t ldw, const float* bias, const void* resid, int64_t ldr,
                                 int resid_dtype, int64_t resid_period, void* y, int64_t ldy, int y_dtype, int64_t M,
                                 int64_t N, int64_t K, int act, void* stream) {
  return linear_impl(x, ldx, x_rows_per_batch, x_batch_stride, w, ldw, bias, resid, ldr, resid_dtype, resid_period, y, ldy,
                     y_dtype, M, N, K, act, stream);
}
