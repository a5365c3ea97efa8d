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

// Stage a 128-row x 64-col bf16 tile (rows row0.., cols k0..k0+63) of G into `tile`.
// Each wave copies 32 rows with 4 instructions of 8 rows x 128 B.
__device__ __forceinline__ void stage_tile(const bf16* __restrict__ G, int64_t ld, int row0, int row_max, int k0,
                                           char* tile, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rbase = wave * 32 + i * 8;
    const int rt = rbase + (lane >> 3);
    const int chunk = swz_pos(rt, lane & 7);  // involution: position p holds chunk p ^ f(row)
    int grow = row0 + rt;
    grow = grow < row_max ? grow : row_max - 1;  // clamp: out-of-range rows are masked at the store
    glds16(G + (int64_t)grow * ld + k0 + chunk * 8, tile + rbase * 128);
  }
}

__device__ __forceinline__ bf16x8 read_frag(const char* tile, int row, int chunk) {
  return *(const bf16x8*)(tile + row * 128 + swz_pos(row, chunk) * 16);
}

template <int ACT, bool YF32>
__global__ __launch_bounds__(256, 2) void linear_bf16_kernel(
    const bf16* __restrict__ X, int64_t ldx, const bf16* __restrict__ W, int64_t ldw, const float* __restrict__ bias,
    const void* resid, int64_t ldr, int resid_f32, void* Y, int64_t ldy, int M, int N, int K,
    int tiles_n) {
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

  const int nk = K / BK;
  stage_tile(X, ldx, m0, M, 0, smem, wave, lane);
  stage_tile(W, ldw, n0, N, 0, smem + TILE_BYTES, wave, lane);
  wait_vmcnt0();
  __syncthreads();

  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    char* xcur = smem + (kt & 1) * 2 * TILE_BYTES;  // buffer b: X tile at 2b, W tile at 2b + 1
    char* wcur = xcur + TILE_BYTES;
    if (kt + 1 < nk) {
      char* xnxt = smem + ((kt + 1) & 1) * 2 * TILE_BYTES;
      stage_tile(X, ldx, m0, M, (kt + 1) * BK, xnxt, wave, lane);
      stage_tile(W, ldw, n0, N, (kt + 1) * BK, xnxt + TILE_BYTES, wave, lane);
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
    if (n >= N) continue;  // N % 4 == 0: a lane's 4 features are all in or all out
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *(const f32x4*)(bias + n);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 64 + i * 16 + fr;
      if (m >= M) continue;
      f32x4 v = acc[j][i] + bv;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = apply_act<ACT, YF32>(v[r]);
      if (resid) {
        if (resid_f32) {
          v += *(const f32x4*)((const float*)resid + (int64_t)m * ldr + n);
        } else {
          const bf16x4 rv = *(const bf16x4*)((const bf16*)resid + (int64_t)m * ldr + n);
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
    }
  }
}

template <bool YF32>
int launch_act(int act, dim3 grid, hipStream_t st, const bf16* X, int64_t ldx, const bf16* W, int64_t ldw,
               const float* bias, const void* resid, int64_t ldr, int resid_f32, void* Y, int64_t ldy, int M, int N,
               int K, int tiles_n) {
#define PM_GO(A)                                                                                                    \
  hipLaunchKernelGGL((linear_bf16_kernel<A, YF32>), grid, dim3(256), 0, st, X, ldx, W, ldw, bias, resid, ldr, resid_f32, \
                     Y, ldy, M, N, K, tiles_n);                                                                     \
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

extern "C" int pm_linear_bf16(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias,
                              const void* resid, int64_t ldr, int resid_dtype, void* y, int64_t ldy, int y_dtype,
                              int64_t M, int64_t N, int64_t K, int act, void* stream) {
  if (!x || !w || !y || M < 0 || N <= 0 || K <= 0) return PM_EINVAL;
  if (M == 0) return PM_OK;
  if (y_dtype != PM_BF16 && y_dtype != PM_F32) return PM_EINVAL;
  if (resid && resid_dtype != PM_BF16 && resid_dtype != PM_F32) return PM_EINVAL;
  if (K % BK != 0 || N % 4 != 0) return PM_EUNSUPPORTED;
  if (ldx < K || ldw < K || ldy < N || (resid && ldr < N)) return PM_EINVAL;
  if (ldx % 8 || ldw % 8 || ldy % 4 || (resid && ldr % 4)) return PM_EALIGN;
  if (((uintptr_t)x | (uintptr_t)w) & 15) return PM_EALIGN;
  if (((uintptr_t)y & (y_dtype == PM_F32 ? 15 : 7)) || (bias && ((uintptr_t)bias & 15))) return PM_EALIGN;
  if (resid && ((uintptr_t)resid & (resid_dtype == PM_F32 ? 15 : 7))) return PM_EALIGN;
  if (M > (1 << 30) || N > (1 << 30) || K > (1 << 30)) return PM_EINVAL;
  const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (int)((N + BN - 1) / BN);
  const int64_t nblk = (int64_t)tiles_m * tiles_n;
  if (nblk > 0x7fffffff) return PM_EINVAL;
  dim3 grid((unsigned)nblk);
  hipStream_t st = (hipStream_t)stream;
  int rc = (y_dtype == PM_F32)
               ? launch_act<true>(act, grid, st, (const bf16*)x, ldx, (const bf16*)w, ldw, bias, resid, ldr,
                                  resid_dtype == PM_F32, y, ldy, (int)M, (int)N, (int)K, tiles_n)
               : launch_act<false>(act, grid, st, (const bf16*)x, ldx, (const bf16*)w, ldw, bias, resid, ldr,
                                   resid_dtype == PM_F32, y, ldy, (int)M, (int)N, (int)K, tiles_n);
  if (rc != PM_OK) return rc;
  PM_CHECK_LAUNCH();
  return PM_OK;
}
