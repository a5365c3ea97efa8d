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
#include <stdlib.h>

#include "common.h"
// Priority: waves 4-7 (the younger wave of each SIMD pair, which loses every age-based arbitration and made waves 0-3
// wait ~600-1200 cycles at each barrier) run at s_setprio 1 for the whole kernel; no per-step flips
// (MI355X_MICROARCH.md, two waves per SIMD, item 4).  out_proj -4.5 %, fc2 -3 %, fc1 / QKV -1 %.  0 = the old flips.
#ifndef PM_STATIC_PRIO
#define PM_STATIC_PRIO 1
#endif

// linear_bf16_wide.hip
int pm_linear_bf16_wide_launch(const void* x, int64_t ldx, int64_t x_rows_per_batch, int64_t x_batch_stride, const void* w,
                               int64_t ldw, const float* bias, const void* resid, int64_t ldr, int64_t resid_period, void* y,
                               int64_t ldy, int64_t M, int64_t N, int64_t K, int act, PmLnFold ln, hipStream_t st);
bool pm_linear_bf16_wide_applies(int64_t M, int64_t N, int64_t K, int act);
// linear_bf16_tile.hip
int pm_linear_bf16_tile_launch(int mi, const void* x, int64_t ldx, int64_t x_rows_per_batch, int64_t x_batch_stride,
                               const void* w, int64_t ldw, const float* bias, const void* resid, int64_t ldr,
                               int64_t resid_period, void* y, int64_t ldy, int64_t M, int64_t N, int64_t K, int act, PmLnFold ln,
                               hipStream_t st);
bool pm_linear_bf16_tile_applies(int64_t M, int64_t N, int64_t K, int act);
// experiments/linear_bf16_sk.hip: stream-K and hybrid forms (tiles cut along K; opt-in, measured slower or batch-position
// dependent: DESIGN.md section 8).  Only in `make experiments` builds; the product keeps whole tiles.
#ifdef PM_EXPERIMENTS
int pm_linear_bf16_sk_launch(const void* x, int64_t ldx, int64_t x_rows_per_batch, int64_t x_batch_stride, const void* w,
                             int64_t ldw, const float* bias, const void* resid, int64_t ldr, int64_t resid_period, void* y,
                             int64_t ldy, int64_t M, int64_t N, int64_t K, int act, PmLnFold ln, void* ws, hipStream_t st,
                             int mode);
bool pm_linear_bf16_hyb_applies(int64_t M, int64_t N, int64_t K, int act);
bool pm_linear_bf16_sk_applies(int64_t M, int64_t N, int64_t K, int act);
int64_t pm_linear_sk_ws_bytes();
#else
static int pm_linear_bf16_sk_launch(const void*, int64_t, int64_t, int64_t, const void*, int64_t, const float*, const void*, int64_t,
                                    int64_t, void*, int64_t, int64_t, int64_t, int64_t, int, PmLnFold, void*, hipStream_t, int) {
  return PM_EUNSUPPORTED;
}
static bool pm_linear_bf16_hyb_applies(int64_t, int64_t, int64_t, int) { return false; }
static bool pm_linear_bf16_sk_applies(int64_t, int64_t, int64_t, int) { return false; }
static int64_t pm_linear_sk_ws_bytes() { return 0; }
#endif

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;  // 16 KiB per operand tile

__device__ __attribute__((aligned(16))) const unsigned short pm_zero_page[8] = {0, 0, 0, 0, 0, 0, 0, 0};

// Stage a 128-row x 64-col bf16 tile of G into `tile`: each wave copies 32 rows with 4 instructions of
// 8 rows x 128 B.  off[i] is the element offset of (this lane's row, this lane's swizzled 16-byte chunk), kc[i] that
// chunk's first column; chunks at or beyond K (a K that is not a multiple of 64) are fed from a zero page.
__device__ __forceinline__ void stage_tile(const bf16* __restrict__ G, const int64_t (&off)[4], const int (&kc)[4], int k0,
                                           int K, char* tile, int wave) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const void* src = (k0 + kc[i] < K) ? (const void*)(G + off[i] + k0) : (const void*)pm_zero_page;
    glds16(src, tile + (wave * 32 + i * 8) * 128);
  }
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
  int kc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rt = wave * 32 + i * 8 + (lane >> 3);
    const int chunk = swz_pos(rt, lane & 7);  // involution: position p holds chunk p ^ f(row)
    kc[i] = chunk * 8;
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
  const int nk = (K + BK - 1) / BK;
  stage_tile(X, xoff, kc, 0, K, smem, wave);
  stage_tile(W, woff, kc, 0, K, smem + TILE_BYTES, wave);
  wait_vmcnt0();
  __syncthreads();

  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    char* xcur = smem + (kt & 1) * 2 * TILE_BYTES;  // buffer b: X tile at 2b, W tile at 2b + 1
    char* wcur = xcur + TILE_BYTES;
    if (kt + 1 < nk) {
      char* xnxt = smem + ((kt + 1) & 1) * 2 * TILE_BYTES;
      stage_tile(X, xoff, kc, (kt + 1) * BK, K, xnxt, wave);
      stage_tile(W, woff, kc, (kt + 1) * BK, K, xnxt + TILE_BYTES, wave);
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

// ---------------------------------------------------------------------------------------------------------------
// Large-M variant: PERSISTENT, 256 (tokens) x 128 (features) x 64 tile, 8 waves as 4 x 2 (each wave still 64 x 64),
// one workgroup per CU, 3-stage LDS ring (3 x 48 KiB = 144 KiB).
//  * The K steps of all tiles a workgroup owns form ONE stream; two steps are always in flight: the wait for step p
//    is a COUNTED s_waitcnt vmcnt(6) (6 = LDS-DMA instructions per wave per step) followed by a raw s_barrier, so the
//    loads of step p+1 stay in flight across the barrier and step p+2 is issued right behind it
//    (cdna_hip_programming.md, "Pipelining across barriers").
//  * Because the stream runs across tile boundaries, the first two K steps of the NEXT tile are already in flight while
//    the current tile's epilogue runs: at K = 768 (12 steps per tile) the per-tile prologue / epilogue bubble was ~40 %
//    of the tile time in the one-tile-per-workgroup form.
//  * Epilogue: bias / activation in the accumulator layout, then each wave transposes its 64 x 64 result IN FP32 through
//    4 KiB of the ring buffer the last step just freed (16 tokens at a time); the residual is read and the result is
//    stored row-wise - 128-byte segments, 16 bytes per lane - and rounded to bf16 exactly once.
//  * XCD locality: the 32 workgroups that share an XCD (blockIdx % 8: a label, speed only) walk one contiguous
//    eighth of the tile list together, so they share token panels and the weight panel in that XCD's L2.
constexpr int LBM = 256, LBN = 128;
constexpr int STAGE_BYTES = (LBM + LBN) * BK * 2;  // 48 KiB
constexpr int PERSIST_WGS = 256;
constexpr int GROUP_M = 4;  // tile order: super-rows of 4 token panels, feature tiles inside, panels innermost

// Tile id -> (token panel, feature tile).  Consecutive ids walk GROUP_M token panels for one feature tile, then the next
// feature tile: the 32 tiles an XCD works on at once form a ~4 x 8 block (4 token panels + 8 weight panels = 3.1 MB at
// K = 768, inside the 4 MB L2) instead of 1.3 x 24 (the whole 4.7 MB weight matrix per panel: measured 66 % L2 hit
// rate and 2.8x the algorithmic bytes crossing the fabric).
__device__ __forceinline__ void tile_coords(int t, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int per = GROUP_M * tiles_n;
  const int sr = t / per, r = t - sr * per;
  const int left = tiles_m - sr * GROUP_M;
  const int gm = left < GROUP_M ? left : GROUP_M;
  tn = r / gm;
  tm = sr * GROUP_M + (r - tn * gm);
}

template <int ACT, bool YF32>
__global__ __launch_bounds__(512, 2) void linear_bf16_persist_kernel(
    const bf16* __restrict__ X, int64_t ldx, const bf16* __restrict__ W, int64_t ldw, const float* __restrict__ bias,
    const void* resid, int64_t ldr, int resid_f32, int resid_period, void* Y, int64_t ldy, int M, int N, int K,
    int tiles_n, int ntiles, int x_rows_per_batch, int64_t x_batch_stride, int vec_ok, PmLnFold ln) {
  __shared__ __attribute__((aligned(16))) char smem[3 * STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // this workgroup's tiles: chunk of XCD group (blockIdx % 8), strided by the 32 workgroups of the group
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, nloc = gridDim.x >> 3;
  const int cq = ntiles >> 3, cr = ntiles & 7;
  const int tbase = xcd * cq + (xcd < cr ? xcd : cr), tcount = cq + (xcd < cr ? 1 : 0);
  const int my_tiles = local < tcount ? (tcount - local + nloc - 1) / nloc : 0;
  const int nk = K / BK;
  const int P = my_tiles * nk;  // K steps in this workgroup's stream
  const int tiles_m = ntiles / tiles_n;

  // staging side of the stream.  Written as a macro over plain locals (not a capturing lambda): with mutable
  // by-reference captures hipcc can keep such state in scratch memory, whose loads/stores are VMEM ops that drain
  // the counted-vmcnt pipeline every step (measured: 4x slower).
  int64_t xoff[4], woff[2];
  int pp = 0, pp_kt = 0, pp_tile = 0, pp_buf = 0;
#define PM_STAGE_NEXT()                                                                                              \
  if (pp < P) {                                                                                                      \
    if (pp_kt == 0) {                                                                                                \
      const int t_ = tbase + local + pp_tile * nloc;                                                                 \
      int tm_, tn_;                                                                                                  \
      tile_coords(t_, tiles_m, tiles_n, tm_, tn_);                                                                   \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                \
        const int rt = wave * 32 + i * 8 + (lane >> 3);                                                              \
        int gm = tm_ * LBM + rt;                                                                                     \
        gm = gm < M ? gm : M - 1;                                                                                    \
        const int chunk = swz_pos(rt, lane & 7);                                                                     \
        if (x_rows_per_batch > 0) {                                                                                  \
          const int bb = gm / x_rows_per_batch;                                                                      \
          xoff[i] = (int64_t)bb * x_batch_stride + (int64_t)(gm - bb * x_rows_per_batch) * ldx + chunk * 8;          \
        } else {                                                                                                     \
          xoff[i] = (int64_t)gm * ldx + chunk * 8;                                                                   \
        }                                                                                                            \
      }                                                                                                              \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                \
        const int rt = wave * 16 + i * 8 + (lane >> 3);                                                              \
        int gn = tn_ * LBN + rt;                                                                                     \
        gn = gn < N ? gn : N - 1;                                                                                    \
        woff[i] = (int64_t)gn * ldw + swz_pos(rt, lane & 7) * 8;                                                     \
      }                                                                                                              \
    }                                                                                                                \
    char* xs_ = smem + pp_buf * STAGE_BYTES;                                                                         \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) glds16_aux<PM_GLDS_X_AUX>(X + xoff[i] + pp_kt * BK, xs_ + (wave * 32 + i * 8) * 128); \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                    \
        glds16_aux<PM_GLDS_W_AUX>(W + woff[i] + pp_kt * BK, xs_ + LBM * 128 + (wave * 16 + i * 8) * 128);                                \
    ++pp;                                                                                                            \
    pp_buf = pp_buf == 2 ? 0 : pp_buf + 1;                                                                           \
    if (++pp_kt == nk) { pp_kt = 0; ++pp_tile; }                                                                     \
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

#if PM_STATIC_PRIO
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  PM_STAGE_NEXT();
  PM_STAGE_NEXT();
  const int fr = lane & 15, fq = lane >> 4;
  const bool staged_epi = vec_ok && !YF32 && !(resid && resid_f32);
  bf16x8 rv[4][2];
  f32x2 lnst[4] = {{0.f, 1.f}, {0.f, 1.f}, {0.f, 1.f}, {0.f, 1.f}};
  int buf = 0, kt = 0, ti = 0;
  for (int pc = 0; pc < P; ++pc) {
    // step pc landed; step pc+1 may stay in flight.  (Exact bookkeeping that also lets the epilogue's stores stay in
    // flight across tile boundaries was measured: no gain, +VGPRs - the epilogue is VALU/LDS-bound, not drain-bound.)
    if (pp - pc >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave's part landed; every wave is past step pc-1: its buffer is free
    PM_STAGE_NEXT();
    if (kt == nk - 1 && staged_epi && resid) {
      // last K step of the tile: request the residual now (coalesced 16-byte loads in the STORE layout) so that its
      // latency hides under this step's 32 MFMAs instead of stalling the epilogue
      int tm_r, tn_r;
      tile_coords(tbase + local + ti * nloc, tiles_m, tiles_n, tm_r, tn_r);
      const int m0r = tm_r * LBM + wm * 64, n0r = tn_r * LBN + wn * 64;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          int mm = m0r + i * 16 + (lane >> 3) + p * 8;
          mm = mm < M ? mm : M - 1;
          int nn = n0r + (lane & 7) * 8;
          nn = nn < N ? nn : N - 8;
          rv[i][p] = *(const bf16x8*)((const bf16*)resid + (int64_t)(resid_period ? mm % resid_period : mm) * ldr + nn);
        }
    }
    if (kt == nk - 1 && staged_epi && ln.stats) {  // LayerNorm fold: (mean, rstd) of this lane's rows m0 + 16 i + fr
      int tm_r, tn_r;
      tile_coords(tbase + local + ti * nloc, tiles_m, tiles_n, tm_r, tn_r);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int mr = tm_r * LBM + wm * 64 + i * 16 + fr;
        mr = mr < M ? mr : M - 1;
        lnst[i] = *(const f32x2*)(ln.stats + 2 * (int64_t)mr);
      }
    }
    const char* xcur = smem + buf * STAGE_BYTES;
    const char* wcur = xcur + LBM * 128;
    {
      // all 16 fragment reads of the K step are issued before its first MFMA; the compiler then waits with counted
      // lgkmcnt(N) per operand, so only the first read's latency is exposed and the rest hides behind the 32 MFMAs
      bf16x8 a[2][4], b[2][4];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int j = 0; j < 4; ++j) a[s][j] = read_frag(wcur, wn * 64 + j * 16 + fr, s * 4 + fq);
#pragma unroll
        for (int i = 0; i < 4; ++i) b[s][i] = read_frag(xcur, wm * 64 + i * 16 + fr, s * 4 + fq);
      }
      __builtin_amdgcn_sched_barrier(0);
#if !PM_STATIC_PRIO
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (s == 0 && kt == 0) {  // a tile's first MFMAs start from the constant 0: no accumulator clearing anywhere
          const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][j], b[s][i], zero, 0, 0, 0);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
              acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][j], b[s][i], acc[j][i], 0, 0, 0);
        }
      }
#if !PM_STATIC_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
    const int cbuf = buf;
    buf = buf == 2 ? 0 : buf + 1;
    if (++kt < nk) continue;

    // ---------------- tile finished: epilogue (the next tile's first two K steps are already in flight)
    kt = 0;
#ifdef PM_ABLATE_EPILOGUE  // experiment only: keep the accumulators live, skip the epilogue
    {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(acc[j][i]));
      ++ti;
      continue;
    }
#endif
    const int t = tbase + local + ti * nloc;
    ++ti;
    int tm, tn;
    tile_coords(t, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * LBM + wm * 64, n0 = tn * LBN + wn * 64;
    if (staged_epi) {
      // fp32 staging through the ring buffer this step just consumed (free once every wave is past its MFMAs;
      // it is not re-filled before the barrier at the top of the next step): 4 KiB per wave.
      __builtin_amdgcn_s_barrier();
      char* stg = smem + cbuf * STAGE_BYTES + wave * 4096;
      const int srow = lane >> 3, sch = lane & 7;  // row-wise side: 8 lanes x 8 features per 64-feature row segment
      // one 64-bit multiply per tile: this lane's rows are m0 + srow + 8 k, every other address a uniform offset from it
      bf16* const ybase = (bf16*)Y + (int64_t)(m0 + srow) * ldy + n0 + sch * 8;
      float* const sbase = ln.row_out ? ln.row_out + ((int64_t)(m0 + srow) * (N >> 6) + (n0 >> 6)) * 2 : nullptr;
      const int mleft = M - m0 - srow;  // row k of this lane exists iff 8 k < mleft
      const bool n_ok = n0 + sch * 8 < N;
      f32x4 bvec[4], svec[4];  // this lane's bias (and LN-fold column sum) values, loaded once per tile
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + j * 16 + fq * 4;
        bvec[j] = (bias && n < N) ? *(const f32x4*)(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        svec[j] = (ln.stats && n < N) ? *(const f32x4*)(ln.s + n) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float mu = lnst[i][0], rstd = lnst[i][1];  // (0, 1) without the LayerNorm fold
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaf(rstd, acc[j][i][r] - mu * svec[j][r], bvec[j][r]);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = apply_act<ACT, false>(v[r]);
          // staging row fr (64 f32 = 256 B), 16-byte chunk c = 4j + fq stored at position c ^ fr
          *(f32x4*)(stg + fr * 256 + (((4 * j + fq) ^ fr) * 16)) = v;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int row = srow + p * 8;
          const f32x4 lo = *(const f32x4*)(stg + row * 256 + (((2 * sch) ^ row) * 16));
          const f32x4 hi = *(const f32x4*)(stg + row * 256 + (((2 * sch + 1) ^ row) * 16));
          bf16x8 o;
          if (resid) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { o[r] = (bf16)(lo[r] + (float)rv[i][p][r]); o[4 + r] = (bf16)(hi[r] + (float)rv[i][p][4 + r]); }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) { o[r] = (bf16)lo[r]; o[4 + r] = (bf16)hi[r]; }
          }
          const bool ok = i * 16 + p * 8 < mleft && n_ok;
          if (ln.row_out) {  // (sum, sum of squares) of this row's 64 ROUNDED outputs: the next LayerNorm's partials
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) { const float f = (float)o[r]; s1 += f; s2 = fmaf(f, f, s2); }
            s1 = sum8_dpp(s1);  // the row's 8 lanes (sch 0..7): DPP adds, no LDS traffic
            s2 = sum8_dpp(s2);
            if (sch == 0 && ok) *(f32x2*)(sbase + (int64_t)(i * 16 + p * 8) * (N >> 6) * 2) = f32x2{s1, s2};
          }
#ifdef PM_ABLATE_STORES  // experiment only
          asm volatile("" ::"v"(o));
#else
          if (ok) store_y((bf16x8*)(ybase + (int64_t)(i * 16 + p * 8) * ldy), o);  // N % 8 == 0 on this path
#endif
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + j * 16 + fq * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = m0 + i * 16 + fr;
          f32x4 v = acc[j][i];
          if (n >= N || m >= M) continue;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (n + r >= N) continue;
            float e = v[r] + (bias ? bias[n + r] : 0.f);
            e = apply_act<ACT, YF32>(e);
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
}

template <bool YF32>
int launch_act(int act, bool big, dim3 grid, hipStream_t st, const bf16* X, int64_t ldx, const bf16* W, int64_t ldw,
               const float* bias, const void* resid, int64_t ldr, int resid_f32, int resid_period, void* Y, int64_t ldy,
               int M, int N, int K, int tiles_n, int xrpb, int64_t xbs, int vec_ok, PmLnFold ln) {
#define PM_GO(A)                                                                                                       \
  if (big)                                                                                                             \
    hipLaunchKernelGGL((linear_bf16_persist_kernel<A, YF32>), dim3(PERSIST_WGS), dim3(512), 0, st, X, ldx, W, ldw, bias, \
                       resid, ldr, resid_f32, resid_period, Y, ldy, M, N, K, tiles_n, (int)grid.x, xrpb, xbs, vec_ok, ln); \
  else                                                                                                                 \
    hipLaunchKernelGGL((linear_bf16_kernel<A, YF32>), grid, dim3(256), 0, st, X, ldx, W, ldw, bias, resid, ldr,          \
                       resid_f32, resid_period, Y, ldy, M, N, K, tiles_n, xrpb, xbs, vec_ok);                          \
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

// ---- which of the three kernels: a wave-quantisation cost model fitted to a sweep of transformer-block shapes
// (tools/dispatch_sweep.py, 96 shapes: mean regret 0.2 %, worst 7 %; the tile-count thresholds it replaces - wide from 1024
// tiles, 256 x 128 from 512 - left 7.8 % on the table, mostly at M = 6 k .. 32 k where a 256 x 256 tiling already fills the
// chip).  Cost = rounds over the resident workgroups x work per tile / relative rate: 256 x 256 tiles on 256 workgroups at
// 1.25, 256 x 128 on 256 at 1.0, 128 x 128 on 512 (two per CU) at 0.7 with a partly filled last round costing
// 0.3 + 0.7 * fill of a full one (co-resident workgroups speed up when their neighbour has finished).
// Stream-K form of the 256 x 256 kernel (linear_bf16_sk.hip): no round quantisation - its cost is the exact tile count over 256
// plus the partial-tile exchange (one 256 KiB store and load per workgroup, ~a quarter of a K = 768 tile), taken when that
// beats the best whole-tile kernel by 3 % (QKV at 6.93 rounds stays whole-tile; linear1 at 9.23, out_proj / linear2 at 2.31 move).
// Hybrid form (linear_bf16_sk.hip, mode 1): the wide kernel's whole tiles + the last round's tiles in two K halves: its cost is
// the whole rounds + half a round + the hand-off of one partial tile.
enum { PM_K_SMALL = 1, PM_K_PERSIST = 2, PM_K_WIDE = 3, PM_K_SK = 4, PM_K_HYB = 5, PM_K_TILE4 = 6, PM_K_TILE5 = 7 };
static int pm_linear_pick_kernel(int64_t M, int64_t N, int64_t K, bool persist_ok, bool wide_ok, bool sk_ok, bool hyb_ok,
                                 bool has_resid, bool tile_ok) {
  static const int forced = [] { const char* e = getenv("PM_GEMM_KERNEL"); return e ? atoi(e) : 0; }();  // experiments: 1 .. 5
  // OFF unless PM_GEMM_STREAMK=1: measured on MI355X (tools/sk_check.py, us, whole-tile 256 x 256 kernel -> stream-K): out_proj
  // 96 -> 118, linear2 316 -> 339, linear1 + GELU 335 -> 352, QKV 196 -> 224, 8192^3 851 -> 935.  The balance is real (every
  // workgroup issues the same MFMAs) but each launch moves 64 MB of fp32 partial tiles out and back (one per workgroup:
  // ~25 us), and contiguous per-workgroup ranges lose the L2 sharing of interleaved neighbouring tiles (-10 % in the K loop).
  static const bool use_sk = [] { const char* e = getenv("PM_GEMM_STREAMK"); return e && atoi(e) != 0; }();
  if (!use_sk && forced != PM_K_SK) sk_ok = false;
  // OFF unless PM_GEMM_HYBRID=1.  It wins where the cost model picks it (linear2 at M = 50432: 264 -> 255 us, ViT-B/16 +0.8 %),
  // but a row in a tail tile is summed as two K halves and a row in a whole tile as one chain: the rounding of a sample then
  // depends on its POSITION in the batch, and the bit-exact batch-permutation invariance this library guarantees (and tests:
  // tests/test_hip_vit.py, tests/test_hip_fuzz.py) is worth more than 0.8 %.
  static const bool use_hyb = [] { const char* e = getenv("PM_GEMM_HYBRID"); return e && atoi(e) != 0; }();
  if (!use_hyb && forced != PM_K_HYB) hyb_ok = false;
  if (forced == PM_K_SK && sk_ok) return PM_K_SK;
  if (forced == PM_K_HYB && hyb_ok) return PM_K_HYB;
  if (forced && forced != PM_K_SK) sk_ok = false;
  if (forced && forced != PM_K_HYB) hyb_ok = false;
  if (forced == PM_K_WIDE && wide_ok) return PM_K_WIDE;
  if ((forced == PM_K_TILE4 || forced == PM_K_TILE5) && tile_ok) return forced;
  static const bool use_tile = [] { const char* e = getenv("PM_GEMM_TILE"); return !e || atoi(e) != 0; }();
  if (!use_tile || forced) tile_ok = false;
  if (forced == PM_K_PERSIST && persist_ok) return PM_K_PERSIST;
  if (forced == PM_K_SMALL) return PM_K_SMALL;
  if (forced == PM_K_PERSIST) wide_ok = false;  // "no wider than": lets a sweep see each kernel's own curve
  const double tm = (double)((M + 255) / 256), tw = tm * (double)((N + 255) / 256), tp = tm * (double)((N + 127) / 128);
  const double ts = (double)((M + 127) / 128) * (double)((N + 127) / 128);
  auto rounds = [](double t, double slots, double frac) {
    const double full = (double)(int64_t)(t / slots), rem = t - full * slots;
    return full + (rem > 0 ? frac + (1.0 - frac) * rem / slots : 0.0);
  };
  // with a residual operand in the epilogue the 256 x 256 tiling keeps less of its edge (second sweep, --resid: rates 1.1 and
  // 0.65 pick within 0.1 % of the best kernel, worst case 3 %)
  const double cs = rounds(ts, 512, 0.3) * 2.0 / (has_resid ? 0.65 : 0.7);
  const double cp = persist_ok ? rounds(tp, 256, 1.0) * 2.0 : 1e30;
  const double cw = wide_ok ? rounds(tw, 256, 1.0) * 4.0 / (has_resid ? 1.1 : 1.25) : 1e30;
  const double csk = sk_ok ? (tw / 256.0 + 0.25 * 768.0 / (double)K) * 4.0 / (has_resid ? 1.1 : 1.25) : 1e30;
  // (64 MI) x 256 tiles (linear_bf16_tile.hip): MI = 4 replaces the wide kernel wherever both apply
  const double t5 = (double)((M + 319) / 320) * (double)((N + 255) / 256);
  const double ct4 = tile_ok ? rounds(tw, 256, 1.0) * 4.0 / (has_resid ? 1.1 : 1.25) : 1e30;
  // measured (tools/tile_check.py, M = 50432): QKV at 7 rounds of 256-row tiles = 6 rounds of 320-row tiles (7.5 units) in the
  // same time, 8192^3 806 against 932 us for 4 against 5 units: per unit of work the taller tile runs ~7 % faster (72 KB instead
  // of 80 KB through the CU's vector-memory path per 320 rows of MFMA work)
  const double ct5 = tile_ok ? rounds(t5, 256, 1.0) * 5.0 / (1.07 * (has_resid ? 1.1 : 1.25)) : 1e30;
  if (tile_ok) {
    const double bt = ct4 <= ct5 ? ct4 : ct5;
    if (bt <= cp && bt <= cs) return ct4 <= ct5 ? PM_K_TILE4 : PM_K_TILE5;
  }
  const double best = cw < cp ? (cw < cs ? cw : cs) : (cp < cs ? cp : cs);
  const double chyb = hyb_ok ? ((double)(int64_t)(tw / 256.0) + 0.5 + 0.25 * 768.0 / (double)K) * 4.0 / (has_resid ? 1.1 : 1.25) : 1e30;
  if (csk < 0.97 * best && csk <= chyb) return PM_K_SK;
  if (chyb < 0.97 * best) return PM_K_HYB;
  if (cw <= cp && cw <= cs) return PM_K_WIDE;
  return cp <= cs ? PM_K_PERSIST : PM_K_SMALL;
}

static int linear_impl(const void* x, int64_t ldx, int64_t x_rows_per_batch, int64_t x_batch_stride, const void* w,
                       int64_t ldw, const float* bias, const void* resid, int64_t ldr, int resid_dtype,
                       int64_t resid_period, void* y, int64_t ldy, int y_dtype, int64_t M, int64_t N, int64_t K, int act,
                       PmLnFold ln, void* stream, void* ws = nullptr, int64_t ws_bytes = 0) {
  if (!x || !w || !y || M < 0 || N <= 0 || K <= 0) return PM_EINVAL;
  const bool want_ln = ln.stats || ln.row_out;
  if ((ln.stats == nullptr) != (ln.s == nullptr)) return PM_EINVAL;
  if (ln.row_out && N % 64) return PM_EUNSUPPORTED;
  if (M == 0) return PM_OK;
  if (y_dtype != PM_BF16 && y_dtype != PM_F32) return PM_EINVAL;
  if (resid && resid_dtype != PM_BF16 && resid_dtype != PM_F32) return PM_EINVAL;
  if (K % 8 != 0) return PM_EUNSUPPORTED;  // 16-byte chunks; a K tail below 64 is zero-filled by the 128 x 128 kernel
  if (ldx < 0 || ldw < K || ldy < N || (resid && ldr < N) || x_rows_per_batch < 0 || resid_period < 0) return PM_EINVAL;
  if (ldx % 8 || ldw % 8 || x_batch_stride % 8) return PM_EALIGN;
  if (((uintptr_t)x | (uintptr_t)w) & 15) return PM_EALIGN;
  const int vec_ok = (N % 4 == 0) && (ldy % 4 == 0) && (!resid || ldr % 4 == 0) &&
                     !((uintptr_t)y & (y_dtype == PM_F32 ? 15 : 7)) && !(bias && ((uintptr_t)bias & 15)) &&
                     !(resid && ((uintptr_t)resid & (resid_dtype == PM_F32 ? 15 : 7)));
  if (M > (1 << 30) || N > (1 << 30) || K > (1 << 30) || resid_period > (1 << 30)) return PM_EINVAL;
  hipStream_t st0 = (hipStream_t)stream;
  const bool out_vec16 = N % 8 == 0 && ldy % 8 == 0 && !((uintptr_t)y & 15);
  const bool wide_base = y_dtype == PM_BF16 && vec_ok && out_vec16 && !(resid && resid_dtype != PM_BF16) &&
                         !(resid && (ldr % 8 || ((uintptr_t)resid & 15))) &&
                         // the 256 x 256 kernels keep 32-bit element offsets per operand
                         (x_rows_per_batch > 0 ? (M / x_rows_per_batch + 1) * x_batch_stride : M * ldx) < (1LL << 32) &&
                         N * ldw < (1LL << 32);
  bool wide_ok = wide_base && pm_linear_bf16_wide_applies(M, N, K, act) && !ln.row_out;
  const bool sk_ok = wide_base && ws && ws_bytes >= pm_linear_sk_ws_bytes() && !((uintptr_t)ws & 15) &&
                     pm_linear_bf16_sk_applies(M, N, K, act) && !(ln.row_out && act != PM_ACT_NONE);
  const bool persist_ok = (K % BK == 0) && (M >= 4096) && (y_dtype == PM_F32 || !vec_ok || out_vec16);
  const bool staged_ok = persist_ok && y_dtype == PM_BF16 && vec_ok && !(resid && resid_dtype == PM_F32);
  const bool hyb_ok = wide_base && ws && ws_bytes >= pm_linear_sk_ws_bytes() && !((uintptr_t)ws & 15) &&
                      pm_linear_bf16_hyb_applies(M, N, K, act) && !(ln.row_out && act != PM_ACT_NONE);
  // (64 MI) x 256 tiles: LDS-DMA pieces of 8 whole rows off uniform row bases (M % 8 == 0, plain row addressing), epilogue
  // modes bias / LayerNorm-fold consumer / residual (+ row partials): a consumer with a residual stays on the 256 x 128 kernel
  const bool tile_ok = wide_base && pm_linear_bf16_tile_applies(M, N, K, act) && M % 8 == 0 && x_rows_per_batch == 0 &&
                       !(ln.stats && (resid || ln.row_out)) && !(ln.row_out && act != PM_ACT_NONE);
  int kernel;
  if (want_ln) {  // the LayerNorm fold lives in the persistent kernels' epilogues (row partials: the tile and 256 x 128 kernels)
    if (!staged_ok) return PM_EUNSUPPORTED;
    kernel = pm_linear_pick_kernel(M, N, K, true, wide_ok, sk_ok, hyb_ok, resid != nullptr, tile_ok);
    if (kernel == PM_K_SMALL) kernel = PM_K_PERSIST;
  } else {
    kernel = pm_linear_pick_kernel(M, N, K, persist_ok, wide_ok, sk_ok, hyb_ok, resid != nullptr, tile_ok);
  }
  if (kernel == PM_K_TILE4 || kernel == PM_K_TILE5) {
    const int rct = pm_linear_bf16_tile_launch(kernel == PM_K_TILE4 ? 4 : 5, x, ldx, x_rows_per_batch, x_batch_stride, w, ldw, bias,
                                               resid, ldr, resid_period, y, ldy, M, N, K, act, ln, st0);
    if (rct != PM_OK) return rct;
    PM_CHECK_LAUNCH();
    return PM_OK;
  }
  if (kernel == PM_K_SK || kernel == PM_K_HYB) {
    const int rck = pm_linear_bf16_sk_launch(x, ldx, x_rows_per_batch, x_batch_stride, w, ldw, bias, resid, ldr, resid_period, y,
                                             ldy, M, N, K, act, ln, ws, st0, kernel == PM_K_HYB ? 1 : 0);
    if (rck != PM_OK) return rck;
    PM_CHECK_LAUNCH();
    return PM_OK;
  }
  if (kernel == PM_K_WIDE) {
    const int rcw = pm_linear_bf16_wide_launch(x, ldx, x_rows_per_batch, x_batch_stride, w, ldw, bias, resid, ldr, resid_period, y,
                                               ldy, M, N, K, act, ln, st0);
    if (rcw != PM_OK) return rcw;
    PM_CHECK_LAUNCH();
    return PM_OK;
  }
  const bool big = kernel == PM_K_PERSIST;
  const int tiles_m = (int)((M + (big ? LBM : BM) - 1) / (big ? LBM : BM)), tiles_n = (int)((N + BN - 1) / BN);
  const int64_t nblk = (int64_t)tiles_m * tiles_n;
  if (nblk > 0x7fffffff) return PM_EINVAL;
  dim3 grid((unsigned)nblk);
  hipStream_t st = (hipStream_t)stream;
  int rc = (y_dtype == PM_F32)
               ? launch_act<true>(act, big, grid, st, (const bf16*)x, ldx, (const bf16*)w, ldw, bias, resid, ldr,
                                  resid_dtype == PM_F32, (int)resid_period, y, ldy, (int)M, (int)N, (int)K, tiles_n,
                                  (int)x_rows_per_batch, x_batch_stride, vec_ok, ln)
               : launch_act<false>(act, big, grid, st, (const bf16*)x, ldx, (const bf16*)w, ldw, bias, resid, ldr,
                                   resid_dtype == PM_F32, (int)resid_period, y, ldy, (int)M, (int)N, (int)K, tiles_n,
                                   (int)x_rows_per_batch, x_batch_stride, vec_ok, ln);
  if (rc != PM_OK) return rc;
  PM_CHECK_LAUNCH();
  return PM_OK;
}

extern "C" int pm_linear_bf16(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias,
                              const void* resid, int64_t ldr, int resid_dtype, void* y, int64_t ldy, int y_dtype,
                              int64_t M, int64_t N, int64_t K, int act, void* stream) {
  if (ldx < K) return PM_EINVAL;
  return linear_impl(x, ldx, 0, 0, w, ldw, bias, resid, ldr, resid_dtype, 0, y, ldy, y_dtype, M, N, K, act, PmLnFold{}, stream);
}

extern "C" int pm_linear_bf16_ex(const void* x, int64_t ldx, int64_t x_rows_per_batch, int64_t x_batch_stride,
                                 const void* w, int64_t ldw, const float* bias, const void* resid, int64_t ldr,
                                 int resid_dtype, int64_t resid_period, void* y, int64_t ldy, int y_dtype, int64_t M,
                                 int64_t N, int64_t K, int act, void* stream) {
  return linear_impl(x, ldx, x_rows_per_batch, x_batch_stride, w, ldw, bias, resid, ldr, resid_dtype, resid_period, y, ldy,
                     y_dtype, M, N, K, act, PmLnFold{}, stream);
}

// ---- LayerNorm fold -------------------------------------------------------------------------------------------
// A pre-norm block computes y = LN(x) W^T + b.  With W' = bf16(gamma (.) W), s[n] = sum_k W'[n][k] and
// c[n] = b[n] + sum_k beta[k] W[n][k] this is rstd[m] * (x W'^T - mean[m] * s[n]) + c[n]: the GEMM runs on the RAW
// residual stream and the normalisation is two fma per output in its epilogue.  mean / rstd come from per-row partial
// sums that the PRODUCER of x (the previous out_proj / linear2 GEMM) emits from its own epilogue (ln_row_out), reduced
// by pm_ln_stats_finalize.  The LayerNorm kernel, its write of LN(x) and the re-read of it disappear.
extern "C" int pm_linear_bf16_ln(const void* x, int64_t ldx, int64_t x_rows_per_batch, int64_t x_batch_stride,
                                 const void* w, int64_t ldw, const float* bias, const void* resid, int64_t ldr,
                                 int resid_dtype, int64_t resid_period, void* y, int64_t ldy, int y_dtype, int64_t M,
                                 int64_t N, int64_t K, int act, const float* ln_stats, const float* ln_s,
                                 float* ln_row_out, void* stream) {
  PmLnFold ln{ln_stats, ln_s, ln_row_out};
  return linear_impl(x, ldx, x_rows_per_batch, x_batch_stride, w, ldw, bias, resid, ldr, resid_dtype, resid_period, y, ldy,
                     y_dtype, M, N, K, act, ln, stream);
}

/* pm_linear_bf16_ln with a caller-owned workspace (pm_linear_ws_bytes() bytes, 16-byte aligned, its first 4096 bytes ZERO
 * before the first call and never written by the caller afterwards): with it the dispatcher may deal a GEMM's K steps out
 * as one stream over the persistent workgroups (csrc/linear_bf16_sk.hip) when whole tiles would leave the last round
 * badly filled.  One workspace serves one stream at a time (calls that may overlap need their own).  ws == NULL: as
 * pm_linear_bf16_ln. */
extern "C" int pm_linear_bf16_ws(const void* x, int64_t ldx, int64_t x_rows_per_batch, int64_t x_batch_stride,
                                 const void* w, int64_t ldw, const float* bias, const void* resid, int64_t ldr,
                                 int resid_dtype, int64_t resid_period, void* y, int64_t ldy, int y_dtype, int64_t M,
                                 int64_t N, int64_t K, int act, const float* ln_stats, const float* ln_s,
                                 float* ln_row_out, void* ws, int64_t ws_bytes, void* stream) {
  PmLnFold ln{ln_stats, ln_s, ln_row_out};
  return linear_impl(x, ldx, x_rows_per_batch, x_batch_stride, w, ldw, bias, resid, ldr, resid_dtype, resid_period, y, ldy,
                     y_dtype, M, N, K, act, ln, stream, ws, ws_bytes);
}

extern "C" int64_t pm_linear_ws_bytes(void) { return pm_linear_sk_ws_bytes(); }

namespace {
__global__ __launch_bounds__(256) void ln_stats_finalize_kernel(const float* __restrict__ part, float* __restrict__ stats,
                                                               int64_t M, int np, float inv_n, float eps) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  float s1 = 0.f, s2 = 0.f;
  if ((np & 1) == 0) {
    // sixteen pairs per trip requested together (branch-free: clamped index, repeats dropped), added in pair order - the
    // one-pair-per-trip loop below compiles to load, vmcnt(0), add, branch: np dependent round trips, the whole 4.3 us
    const float* pr = part + m * np * 2;
    // two 16-byte requests per trip: within 16 registers this kernel fits on a SIMD beside the other stream's persistent GEMM
    // (2 waves x 240 of 512 registers) instead of waiting for that GEMM's workgroups to leave (PM_FINALIZE_WIDE=1: 8 per trip)
    for (int p0 = 0; p0 < np; p0 += 4) {
      f32x4 v[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int p = p0 + 2 * j;
        v[j] = *(const f32x4*)(pr + (p < np ? p : np - 2) * 2);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bool ok = p0 + 2 * j < np;
        s1 += ok ? v[j][0] : 0.f;
        s2 += ok ? v[j][1] : 0.f;
        s1 += ok ? v[j][2] : 0.f;
        s2 += ok ? v[j][3] : 0.f;
      }
    }
  } else {
    for (int p = 0; p < np; ++p) {  // fixed order: deterministic
      s1 += part[(m * np + p) * 2];
      s2 += part[(m * np + p) * 2 + 1];
    }
  }
  const float mean = s1 * inv_n;
  const float var = fmaxf(s2 * inv_n - mean * mean, 0.f);
  stats[2 * m] = mean;
  stats[2 * m + 1] = rsqrtf(var + eps);
}
}  // namespace

extern "C" int pm_ln_stats_finalize(const float* row_partials, float* stats, int64_t M, int64_t N, float eps, void* stream) {
  if (!row_partials || !stats || M < 0 || N <= 0) return PM_EINVAL;
  if (M == 0) return PM_OK;
  if (N % 64) return PM_EUNSUPPORTED;
  hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, row_partials,
                     stats, M, (int)(N / 64), 1.0f / (float)N, eps);
  PM_CHECK_LAUNCH();
  return PM_OK;
}

/* 1 if pm_linear_bf16_ln can serve this shape (as a consumer of ln_stats when produce == 0, as a producer of
 * ln_row_out when produce == 1), else 0: callers fall back to pm_layernorm + pm_linear_bf16. */
extern "C" int pm_linear_ln_supported(int64_t M, int64_t N, int64_t K, int act, int produce) {
  // shapes on which the fold is served AND worth it: a persistent kernel must be at least three quarters of a round full
  // (below that the forced 256 x 128 tiling costs more than the LayerNorm launch it saves)
  if (K % 64 || N % 8 || M < 4096) return 0;
  if (((M + LBM - 1) / LBM) * ((N + LBN - 1) / LBN) < 192) return 0;
  if (produce) return N % 64 == 0 ? 1 : 0;
  return 1;
}
