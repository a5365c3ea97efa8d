// linear_bf16_wide.hip - the wide-N variant of pm_linear_bf16: persistent 256 (tokens) x 256 (features) x 32 tiles.
// (same operation and reference sites as linear_bf16.hip: pytorch_models/transformer.py:28-31,47-53,59-66)
//
// Why a second tile shape.  Every K step a workgroup pulls (BM + BN) * BK * 2 bytes from L2 through its CU's vector
// memory path (~64 B/clk) and issues 2 * BM * BN * BK flop on the matrix pipe (4096 flop/clk/CU): 256 x 128 tiles need
// 768 memory-path clocks per 1024 matrix clocks, 256 x 256 tiles 512 per 1024.  With the 256 x 128 kernel the matrix
// pipe was measured 34 % (K = 768) to 52 % (K = 8192) busy with zero LDS bank conflicts; the load path was the peer
// limiter.  256 x 256 halves... cuts that pressure by a third, needs 12 instead of 16 fragment reads per 32 MFMAs, and 4
// instead of 6 LDS-DMA pieces per wave and step.  It is used where there are enough 256 x 256 tiles to balance 256
// persistent workgroups (QKV, fc1: N = 2304, 3072); N = 768 layers stay on the 256 x 128 kernel (591 tiles would
// leave a 3-vs-2 tail).
//
// Structure (see linear_bf16.hip for the shared ideas): 8 waves as 4 (tokens) x 2 (features), each wave 64 x 128 =
// 4 x 8 MFMA 16x16x32 tiles (128 accumulator VGPRs); K step = 32 (64-byte rows), 4-stage LDS ring of 32 KiB stages,
// three steps in flight behind counted vmcnt + raw s_barrier, one stream across all of a workgroup's tiles; 64-byte
// rows are XOR-swizzled on the LDS-DMA source address (chunk c of row r at c ^ 3*((r >> 3) & 1): conflict-free for the
// 16-row x 4-chunk fragment read); epilogue = bias / GELU in the accumulator layout, fp32 transposition of 16 x 64
// blocks through the ring buffer the last step freed, residual + single rounding + 16-byte row-wise stores.
#include <cstdlib>

#include "common.h"

namespace {

#ifndef PM_WIDE_BK
#define PM_WIDE_BK 64  // 32 = the earlier form: 64-byte rows, four 32 KiB stages, a barrier per 32 of K
#endif
constexpr int WBM = 256, WBN = 256, WBK = PM_WIDE_BK;
constexpr int WSTAGE = (WBM + WBN) * WBK * 2;  // 32 KiB (64 KiB at WBK = 64)
constexpr int WRING = WBK == 32 ? 4 : 2;
constexpr int WROWB = WBK * 2;          // bytes per LDS row
constexpr int WPIECE_ROWS = 1024 / WROWB;  // rows per 1 KiB LDS-DMA piece: 16 or 8
constexpr int WNPIECE = 32 / WPIECE_ROWS;  // pieces per operand and wave (32 rows each): 2 or 4
#ifndef PM_WGROUP_M
#define PM_WGROUP_M 4
#endif
constexpr int WGROUP_M = PM_WGROUP_M;
// Priority: waves 4-7 (the younger wave of each SIMD pair, which loses every age-based arbitration and made waves 0-3
// wait ~600-1200 cycles at each barrier) run at s_setprio 1 for the whole kernel; no per-step flips
// (MI355X_MICROARCH.md, two waves per SIMD, item 4).  out_proj -4.5 %, fc2 -3 %, fc1 / QKV -1 %.  0 = the old flips.
#ifndef PM_STATIC_PRIO
#define PM_STATIC_PRIO 1
#endif

__device__ __forceinline__ int swz64(int row, int chunk) { return chunk ^ (((row >> 3) & 1) * 3); }

__device__ __forceinline__ bf16x8 wread(const char* tile, int row, int chunk) {
  if constexpr (WBK == 32) return *(const bf16x8*)(tile + row * 64 + swz64(row, chunk) * 16);
  else return *(const bf16x8*)(tile + row * 128 + swz_pos(row, chunk) * 16);
}

__device__ __forceinline__ void wtile_coords(int t, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int per = WGROUP_M * tiles_n;
  const int sr = t / per, r = t - sr * per;
  const int left = tiles_m - sr * WGROUP_M;
  const int gm = left < WGROUP_M ? left : WGROUP_M;
  tn = r / gm;
  tm = sr * WGROUP_M + (r - tn * gm);
}

template <int ACT>
__global__ __launch_bounds__(512, 2) void linear_bf16_wide_kernel(
    const bf16* __restrict__ X, int64_t ldx, const bf16* __restrict__ W, int64_t ldw, const float* __restrict__ bias,
    const bf16* resid, int64_t ldr, int resid_period, bf16* Y, int64_t ldy, int M, int N, int K, int tiles_m, int tiles_n,
    int x_rows_per_batch, int64_t x_batch_stride, PmLnFold ln) {
  __shared__ __attribute__((aligned(16))) char smem[WRING * WSTAGE + 8 * 4096];  // ring + 4 KiB of epilogue staging per wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntiles = tiles_m * tiles_n;

  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, nloc = gridDim.x >> 3;
  const int cq = ntiles >> 3, cr = ntiles & 7;
  const int tbase = xcd * cq + (xcd < cr ? xcd : cr), tcount = cq + (xcd < cr ? 1 : 0);
  const int my_tiles = local < tcount ? (tcount - local + nloc - 1) / nloc : 0;
  const int nk = K / WBK;
  const int P = my_tiles * nk;

  // staging side of the stream (a macro over plain locals: see linear_bf16.hip)
  uint32_t xoff[WNPIECE], woff[WNPIECE];  // element offsets (the dispatcher keeps this path under 2^32 elements per operand)
  int pp = 0, pp_kt = 0, pp_tile = 0, pp_buf = 0;
#define PM_WSTAGE_NEXT()                                                                                             \
  if (pp < P) {                                                                                                      \
    if (pp_kt == 0) {                                                                                                \
      int tm_, tn_;                                                                                                  \
      wtile_coords(tbase + local + pp_tile * nloc, tiles_m, tiles_n, tm_, tn_);                                      \
      _Pragma("unroll") for (int i = 0; i < WNPIECE; ++i) {                                                          \
        const int rt = wave * 32 + i * WPIECE_ROWS + (WBK == 32 ? (lane >> 2) : (lane >> 3));                        \
        const int chunk = WBK == 32 ? swz64(rt, lane & 3) : swz_pos(rt, lane & 7);                                   \
        int gm = tm_ * WBM + rt;                                                                                     \
        gm = gm < M ? gm : M - 1;                                                                                    \
        if (x_rows_per_batch > 0) {                                                                                  \
          const int bb = gm / x_rows_per_batch;                                                                      \
          xoff[i] = (uint32_t)((int64_t)bb * x_batch_stride + (int64_t)(gm - bb * x_rows_per_batch) * ldx + chunk * 8); \
        } else {                                                                                                     \
          xoff[i] = (uint32_t)((int64_t)gm * ldx + chunk * 8);                                                        \
        }                                                                                                            \
        int gn = tn_ * WBN + rt;                                                                                     \
        gn = gn < N ? gn : N - 1;                                                                                    \
        woff[i] = (uint32_t)((int64_t)gn * ldw + chunk * 8);                                                          \
      }                                                                                                              \
    }                                                                                                                \
    char* xs_ = smem + pp_buf * WSTAGE;                                                                              \
    _Pragma("unroll") for (int i = 0; i < WNPIECE; ++i)                                                              \
        glds16_aux<PM_GLDS_X_AUX>(X + xoff[i] + pp_kt * WBK, xs_ + (wave * 32 + i * WPIECE_ROWS) * WROWB);                               \
    _Pragma("unroll") for (int i = 0; i < WNPIECE; ++i)                                                              \
        glds16_aux<PM_GLDS_W_AUX>(W + woff[i] + pp_kt * WBK, xs_ + WBM * WROWB + (wave * 32 + i * WPIECE_ROWS) * WROWB);                 \
    ++pp;                                                                                                            \
    pp_buf = (pp_buf + 1) & (WRING - 1);                                                                             \
    if (++pp_kt == nk) { pp_kt = 0; ++pp_tile; }                                                                     \
  }

  f32x4 acc[8][4];  // [feature subtile j][token subtile i]
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

#if PM_STATIC_PRIO
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);  // the younger wave of each SIMD pair loses every arbitration otherwise
#endif
  PM_WSTAGE_NEXT();
  if (WRING == 4) { PM_WSTAGE_NEXT(); PM_WSTAGE_NEXT(); }
  const int fr = lane & 15, fq = lane >> 4;
  int buf = 0, kt = 0, ti = 0;
  f32x2 lnst[4] = {{0.f, 1.f}, {0.f, 1.f}, {0.f, 1.f}, {0.f, 1.f}};
  for (int pc = 0; pc < P; ++pc) {
    // step pc landed (this wave's part); up to two younger steps (4 LDS-DMA pieces each) stay in flight
    const int younger = pp - pc - 1;
    if (WRING == 4) {
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {  // two stages: only step pc itself was in flight
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // every wave's part landed; every wave is past step pc-1: its buffer is free
    PM_WSTAGE_NEXT();
    if (kt == nk - 1 && ln.stats) {
      // LayerNorm fold, last K step of the tile: request (mean, rstd) of this lane's four token rows
      // (m0 + 16 i + fr) now, so the latency hides under this step's MFMAs instead of stalling the epilogue
      int tm_r, tn_r;
      wtile_coords(tbase + local + ti * nloc, tiles_m, tiles_n, tm_r, tn_r);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int mr = tm_r * WBM + wm * 64 + i * 16 + fr;
        mr = mr < M ? mr : M - 1;
        lnst[i] = *(const f32x2*)(ln.stats + 2 * (int64_t)mr);
      }
    }
    const char* xcur = smem + buf * WSTAGE;
    const char* wcur = xcur + WBM * WROWB;
#pragma unroll
    for (int ss = 0; ss < WBK / 32; ++ss) {
      bf16x8 a[8], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = wread(xcur, wm * 64 + i * 16 + fr, ss * 4 + fq);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = wread(wcur, wn * 128 + j * 16 + fr, ss * 4 + fq);
      __builtin_amdgcn_sched_barrier(0);
#if !PM_STATIC_PRIO
      __builtin_amdgcn_s_setprio(1);
#endif
      if (kt == 0 && ss == 0) {  // a tile's first step starts from the constant 0: nobody has to clear 128 accumulator registers
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], b[i], zero, 0, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], b[i], acc[j][i], 0, 0, 0);
      }
#if !PM_STATIC_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
    buf = (buf + 1) & (WRING - 1);
    if (++kt < nk) continue;

    // ---------------- tile finished: epilogue (the next tile's first three K steps are already in flight)
    kt = 0;
    int tm, tn;
    wtile_coords(tbase + local + ti * nloc, tiles_m, tiles_n, tm, tn);
    ++ti;
    const int m0 = tm * WBM + wm * 64, n0 = tn * WBN + wn * 128;
    // every wave stages through its OWN 4 KiB behind the ring: no barrier in front of the epilogue, so a wave that
    // finished its MFMAs early (waves 0-3 win the arbitration) converts and stores while its SIMD partner still computes
    char* stg = smem + WRING * WSTAGE + wave * 4096;
    const int srow = lane >> 3, sch = lane & 7;
    // row-wise side: this lane stores rows m0 + srow + 8 k (k = 0..7), 8 features at n0 + 64 hf + 8 sch.  One 64-bit
    // multiply per tile; every other address is that base plus a wave-uniform offset (the per-store form cost ~10
    // integer VALU instructions a store)
    bf16* const ybase = Y + (int64_t)(m0 + srow) * ldy + n0 + sch * 8;
    const bf16* const rbase = resid && !resid_period ? resid + (int64_t)(m0 + srow) * ldr + n0 + sch * 8 : nullptr;
    const int mleft = M - m0 - srow;  // row k of this lane exists iff 8 k < mleft
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {  // 64-feature halves of the wave's 128 features
      f32x4 bvec[4], svec[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int n = n0 + hf * 64 + jj * 16 + fq * 4;
        bvec[jj] = (bias && n < N) ? *(const f32x4*)(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        svec[jj] = (ln.stats && n < N) ? *(const f32x4*)(ln.s + n) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float mu = lnst[i][0], rstd = lnst[i][1];  // (0, 1) without the LayerNorm fold
        bf16x8 rv[2];
        if (rbase) {  // coalesced 16-byte loads in the store layout; used after the staging round trip below
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const bool ok = i * 16 + p * 8 < mleft && n0 + hf * 64 + sch * 8 < N;
            rv[p] = ok ? *(const bf16x8*)(rbase + (int64_t)(i * 16 + p * 8) * ldr + hf * 64) : bf16x8{};
          }
        } else if (resid) {  // periodic residual (row m reads row m % resid_period: a position table)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            int mm = m0 + i * 16 + srow + p * 8;
            mm = mm < M ? mm : M - 1;
            int nn = n0 + hf * 64 + sch * 8;
            nn = nn < N ? nn : N - 8;
            rv[p] = *(const bf16x8*)(resid + (int64_t)(mm % resid_period) * ldr + nn);
          }
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaf(rstd, acc[hf * 4 + jj][i][r] - mu * svec[jj][r], bvec[jj][r]);
          v = apply_act4<ACT>(v);
          *(f32x4*)(stg + fr * 256 + (((4 * jj + fq) ^ fr) * 16)) = v;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int row = srow + p * 8;
          const f32x4 lo = *(const f32x4*)(stg + row * 256 + (((2 * sch) ^ row) * 16));
          const f32x4 hi = *(const f32x4*)(stg + row * 256 + (((2 * sch + 1) ^ row) * 16));
          bf16x8 o;
          if (resid) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { o[r] = (bf16)(lo[r] + (float)rv[p][r]); o[4 + r] = (bf16)(hi[r] + (float)rv[p][4 + r]); }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) { o[r] = (bf16)lo[r]; o[4 + r] = (bf16)hi[r]; }
          }
          if (i * 16 + p * 8 < mleft && n0 + hf * 64 + sch * 8 < N)  // N % 8 == 0 on this path
            store_y((bf16x8*)(ybase + (int64_t)(i * 16 + p * 8) * ldy + hf * 64), o);
        }
      }
    }
  }
#undef PM_WSTAGE_NEXT
}

}  // namespace

// Internal entry (called from linear_bf16.hip's dispatcher; arguments already validated there).
int pm_linear_bf16_wide_launch(const void* x, int64_t ldx, int64_t x_rows_per_batch, int64_t x_batch_stride, const void* w,
                               int64_t ldw, const float* bias, const void* resid, int64_t ldr, int64_t resid_period, void* y,
                               int64_t ldy, int64_t M, int64_t N, int64_t K, int act, PmLnFold ln, hipStream_t st) {
  const int tiles_m = (int)((M + WBM - 1) / WBM), tiles_n = (int)((N + WBN - 1) / WBN);
#define PM_WGO(A)                                                                                                      \
  hipLaunchKernelGGL((linear_bf16_wide_kernel<A>), dim3(256), dim3(512), 0, st, (const bf16*)x, ldx, (const bf16*)w, ldw, \
                     bias, (const bf16*)resid, ldr, (int)resid_period, (bf16*)y, ldy, (int)M, (int)N, (int)K, tiles_m,  \
                     tiles_n, (int)x_rows_per_batch, x_batch_stride, ln)
  if (act == PM_ACT_NONE) PM_WGO(PM_ACT_NONE);
  else if (act == PM_ACT_GELU) PM_WGO(PM_ACT_GELU);
  else return PM_EUNSUPPORTED;
#undef PM_WGO
  return PM_OK;
}

bool pm_linear_bf16_wide_applies(int64_t M, int64_t N, int64_t K, int act) {
  // shape eligibility only; whether the 256 x 256 tiling is the FASTEST of the three kernels is decided by the cost
  // model of linear_impl (pm_linear_pick_kernel)
  if (act != PM_ACT_NONE && act != PM_ACT_GELU) return false;
  if (K % WBK || N % 8) return false;
  return M >= 4096;
}
