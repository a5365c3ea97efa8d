// linear_bf16_8ph.hip - EXPERIMENT (not on the product path; pm_gemm8ph_bench is not in the header): the 256 x 256 x 64 bf16
// GEMM K loop as a phase-interleaved schedule with the two wave groups of a workgroup staggered by one barrier
// (cdna_hip_programming.md 5, "The 256^2 8-phase template"), to price that structure against linear_bf16_wide.hip's
// one-barrier-per-K-step loop before anything is built on it.  y[M, N] = x[M, K] w[N, K]^T, bf16 in, bf16 out.
//
// 8 waves as 2 (token halves, wr) x 4 (feature quarters, wc): a wave owns 128 tokens x 64 features = 8 x 4 MFMA
// 16x16x32 tiles (128 accumulator VGPRs).  LDS = two K tiles of four 16 KiB half-tiles [X0 | X1 | W0 | W1] (128 rows of
// 128 bytes each, XOR-swizzled chunks as everywhere in this library).  A K tile is four phases, one 64-token x 32-feature
// quadrant x K = 64 each = 16 MFMAs; a phase is {fragment reads of the quadrant + 2 LDS-DMA pieces of a half-tile two K
// tiles ahead} barrier {16 MFMAs} barrier.  Group wr = 1 runs one barrier behind group wr = 0: while one wave of a SIMD
// issues MFMAs its partner reads fragments and issues loads.
//   reads:  phase 1: x rows 0-63 (8 reads) + w cols 0-31 (4); phase 2: w cols 32-63 (4); phase 3: x rows 64-127 (8);
//           phase 4: none (w cols 0-31 are still in registers).  Quadrants: (x0,w0) (x0,w1) (x1,w1) (x1,w0).
//   stages: tile t phase 4: W0(t+2); tile t+1 phase 1: W1(t+2); phase 2: X0(t+2); phase 3: X1(t+2) - each at least two
//           barriers after the last read of the half-tile it overwrites, counting the lagging group.
//   waits:  with segments numbered globally (a barrier between consecutive ones; G0's load segment of (t, p) is
//           8t + 2(p-1), its MFMA segment one later; G1 one later still): before the barrier ending segment 8t+15 every
//           wave waits until all but its two youngest half-tile pieces have landed (W0, W1, X0 of tile t+2 are needed by
//           G0's reads in 8t+16), before the barrier ending 8t+16 until X1(t+2) has landed (G1 reads it in 8t+17).
#include "../common.h"
#include "../../../include/pm_mi355x_experiments.h"

namespace {

constexpr int PBM = 256, PBN = 256, PBK = 64;
constexpr int PHALF = 128 * PBK * 2;   // 16 KiB
constexpr int PTILE = 4 * PHALF;       // 64 KiB: X0 X1 W0 W1

__device__ __forceinline__ void wait_vm(int n) {  // n in {0, 2, 4}: wave-uniform
  if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__device__ __forceinline__ bf16x8 pread(const char* half_tile, int row, int chunk) {
  return *(const bf16x8*)(half_tile + row * 128 + swz_pos(row, chunk) * 16);
}

__global__ __launch_bounds__(512, 2) void gemm8ph_kernel(const bf16* __restrict__ X, int64_t ldx, const bf16* __restrict__ W,
                                                         int64_t ldw, bf16* __restrict__ Y, int64_t ldy, int M, int N, int K,
                                                         int tiles_n) {
  __shared__ __attribute__((aligned(16))) char smem[2 * PTILE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  // 4-panel super-rows, as the product kernels order their tiles: consecutive tile ids share token panels and walk the
  // feature tiles slowly, so the 32 workgroups of an XCD keep a small set of panels in its L2
  const int tiles_m = (M + PBM - 1) / PBM;
  int tm, tn;
  {
    const int per = 4 * tiles_n, sr = tile / per, r = tile - sr * per;
    const int left = tiles_m - sr * 4, gm = left < 4 ? left : 4;
    tn = r / gm;
    tm = sr * 4 + (r - tn * gm);
  }
  const int nk = K / PBK;
  const int fr = lane & 15, fq = lane >> 4;

  // staging: a half-tile = 128 rows x 8 chunks = 1024 chunks of 16 B = 2 per thread; piece i of wave w covers rows
  // w * 16 + i * 8 + (lane >> 3), chunk position lane & 7 (source chunk un-swizzled: LDS-DMA writes lane-linear)
  const bf16* xsrc[2][2];  // [half][piece]
  const bf16* wsrc[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rt = wave * 16 + i * 8 + (lane >> 3);
      const int chunk = swz_pos(rt, lane & 7);
      int gm = tm * PBM + h * 128 + rt;
      gm = gm < M ? gm : M - 1;
      int gn = tn * PBN + h * 128 + rt;
      gn = gn < N ? gn : N - 1;
      xsrc[h][i] = X + (int64_t)gm * ldx + chunk * 8;
      wsrc[h][i] = W + (int64_t)gn * ldw + chunk * 8;
    }
  auto stage = [&](int kt, int which) {  // which: 0 = X0, 1 = X1, 2 = W0, 3 = W1
    char* dst = smem + (kt & 1) * PTILE + which * PHALF + (wave * 16) * 128;
    const int h = which & 1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      glds16((which < 2 ? xsrc[h][i] : wsrc[h][i]) + kt * PBK, dst + i * 8 * 128);
  };

  f32x4 acc[8][4];  // [token subtile][feature subtile]
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // prologue: K tiles 0 and 1 complete
  for (int kt = 0; kt < 2 && kt < nk; ++kt)
#pragma unroll
    for (int which = 0; which < 4; ++which) stage(kt, which);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // the lagging group: one barrier behind from here on

  bf16x8 xf[2][4], wf[2][2][2];  // x: [k half][token subtile of the current 64-row half]; w: [32-col half][k half][subtile]
  for (int kt = 0; kt < nk; ++kt) {
    const char* base = smem + (kt & 1) * PTILE;
    const char* xt = base + wr * PHALF;
    const char* wt = base + (2 + (wc >> 1)) * PHALF;
    const int wrow0 = (wc & 1) * 64;
    const bool more = kt + 2 < nk;
#define PM_XREAD(MI)                                                                                     \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < 4; ++i)          \
      xf[ks][i] = pread(xt, (MI) * 64 + i * 16 + fr, ks * 4 + fq);
#define PM_WREAD(NI)                                                                                     \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int j = 0; j < 2; ++j)          \
      wf[NI][ks][j] = pread(wt, wrow0 + (NI) * 32 + j * 16 + fr, ks * 4 + fq);
#define PM_MFMA(MI, NI)                                                                                  \
  __builtin_amdgcn_s_setprio(1);                                                                         \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < 4; ++i)          \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[(MI) * 4 + i][(NI) * 2 + j] =                     \
          __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[NI][ks][j], xf[ks][i], acc[(MI) * 4 + i][(NI) * 2 + j], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);
#define PM_SYNC_LOAD()                                     \
  __builtin_amdgcn_s_barrier();                            \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
  __builtin_amdgcn_sched_barrier(0);
    // ---- phase 1
    PM_WREAD(0)
    __builtin_amdgcn_sched_barrier(0);
    PM_XREAD(0)
    const bool mid = kt > 0 && kt + 1 < nk;  // tile kt + 1's W1, X0, X1 are staged during this tile (tiles 0, 1: prologue)
    if (mid) stage(kt + 1, 3);
    // the wait that lets the lagging group read X1 of THIS tile (header: "before the barrier ending 8t+16"); for the
    // leading group it sits here, at the end of its load segment of phase 1: only W0 / W1 of the next tile may be in flight
    if (wr == 0) wait_vm(2 * ((kt + 1 < nk && kt > 0 ? 1 : 0) + (mid ? 1 : 0)));
    PM_SYNC_LOAD()
    PM_MFMA(0, 0)
    __builtin_amdgcn_s_barrier();
    // ---- phase 2
    PM_WREAD(1)
    if (mid) stage(kt + 1, 0);
    PM_SYNC_LOAD()
    PM_MFMA(0, 1)
    __builtin_amdgcn_s_barrier();
    // ---- phase 3
    PM_XREAD(1)
    if (mid) stage(kt + 1, 1);
    PM_SYNC_LOAD()
    PM_MFMA(1, 1)
    __builtin_amdgcn_s_barrier();
    // ---- phase 4
    if (more) stage(kt + 2, 2);
    // everything of tile kt + 1 but its X1 (and what was just issued) has landed before the barrier that ends this segment
    // for the lagging group / the MFMA segment for the leading group
    const int inflight = 2 * ((mid ? 1 : 0) + (more ? 1 : 0));  // X1(kt + 1) and W0(kt + 2), where they were issued
    if (wr == 1) wait_vm(inflight);
    __builtin_amdgcn_s_barrier();
    PM_MFMA(1, 0)
    if (wr == 0) wait_vm(inflight);
    else wait_vm(more ? 2 : 0);  // the lagging group: X1(kt + 1) landed, for its own reads right behind this barrier
    __builtin_amdgcn_s_barrier();
  }
#undef PM_XREAD
#undef PM_WREAD
#undef PM_MFMA
#undef PM_SYNC_LOAD
  if (wr == 0) __builtin_amdgcn_s_barrier();  // the leading group meets the lagging one's last barrier

  // D[row = feature 4 fq + r][col = token fr]
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = tm * PBM + wr * 128 + i * 16 + fr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = tn * PBN + wc * 64 + j * 16 + fq * 4;
      if (m < M && n + 3 < N) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16)acc[i][j][r];
        *(bf16x4*)(Y + (int64_t)m * ldy + n) = o;
      }
    }
  }
}

}  // namespace

extern "C" int pm_gemm8ph_bench(const void* x, int64_t ldx, const void* w, int64_t ldw, void* y, int64_t ldy, int64_t M, int64_t N,
                                int64_t K, void* stream) {
  if (!x || !w || !y || M <= 0 || N <= 0 || K < 128 || K % PBK || N % 4) return PM_EINVAL;
  const int tiles_m = (int)((M + PBM - 1) / PBM), tiles_n = (int)((N + PBN - 1) / PBN);
  hipLaunchKernelGGL(gemm8ph_kernel, dim3((unsigned)(tiles_m * tiles_n)), dim3(512), 0, (hipStream_t)stream, (const bf16*)x, ldx,
                     (const bf16*)w, ldw, (bf16*)y, ldy, (int)M, (int)N, (int)K, tiles_n);
  PM_CHECK_LAUNCH();
  return PM_OK;
}
