// grouped_conv.hip - grouped Conv1d over time with few channels per group and a long kernel: the positional conv of
// Wav2Vec2 / HuBERT (d = 768: 16 groups x 48 channels, k = 128; reference pytorch_models/audio/wav2vec2.py:70-74),
// data2vec-audio (k = 19, data2vec_audio.py:25) and SEW (k = 31, stride 2, sew.py:24), fused with bias, GELU and the
// residual add.
//
// Per group the conv is a GEMM  y[(clip, t), n] = sum_{tap, c} x[clip, t*stride + tap, c] * w[n, tap, c]  with
// N = channels per group (48) and K = k * 48 = 6144.  Run as G generic GEMMs (pm_linear_bf16_ex over the regrouped
// buffer of pm_group_windows) it is bound by the DATA side, not the matrix cores: the A-operand rows of neighbouring
// steps overlap in all but one tap, yet every K step re-fetches its 128 x 64 window from L2 - 1.5 MB of loads per tile
// for 24 KB of distinct data - and N = 48 fills 3/8 of a 128-wide tile (measured: 105 TFLOP/s, 16 launches, 16 % of a
// wav2vec2-base forward).  Here the Toeplitz structure is used instead:
//   * a workgroup owns 64*MI consecutive steps of one (clip, group) and copies the DISTINCT rows it touches -
//     (64*MI - 1) * stride + k rows of cgp channels - into LDS once (43 KB at MI = 4, d = 768);
//   * the X fragment of (step t, tap j) is then just LDS row t*stride + j: every K step reads its fragments straight
//     from that resident span, no X traffic at all after the prologue;
//   * only the group's weight streams (K-major (n, tap, c) bf16, 590 KB per group, shared by all tiles of the group
//     through the XCD's L2: the tile order keeps a group on one XCD), 64 K-elements per stage through a 4-deep
//     global_load_lds ring with counted vmcnt and one barrier per stage;
//   * LDS image of the span: ds_read_b128 serves four fixed 16-lane groups that mix rows of two neighbouring K chunks
//     (MI355X_MICROARCH.md, LDS), so "16 consecutive rows on distinct banks" is not enough - the first layout (rows padded
//     by 16 B) measured 37 % of its LDS cycles as bank conflicts.  Enumerating the real groups over all row offsets and K
//     steps: at stride 1, 128-byte rows with chunk c at position c ^ (row & 7) are conflict-free; at stride 2 the padded
//     rows are.  The kernel takes the layout as a template parameter.
// MFMA 16x16x32 with the weight as the A operand, so a lane owns 4 consecutive output channels of one step and the
// epilogue (bias, GELU, + residual, bf16) stores 8 bytes per lane straight into the (clip, step, d) activation.
// Algorithmic work: 2 * B * T_out * d * k * cg flop (151 GFLOP for 32 x 10 s at d = 768); MFMA-bound by intent,
// LDS-read-bound in this first form (MI + NJ fragment reads per MI * NJ MFMAs).
#include "common.h"

namespace {

constexpr int GC_WSTAGES = 4;
constexpr int GC_THREADS = 256;

template <int CPT, int NJ, int MI, int ACT, bool SWZ>
__global__ __launch_bounds__(GC_THREADS) void grouped_conv_kernel(const bf16* __restrict__ xg, const bf16* __restrict__ w,
                                                                   const float* __restrict__ bias, const bf16* __restrict__ resid,
                                                                   int64_t ldr, bf16* __restrict__ y, int64_t ldy, int G, int Tp,
                                                                   int To, int cg, int stride, int Kp, int tiles_t, int span_rows) {
  extern __shared__ __attribute__((aligned(16))) char gc_smem[];
  constexpr int RS = SWZ ? 128 : CPT * 16 + 16;  // LDS row: 8 XOR-swizzled chunk slots, or CPT chunks + 16 B of padding
  constexpr int WT = NJ * 16 * 128;       // one weight stage: NJ*16 output channels x 64 K-elements
  constexpr int BM = 64 * MI;
  char* wring = gc_smem;
  char* xl = gc_smem + GC_WSTAGES * WT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // tile order: group slowest, so that the contiguous per-XCD ranges of xcd_remap keep a group's weight in one L2
  int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tt = wg % tiles_t;
  wg /= tiles_t;
  const int B = gridDim.x / (tiles_t * G);
  const int b = wg % B, g = wg / B;
  const int t0 = tt * BM;

  // ---- weight ring: wave v < NJ streams output channels [16 v, 16 v + 16) of every stage (2 x 1 KiB per stage)
  const bf16* wgp = w + (int64_t)g * cg * Kp;
  int64_t woff[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int rt = wave * 16 + h * 8 + (lane >> 3);
    const int n = rt < cg ? rt : cg - 1;  // rows past the group's channels: finite duplicates, masked at the store
    woff[h] = (int64_t)n * Kp + swz_pos(rt, lane & 7) * 8;
  }
  const int nk = Kp / 64;
  auto stage_w = [&](int kt) {
    if (wave < NJ) {
      char* dst = wring + (kt % GC_WSTAGES) * WT + wave * 16 * 128;
#pragma unroll
      for (int h = 0; h < 2; ++h) glds16(wgp + woff[h] + kt * 64, dst + h * 8 * 128);
    }
  };
#pragma unroll
  for (int p = 0; p < GC_WSTAGES - 1; ++p)
    if (p < nk) stage_w(p);

  // ---- the distinct input rows of this tile, once: slab row t0*stride + r -> LDS row r (zeros past the slab)
  {
    const bf16* slab = xg + ((int64_t)b * G + g) * Tp * (CPT * 8);
    const int r0 = t0 * stride;
    const int nchunks = span_rows * CPT;
    for (int i = tid; i < nchunks; i += GC_THREADS) {
      const int r = i / CPT, c = i - r * CPT;
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
      if (r0 + r < Tp) v = *(const bf16x8*)(slab + ((int64_t)(r0 + r) * CPT + c) * 8);
      *(bf16x8*)(xl + r * RS + (SWZ ? (c ^ (r & 7)) : c) * 16) = v;
    }
  }

  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // span written; the first barrier of the K loop publishes it

  f32x4 acc[NJ][MI];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  const int row0 = (wave * MI * 16 + fr) * stride;              // this lane's span row within subtile 0, before the tap
  const char* xrow = xl + row0 * RS;
  const int sub = 16 * stride * RS;                             // next 16-step subtile
  int tap = fq / CPT, cc = fq % CPT;                            // this lane's K chunk (tap, 8-channel chunk), advanced 4 chunks per MFMA

  for (int kt = 0; kt < nk; ++kt) {
    // stages kt .. kt+2 are in flight (2 loads each on the streaming waves); stage kt must have landed
    if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // stage kt visible to all waves; everyone is done with stage kt-1, whose slot is refilled next
    if (kt + GC_WSTAGES - 1 < nk) stage_w(kt + GC_WSTAGES - 1);
    const char* wt = wring + (kt % GC_WSTAGES) * WT;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[NJ], bx[MI];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int row = j * 16 + fr;
        a[j] = *(const bf16x8*)(wt + row * 128 + swz_pos(row, s * 4 + fq) * 16);
      }
      const char* xp = xrow + tap * RS + (SWZ ? (cc ^ ((row0 + tap) & 7)) : cc) * 16;  // 16 * stride rows further: same row & 7
#pragma unroll
      for (int i = 0; i < MI; ++i) bx[i] = *(const bf16x8*)(xp + i * sub);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], bx[i], acc[j][i], 0, 0, 0);
      cc += 4;
      tap += cc / CPT;
      cc %= CPT;
    }
  }

  // ---- epilogue: lane = (step fr of subtile i, channels j*16 + fq*4 .. +3)
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int t = t0 + (wave * MI + i) * 16 + fr;
    if (t >= To) continue;
    const int64_t row = (int64_t)b * To + t;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = j * 16 + fq * 4;
      if (n >= cg) continue;  // cg % 4 == 0: a lane's 4 channels are all inside or all outside
      const int f = g * cg + n;
      f32x4 v = acc[j][i];
      if (bias) v += *(const f32x4*)(bias + f);
      v = apply_act4<ACT>(v);
      if (resid) {
        const bf16x4 r = *(const bf16x4*)(resid + row * ldr + f);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
      }
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
      *(bf16x4*)(y + row * ldy + f) = o;
    }
  }
}

template <int CPT, int NJ, int MI, bool SWZ>
int launch_act(int act, dim3 grid, size_t lds, hipStream_t st, const bf16* xg, const bf16* w, const float* bias, const bf16* resid,
               int64_t ldr, bf16* y, int64_t ldy, int G, int Tp, int To, int cg, int stride, int Kp, int tiles_t, int span_rows) {
#define PM_GC(A)                                                                                                             \
  do {                                                                                                                       \
    auto kern = grouped_conv_kernel<CPT, NJ, MI, A, SWZ>;                                                                    \
    static int lds_allowed = 0; /* per instantiation: raise the dynamic-LDS limit once, outside any stream capture */       \
    if ((int)lds > lds_allowed) {                                                                                            \
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)     \
        return PM_ELAUNCH;                                                                                                   \
      lds_allowed = 160 * 1024;                                                                                              \
    }                                                                                                                        \
    hipLaunchKernelGGL(kern, grid, dim3(GC_THREADS), lds, st, xg, w, bias, resid, ldr, y, ldy, G, Tp, To, cg, stride, Kp,   \
                       tiles_t, span_rows);                                                                                  \
  } while (0)
  if (act == PM_ACT_GELU) PM_GC(PM_ACT_GELU);
  else if (act == PM_ACT_NONE) PM_GC(PM_ACT_NONE);
  else return PM_EUNSUPPORTED;
#undef PM_GC
  return PM_OK;
}

template <int CPT, int NJ>
int launch_mi(int mi, int act, int64_t nb, size_t lds, hipStream_t st, const bf16* xg, const bf16* w, const float* bias,
              const bf16* resid, int64_t ldr, bf16* y, int64_t ldy, int G, int Tp, int To, int cg, int stride, int Kp, int tiles_t,
              int span_rows) {
  dim3 grid((unsigned)nb);
  const bool swz = stride == 1 && CPT >= 2;  // must match the LDS size computed by the caller
#define PM_GC_MI(M_, S_) return launch_act<CPT, NJ, M_, S_>(act, grid, lds, st, xg, w, bias, resid, ldr, y, ldy, G, Tp, To, cg, stride, Kp, tiles_t, span_rows)
  if (mi == 4 && swz) PM_GC_MI(4, true);
  if (mi == 4) PM_GC_MI(4, false);
  if (swz) PM_GC_MI(1, true);
  PM_GC_MI(1, false);
#undef PM_GC_MI
}

}  // namespace

extern "C" int pm_grouped_conv_supported(int64_t cg, int64_t cgp) {
  if (cg <= 0 || cg % 4 || cgp < cg) return 0;
  return (cgp == 8 || cgp == 16 || cgp == 32 || cgp == 48 || cgp == 64) && cgp - cg < 8;
}

extern "C" int pm_grouped_conv_bf16(const void* xg, const void* w, const float* bias, const void* resid, int64_t ldr, void* y,
                                    int64_t ldy, int64_t B, int64_t G, int64_t Tp, int64_t cg, int64_t cgp, int64_t k,
                                    int64_t stride, int64_t Kp, int act, void* stream) {
  if (!xg || !w || !y || B < 0 || G <= 0 || Tp <= 0 || k <= 0 || stride <= 0) return PM_EINVAL;
  if (!pm_grouped_conv_supported(cg, cgp)) return PM_EUNSUPPORTED;
  if (Kp % 64 || Kp < k * cgp || Kp - k * cgp >= 64) return PM_EINVAL;
  if (Tp < k) return PM_EINVAL;
  if (ldy < G * cg || (resid && ldr < G * cg)) return PM_EINVAL;
  if (ldy % 4 || (resid && ldr % 4) || (((uintptr_t)y | (uintptr_t)resid) & 7) || (bias && ((uintptr_t)bias & 15)) ||
      (((uintptr_t)xg | (uintptr_t)w) & 15))
    return PM_EALIGN;
  if (B == 0) return PM_OK;
  const int64_t To = (Tp - k) / stride + 1;
  const int mi = To > 128 ? 4 : 1;
  const int64_t bm = 64 * mi, tiles_t = (To + bm - 1) / bm;
  const int cpt = (int)(cgp / 8), nj = (int)((cg + 15) / 16);
  const int64_t taps_p = (Kp / 8 + cpt - 1) / cpt;  // taps the (zero-weighted) K padding still reads
  const int64_t span_rows = (bm - 1) * stride + taps_p;
  const size_t row_bytes = (stride == 1 && cpt >= 2) ? 128 : cpt * 16 + 16;
  const size_t lds = (size_t)GC_WSTAGES * nj * 16 * 128 + (size_t)span_rows * row_bytes;
  if (lds > 160 * 1024) return PM_EUNSUPPORTED;
  const int64_t nb = B * G * tiles_t;
  if (nb > 0x7fffffff || Tp > 0x7fffffff / 64 || Kp > 0x7fffffff) return PM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int rc;
#define PM_GC_GO(CPT, NJ)                                                                                                    \
  rc = launch_mi<CPT, NJ>(mi, act, nb, lds, st, (const bf16*)xg, (const bf16*)w, bias, (const bf16*)resid, ldr, (bf16*)y, ldy, \
                          (int)G, (int)Tp, (int)To, (int)cg, (int)stride, (int)Kp, (int)tiles_t, (int)span_rows)
  if (cpt == 1) PM_GC_GO(1, 1);
  else if (cpt == 2) PM_GC_GO(2, 1);
  else if (cpt == 4) PM_GC_GO(4, 2);
  else if (cpt == 6) PM_GC_GO(6, 3);
  else PM_GC_GO(8, 4);
#undef PM_GC_GO
  if (rc != PM_OK) return rc;
  PM_CHECK_LAUNCH();
  return PM_OK;
}
