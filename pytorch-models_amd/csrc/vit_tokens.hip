// vit_tokens.hip - ViT token assembly: im2col-free Conv2d(3, d, 16, stride 16) patch projection on
// MFMA with the "+ pe" and "prepend cls" of the reference fused into the epilogue.
// (reference: pytorch_models/image/vit.py:64,78-81)
//
// Roofline: the stage reads every input pixel (fp32 NCHW) once from HBM - 3*H*W*4 B per image - and
// writes (L + cls)*d*2 B; at d = 768 it is 302 flop/B, i.e. at the MFMA/HBM ridge, so it is priced in
// GB/s of algorithmic bytes.
//
// GEMM view: rows = patches (n, gy, gx), K = (c, ph, pw) = 768, cols = features.  A 64-deep K step is
// one channel x 4 patch rows x 16 pixels, so a patch contributes four 64-byte pixel runs per step and
// 16 consecutive patches of one image row contribute one contiguous 1 KiB run: a wave's float4 loads
// (4 lanes per patch run, 16 patches) are full-line coalesced.  Pixels are converted to bf16 on the
// way into the swizzled LDS tile, two K steps after their loads were issued; the weight tile goes through
// registers as well (see load_w: one kind of load in flight keeps the compiler's waits exact).  The
// generic-patch kernel below keeps the simple single-buffered form with global_load_lds weights.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64, P = 16;
constexpr int TILE_BYTES = 128 * BK * 2;

__device__ __forceinline__ void stage_w(const bf16* __restrict__ W, int64_t ld, int row0, int row_max, int k0,
                                        char* tile, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rbase = wave * 32 + i * 8;
    const int rt = rbase + (lane >> 3);
    const int chunk = swz_pos(rt, lane & 7);
    int grow = row0 + rt;
    grow = grow < row_max ? grow : row_max - 1;
    glds16(W + (int64_t)grow * ld + k0 + chunk * 8, tile + rbase * 128);
  }
}

// the same tile through registers (vit_tokens_kernel: its pixel loads are ordinary loads, and the compiler's wait insertion only
// counts exactly when every vector-memory operation in flight is of one kind - with LDS-DMA in the mix it waits vmcnt(0))
__device__ __forceinline__ void load_w(const bf16* __restrict__ W, int64_t ld, int row0, int row_max, int k0, int wave, int lane,
                                       bf16x8 (&regs)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rt = wave * 32 + i * 8 + (lane >> 3);
    int grow = row0 + rt;
    grow = grow < row_max ? grow : row_max - 1;
    regs[i] = *(const bf16x8*)(W + (int64_t)grow * ld + k0 + (lane & 7) * 8);
  }
}
__device__ __forceinline__ void write_w(char* tile, int wave, int lane, const bf16x8 (&regs)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rt = wave * 32 + i * 8 + (lane >> 3);
    *(bf16x8*)(tile + rt * 128 + swz_pos(rt, lane & 7) * 16) = regs[i];
  }
}

__device__ __forceinline__ bf16x8 read_frag(const char* tile, int row, int chunk) {
  return *(const bf16x8*)(tile + row * 128 + swz_pos(row, chunk) * 16);
}

__global__ __launch_bounds__(256, 2) void vit_tokens_kernel(const float* __restrict__ imgs, const bf16* __restrict__ W,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ pe, const float* __restrict__ cls,
                                                            bf16* __restrict__ out, int64_t Mtot, int Himg, int Wimg,
                                                            int gw, int Lp, int d, int tiles_n) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
  const int64_t m0 = (int64_t)tm * BM;
  const int n0 = tn * BN;
  const int wm = wave >> 1, wn = wave & 1;
  const int K = 3 * P * P;

  // pixel-run ownership: thread -> (patch tid/4 [+64], quarter tid%4) for each of the 4 patch rows of a K step
  const int quarter = tid & 3;
  int64_t pix_base[2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    int64_t pm = m0 + (tid >> 2) + hf * 64;
    pm = pm < Mtot ? pm : Mtot - 1;
    const int64_t n = pm / Lp;
    const int p = (int)(pm - n * Lp);
    const int gy = p / gw, gx = p - gy * gw;
    pix_base[hf] = (n * 3 * Himg + gy * P) * (int64_t)Wimg + gx * P + quarter * 4;
  }
  // pixels travel two K steps ahead of their use (two register sets), the weight tile one step: when step kt ends, the pixel
  // loads of step kt + 2 are the youngest vector-memory operations and stay in flight while the compiler's exact wait covers
  // the weights and pixels of step kt + 1.  With one set and vmcnt(0) per step every step exposed a full HBM latency
  // (1.4 TB/s of algorithmic bytes).
  f32x4 areg[2][8];
  auto load_a = [&](int kt, f32x4 (&regs)[8]) {
    const int k0 = kt * BK;
    const int c = k0 >> 8, ph0 = (k0 & 255) >> 4;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int ph = ph0 + (it >> 1);
      regs[it] = *(const f32x4*)(imgs + pix_base[it & 1] + ((int64_t)c * Himg + ph) * Wimg);
    }
  };
  auto write_a = [&](char* tile, const f32x4 (&regs)[8]) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = (tid >> 2) + (it & 1) * 64;
      const int chunk = 2 * (it >> 1) + (quarter >> 1);
      bf16x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (bf16)regs[it][j];
      *(bf16x4*)(tile + row * 128 + swz_pos(row, chunk) * 16 + (quarter & 1) * 8) = v;
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int nk = 3 * P * P / BK;  // 12, compile-time: the loop is fully unrolled so that the compiler's own waits for the pixel
                                      // registers are exact counts (across a loop back-edge it falls back to vmcnt(0))
  bf16x8 wreg[4];
  load_a(0, areg[0]);
  load_w(W, K, n0, d, 0, wave, lane, wreg);
  load_a(1, areg[1]);
  __builtin_amdgcn_sched_barrier(0);
  write_a(smem, areg[0]);
  write_w(smem + TILE_BYTES, wave, lane, wreg);
  // raw barrier: __syncthreads() carries a fence that waits for EVERY outstanding load (vmcnt(0)), the prefetch included
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int kt = 0; kt < nk; ++kt) {
    char* acur = smem + (kt & 1) * 2 * TILE_BYTES;
    char* wcur = acur + TILE_BYTES;
    char* anxt = smem + ((kt + 1) & 1) * 2 * TILE_BYTES;
    if (kt + 1 < nk) load_w(W, K, n0, d, (kt + 1) * BK, wave, lane, wreg);
    if (kt + 2 < nk) load_a(kt + 2, areg[kt & 1]);  // the set that fed step kt (written to LDS one step ago)
    __builtin_amdgcn_sched_barrier(0);  // the scheduler otherwise sinks these loads down to their use, two steps later
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[4], b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = read_frag(wcur, wn * 64 + j * 16 + fr, s * 4 + fq);
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = read_frag(acur, wm * 64 + i * 16 + fr, s * 4 + fq);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], b[i], acc[j][i], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      write_a(anxt, areg[(kt + 1) & 1]);
      write_w(anxt + TILE_BYTES, wave, lane, wreg);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  const int has_cls = cls != nullptr;
  const int Lt = Lp + has_cls;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t pm = m0 + wm * 64 + i * 16 + fr;
    if (pm >= Mtot) continue;
    const int64_t n = pm / Lp;
    const int p = (int)(pm - n * Lp);
    bf16* orow = out + (n * Lt + has_cls + p) * (int64_t)d;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = n0 + wn * 64 + j * 16 + fq * 4;
      if (f >= d) continue;
      const f32x4 v = acc[j][i] + *(const f32x4*)(bias + f) + *(const f32x4*)(pe + (int64_t)p * d + f);
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (bf16)v[r];
      *(bf16x4*)(orow + f) = o;
      if (has_cls && p == 0) {  // the cls row of image n, broadcast over the batch
        const f32x4 cv = *(const f32x4*)(cls + f);
        bf16x4 co;
#pragma unroll
        for (int r = 0; r < 4; ++r) co[r] = (bf16)cv[r];
        *(bf16x4*)(out + n * Lt * (int64_t)d + f) = co;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Any patch size (14: DINOv2, 32, 8): same GEMM, but the pixel gather walks (channel, row, column) element by element
// with scalar loads and the K extent is zero-padded to a multiple of 64 (the packed weight is padded the same way).
// Not on the benchmark path (every BASELINE config uses patch 16); correctness-first.
__global__ __launch_bounds__(256, 2) void vit_tokens_generic_kernel(const float* __restrict__ imgs, const bf16* __restrict__ W,
                                                                    int64_t ldw, const float* __restrict__ bias,
                                                                    const float* __restrict__ pe,
                                                                    const float* __restrict__ cls, bf16* __restrict__ out,
                                                                    int64_t Mtot, int Himg, int Wimg, int Pp, int gw, int Lp,
                                                                    int d, int tiles_n) {
  __shared__ __attribute__((aligned(16))) char smem[2 * TILE_BYTES];  // single-buffered: A tile, W tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
  const int64_t m0 = (int64_t)tm * BM;
  const int n0 = tn * BN;
  const int wm = wave >> 1, wn = wave & 1;
  const int K = 3 * Pp * Pp, Kpad = (K + BK - 1) / BK * BK;
  char* atile = smem;
  char* wtile = smem + TILE_BYTES;

  // thread -> (patch row tid / 2, 32 consecutive k of the step)
  const int prow = tid >> 1, khalf = tid & 1;
  int64_t pm = m0 + prow;
  pm = pm < Mtot ? pm : Mtot - 1;
  const int64_t nimg = pm / Lp;
  const int pidx = (int)(pm - nimg * Lp);
  const int gy = pidx / gw, gx = pidx - gy * gw;
  const float* pbase = imgs + (nimg * 3 * Himg + (int64_t)gy * Pp) * Wimg + gx * Pp;

  f32x4 acc[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < Kpad / BK; ++kt) {
    const int k0 = kt * BK;
    stage_w(W, ldw, n0, d, k0, wtile, wave, lane);
    int k = k0 + khalf * 32;
    int c = k / (Pp * Pp), rem = k - c * Pp * Pp;
    int ph = rem / Pp, pw = rem - ph * Pp;
#pragma unroll 4
    for (int e = 0; e < 32; ++e) {
      const float v = (k + e < K) ? pbase[((int64_t)c * Himg + ph) * Wimg + pw] : 0.f;
      const int kk = khalf * 32 + e;
      *(bf16*)(atile + prow * 128 + swz_pos(prow, kk >> 3) * 16 + (kk & 7) * 2) = (bf16)v;
      if (++pw == Pp) { pw = 0; if (++ph == Pp) { ph = 0; ++c; } }
    }
    wait_vmcnt0();
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[4], b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = read_frag(wtile, wn * 64 + j * 16 + fr, s * 4 + fq);
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = read_frag(atile, wm * 64 + i * 16 + fr, s * 4 + fq);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], b[i], acc[j][i], 0, 0, 0);
    }
    __syncthreads();
  }

  const int has_cls = cls != nullptr;
  const int Lt = Lp + has_cls;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t pmo = m0 + wm * 64 + i * 16 + fr;
    if (pmo >= Mtot) continue;
    const int64_t n = pmo / Lp;
    const int p = (int)(pmo - n * Lp);
    bf16* orow = out + (n * Lt + has_cls + p) * (int64_t)d;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = n0 + wn * 64 + j * 16 + fq * 4;
      if (f >= d) continue;
      const f32x4 v = acc[j][i] + *(const f32x4*)(bias + f) + *(const f32x4*)(pe + (int64_t)p * d + f);
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (bf16)v[r];
      *(bf16x4*)(orow + f) = o;
      if (has_cls && p == 0) {
        const f32x4 cv = *(const f32x4*)(cls + f);
        bf16x4 co;
#pragma unroll
        for (int r = 0; r < 4; ++r) co[r] = (bf16)cv[r];
        *(bf16x4*)(out + n * Lt * (int64_t)d + f) = co;
      }
    }
  }
}

}  // namespace

extern "C" int pm_vit_tokens(const float* imgs, const void* w, const float* bias, const float* pe, const float* cls,
                             void* out, int64_t N, int64_t Himg, int64_t Wimg, int64_t Pp, int64_t d, void* stream) {
  if (!imgs || !w || !bias || !pe || !out || N < 0 || Himg <= 0 || Wimg <= 0 || d <= 0) return PM_EINVAL;
  if (N == 0) return PM_OK;
  if (Pp != P) return PM_EUNSUPPORTED;  // patch 16 only (every BASELINE config); 14 / 32 / 8 are "next" rows
  if (Himg % P || Wimg % P || d % 4) return PM_EUNSUPPORTED;
  if (((uintptr_t)imgs | (uintptr_t)w | (uintptr_t)bias | (uintptr_t)pe | (uintptr_t)cls) & 15) return PM_EALIGN;
  if ((uintptr_t)out & 7) return PM_EALIGN;
  const int gw = (int)(Wimg / P), gh = (int)(Himg / P);
  const int64_t Lp = (int64_t)gw * gh;
  const int64_t Mtot = N * Lp;
  const int64_t tiles_m = (Mtot + BM - 1) / BM, tiles_n = (d + BN - 1) / BN;
  if (tiles_m * tiles_n > 0x7fffffff || Lp > (1 << 24)) return PM_EINVAL;
  hipLaunchKernelGGL(vit_tokens_kernel, dim3((unsigned)(tiles_m * tiles_n)), dim3(256), 0, (hipStream_t)stream, imgs,
                     (const bf16*)w, bias, pe, cls, (bf16*)out, Mtot, (int)Himg, (int)Wimg, gw, (int)Lp, (int)d,
                     (int)tiles_n);
  PM_CHECK_LAUNCH();
  return PM_OK;
}

extern "C" int pm_vit_tokens_generic(const float* imgs, const void* w, int64_t ldw, const float* bias, const float* pe,
                                     const float* cls, void* out, int64_t N, int64_t Himg, int64_t Wimg, int64_t Pp,
                                     int64_t d, void* stream) {
  if (!imgs || !w || !bias || !pe || !out || N < 0 || Himg <= 0 || Wimg <= 0 || d <= 0 || Pp <= 0) return PM_EINVAL;
  if (N == 0) return PM_OK;
  if (Himg % Pp || Wimg % Pp || d % 4 || Pp > 64) return PM_EUNSUPPORTED;
  const int64_t K = 3 * Pp * Pp, Kpad = (K + BK - 1) / BK * BK;
  if (ldw < Kpad || ldw % 8) return PM_EINVAL;
  if (((uintptr_t)w | (uintptr_t)bias | (uintptr_t)pe | (uintptr_t)cls) & 15) return PM_EALIGN;
  if ((uintptr_t)out & 7) return PM_EALIGN;
  const int gw = (int)(Wimg / Pp), gh = (int)(Himg / Pp);
  const int64_t Lp = (int64_t)gw * gh, Mtot = N * Lp;
  const int64_t tiles_m = (Mtot + BM - 1) / BM, tiles_n = (d + BN - 1) / BN;
  if (tiles_m * tiles_n > 0x7fffffff || Lp > (1 << 24)) return PM_EINVAL;
  hipLaunchKernelGGL(vit_tokens_generic_kernel, dim3((unsigned)(tiles_m * tiles_n)), dim3(256), 0, (hipStream_t)stream, imgs,
                     (const bf16*)w, ldw, bias, pe, cls, (bf16*)out, Mtot, (int)Himg, (int)Wimg, (int)Pp, gw, (int)Lp, (int)d,
                     (int)tiles_n);
  PM_CHECK_LAUNCH();
  return PM_OK;
}
