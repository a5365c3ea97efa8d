// whisper_stem1.hip - first stem conv of WhisperEncoder: Conv1d(n_mels, d, 3, stride 1, pad 1) + GELU on a
// channel-major fp32 log-mel (B, C, T), written TIME-major in bf16 as (B, T + 2, d) with zero rows 0 and T+1.
// (reference: pytorch_models/audio2text/whisper.py:16-18,30)
//
// Why this layout: with rows = time and K = (tap, channel) the SECOND conv (k 3, stride 2, pad 1) becomes a
// plain K-contiguous GEMM over this buffer - row t' of its A operand is the 3*d contiguous values starting at
// buffer row 2*t' - so it runs on linear_bf16.hip unchanged (x_row_stride = 2*d, K = 3*d), GELU and pos_embs in
// its epilogue, and the transpose(1, 2) of the reference never happens.
//
// GEMM view here: rows = (clip, t), K = (tap, channel padded to Cpad), cols = d.  The A tile is gathered from
// the channel-major input (coalesced along t), converted to bf16 and transposed into the swizzled LDS tile; the
// packed weight (d, 3 * Cpad) streams in with global_load_lds.  0.74 GFLOP per 30 s clip at d = 512: <1 % of the
// encoder; priced in DESIGN.md as HBM bytes (0.96 MB in, 3.07 MB out per clip).
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;

__device__ __forceinline__ bf16x8 read_frag(const char* tile, int row, int chunk) {
  return *(const bf16x8*)(tile + row * 128 + swz_pos(row, chunk) * 16);
}

__global__ __launch_bounds__(256, 2) void whisper_stem1_kernel(const float* __restrict__ x, const bf16* __restrict__ W,
                                                               const float* __restrict__ bias, bf16* __restrict__ out,
                                                               int C, int Cpad, int T, int d, int tiles_t, int tiles_n) {
  __shared__ __attribute__((aligned(16))) char smem[2 * TILE_BYTES];  // A tile, W tile (single-buffered: K is short)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = wg % tiles_n;
  wg /= tiles_n;
  const int tt = wg % tiles_t, b = wg / tiles_t;
  const int t0 = tt * BM, n0 = tn * BN;
  const int wm = wave >> 1, wn = wave & 1;
  const int K = 3 * Cpad;
  const float* xb = x + (int64_t)b * C * T;
  char* atile = smem;
  char* wtile = smem + TILE_BYTES;

  f32x4 acc[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int tx = tid & 31, cy = tid >> 5;  // 32 threads x 4 consecutive t = 128 rows; 8 channels per pass
  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < K / BK; ++kt) {
    const int k0 = kt * BK;
    const int tap = k0 / Cpad, c0 = k0 - tap * Cpad;
    // weight tile via LDS-DMA
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rbase = wave * 32 + i * 8;
      const int rt = rbase + (lane >> 3);
      int grow = n0 + rt;
      grow = grow < d ? grow : d - 1;
      glds16(W + (int64_t)grow * K + k0 + swz_pos(rt, lane & 7) * 8, wtile + rbase * 128);
    }
    // A tile: A[t_local][c_local] = x[b][c0 + c_local][t0 + t_local + tap - 1]
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int cl = pass * 8 + cy;
      const int c = c0 + cl;
      const float* src = xb + (int64_t)c * T;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int tl = tx * 4 + e;
        const int t = t0 + tl + tap - 1;
        const float v = (c < C && t >= 0 && t < T) ? src[t] : 0.f;
        *(bf16*)(atile + tl * 128 + swz_pos(tl, cl >> 3) * 16 + (cl & 7) * 2) = (bf16)v;
      }
    }
    wait_vmcnt0();
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[4], bb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = read_frag(wtile, wn * 64 + j * 16 + fr, s * 4 + fq);
#pragma unroll
      for (int i = 0; i < 4; ++i) bb[i] = read_frag(atile, wm * 64 + i * 16 + fr, s * 4 + fq);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], bb[i], acc[j][i], 0, 0, 0);
    }
    __syncthreads();
  }

  bf16* ob = out + (int64_t)b * (T + 2) * d;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = t0 + wm * 64 + i * 16 + fr;
    if (t >= T) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = n0 + wn * 64 + j * 16 + fq * 4;
      if (f >= d) continue;
      const f32x4 v = acc[j][i] + *(const f32x4*)(bias + f);
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (bf16)apply_act<PM_ACT_GELU, false>(v[r]);
      *(bf16x4*)(ob + (int64_t)(t + 1) * d + f) = o;
      const bf16x4 z = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
      if (t == 0) *(bf16x4*)(ob + f) = z;                                // left zero-padding row
      if (t == T - 1) *(bf16x4*)(ob + (int64_t)(T + 1) * d + f) = z;     // right zero-padding row
    }
  }
}

}  // namespace

extern "C" int pm_whisper_stem1(const float* x, const void* w, const float* bias, void* out, int64_t B, int64_t C,
                                int64_t Cpad, int64_t T, int64_t d, void* stream) {
  if (!x || !w || !bias || !out || B < 0 || C <= 0 || T <= 0 || d <= 0) return PM_EINVAL;
  if (B == 0) return PM_OK;
  if (Cpad % BK || Cpad < C || d % 4) return PM_EUNSUPPORTED;
  if (((uintptr_t)w | (uintptr_t)bias) & 15 || ((uintptr_t)out & 7)) return PM_EALIGN;
  const int64_t tiles_t = (T + BM - 1) / BM, tiles_n = (d + BN - 1) / BN;
  if (B * tiles_t * tiles_n > 0x7fffffff) return PM_EINVAL;
  hipLaunchKernelGGL(whisper_stem1_kernel, dim3((unsigned)(B * tiles_t * tiles_n)), dim3(256), 0, (hipStream_t)stream, x,
                     (const bf16*)w, bias, (bf16*)out, (int)C, (int)Cpad, (int)T, (int)d, (int)tiles_t, (int)tiles_n);
  PM_CHECK_LAUNCH();
  return PM_OK;
}
