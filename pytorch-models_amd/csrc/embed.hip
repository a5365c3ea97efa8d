// embed.hip - token embedding gather + positional add of WhisperDecoder.
// (reference: pytorch_models/audio2text/whisper.py:48-49: token_embs(x) + pos_embs[:L])
// HBM-bound and tiny: B*L rows of d values.  One wave per (b, l) row, 16-byte loads.
#include "common.h"

namespace {

template <bool YF32, typename ET = bf16>
__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ tok, const ET* __restrict__ E,
                                                    const float* __restrict__ pos, void* __restrict__ out, int64_t rows,
                                                    int L, int pos0, int d, int V) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  int64_t t = tok[row];
  t = t < 0 ? 0 : (t >= V ? V - 1 : t);  // ops.embed_tokens raises on ids outside [0, V); clamp so that a raw C-ABI caller's bad id cannot fault
  const int l = (int)(row % L) + pos0;
  for (int c = lane; c < d / 8; c += 64) {
    float e[8];
    if constexpr (sizeof(ET) == 2) {
      const bf16x8 ev = *(const bf16x8*)(E + t * d + c * 8);
#pragma unroll
      for (int i = 0; i < 8; ++i) e[i] = (float)ev[i];
    } else {
      const f32x4 e0 = *(const f32x4*)(E + t * d + c * 8), e1 = *(const f32x4*)(E + t * d + c * 8 + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { e[i] = e0[i]; e[4 + i] = e1[i]; }
    }
    f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = p0;
    if (pos) {  // no absolute positions: T5 (text/t5.py:145)
      p0 = *(const f32x4*)(pos + (int64_t)l * d + c * 8);
      p1 = *(const f32x4*)(pos + (int64_t)l * d + c * 8 + 4);
    }
    float v[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = e[i] + p0[i]; v[4 + i] = e[4 + i] + p1[i]; }
    if constexpr (YF32) {
      *(f32x4*)((float*)out + row * d + c * 8) = f32x4{v[0], v[1], v[2], v[3]};
      *(f32x4*)((float*)out + row * d + c * 8 + 4) = f32x4{v[4], v[5], v[6], v[7]};
    } else {
      bf16x8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = (bf16)v[i];
      *(bf16x8*)((bf16*)out + row * d + c * 8) = o;
    }
  }
}

}  // namespace

extern "C" int pm_embed_tokens(const int64_t* tokens, const void* emb, const float* pos, void* out, int out_dtype,
                               int64_t B, int64_t L, int64_t pos0, int64_t d, int64_t V, void* stream) {
  if (!tokens || !emb || !out || B < 0 || L < 0 || d <= 0 || V <= 0 || pos0 < 0) return PM_EINVAL;
  if (B == 0 || L == 0) return PM_OK;
  if (d % 8) return PM_EUNSUPPORTED;
  if (((uintptr_t)emb | (uintptr_t)pos | (uintptr_t)out) & 15) return PM_EALIGN;
  const int64_t rows = B * L, nblk = (rows + 3) / 4;
  if (nblk > 0x7fffffff) return PM_EINVAL;
  if (out_dtype == PM_F32)
    hipLaunchKernelGGL((embed_kernel<true>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, tokens,
                       (const bf16*)emb, pos, out, rows, (int)L, (int)pos0, (int)d, (int)V);
  else if (out_dtype == PM_BF16)
    hipLaunchKernelGGL((embed_kernel<false>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, tokens,
                       (const bf16*)emb, pos, out, rows, (int)L, (int)pos0, (int)d, (int)V);
  else
    return PM_EINVAL;
  PM_CHECK_LAUNCH();
  return PM_OK;
}

/* pm_embed_tokens from an fp32 table into fp32 rows: the embedding of modules whose parameters are fp32. */
extern "C" int pm_embed_tokens_f32(const int64_t* tokens, const float* emb, const float* pos, float* out, int64_t B, int64_t L,
                                   int64_t pos0, int64_t d, int64_t V, void* stream) {
  if (!tokens || !emb || !out || B < 0 || L < 0 || d <= 0 || V <= 0 || pos0 < 0) return PM_EINVAL;
  if (B == 0 || L == 0) return PM_OK;
  if (d % 8) return PM_EUNSUPPORTED;
  if (((uintptr_t)emb | (uintptr_t)pos | (uintptr_t)out) & 15) return PM_EALIGN;
  const int64_t rows = B * L, nblk = (rows + 3) / 4;
  if (nblk > 0x7fffffff) return PM_EINVAL;
  hipLaunchKernelGGL((embed_kernel<true, float>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, tokens, emb, pos,
                     (void*)out, rows, (int)L, (int)pos0, (int)d, (int)V);
  PM_CHECK_LAUNCH();
  return PM_OK;
}
