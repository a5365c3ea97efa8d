// attention_f32.hip - softmax(q k^T / 8) v on FP32 operands, head_dim 64, on the f32-input matrix pipe
// (reference: pytorch_models/transformer.py:52, F.scaled_dot_product_attention of fp32 modules; no mask, no bias, not causal:
// the encoder of the reference-accuracy Whisper pipeline and fp32 ViTs.  Other forms stay on attention_generic.hip).
//
// Roofline: MFMA (f32 in: 157 TFLOP/s = 64 flop / clk / SIMD, 1/16 of bf16): 4 * B * H * Lq * Lk * 64 flop; Whisper-base
// encoder, 32 clips: 147 GF per layer = 0.94 ms at the f32 matrix peak.  The VALU kernel it replaces took 16.9 ms per layer.
//
// v_mfma_f32_32x32x2_f32 is bit for bit a k-ordered fmaf chain, so this is fp32 arithmetic throughout (products and sums; the
// exponentials are expf).  Structure (the bf16 kernel's idea in fp32): a wave owns 32 queries and computes S^T = K Q^T - the
// QUERY ends on the lane (D column), 16 of a 32-key block's scores per lane in registers - so the row maximum / sum are
// in-lane plus one exchange with lane ^ 32, the rescale factor of the running output is a per-lane scalar, and the
// exponentials ARE the B operand of O^T += V^T P^T as they stand (accumulator register r of lane half h is key
// (r & 3) + 8 (r >> 2) + 4 h: the V^T operand is read in that key order).  Q lives in registers (pre-scaled by 1/8, exact),
// K / V tiles of 32 keys in LDS with a 65-float row pitch (conflict-free for both operand reads).
#include "common.h"

namespace {

constexpr int FA_KEYS = 32, FA_PITCH = 65, FA_QPW = 32, FA_QPB = 128;

__global__ __launch_bounds__(256) void attn_f32_hd64_kernel(const float* __restrict__ q, int64_t qsb, int64_t qst,
                                                            const float* __restrict__ k, int64_t ksb, int64_t kst,
                                                            const float* __restrict__ v, int64_t vsb, int64_t vst,
                                                            float* __restrict__ o, int64_t osb, int64_t ost, int H, int Lq,
                                                            int Lk, int nqb) {
  __shared__ float Ks[FA_KEYS * FA_PITCH], Vs[FA_KEYS * FA_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qb = blockIdx.x % nqb, bh = blockIdx.x / nqb;
  const int b = bh / H, h = bh - b * H;
  const int ql = lane & 31, half = lane >> 5;
  const int qi = qb * FA_QPB + wave * FA_QPW + ql;
  const int qrow = qi < Lq ? qi : Lq - 1;
  const float* qp = q + (int64_t)b * qsb + (int64_t)qrow * qst + h * 64;
  float qr[32];  // Q[query][2 s + half] / 8
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const f32x4 t = *(const f32x4*)(qp + 4 * g);  // dims 4g .. 4g+3: steps 2g (dims 4g, 4g+1) and 2g+1 (4g+2, 4g+3)
    qr[2 * g] = (half ? t[1] : t[0]) * 0.125f;
    qr[2 * g + 1] = (half ? t[3] : t[2]) * 0.125f;
  }
  f32x16 oacc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.f; oacc[1][r] = 0.f; }
  float m = -INFINITY, lsum = 0.f;
  const int lkey = tid >> 3, ld0 = (tid & 7) * 8;  // staging: this thread's key row and first of its 8 dims
  const float* kbp = k + (int64_t)b * ksb + h * 64 + ld0;
  const float* vbp = v + (int64_t)b * vsb + h * 64 + ld0;
  for (int k0 = 0; k0 < Lk; k0 += FA_KEYS) {
    int key = k0 + lkey;
    key = key < Lk ? key : Lk - 1;
    const f32x4 ka = *(const f32x4*)(kbp + (int64_t)key * kst), kb2 = *(const f32x4*)(kbp + (int64_t)key * kst + 4);
    const f32x4 va = *(const f32x4*)(vbp + (int64_t)key * vst), vb2 = *(const f32x4*)(vbp + (int64_t)key * vst + 4);
    __syncthreads();  // the previous tile has been consumed by every wave
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      Ks[lkey * FA_PITCH + ld0 + e] = ka[e];
      Ks[lkey * FA_PITCH + ld0 + 4 + e] = kb2[e];
      Vs[lkey * FA_PITCH + ld0 + e] = va[e];
      Vs[lkey * FA_PITCH + ld0 + 4 + e] = vb2[e];
    }
    __syncthreads();
    // ---- S^T = K Q^T: s[r] = score of key k0 + (r & 3) + 8 (r >> 2) + 4 half with query ql
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int st = 0; st < 32; ++st) s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[ql * FA_PITCH + 2 * st + half], qr[st], s, 0, 0, 0);
    float mloc = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kk = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      s[r] = kk < Lk ? s[r] : -INFINITY;
      mloc = fmaxf(mloc, s[r]);
    }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float mnew = fmaxf(m, mloc);  // finite: tile 0 holds key 0
    const float alpha = expf(m - mnew);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = expf(s[r] - mnew);
      psum += s[r];
    }
    lsum = fmaf(lsum, alpha, psum);
    m = mnew;
#pragma unroll
    for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; }
    // ---- O^T += V^T P^T, keys in the accumulator's own order
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kr = (r & 3) + 8 * (r >> 2) + 4 * half;
      oacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[kr * FA_PITCH + ql], s[r], oacc[0], 0, 0, 0);
      oacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[kr * FA_PITCH + 32 + ql], s[r], oacc[1], 0, 0, 0);
    }
  }
  lsum += __shfl_xor(lsum, 32, 64);
  if (qi < Lq) {
    const float inv = 1.0f / lsum;
    float* op = o + (int64_t)b * osb + (int64_t)qi * ost + h * 64;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int g = 0; g < 4; ++g) {  // registers 4g .. 4g+3 = dims 8g + 4 half .. + 3 (+ 32 hf)
        f32x4 t;
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = oacc[hf][4 * g + e] * inv;
        *(f32x4*)(op + 32 * hf + 8 * g + 4 * half) = t;
      }
  }
}

}  // namespace

// Internal entry (attention_generic.hip dispatches here for head_dim 64 without mask / bias; arguments validated there).
int pm_attention_f32_hd64_launch(const float* q, int64_t qsb, int64_t qst, const float* k, int64_t ksb, int64_t kst, const float* v,
                                 int64_t vsb, int64_t vst, float* o, int64_t osb, int64_t ost, int64_t B, int64_t H, int64_t Lq,
                                 int64_t Lk, hipStream_t st) {
  const int nqb = (int)((Lq + FA_QPB - 1) / FA_QPB);
  const int64_t nblk = B * H * nqb;
  if (nblk > 0x7fffffff) return PM_EINVAL;
  hipLaunchKernelGGL(attn_f32_hd64_kernel, dim3((unsigned)nblk), dim3(256), 0, st, q, qsb, qst, k, ksb, kst, v, vsb, vst, o, osb,
                     ost, (int)H, (int)Lq, (int)Lk, nqb);
  return PM_OK;
}
