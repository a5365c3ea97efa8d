// decode.hip - the KV-cached greedy decode step of WhisperDecoder (a capability the reference lacks; its
// semantics are "argmax of the last position's logits, append" as in pytorch_models/text/generator.py:23-35,
// over the layer algebra of pytorch_models/transformer.py:96-100 and audio2text/whisper.py:47-53).
//
// One step = one new token per sequence.  Activations are fp32 end to end (bit-exact greedy ids against the
// fp32 oracle need it: SURVEY.md 7.3); weights and the K/V caches are bf16.  The step is HBM-bound: at batch
// 32 it streams the cross-attention K/V of every layer (24.6 MB per sequence for "base") plus every weight once.
//
//  dec_embed          x = E[token] + pos[t]                                      (whisper.py:48-49)
//  dec_linear         y = [LayerNorm](x) W^T + b [GELU] [+ resid]  for <= 64 rows: the fp32 activations are split
//                     into three bf16 terms (hi + mid + lo = x exactly) so the bf16 MFMA reproduces an fp32 GEMM on
//                     bf16 weights: 3 MFMAs per K step, free under the weight stream.  Workgroup = 16 output
//                     features, 4 waves split K and reduce through LDS in a fixed order (deterministic).
//                     mode QKV scatters k, v (rounded to bf16) into the caches at position t;
//                     mode ARGMAX reduces the tile's logits to (max, index) per sequence instead of storing them.
//  dec_attn           softmax(q K^T / 8) V for one query per (sequence, head) over a bf16 K/V with arbitrary strides
//                     (self cache or the packed cross projection): exact two-pass softmax, scores parked in LDS.
//  dec_argmax_reduce  per-sequence winner over all tiles (lowest index on ties, like torch.argmax), teacher-forces
//                     the prompt, appends to the output.
//  dec_advance        ++t (its own launch: every workgroup of the step has read t by then).
// The position t and the current tokens live in device memory, so ONE captured hipGraph replays every step.
#include "common.h"

namespace {

constexpr int DL_FEATS = 16;  // output features per workgroup

__device__ __forceinline__ void split3(const float (&v)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const bf16 h = (bf16)v[i];
    const float r1 = v[i] - (float)h;
    const bf16 m = (bf16)r1;
    const float r2 = r1 - (float)m;
    hi[i] = h;
    mid[i] = m;
    lo[i] = (bf16)r2;
  }
}

enum { DL_PLAIN = 0, DL_QKV = 1, DL_ARGMAX = 2 };

template <int ACT, int MT, int FT>
__global__ __launch_bounds__(256) void dec_linear_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps,
                                                         const bf16* __restrict__ W, int64_t ldw,
                                                         const float* __restrict__ bias, const float* resid, int ldr,
                                                         float* out, int ldo, int M, int N, int K, int mode,
                                                         bf16* __restrict__ kcache, bf16* __restrict__ vcache, int inner,
                                                         int H, int Tmax, const int* __restrict__ pos_ptr,
                                                         float* __restrict__ ws_val, int* __restrict__ ws_idx, int nwg) {
  // FT = 16-feature MFMA row tiles per workgroup (1 for the layer projections: more workgroups pull the weight
  // stream; 4 for the 51865-row vocabulary: the x loads and the LayerNorm are amortised over 64 features)
  __shared__ float stats[64 * 2];
  __shared__ __attribute__((aligned(16))) float red[4 * FT * MT * 64 * 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = blockIdx.x * (DL_FEATS * FT);
  const bool ln = gamma != nullptr;

  if (ln) {  // per-row mean and rstd (two passes, like the reference's mean / biased variance): 8 threads per row
    for (int r0 = 0; r0 < M; r0 += 32) {
      const int row = r0 + (tid >> 3), part = tid & 7;
      const int rr = row < M ? row : M - 1;
      const float* xr = x + (int64_t)rr * ldx;
      float s = 0.f;
      for (int k = part * 4; k < K; k += 32) {
        const f32x4 v = *(const f32x4*)(xr + k);
        s += (v[0] + v[1]) + (v[2] + v[3]);
      }
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      const float mean = s / (float)K;
      float q = 0.f;
      for (int k = part * 4; k < K; k += 32) {
        const f32x4 v = *(const f32x4*)(xr + k);
#pragma unroll
        for (int i = 0; i < 4; ++i) q = fmaf(v[i] - mean, v[i] - mean, q);
      }
      q += __shfl_xor(q, 1, 64);
      q += __shfl_xor(q, 2, 64);
      q += __shfl_xor(q, 4, 64);
      if (part == 0 && row < M) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = rsqrtf(q / (float)K + eps);
      }
    }
    __syncthreads();
  }

  const int fi = lane & 15, kq = lane >> 4;
  f32x4 acc[FT][MT];
#pragma unroll
  for (int f = 0; f < FT; ++f)
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16* wp[FT];
#pragma unroll
  for (int f = 0; f < FT; ++f) {
    int wrow = n0 + f * 16 + fi;
    wrow = wrow < N ? wrow : N - 1;
    wp[f] = W + (int64_t)wrow * ldw + kq * 8;
  }
  const float* xrow[MT];
  float mean[MT], rstd[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    int row = t * 16 + fi;
    row = row < M ? row : M - 1;
    xrow[t] = x + (int64_t)row * ldx + kq * 8;
    mean[t] = ln ? stats[2 * row] : 0.f;
    rstd[t] = ln ? stats[2 * row + 1] : 1.f;
  }
  const int ksteps = K >> 5;
  // The step is latency-bound: issue the loads of U k-steps (weights, activations, LayerNorm affine) back to back,
  // then do their MFMAs - one memory round trip per block of U instead of one per k-step.
  constexpr int U = FT == 1 ? 4 : 2;
  for (int sb = wave; sb < ksteps; sb += 4 * U) {
    bf16x8 a[U][FT];
    f32x4 xv[U][MT][2], gv[U][2], bv[U][2];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int s = sb + 4 * u;
      s = s < ksteps ? s : ksteps - 1;
#pragma unroll
      for (int f = 0; f < FT; ++f) a[u][f] = *(const bf16x8*)(wp[f] + s * 32);
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        xv[u][t][0] = *(const f32x4*)(xrow[t] + s * 32);
        xv[u][t][1] = *(const f32x4*)(xrow[t] + s * 32 + 4);
      }
      if (ln) {
        const int k0 = s * 32 + kq * 8;
        gv[u][0] = *(const f32x4*)(gamma + k0);
        gv[u][1] = *(const f32x4*)(gamma + k0 + 4);
        bv[u][0] = *(const f32x4*)(beta + k0);
        bv[u][1] = *(const f32x4*)(beta + k0 + 4);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (sb + 4 * u >= ksteps) break;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = xv[u][t][0][i]; v[4 + i] = xv[u][t][1][i]; }
        if (ln) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = (v[i] - mean[t]) * rstd[t] * gv[u][i >> 2][i & 3] + bv[u][i >> 2][i & 3];
        }
        bf16x8 hi, mid, lo;
        split3(v, hi, mid, lo);
#pragma unroll
        for (int f = 0; f < FT; ++f) {
          acc[f][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][f], hi, acc[f][t], 0, 0, 0);
          acc[f][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][f], mid, acc[f][t], 0, 0, 0);
          acc[f][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][f], lo, acc[f][t], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int f = 0; f < FT; ++f)
#pragma unroll
    for (int t = 0; t < MT; ++t) *(f32x4*)(red + (((wave * FT + f) * MT + t) * 64 + lane) * 4) = acc[f][t];
  __syncthreads();
  if (wave != 0) return;

  // D[row = feature 4*kq + r][col = sequence fi]; partial sums added in wave order 0..3
  const int tpos = (mode == DL_QKV) ? *pos_ptr : 0;
  float best_v[MT];
  int best_i[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) { best_v[t] = -INFINITY; best_i[t] = 0x7fffffff; }
#pragma unroll
  for (int f = 0; f < FT; ++f) {
    const int n = n0 + f * 16 + kq * 4;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      f32x4 v = *(const f32x4*)(red + (((0 * FT + f) * MT + t) * 64 + lane) * 4);
#pragma unroll
      for (int w = 1; w < 4; ++w) v += *(const f32x4*)(red + (((w * FT + f) * MT + t) * 64 + lane) * 4);
      const int row = t * 16 + fi;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nn = n + r;
        if (nn >= N) continue;
        float e = v[r] + (bias ? bias[nn] : 0.f);
        e = apply_act<ACT, true>(e);
        if (mode == DL_ARGMAX) {
          if (e > best_v[t]) { best_v[t] = e; best_i[t] = nn; }  // ascending nn: strict > keeps the lowest index
          continue;
        }
        if (row >= M) continue;
        if (mode == DL_PLAIN) {
          if (resid) e += resid[(int64_t)row * ldr + nn];
          out[(int64_t)row * ldo + nn] = e;
        } else {  // DL_QKV: [q | k | v] column blocks of width inner
          const int which = nn / inner, c = nn - which * inner;
          if (which == 0) {
            out[(int64_t)row * ldo + c] = e;
          } else {
            bf16* cache = which == 1 ? kcache : vcache;
            const int h = c >> 6, dd = c & 63;
            cache[(((int64_t)row * H + h) * Tmax + tpos) * 64 + dd] = (bf16)e;
          }
        }
      }
    }
  }
  if (mode == DL_ARGMAX) {
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      float bv = best_v[t];
      int bi = best_i[t];
#pragma unroll
      for (int o = 16; o < 64; o <<= 1) {  // the 4 lanes sharing a sequence (kq = 0..3)
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      const int row = t * 16 + fi;
      if (kq == 0 && row < M) {
        ws_val[(int64_t)row * nwg + blockIdx.x] = bv;
        ws_idx[(int64_t)row * nwg + blockIdx.x] = bi;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dec_embed_kernel(const int64_t* __restrict__ tok, const bf16* __restrict__ E,
                                                        const float* __restrict__ pos, const int* __restrict__ pos_ptr,
                                                        float* __restrict__ x, int B, int d, int V) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const int t = *pos_ptr;
  int64_t id = tok[row];
  id = id < 0 ? 0 : (id >= V ? V - 1 : id);
  for (int c = lane; c < d / 8; c += 64) {
    const bf16x8 e = *(const bf16x8*)(E + id * d + c * 8);
    const f32x4 p0 = *(const f32x4*)(pos + (int64_t)t * d + c * 8), p1 = *(const f32x4*)(pos + (int64_t)t * d + c * 8 + 4);
    f32x4 o0, o1;
#pragma unroll
    for (int i = 0; i < 4; ++i) { o0[i] = (float)e[i] + p0[i]; o1[i] = (float)e[4 + i] + p1[i]; }
    *(f32x4*)(x + (int64_t)row * d + c * 8) = o0;
    *(f32x4*)(x + (int64_t)row * d + c * 8 + 4) = o1;
  }
}

// ---------------------------------------------------------------------------------------------------------------
constexpr int DA_MAXK = 4096;

__device__ __forceinline__ float block_reduce(float v, float* scratch, bool is_max) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  float r = scratch[0];
  for (int w = 1; w < 4; ++w) r = is_max ? fmaxf(r, scratch[w]) : r + scratch[w];
  return r;
}

__global__ __launch_bounds__(256) void dec_attn_kernel(const float* __restrict__ q, const bf16* __restrict__ Kc,
                                                       const bf16* __restrict__ Vc, int64_t sb, int64_t sh, int64_t sk,
                                                       const int* __restrict__ lk_ptr, int lk_add, float* __restrict__ out,
                                                       int H) {
  __shared__ float sc[DA_MAXK];
  __shared__ float scratch[4];
  __shared__ float part[4 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int Lk = (lk_ptr ? *lk_ptr : 0) + lk_add;
  const int c = lane & 7, ks = lane >> 3;
  const float* qp = q + ((int64_t)b * H + h) * 64 + c * 8;
  const f32x4 q0 = *(const f32x4*)qp, q1 = *(const f32x4*)(qp + 4);
  const bf16* kb = Kc + b * sb + h * sh + c * 8;
  const bf16* vb = Vc + b * sb + h * sh + c * 8;

  // ---- scores: 8 lanes per key, 32 keys per workgroup pass
  for (int k0 = 0; k0 < Lk; k0 += 128) {
    bf16x8 kv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int key = k0 + u * 32 + wave * 8 + ks;
      key = key < Lk ? key : Lk - 1;
      kv[u] = *(const bf16x8*)(kb + key * sk);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) s = fmaf(q0[i], (float)kv[u][i], s);
#pragma unroll
      for (int i = 0; i < 4; ++i) s = fmaf(q1[i], (float)kv[u][4 + i], s);
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      const int key = k0 + u * 32 + wave * 8 + ks;
      if (c == 0 && key < Lk) sc[key] = s * 0.125f;
    }
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int k = tid; k < Lk; k += 256) mx = fmaxf(mx, sc[k]);
  mx = block_reduce(mx, scratch, true);
  float sum = 0.f;
  for (int k = tid; k < Lk; k += 256) {
    const float p = expf(sc[k] - mx);
    sc[k] = p;
    sum += p;
  }
  sum = block_reduce(sum, scratch, false);  // its barriers also publish the p values

  // ---- P.V
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < Lk; k0 += 128) {
    bf16x8 vv[4];
    float p[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int key = k0 + u * 32 + wave * 8 + ks;
      const int kc = key < Lk ? key : Lk - 1;
      vv[u] = *(const bf16x8*)(vb + kc * sk);
      p[u] = key < Lk ? sc[key] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = fmaf(p[u], (float)vv[u][i], acc[i]);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    acc[i] += __shfl_xor(acc[i], 8, 64);
    acc[i] += __shfl_xor(acc[i], 16, 64);
    acc[i] += __shfl_xor(acc[i], 32, 64);
  }
  if (ks == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) part[wave * 64 + c * 8 + i] = acc[i];
  }
  __syncthreads();
  if (tid < 64) {
    const float o = (part[tid] + part[64 + tid]) + (part[128 + tid] + part[192 + tid]);
    out[((int64_t)b * H + h) * 64 + tid] = o / sum;
  }
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dec_argmax_reduce_kernel(const float* __restrict__ ws_val, const int* __restrict__ ws_idx,
                                                                int nwg, const int* __restrict__ pos_ptr,
                                                                const int64_t* __restrict__ prompt, int P,
                                                                int64_t* __restrict__ tok_cur, int64_t* __restrict__ tokens_out,
                                                                int Ttot, float* __restrict__ margin_out) {
  __shared__ float sv[4];
  __shared__ int si[4];
  __shared__ float s2[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float bv = -INFINITY, second = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = tid; i < nwg; i += 256) {
    const float v = ws_val[(int64_t)b * nwg + i];
    const int ix = ws_idx[(int64_t)b * nwg + i];
    if (v > bv || (v == bv && ix < bi)) { second = fmaxf(second, bv); bv = v; bi = ix; }
    else second = fmaxf(second, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    const float o2 = __shfl_xor(second, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) { second = fmaxf(fmaxf(second, o2), bv); bv = ov; bi = oi; }
    else second = fmaxf(fmaxf(second, o2), ov);
  }
  if (lane == 0) { sv[wave] = bv; si[wave] = bi; s2[wave] = second; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w) {
      if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { second = fmaxf(fmaxf(second, s2[w]), bv); bv = sv[w]; bi = si[w]; }
      else second = fmaxf(fmaxf(second, s2[w]), sv[w]);
    }
    const int t = *pos_ptr;  // the token just consumed sits at position t; this step decides position t + 1
    const int64_t next = (t + 1 < P) ? prompt[(int64_t)b * P + t + 1] : (int64_t)bi;
    tok_cur[b] = next;
    if (t + 1 < Ttot) tokens_out[(int64_t)b * Ttot + t + 1] = next;
    // NOTE: `second` is the runner-up among per-tile winners, i.e. a lower bound on the true top1 - top2 margin's
    // complement; it is diagnostic only (tests classify near-ties with it).
    if (margin_out && t + 1 < Ttot) margin_out[(int64_t)b * Ttot + t + 1] = bv - second;
  }
}

__global__ void dec_advance_kernel(int* pos_ptr) { *pos_ptr += 1; }

}  // namespace

// =================================================================================================================
extern "C" int pm_dec_embed(const int64_t* tok_cur, const void* emb, const float* pos, const int32_t* pos_ptr, float* x,
                            int64_t B, int64_t d, int64_t V, void* stream) {
  if (!tok_cur || !emb || !pos || !pos_ptr || !x || B <= 0 || d <= 0 || V <= 0) return PM_EINVAL;
  if (d % 8) return PM_EUNSUPPORTED;
  if (((uintptr_t)emb | (uintptr_t)pos | (uintptr_t)x) & 15) return PM_EALIGN;
  hipLaunchKernelGGL(dec_embed_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, tok_cur,
                     (const bf16*)emb, pos, (const int*)pos_ptr, x, (int)B, (int)d, (int)V);
  PM_CHECK_LAUNCH();
  return PM_OK;
}

template <int ACT, int FT>
static void dl_launch(int mt, dim3 grid, hipStream_t st, const float* x, int ldx, const float* gamma, const float* beta,
                      float eps, const bf16* W, int64_t ldw, const float* bias, const float* resid, int ldr, float* out,
                      int ldo, int M, int N, int K, int mode, bf16* kc, bf16* vc, int inner, int H, int Tmax,
                      const int* pos_ptr, float* wv, int* wi, int nwg) {
#define PM_DL(T)                                                                                                      \
  hipLaunchKernelGGL((dec_linear_kernel<ACT, T, FT>), grid, dim3(256), 0, st, x, ldx, gamma, beta, eps, W, ldw, bias, resid, \
                     ldr, out, ldo, M, N, K, mode, kc, vc, inner, H, Tmax, pos_ptr, wv, wi, nwg)
  if (mt == 1) PM_DL(1);
  else if (mt == 2) PM_DL(2);
  else if (mt == 3) PM_DL(3);
  else PM_DL(4);
#undef PM_DL
}

extern "C" int pm_dec_linear(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, const void* w,
                             int64_t ldw, const float* bias, const float* resid, int64_t ldr, float* out, int64_t ldo,
                             int64_t M, int64_t N, int64_t K, int act, int mode, void* kcache, void* vcache,
                             int64_t inner, int64_t H, int64_t Tmax, const int32_t* pos_ptr, float* ws_val,
                             int32_t* ws_idx, void* stream) {
  if (!x || !w || M <= 0 || N <= 0 || K <= 0) return PM_EINVAL;
  if (M > 64 || K % 32) return PM_EUNSUPPORTED;
  if (ldx < K || ldw < K || ldx % 4 || ldw % 8) return PM_EALIGN;
  if (((uintptr_t)x | (uintptr_t)w) & 15) return PM_EALIGN;
  if ((gamma == nullptr) != (beta == nullptr)) return PM_EINVAL;
  if (gamma && (((uintptr_t)gamma | (uintptr_t)beta) & 15)) return PM_EALIGN;
  if (mode == DL_PLAIN) {
    if (!out || ldo < N || (resid && ldr < N)) return PM_EINVAL;
  } else if (mode == DL_QKV) {
    if (!out || !kcache || !vcache || !pos_ptr || inner <= 0 || N != 3 * inner || inner != H * 64 || ldo < inner || Tmax <= 0)
      return PM_EINVAL;
  } else if (mode == DL_ARGMAX) {
    if (!ws_val || !ws_idx) return PM_EINVAL;
  } else {
    return PM_EINVAL;
  }
  if (act != PM_ACT_NONE && act != PM_ACT_GELU) return PM_EUNSUPPORTED;
  const int ft = mode == DL_ARGMAX ? 4 : 1;
  const int nwg = (int)((N + DL_FEATS * ft - 1) / (DL_FEATS * ft));
  const int mt = (int)((M + 15) / 16);
  hipStream_t st = (hipStream_t)stream;
  if (mode == DL_ARGMAX)
    dl_launch<PM_ACT_NONE, 4>(mt, dim3(nwg), st, x, (int)ldx, gamma, beta, eps, (const bf16*)w, ldw, bias, resid, (int)ldr, out,
                              (int)ldo, (int)M, (int)N, (int)K, mode, (bf16*)kcache, (bf16*)vcache, (int)inner, (int)H,
                              (int)Tmax, (const int*)pos_ptr, ws_val, (int*)ws_idx, nwg);
  else if (act == PM_ACT_GELU)
    dl_launch<PM_ACT_GELU, 1>(mt, dim3(nwg), st, x, (int)ldx, gamma, beta, eps, (const bf16*)w, ldw, bias, resid, (int)ldr, out,
                           (int)ldo, (int)M, (int)N, (int)K, mode, (bf16*)kcache, (bf16*)vcache, (int)inner, (int)H,
                           (int)Tmax, (const int*)pos_ptr, ws_val, (int*)ws_idx, nwg);
  else
    dl_launch<PM_ACT_NONE, 1>(mt, dim3(nwg), st, x, (int)ldx, gamma, beta, eps, (const bf16*)w, ldw, bias, resid, (int)ldr, out,
                           (int)ldo, (int)M, (int)N, (int)K, mode, (bf16*)kcache, (bf16*)vcache, (int)inner, (int)H,
                           (int)Tmax, (const int*)pos_ptr, ws_val, (int*)ws_idx, nwg);
  PM_CHECK_LAUNCH();
  return PM_OK;
}

extern "C" int pm_dec_attention(const float* q, const void* kc, const void* vc, int64_t stride_b, int64_t stride_h,
                                int64_t stride_k, const int32_t* lk_ptr, int64_t lk_add, int64_t lk_max, float* out,
                                int64_t B, int64_t H, void* stream) {
  if (!q || !kc || !vc || !out || B <= 0 || H <= 0 || lk_add < 0 || lk_max <= 0) return PM_EINVAL;
  if (lk_max > DA_MAXK) return PM_EUNSUPPORTED;
  if (!lk_ptr && lk_add <= 0) return PM_EINVAL;
  if ((stride_b | stride_h | stride_k) % 8) return PM_EALIGN;
  if (((uintptr_t)q | (uintptr_t)kc | (uintptr_t)vc | (uintptr_t)out) & 15) return PM_EALIGN;
  if (B * H > 0x7fffffff) return PM_EINVAL;
  hipLaunchKernelGGL(dec_attn_kernel, dim3((unsigned)(B * H)), dim3(256), 0, (hipStream_t)stream, q, (const bf16*)kc,
                     (const bf16*)vc, stride_b, stride_h, stride_k, (const int*)lk_ptr, (int)lk_add, out, (int)H);
  PM_CHECK_LAUNCH();
  return PM_OK;
}

extern "C" int pm_dec_argmax_reduce(const float* ws_val, const int32_t* ws_idx, int64_t n_tiles, const int32_t* pos_ptr,
                                    const int64_t* prompt, int64_t P, int64_t* tok_cur, int64_t* tokens_out,
                                    int64_t Ttot, float* margin_out, int64_t B, void* stream) {
  if (!ws_val || !ws_idx || !pos_ptr || !prompt || !tok_cur || !tokens_out || n_tiles <= 0 || P <= 0 || B <= 0 || Ttot < P)
    return PM_EINVAL;
  hipLaunchKernelGGL(dec_argmax_reduce_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, ws_val,
                     (const int*)ws_idx, (int)n_tiles, (const int*)pos_ptr, prompt, (int)P, tok_cur, tokens_out, (int)Ttot,
                     margin_out);
  PM_CHECK_LAUNCH();
  return PM_OK;
}

extern "C" int pm_dec_advance(int32_t* pos_ptr, void* stream) {
  if (!pos_ptr) return PM_EINVAL;
  hipLaunchKernelGGL(dec_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (int*)pos_ptr);
  PM_CHECK_LAUNCH();
  return PM_OK;
}
