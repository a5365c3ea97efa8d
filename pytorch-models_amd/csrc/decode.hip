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
#include <cstdlib>

#include "common.h"

namespace {

constexpr int DL_FEATS = 16;  // output features per workgroup

// Cross-lane exchanges of the fused attention block as DPP modifiers of the add / max itself (v_add_f32 ... quad_perm / row_ror
// / row_half_mirror) wherever the partner lane sits in the same row of 16: hipcc turns EVERY __shfl_xor into ds_bpermute_b32
// - an LDS round trip of ~100 cycles - and this block's critical path is a chain of them (two LayerNorm reductions, the
// projection's 8-lane sums, the scores' 8-lane sums, the softmax's max and sum: ~33 dependent exchanges, ~1.4 us of a
// 10 us block).  Partners 16 and 32 lanes away: gfx950's v_permlane16_swap / v_permlane32_swap (one vector-ALU instruction
// each: with both operands the same value the two results are a lane's own value and its partner's).  No exchange of the step's
// kernels goes through LDS any more.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// lane i + lane i ^ 1, ^ 2 (quad permutes), ^ 4 (mirror of the half row: the other quad's sum, uniform by then): every lane of
// an aligned group of 8 gets the group's sum, added in the order of the xor butterfly (same bits)
__device__ __forceinline__ float sum8_dpp(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1, 0, 3, 2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2, 3, 0, 1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  return v;
}
__device__ __forceinline__ float xor8_dpp(float v) { return dpp_mov<0x128>(v); }  // row_ror:8 = lane i ^ 8
// v[i] (+ | max) v[i ^ 16] and v[i ^ 32]: a + b and max(a, b) do not care which of the two results is the lane's own
__device__ __forceinline__ float add_xor16(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float add_xor32(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float max_xor16(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float max_xor32(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// the better of a lane's (value, index) pair and its partner's 16 / 32 lanes away (higher value, lower index on a tie)
template <int O>
__device__ __forceinline__ void argmax_xor(float& bv, int& bi) {
  static_assert(O == 16 || O == 32, "partners in another row of 16");
  const auto rv = O == 16 ? __builtin_amdgcn_permlane16_swap(__float_as_uint(bv), __float_as_uint(bv), false, false)
                          : __builtin_amdgcn_permlane32_swap(__float_as_uint(bv), __float_as_uint(bv), false, false);
  const auto ri = O == 16 ? __builtin_amdgcn_permlane16_swap((unsigned)bi, (unsigned)bi, false, false)
                          : __builtin_amdgcn_permlane32_swap((unsigned)bi, (unsigned)bi, false, false);
  const float v0 = __uint_as_float(rv[0]), v1 = __uint_as_float(rv[1]);
  const int i0 = (int)ri[0], i1 = (int)ri[1];
  const bool second = v1 > v0 || (v1 == v0 && i1 < i0);
  bv = second ? v1 : v0;
  bi = second ? i1 : i0;
}
__device__ __forceinline__ float dwave_sum(float v) {
  v = sum8_dpp(v);
  v += xor8_dpp(v);
  v = add_xor16(v);
  return add_xor32(v);
}
__device__ __forceinline__ float dwave_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, xor8_dpp(v));
  v = max_xor16(v);
  return max_xor32(v);
}


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

enum { DL_PLAIN = 0, DL_QKV = 1, DL_ARGMAX = 2, DL_PARTS = 3 };

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() carries a fence that waits for EVERY outstanding
// memory operation - also the global loads a kernel has deliberately left in flight (its weight stream) - so a reduction in the
// middle of a kernel would drain them.  This one waits for the LDS traffic alone.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// In-kernel phase stamps of the fused attention block (variant builds only: tools/build_variant.sh stamps -DPM_DF_STAMPS=1;
// tools/chain_stamps.py reads them through pm_debug_df_stamps)
#ifndef PM_DF_STAMPS
#define PM_DF_STAMPS 0
#endif
#if PM_DF_STAMPS
__device__ unsigned long long g_df_stamps[1024 * 16];
#define PM_STAMP(i_)                                                                                  \
  do {                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_df_stamps[blockIdx.x * 16 + (i_)] = wall_clock64();  \
    __builtin_amdgcn_sched_barrier(0);                                                                \
  } while (0)
extern "C" int pm_debug_df_stamps(void* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_df_stamps), sizeof(unsigned long long) * 1024 * 16) == hipSuccess ? 0 : 1;
}
#else
#define PM_STAMP(i_)
#endif

// NSTEP = K steps of 32 per wave held in registers at once.  LayerNorm kernels (K = d_model) keep the wave's WHOLE
// share of x in registers: the row statistics come from those registers (two exchanges through LDS: mean, then the
// centred second moment - the reference's two-pass form), so x is read from memory exactly once and every load of the
// kernel (x, weights) is issued before the first wait: one memory round trip.  The earlier form re-read x for the
// statistics in a scalar loop whose loads each paid a full L2 round trip (~15 us per launch, all latency).
template <int ACT, int MT, int FT, int NSTEP, bool LN>
__global__ __launch_bounds__(256) void dec_linear_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps,
                                                         const bf16* __restrict__ W, int64_t ldw,
                                                         const float* __restrict__ bias, const float* resid, int ldr,
                                                         float* out, int ldo, int M, int N, int K, int mode,
                                                         bf16* __restrict__ kcache, bf16* __restrict__ vcache, int inner,
                                                         int H, int Tmax, const int* __restrict__ pos_ptr,
                                                         float* __restrict__ ws_val, int* __restrict__ ws_idx, int nwg) {
  __shared__ float part[4 * 64];
  constexpr int GBK = 128 * NSTEP;  // the largest K this instantiation serves (host: ksteps <= 4 * NSTEP)
  __shared__ float gb[LN ? 2 * GBK : 2];
  __shared__ __attribute__((aligned(16))) float red[4 * FT * MT * 64 * 4];
  const int tid = threadIdx.x, lane = tid & 63;
  PM_STAMP(0);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = blockIdx.x * (DL_FEATS * FT);
  const int fi = lane & 15, kq = lane >> 4;
  const int ksteps = K >> 5;
  const int rbase = blockIdx.z * (16 * MT);  // row tiles may be spread over gridDim.z workgroups (independent rows)
  // K may be split over gridDim.y workgroups (plain mode, no LayerNorm): part kp takes K steps [ks0, ks1) - every
  // workgroup then reads 1 / gridDim.y of the activations instead of all of them (fc2: 256 KB per workgroup through one
  // CU's L2 path was the kernel's time) - and the parts are combined by the last one to finish (below)
  const int kparts = gridDim.y, kp = blockIdx.y;
  const int ks0 = ksteps * kp / kparts, ks1 = ksteps * (kp + 1) / kparts;

  // ---- the epilogue's operands (bias, residual) are requested before anything else by the wave that will run it: issued
  // where they are used they cost one exposed L2 round trip per launch, ~1 us of a 4-9 us kernel (the residual only while it
  // fits a few registers: out_proj / fc2 shapes)
  const int tpos = (mode == DL_QKV) ? *pos_ptr : 0;  // cache position of the k / v rows written by the epilogue
  constexpr bool PRE_R = FT * MT <= 4;
  constexpr bool PRE_B = FT * MT <= 8;  // the 4 x 4 vocabulary tiles have no registers to spare (and no bias)
  f32x4 bpre[PRE_B ? FT : 1], rpre[PRE_R ? FT : 1][PRE_R ? MT : 1];
  const bool pre_resid = PRE_R && resid && mode == DL_PLAIN;
  if constexpr (PRE_B || PRE_R)
  if (wave == 0) {
#pragma unroll
    for (int f = 0; f < FT; ++f) {
      const int n = n0 + f * 16 + kq * 4;
      if constexpr (PRE_B) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bpre[f][r] = (bias && n + r < N) ? bias[n + r] : 0.f;
      }
      if constexpr (PRE_R) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const int row = rbase + t * 16 + fi;
#pragma unroll
          for (int r = 0; r < 4; ++r) rpre[f][t][r] = (pre_resid && row < M && n + r < N) ? resid[(int64_t)row * ldr + n + r] : 0.f;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }

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
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    int row = rbase + t * 16 + fi;
    row = row < M ? row : M - 1;
    xrow[t] = x + (int64_t)row * ldx + kq * 8;
  }

  // LN kernels: exactly ONE trip for every wave (host guarantees ksteps <= 4*NSTEP), even a wave that owns no K step:
  // the trip contains workgroup barriers
  for (int sb = ks0 + wave, trip = 0; LN ? trip < 1 : sb < ks1; sb += 4 * NSTEP, ++trip) {
    bf16x8 a[NSTEP][FT];
    f32x4 xv[NSTEP][MT][2];
    // LayerNorm kernels: the activations first, then gamma / beta, then the weights - loads return in order and the row
    // statistics need only x, so they run while the weights stream (with LDS-only barriers: see lds_barrier)
#pragma unroll
    for (int u = 0; u < NSTEP; ++u) {
      int s = sb + 4 * u;
      s = s < ks1 ? s : ks1 - 1;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        xv[u][t][0] = *(const f32x4*)(xrow[t] + s * 32);
        xv[u][t][1] = *(const f32x4*)(xrow[t] + s * 32 + 4);
      }
      if constexpr (!LN) {
#pragma unroll
        for (int f = 0; f < FT; ++f) a[u][f] = *(const bf16x8*)(wp[f] + s * 32);
      }
    }
    // gamma / beta of this lane's K positions: in REGISTERS for the small instantiations (requested behind x like any operand:
    // one round trip for everything).  Staged through LDS by a `gb[k] = gamma[k]` loop - still the form of the large ones,
    // which have no registers for it - hipcc waits vmcnt(0) in front of every LDS write: x and a round trip per loop trip
    // BEFORE the weights are even requested (ISA of the fc1 launch: 2 loads, wait, ds_write, 2 loads, wait, ...)
    constexpr bool GREG = LN && NSTEP <= 4;
    f32x4 gv[GREG ? NSTEP : 1][2], bv[GREG ? NSTEP : 1][2];
    if constexpr (GREG) {
#pragma unroll
      for (int u = 0; u < NSTEP; ++u) {
        int s = sb + 4 * u;
        s = s < ks1 ? s : ks1 - 1;
        const int k0 = s * 32 + kq * 8;
        gv[u][0] = *(const f32x4*)(gamma + k0);
        gv[u][1] = *(const f32x4*)(gamma + k0 + 4);
        bv[u][0] = *(const f32x4*)(beta + k0);
        bv[u][1] = *(const f32x4*)(beta + k0 + 4);
      }
    }
    if constexpr (LN) {
      // large instantiations: gamma / beta through registers into LDS - every request of the trip leaves before the first LDS
      // write (the staging loop `gb[k] = gamma[k]` waited vmcnt(0) per trip: K / 256 dependent round trips in front of the weights)
      constexpr int GN = GREG ? 1 : (GBK + 255) / 256;
      float gr[GN], br[GN];
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!GREG) {
#pragma unroll
        for (int j = 0; j < GN; ++j) {
          const int k = tid + 256 * j;
          const int kc = k < K ? k : K - 1;
          gr[j] = gamma[kc];
          br[j] = beta[kc];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < NSTEP; ++u) {
        int s = sb + 4 * u;
        s = s < ks1 ? s : ks1 - 1;
#pragma unroll
        for (int f = 0; f < FT; ++f) a[u][f] = *(const bf16x8*)(wp[f] + s * 32);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!GREG) {
#pragma unroll
        for (int j = 0; j < GN; ++j) {
          const int k = tid + 256 * j;
          if (k < K) { gb[k] = gr[j]; gb[GBK + k] = br[j]; }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    } else {
      // every load of the trip is REQUESTED before the first is used: without this pin hipcc fuses the request loop with the
      // compute loop below (whose steps end in a conditional break) and sinks each K step's loads to their use - four dependent
      // memory round trips per trip instead of one (seen in the ISA: load x, load w, wait, 3 MFMAs, branch, load ...)
      __builtin_amdgcn_sched_barrier(0);
    }
    PM_STAMP(1);
    float mean[MT], rstd[MT];
    if constexpr (LN) {
      // ---- mean: lane partial over its elements, then over the 4 kq lanes, then over the 4 waves through LDS
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        float sm = 0.f;
#pragma unroll
        for (int u = 0; u < NSTEP; ++u)
          if (sb + 4 * u < ks1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) sm += xv[u][t][0][i] + xv[u][t][1][i];
          }
        sm = add_xor16(sm);
        sm = add_xor32(sm);
        if (kq == 0) part[wave * 64 + t * 16 + fi] = sm;
      }
      lds_barrier();
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int r = t * 16 + fi;
        mean[t] = ((part[r] + part[64 + r]) + (part[128 + r] + part[192 + r])) / (float)K;
      }
      lds_barrier();
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        float q = 0.f;
#pragma unroll
        for (int u = 0; u < NSTEP; ++u)
          if (sb + 4 * u < ks1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float d0 = xv[u][t][0][i] - mean[t], d1 = xv[u][t][1][i] - mean[t];
              q = fmaf(d0, d0, q);
              q = fmaf(d1, d1, q);
            }
          }
        q = add_xor16(q);
        q = add_xor32(q);
        if (kq == 0) part[wave * 64 + t * 16 + fi] = q;
      }
      lds_barrier();
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int r = t * 16 + fi;
        rstd[t] = rsqrtf(((part[r] + part[64 + r]) + (part[128 + r] + part[192 + r])) / (float)K + eps);
      }
    }
    PM_STAMP(2);
#pragma unroll
    for (int u = 0; u < NSTEP; ++u) {
      // NO branch around a K step this wave does not own (K not a multiple of 128 NSTEP): its activations become zeros by a
      // select and its MFMAs add nothing.  With `break` here hipcc sank every step's loads into the step's guarded block -
      // load x, load w, wait, 3 MFMAs, branch, load ...: four dependent memory round trips per launch instead of one (ISA of
      // dec_linear_kernel<0, 1, 1, 4, false>, the out_proj / fc2 launches of the decode step)
      // Only the small instantiations (12 load registers sets: the decode step's plain projections at 16 rows x 16 features):
      // at d = 1280 the request-everything-first form needs the registers the occupancy lives on (fc2 part 15.8 -> 21.8 us)
      constexpr bool NOBR = !LN && NSTEP * (2 * MT + FT) <= 12;
      const bool valid = sb + 4 * u < ks1;
      if constexpr (!NOBR) {
        if (!valid) break;
      }
      const int k0 = (sb + 4 * u) * 32 + kq * 8;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = xv[u][t][0][i]; v[4 + i] = xv[u][t][1][i]; }
        if constexpr (GREG) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = (v[i] - mean[t]) * rstd[t] * gv[u][i >> 2][i & 3] + bv[u][i >> 2][i & 3];
        } else if constexpr (LN) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = (v[i] - mean[t]) * rstd[t] * gb[k0 + i] + gb[GBK + k0 + i];
        }
        if constexpr (NOBR) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = valid ? v[i] : 0.f;
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
  PM_STAMP(3);
#pragma unroll
  for (int f = 0; f < FT; ++f)
#pragma unroll
    for (int t = 0; t < MT; ++t) *(f32x4*)(red + (((wave * FT + f) * MT + t) * 64 + lane) * 4) = acc[f][t];
  __syncthreads();
  if (wave != 0) return;
  PM_STAMP(4);

  // D[row = feature 4*kq + r][col = sequence fi]; partial sums added in wave order 0..3
  float best_v[MT];
  int best_i[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) { best_v[t] = -INFINITY; best_i[t] = 0x7fffffff; }
  f32x4 vs[FT][MT];
#pragma unroll
  for (int f = 0; f < FT; ++f)
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      f32x4 v = *(const f32x4*)(red + (((0 * FT + f) * MT + t) * 64 + lane) * 4);
#pragma unroll
      for (int w = 1; w < 4; ++w) v += *(const f32x4*)(red + (((w * FT + f) * MT + t) * 64 + lane) * 4);
      vs[f][t] = v;
    }
  if (mode == DL_PARTS) {
    // K split whose parts are added by the consumer (dec_attn_fused_kernel CHAIN: the next block forms x + bias + parts itself):
    // part kp goes to out + kp * ldr as it stands - no ticket, no second pass, nothing after the store
#pragma unroll
    for (int f = 0; f < FT; ++f) {
      const int n = n0 + f * 16 + kq * 4;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int row = rbase + t * 16 + fi;
        float* o = out + (int64_t)kp * ldr + (int64_t)row * ldo + n;
        if (row < M && n + 3 < N) *(f32x4*)o = vs[f][t];
        else if (row < M) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < N) o[r] = vs[f][t][r];
        }
      }
    }
    PM_STAMP(5);
    return;
  }
  if (kparts > 1) {
    // K split: publish this part's tile (sc1 stores), take a ticket, and only the last part to finish goes on: it adds
    // the parts in part order (deterministic) and runs the epilogue.  Hand-off as in MI355X_MICROARCH.md (first row of
    // the measured table): every byte stored and loaded with sc1, vmcnt(0) before the agent-scope add, the adding wave
    // is the only reader.  ws_val = (tiles * parts * FT * MT * 64) float4 partials, ws_idx = one int32 ticket per tile.
    const int tile_id = blockIdx.x + gridDim.x * blockIdx.z;
    f32x4* my = (f32x4*)ws_val + ((int64_t)(tile_id * kparts + kp) * FT * MT) * 64 + lane;
#pragma unroll
    for (int f = 0; f < FT; ++f)
#pragma unroll
      for (int t = 0; t < MT; ++t) store_sc1_x4(my + (f * MT + t) * 64, vs[f][t]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(ws_idx + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket != kparts - 1) return;
#pragma unroll
    for (int f = 0; f < FT; ++f)
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        // parts in groups of four: the loads of a group are in flight together, one wait, then added in part order
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int q0 = 0; q0 < kparts; q0 += 4) {
          f32x4 pq[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int q = q0 + j < kparts ? q0 + j : kparts - 1;
            pq[j] = load_sc1_x4_async((const f32x4*)ws_val + ((int64_t)(tile_id * kparts + q) * FT * MT + f * MT + t) * 64 + lane);
          }
          wait_loads(pq[0], pq[1], pq[2], pq[3]);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (q0 + j < kparts) v += pq[j];
        }
        vs[f][t] = v;
      }
    if (lane == 0) __hip_atomic_store(ws_idx + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int f = 0; f < FT; ++f) {
    const int n = n0 + f * 16 + kq * 4;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const f32x4 v = vs[f][t];
      const int row = rbase + t * 16 + fi;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nn = n + r;
        if (nn >= N) continue;
        float e = v[r] + (PRE_B ? bpre[PRE_B ? f : 0][r] : (bias ? bias[nn] : 0.f));
        e = apply_act<ACT, true>(e);
        if (mode == DL_ARGMAX) {
          if (e > best_v[t]) { best_v[t] = e; best_i[t] = nn; }  // ascending nn: strict > keeps the lowest index
          continue;
        }
        if (row >= M) continue;
        if (mode == DL_PLAIN) {
          if constexpr (PRE_R) e += rpre[f][t][r];
          else if (resid) e += resid[(int64_t)row * ldr + nn];
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
  PM_STAMP(5);
  if (mode == DL_ARGMAX) {
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      float bv = best_v[t];
      int bi = best_i[t];
      argmax_xor<16>(bv, bi);  // the 4 lanes sharing a sequence (kq = 0..3)
      argmax_xor<32>(bv, bi);
      const int row = rbase + t * 16 + fi;
      if (kq == 0 && row < M) {
        ws_val[(int64_t)row * nwg + blockIdx.x] = bv;
        ws_idx[(int64_t)row * nwg + blockIdx.x] = bi;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The vocabulary projection of a step (final LayerNorm + logits + per-tile arg-max) as a PERSISTENT kernel: the step's fattest
// launch after the cross-attention blocks.  As dec_linear_kernel<.., ARGMAX> it ran ceil(V / 64) workgroups that EACH loaded
// and normalised all of x (64 KB of fp32 for 32 rows of 512: as many bytes as the workgroup's weight rows; 52 + 53 MB per
// launch through the CUs' load paths, 26.9 us at Whisper-base).  Here a workgroup normalises x ONCE, keeps its three bf16
// terms in registers and walks vocabulary tiles: 32 weight rows per inner step, the next step's rows requested before this
// step's MFMAs, partial sums of the four waves (each owns every fourth K step, as in dec_linear_kernel) through a double-buffered
// LDS area - one barrier per step, wave 0 reduces step s while everyone computes s + 1.  Every sum is formed in
// dec_linear_kernel's order, so the logits and the (max, index) pairs are bit-identical to that kernel's.
// NW waves share a row's K steps (wave w: steps w, w + NW, ...), NF 16-feature tiles per inner step.  <4, 2>: two 256-thread
// workgroups per CU (round 2).  <8, 4> (round 3, the default): ONE 512-thread workgroup per CU - the x rows are loaded,
// normalised and split once per CU instead of twice (stamps: 8.3 of the kernel's 20 us went by before its first MFMA, 128 KB of
// x and ~2.3 us of VALU per CU), a step is a whole 64-feature arg-max tile.
template <int MT, int NW, int NF>
__global__ __launch_bounds__(64 * NW) void dec_logits_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps,
                                                         const bf16* __restrict__ W, int64_t ldw,
                                                         const float* __restrict__ bias, int M, int N, int K,
                                                         float* __restrict__ ws_val, int* __restrict__ ws_idx, int ntiles,
                                                         int halves) {
  constexpr int NSTEP = 16 / NW;  // K <= 512: at most 16 K steps of 32
  __shared__ float part[NW * 64];
  __shared__ __attribute__((aligned(16))) float red[2][NW * NF * MT * 64 * 4];
  __shared__ float tbv[2][NF][MT * 16];  // NW == 8: per step and feature tile, the rows' (max, index)
  __shared__ int tbi[2][NF][MT * 16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 15, kq = lane >> 4;
  const int ksteps = K >> 5;

  // ---- x: this wave's K steps of every row, LayerNorm (the reference's two-pass statistics), three bf16 terms
  f32x4 xv[NSTEP][MT][2];
#pragma unroll
  for (int u = 0; u < NSTEP; ++u) {
    int s = wave + NW * u;
    s = s < ksteps ? s : ksteps - 1;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      int row = t * 16 + fi;
      row = row < M ? row : M - 1;
      const float* xr = x + (int64_t)row * ldx + kq * 8 + s * 32;
      xv[u][t][0] = *(const f32x4*)xr;
      xv[u][t][1] = *(const f32x4*)(xr + 4);
    }
  }
  // gamma / beta of this lane's K positions, in registers (see dec_linear_kernel: the LDS staging loop made hipcc wait for
  // everything in flight - here the first weight rows, from HBM - in front of the LayerNorm)
  f32x4 gv[NSTEP][2], bv[NSTEP][2];
#pragma unroll
  for (int u = 0; u < NSTEP; ++u) {
    int s = wave + NW * u;
    s = s < ksteps ? s : ksteps - 1;
    const int k0 = s * 32 + kq * 8;
    gv[u][0] = *(const f32x4*)(gamma + k0);
    gv[u][1] = *(const f32x4*)(gamma + k0 + 4);
    bv[u][0] = *(const f32x4*)(beta + k0);
    bv[u][1] = *(const f32x4*)(beta + k0 + 4);
  }
  // ---- vocabulary tiles: inner step q = (tile, half): 32 weight rows.  The first step's rows are requested HERE, right behind x
  // (loads return in order: x first) and in front of the LayerNorm, whose barriers no load crosses: requested behind it
  // the kernel's first microseconds moved no weight bytes
  const int total = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x * halves;  // inner steps of this workgroup
  const int tile_feats = halves * 16 * NF;
  auto wrow_ptr = [&](int q, int f) {
    const int tile = blockIdx.x + (q / halves) * gridDim.x, hs = q - (q / halves) * halves;
    int n = tile * tile_feats + hs * 16 * NF + f * 16 + fi;
    n = n < N ? n : N - 1;
    return W + (int64_t)n * ldw + kq * 8;
  };
  bf16x8 a[NSTEP][NF], an[NSTEP][NF];
  if (total > 0) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const bf16* wp = wrow_ptr(0, f);
#pragma unroll
      for (int u = 0; u < NSTEP; ++u) {
        int s = wave + NW * u;
        s = s < ksteps ? s : ksteps - 1;
        a[u][f] = *(const bf16x8*)(wp + s * 32);
      }
    }
  }
  auto part_sum = [&](int r) {  // over the waves, pairwise (NW = 4: the order of the 4-wave kernels)
    float h0 = (part[r] + part[64 + r]) + (part[128 + r] + part[192 + r]);
    if constexpr (NW == 8) h0 += (part[256 + r] + part[320 + r]) + (part[384 + r] + part[448 + r]);
    return h0;
  };
  float mean[MT], rstd[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    float sm = 0.f;
#pragma unroll
    for (int u = 0; u < NSTEP; ++u)
      if (wave + NW * u < ksteps) {
#pragma unroll
        for (int i = 0; i < 4; ++i) sm += xv[u][t][0][i] + xv[u][t][1][i];
      }
    sm = add_xor16(sm);
    sm = add_xor32(sm);
    if (kq == 0) part[wave * 64 + t * 16 + fi] = sm;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int r = t * 16 + fi;
    mean[t] = part_sum(r) / (float)K;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < NSTEP; ++u)
      if (wave + NW * u < ksteps) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float d0 = xv[u][t][0][i] - mean[t], d1 = xv[u][t][1][i] - mean[t];
          q = fmaf(d0, d0, q);
          q = fmaf(d1, d1, q);
        }
      }
    q = add_xor16(q);
    q = add_xor32(q);
    if (kq == 0) part[wave * 64 + t * 16 + fi] = q;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int r = t * 16 + fi;
    rstd[t] = rsqrtf(part_sum(r) / (float)K + eps);
  }
  bf16x8 xh[NSTEP][MT], xm[NSTEP][MT], xl[NSTEP][MT];
#pragma unroll
  for (int u = 0; u < NSTEP; ++u) {
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = xv[u][t][0][i]; v[4 + i] = xv[u][t][1][i]; }
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (v[i] - mean[t]) * rstd[t] * gv[u][i >> 2][i & 3] + bv[u][i >> 2][i & 3];
      split3(v, xh[u][t], xm[u][t], xl[u][t]);
    }
  }

  // ---- vocabulary tiles (set up in front of the LayerNorm: the first step's weight rows are already on their way)
  float best_v[MT];
  int best_i[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) { best_v[t] = -INFINITY; best_i[t] = 0x7fffffff; }
#pragma unroll 1
  for (int q = 0; q < total; ++q) {
    if (q + 1 < total) {  // the next step's weight rows fly during this step's MFMAs
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const bf16* wp = wrow_ptr(q + 1, f);
#pragma unroll
        for (int u = 0; u < NSTEP; ++u) {
          int s = wave + NW * u;
          s = s < ksteps ? s : ksteps - 1;
          an[u][f] = *(const bf16x8*)(wp + s * 32);
        }
      }
    }
    f32x4 acc[NF][MT];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NSTEP; ++u) {
      if (wave + NW * u >= ksteps) break;
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          acc[f][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][f], xh[u][t], acc[f][t], 0, 0, 0);
          acc[f][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][f], xm[u][t], acc[f][t], 0, 0, 0);
          acc[f][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][f], xl[u][t], acc[f][t], 0, 0, 0);
        }
    }
    float* rb = red[q & 1];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int t = 0; t < MT; ++t) *(f32x4*)(rb + (((wave * NF + f) * MT + t) * 64 + lane) * 4) = acc[f][t];
    lds_barrier();  // (not __syncthreads: the next step's weight loads stay in flight)
    if constexpr (NW == 8) {
      // the step is a whole arg-max tile: wave f < NF adds the 8 waves' partial sums of feature tile f and leaves its rows'
      // (max, index); wave 0 picks the tile's winner among the NF one step later (behind that step's barrier) - one wave
      // adding all 64 partial tiles of a step (the <4, 2> form's way) left the other seven waiting at the next barrier
      if (wave == 0 && q > 0) {
        const int tile = blockIdx.x + (q - 1) * gridDim.x;
        if (lane < MT * 16 && lane < M) {
          float bv = tbv[(q - 1) & 1][0][lane];
          int bi = tbi[(q - 1) & 1][0][lane];
#pragma unroll
          for (int f = 1; f < NF; ++f) {
            const float ov = tbv[(q - 1) & 1][f][lane];
            if (ov > bv) { bv = ov; bi = tbi[(q - 1) & 1][f][lane]; }  // ascending features: strict > keeps the lowest index
          }
          ws_val[(int64_t)lane * ntiles + tile] = bv;
          ws_idx[(int64_t)lane * ntiles + tile] = bi;
        }
      }
      if (wave < NF) {
        const int tile = blockIdx.x + q * gridDim.x, f = wave;
        const int n = tile * tile_feats + f * 16 + kq * 4;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          f32x4 v = *(const f32x4*)(rb + (((0 * NF + f) * MT + t) * 64 + lane) * 4);
#pragma unroll
          for (int w = 1; w < NW; ++w) v += *(const f32x4*)(rb + (((w * NF + f) * MT + t) * 64 + lane) * 4);
          float bv = -INFINITY;
          int bi = 0x7fffffff;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int nn = n + r;
            if (nn >= N) continue;
            const float e = v[r] + (bias ? bias[nn] : 0.f);
            if (e > bv) { bv = e; bi = nn; }
          }
          argmax_xor<16>(bv, bi);  // the 4 lanes sharing a sequence (kq = 0..3)
          argmax_xor<32>(bv, bi);
          if (kq == 0) { tbv[q & 1][f][t * 16 + fi] = bv; tbi[q & 1][f][t * 16 + fi] = bi; }
        }
      }
    } else
    if (wave == 0) {
      const int tile = blockIdx.x + (q / halves) * gridDim.x, hs = q - (q / halves) * halves;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int n = tile * tile_feats + hs * 16 * NF + f * 16 + kq * 4;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          f32x4 v = *(const f32x4*)(rb + (((0 * NF + f) * MT + t) * 64 + lane) * 4);
#pragma unroll
          for (int w = 1; w < NW; ++w) v += *(const f32x4*)(rb + (((w * NF + f) * MT + t) * 64 + lane) * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int nn = n + r;
            if (nn >= N) continue;
            const float e = v[r] + (bias ? bias[nn] : 0.f);
            if (e > best_v[t]) { best_v[t] = e; best_i[t] = nn; }  // ascending nn: strict > keeps the lowest index
          }
        }
      }
      if (hs == halves - 1) {  // the arg-max tile is complete: winner over the 4 lanes sharing a sequence, publish, reset
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          float bv = best_v[t];
          int bi = best_i[t];
          argmax_xor<16>(bv, bi);  // the 4 lanes sharing a sequence (kq = 0..3)
          argmax_xor<32>(bv, bi);
          const int row = t * 16 + fi;
          if (kq == 0 && row < M) {
            ws_val[(int64_t)row * ntiles + tile] = bv;
            ws_idx[(int64_t)row * ntiles + tile] = bi;
          }
          best_v[t] = -INFINITY;
          best_i[t] = 0x7fffffff;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < NSTEP; ++u)
#pragma unroll
      for (int f = 0; f < NF; ++f) a[u][f] = an[u][f];
  }
  if constexpr (NW == 8) {
    if (total > 0) {  // the last tile's winner
      lds_barrier();
      if (wave == 0 && lane < MT * 16 && lane < M) {
        const int q = total - 1, tile = blockIdx.x + q * gridDim.x;
        float bv = tbv[q & 1][0][lane];
        int bi = tbi[q & 1][0][lane];
#pragma unroll
        for (int f = 1; f < NF; ++f) {
          const float ov = tbv[q & 1][f][lane];
          if (ov > bv) { bv = ov; bi = tbi[q & 1][f][lane]; }
        }
        ws_val[(int64_t)lane * ntiles + tile] = bv;
        ws_idx[(int64_t)lane * ntiles + tile] = bi;
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
      s = sum8_dpp(s);
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
    acc[i] += xor8_dpp(acc[i]);
    acc[i] = add_xor16(acc[i]);
    acc[i] = add_xor32(acc[i]);
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
// Fused attention block of the decode step: one 512-thread workgroup per (sequence b, head h) does
//   LayerNorm(x[b]) -> q_h (cross) or q_h, k_h, v_h (self; k_h, v_h rounded to bf16 and appended to the cache at t)
//   -> softmax(q_h K_h^T / 8) V_h over the cache (self: t + 1 keys, the new one taken from LDS) or over the packed
//   cross K/V (1500 keys) -> att[b, h*64 : h*64 + 64].
// The cross K/V is read once per step by exactly one CU: it is loaded NON-TEMPORALLY so that the 0.8 GB per step of
// streamed K/V does not evict the decoder's weights (120 MB + 53 MB of embeddings) from the 256 MiB Infinity Cache.
// This replaces the [LN + projection] dec_linear launch in front of each attention (2 of 8 launches per layer): the
// projection for one (b, h) is a 64 x d (or 192 x d) GEMV whose weights sit in L2, shared by the 32 sequences.
// The K/V stream is what bounds the cross block (HBM): 8 waves x 8 x 16 B per lane = 64 KiB in flight per CU.
#ifndef PM_CROSS_NKU
#define PM_CROSS_NKU 8
#endif
constexpr int DF_THREADS = 512, DF_WAVES = 8;
#ifndef PM_CROSS_DB
#define PM_CROSS_DB 1
#endif
#ifndef PM_CHAIN_ABL
#define PM_CHAIN_ABL 0  // ablation builds of the chain's OUT side (tools/chain_bench.py): 1 no W_o loads, 2 one row of eight, 3 no store
#endif

// Block-wide sum / max over 8 waves through ONE barrier: every call site owns its 8-float slot of the scratch array, so
// no barrier is needed to protect the slot's previous use (the kernel runs each reduction once).
__device__ __forceinline__ float block_reduce8(float v, float* slot, bool is_max) {
  v = is_max ? dwave_max(v) : dwave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) slot[wave] = v;
  __syncthreads();
  float r = slot[0];
#pragma unroll
  for (int w = 1; w < DF_WAVES; ++w) r = is_max ? fmaxf(r, slot[w]) : r + slot[w];
  return r;
}

// KT = the element type of the K / V caches: bf16 (the throughput path: k, v rounded once when cached) or float (the
// reference-accuracy path, Whisper.generate(exact=True): nothing is rounded anywhere, twice the bytes per key).
typedef __attribute__((ext_vector_type(8))) float f32x8;
template <typename KT> struct KV8;
template <> struct KV8<bf16> { typedef bf16x8 type; };
template <> struct KV8<float> { typedef f32x8 type; };

// CHAIN: the block takes part in the step's deferred sums, on both sides, so that two launches per layer and the K-split's
// ticket disappear from the chain.
//  * IN: the residual stream row arrives as  x[b] + x_bias + sum_p x_parts[p][b]  (p in order: a fixed order, so nothing depends
//    on timing) - the producer was a K-split projection that left its parts (pm_dec_linear_kparts: fc2 of the previous layer)
//    or the previous attention block's per-head projection partials (below).  Every (b, h) workgroup forms the sum itself
//    (n_parts + 2 loads in flight together instead of one: no extra round trip); workgroup (b, 0) also writes the row to
//    x_out, which must not alias x (the other heads of b may still be reading it).
//  * OUT (Wo != null): instead of its 64 attention outputs the workgroup writes their product with its 64 columns of W_o
//    (d x 64, requested before the softmax: long landed when used) as d partial sums to head_parts[b][h][:]; the consumer
//    adds the heads in order, with W_o's bias and the residual.
// Measured first as "the last of a sequence's heads to finish adds them" (agent-scope ticket): the store -> ticket -> load
// round trips cost the block 5-6 us, as much as the launch they replaced.
template <bool SELF, int NCH, typename KT, bool CHAIN>
__global__ __launch_bounds__(DF_THREADS) void dec_attn_fused_kernel(
    const float* x, int d, const float* gamma, const float* beta, float eps,
    const bf16* Wp, const float* bp,  // SELF: packed [q|k|v] (3*inner, d); cross: q (inner, d)
    KT* Kc, KT* Vc, int64_t sb, int64_t sh, int64_t sk, const int* __restrict__ pos_ptr, int lk_const,
    float* __restrict__ out, int H,
    // CHAIN only:
    const float* xparts, int np, int64_t xpstride, int64_t xprow, const float* xbias,
    float* __restrict__ xout, const bf16* __restrict__ Wo, float* __restrict__ hparts) {
  typedef typename KV8<KT>::type kv8;
  __shared__ float sc[DA_MAXK];
  __shared__ float xn[1280];
  __shared__ float qkv[192];
  __shared__ float scratch[4 * DF_WAVES];  // one slot per reduction: mean, variance, max, sum
  __shared__ float part[DF_WAVES * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int inner = H * 64;
  const int tpos = SELF ? *pos_ptr : 0;
  const int Lk = SELF ? tpos + 1 : lk_const;

  // ---- everything small that the chain LayerNorm -> projection needs is requested FIRST, in the order of its use: loads return
  // in order, so whatever is issued ahead of the row (weights, the K stream) delays the LayerNorm, and a gamma / beta / bias
  // load issued where it is used exposes one L2 latency each (measured: the block spent ~2 us of its 10-25 us waiting on them)
  PM_STAMP(0);
  const float* xr = x + (int64_t)b * d;
  // The block's first requests are BRANCH-FREE (clamped indices, values dropped by selects) and their sources carry NO
  // __restrict__: one basic block, so hipcc's waits in front of the LayerNorm are exact counts (with a block per `cond ? load :
  // 0` it waited vmcnt(12): half the projection weights requested BEHIND the row), and loads that may alias the kernel's stores
  // keep their place relative to a sched_barrier - read-only __restrict__ loads float past it when the DAG is linearised, and the
  // parts ended up behind the 128 KB of weights and keys.
  constexpr int XR = (NCH * 64 + DF_THREADS - 1) / DF_THREADS;  // row elements per thread of this instantiation
  float xe[XR], ge[XR], be[XR];
#pragma unroll
  for (int i = 0; i < XR; ++i) {
    const int k = tid + i * DF_THREADS;
    xe[i] = xr[k < d ? k : d - 1];
  }
  // CHAIN, IN side: the first eight parts and the bias are REQUESTED here, behind the row, and added only after the projection
  // weights and the first K passes have been requested too (below): consumed here they held every later request back by one
  // memory round trip (+0.6-1.0 us per block, measured)
  constexpr int XI = CHAIN ? (NCH * 64 + DF_THREADS - 1) / DF_THREADS : 1;  // row elements per thread of this instantiation
  float xb[XI], pv[8][XI];
  if constexpr (CHAIN) {
    // BRANCH-FREE requests (clamped indices, a readable stand-in row where there are no parts): written as
    // `cond ? load : 0` hipcc put each load in a block of its own and sank the first addition of the sum into the first
    // part's block - `global_load; s_waitcnt vmcnt(0); v_add` right here, i.e. the row, the bias and part 0 waited for
    // BEFORE the remaining parts, the weights and the K stream were even requested (the +1 us of the IN side)
    const bool has = np > 0;
    const float* pr = has ? xparts + (int64_t)b * xprow : xr;
    const float* bsrc = (has && xbias) ? xbias : xr;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int k = tid + i * DF_THREADS;
      const int kc = k < d ? k : d - 1;
      xb[i] = bsrc[kc];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int pp = j < np ? j : (has ? np - 1 : 0);
        pv[j][i] = pr[pp * xpstride + kc];
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);  // the row and its parts first: what the LayerNorm waits for
#pragma unroll
  for (int i = 0; i < XR; ++i) {
    const int k = tid + i * DF_THREADS;
    ge[i] = gamma[k < d ? k : d - 1];
    be[i] = beta[k < d ? k : d - 1];
  }
  float bpe[SELF ? 3 : 1];
#pragma unroll
  for (int o = 0; o < (SELF ? 3 : 1); ++o) {
    const float bvl = (bp ? bp : gamma)[bp ? o * inner + h * 64 + (tid >> 3) : 0];
    bpe[o] = bp ? bvl : 0.f;
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- projection weights: 8 lanes per output row, lane p owns 16-byte chunks p, p+8, ...
  const int prow = tid >> 3, pl = tid & 7;
  const int nch = d >> 6;  // chunks per lane (d % 64 == 0)
  // NCH = compile-time bound on nch; all projections' weights are requested up front unless that would not fit the
  // register file (self block at d_model 1280: one projection at a time)
  constexpr bool UPFRONT = !(SELF && NCH > 16);
  bf16x8 wv[UPFRONT ? (SELF ? 3 : 1) : 1][NCH];
  if constexpr (UPFRONT) {
#pragma unroll
    for (int o = 0; o < (SELF ? 3 : 1); ++o) {
      const bf16* wr = Wp + ((int64_t)o * inner + h * 64 + prow) * d + pl * 8;
#pragma unroll
      for (int i = 0; i < NCH; ++i)
        wv[o][i] = *(const bf16x8*)(wr + (i < nch ? i : nch - 1) * 64);  // (chunks past d: a repeat, multiplied by zeros below)
    }
  }
  // ---- the K stream does not depend on q: request its first NKU passes now - AFTER the projection weights, so the
  // weights are not queued behind it - and let HBM stream while the LayerNorm and the projection run.
  constexpr int NKU = SELF ? 4 : PM_CROSS_NKU;
  const int c = lane & 7, ks = lane >> 3;
  const KT* kb = Kc + b * sb + h * sh + c * 8;
  const KT* vb = Vc + b * sb + h * sh + c * 8;
  const int Lc = SELF ? Lk - 1 : Lk;  // keys read from memory (self: the newest one comes from LDS)
  kv8 kv[NKU];
#pragma unroll
  for (int u = 0; u < NKU; ++u) {
    int key = u * 64 + wave * 8 + ks;
    key = key < Lc ? key : (Lc > 0 ? Lc - 1 : 0);
    kv[u] = SELF ? *(const kv8*)(kb + key * sk) : __builtin_nontemporal_load((const kv8*)(kb + key * sk));
  }
  if constexpr (CHAIN) {
    if (np > 0) {
      __builtin_amdgcn_sched_barrier(0);  // (keep the requests above where they are)
      float sp[XI];
#pragma unroll
      for (int i = 0; i < XI; ++i) sp[i] = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < np) {
#pragma unroll
          for (int i = 0; i < XI; ++i) sp[i] += pv[j][i];
        }
      const float* pr = xparts + (int64_t)b * xprow;
      for (int p0 = 8; p0 < np; p0 += 8) {  // more than eight parts (20 heads at d_model 1280): further groups, in part order
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int pp = p0 + j < np ? p0 + j : np - 1;
#pragma unroll
          for (int i = 0; i < XI; ++i) {
            const int k = tid + i * DF_THREADS;
            pv[j][i] = pr[pp * xpstride + (k < d ? k : d - 1)];
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (p0 + j < np) {
#pragma unroll
            for (int i = 0; i < XI; ++i) sp[i] += pv[j][i];
          }
      }
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        const int k = tid + i * DF_THREADS;
        xe[i] = (sp[i] + (xbias ? xb[i] : 0.f)) + xe[i];
        if (h == 0 && k < d) xout[(int64_t)b * d + k] = xe[i];
      }
    }
  }
  PM_STAMP(1);
  // ---- LayerNorm of row b (two-pass from registers)
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < XR; ++i) s += tid + i * DF_THREADS < d ? xe[i] : 0.f;
  const float mean = block_reduce8(s, scratch, false) / (float)d;
  float q2 = 0.f;
#pragma unroll
  for (int i = 0; i < XR; ++i) {
    const int k = tid + i * DF_THREADS;
    const float dv = k < d ? xe[i] - mean : 0.f;
    q2 = fmaf(dv, dv, q2);
  }
  const float rstd = rsqrtf(block_reduce8(q2, scratch + DF_WAVES, false) / (float)d + eps);
#pragma unroll
  for (int i = 0; i < XR; ++i) {  // (zeros from d up to the instantiation's width: the projection runs all NCH chunks)
    const int k = tid + i * DF_THREADS;
    if (k < NCH * 64) xn[k] = k < d ? (xe[i] - mean) * rstd * ge[i] + be[i] : 0.f;
  }
  __syncthreads();
  PM_STAMP(2);
  // ---- q (k, v) = W xn + b : fp32 FMA over this lane's chunks, then across the 8 lanes of the row
#pragma unroll
  for (int o = 0; o < (SELF ? 3 : 1); ++o) {
    float acc = 0.f;
    if constexpr (!UPFRONT) {
      const bf16* wr = Wp + ((int64_t)o * inner + h * 64 + prow) * d + pl * 8;
#pragma unroll
      for (int i = 0; i < NCH; ++i)
        wv[0][i] = *(const bf16x8*)(wr + (i < nch ? i : nch - 1) * 64);
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const float* xp = xn + i * 64 + pl * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc = fmaf((float)wv[UPFRONT ? o : 0][i][e], xp[e], acc);
    }
    acc = sum8_dpp(acc);
    if (pl == 0) {
      float v = acc + bpe[o];
      if (SELF && o > 0) {  // cached k / v: round once to the cache's type (bf16; float = no rounding), store, use the stored value
        const KT r = (KT)v;
        (o == 1 ? Kc : Vc)[b * sb + h * sh + (int64_t)tpos * sk + prow] = r;
        v = (float)r;
      }
      qkv[o * 64 + prow] = v;
    }
  }
  __syncthreads();

  PM_STAMP(3);
  float oval = 0.f;  // threads 0..63: the block's output element
  // CHAIN with W_o: the head's 64 columns of W_o, 8 lanes per row (16 bytes each: whole 128-byte lines per request; one row per
  // lane touched 64 lines per request and cost the block ~2 us), rows  pass * 512 + wave * 64 + i * 8 + (lane >> 3);  queued behind the first K / V groups
  constexpr int NO = CHAIN ? (NCH * 64 + DF_THREADS - 1) / DF_THREADS : 1;
  constexpr bool WO_EARLY = CHAIN && NO == 1;  // wider models: requested at the tail (NO x 32 registers would not fit here)
  bf16x8 wo[NO][8];
  if constexpr (WO_EARLY) {
    if (Wo && PM_CHAIN_ABL != 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int n = wave * 64 + i * 8 + (lane >> 3);
        wo[0][i] = *(const bf16x8*)(Wo + (int64_t)(n < d ? n : d - 1) * inner + h * 64 + (lane & 7) * 8);
      }
    }
  }
  // ---- scores.  The cross block's 1500 keys are six groups of NKU x 64: the NEXT group is requested before this one is used
  // (two register sets, the loop unrolled by two), so that 32-64 KB per CU stay in flight across the group boundary - with the
  // request and its use in the same iteration every group exposed one full memory latency (PM_CROSS_DB=0 builds that form)
  const f32x4 q0 = *(const f32x4*)(qkv + c * 8), q1 = *(const f32x4*)(qkv + c * 8 + 4);
  constexpr int KG = 64 * NKU;
#define PM_KV_LOAD(dst_, base_, k0_)                                                                                   \
  _Pragma("unroll") for (int u = 0; u < NKU; ++u) {                                                                    \
    int key_ = (k0_) + u * 64 + wave * 8 + ks;                                                                         \
    key_ = key_ < Lc ? key_ : Lc - 1;                                                                                  \
    dst_[u] = SELF ? *(const kv8*)(base_ + key_ * sk) : __builtin_nontemporal_load((const kv8*)(base_ + key_ * sk));    \
  }
#define PM_K_SCORES(src_, k0_)                                                                                         \
  _Pragma("unroll") for (int u = 0; u < NKU; ++u) {                                                                    \
    float sv = 0.f;                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) sv = fmaf(q0[i], (float)src_[u][i], sv);                             \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) sv = fmaf(q1[i], (float)src_[u][4 + i], sv);                         \
    sv = sum8_dpp(sv);                                                                                                 \
    const int key_ = (k0_) + u * 64 + wave * 8 + ks;                                                                   \
    if (c == 0 && key_ < Lc) sc[key_] = sv * 0.125f;                                                                   \
    lmax = fmaxf(lmax, key_ < Lc ? sv * 0.125f : -INFINITY);                                                           \
  }
  // the scores' maximum rides along with the scores (a lane's running maximum, the wave's by DPP / permlane at the end, the 8
  // waves' through scratch behind the barrier that publishes the scores anyway): the separate pass over the score array and
  // its block reduction - one workgroup barrier and two LDS round trips of the block's critical path - are gone
  float lmax = -INFINITY;
  if constexpr (!SELF && PM_CROSS_DB) {
    kv8 kw[NKU];
    for (int k0 = 0; k0 < Lc;) {
      if (k0 + KG < Lc) PM_KV_LOAD(kw, kb, k0 + KG);
      PM_K_SCORES(kv, k0);
      k0 += KG;
      if (k0 >= Lc) break;
      if (k0 + KG < Lc) PM_KV_LOAD(kv, kb, k0 + KG);
      PM_K_SCORES(kw, k0);
      k0 += KG;
    }
  } else {
    for (int k0 = 0; k0 < Lc; k0 += KG) {
      if (k0 > 0) PM_KV_LOAD(kv, kb, k0);
      PM_K_SCORES(kv, k0);
    }
  }
  if (SELF && tid < 8) {  // the new key (position t), same summation shape as above
    float sv = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) sv = fmaf(q0[i], qkv[64 + c * 8 + i], sv);
#pragma unroll
    for (int i = 0; i < 4; ++i) sv = fmaf(q1[i], qkv[64 + c * 8 + 4 + i], sv);
    sv = sum8_dpp(sv);
    if (c == 0) sc[Lk - 1] = sv * 0.125f;
    lmax = fmaxf(lmax, sv * 0.125f);
  }
  {
    const float wmax = dwave_max(lmax);
    if (lane == 0) scratch[2 * DF_WAVES + wave] = wmax;
  }
  PM_STAMP(4);
  // V does not depend on the scores: request the first NKU passes now, so they fly during the softmax reductions
  kv8 vv[NKU];
#pragma unroll
  for (int u = 0; u < NKU; ++u) {
    int key = u * 64 + wave * 8 + ks;
    key = key < Lc ? key : (Lc > 0 ? Lc - 1 : 0);
    vv[u] = SELF ? *(const kv8*)(vb + key * sk) : __builtin_nontemporal_load((const kv8*)(vb + key * sk));
  }
  __syncthreads();
  float mx = scratch[2 * DF_WAVES];
#pragma unroll
  for (int w = 1; w < DF_WAVES; ++w) mx = fmaxf(mx, scratch[2 * DF_WAVES + w]);
  float sum = 0.f;
  for (int k = tid; k < Lk; k += DF_THREADS) {
    const float p = expf(sc[k] - mx);
    sc[k] = p;
    sum += p;
  }
  sum = block_reduce8(sum, scratch + 3 * DF_WAVES, false);  // its barrier also publishes the p values

  PM_STAMP(5);
  // ---- P.V
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#define PM_PV(src_, k0_)                                                                                               \
  _Pragma("unroll") for (int u = 0; u < NKU; ++u) {                                                                    \
    const int key_ = (k0_) + u * 64 + wave * 8 + ks;                                                                   \
    const float p_ = key_ < Lc ? sc[key_] : 0.f;                                                                       \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) acc[i] = fmaf(p_, (float)src_[u][i], acc[i]);                        \
  }
  if constexpr (!SELF && PM_CROSS_DB) {
    kv8 vw[NKU];
    for (int k0 = 0; k0 < Lc;) {
      if (k0 + KG < Lc) PM_KV_LOAD(vw, vb, k0 + KG);
      PM_PV(vv, k0);
      k0 += KG;
      if (k0 >= Lc) break;
      if (k0 + KG < Lc) PM_KV_LOAD(vv, vb, k0 + KG);
      PM_PV(vw, k0);
      k0 += KG;
    }
  } else {
    for (int k0 = 0; k0 < Lc; k0 += KG) {
      if (k0 > 0) PM_KV_LOAD(vv, vb, k0);
      PM_PV(vv, k0);
    }
  }
#undef PM_PV
#undef PM_K_SCORES
#undef PM_KV_LOAD
  PM_STAMP(6);
  if (SELF && wave == 0 && ks == 0) {
    const float p = sc[Lk - 1];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = fmaf(p, qkv[128 + c * 8 + i], acc[i]);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    acc[i] += xor8_dpp(acc[i]);
    acc[i] = add_xor16(acc[i]);
    acc[i] = add_xor32(acc[i]);
  }
  if (ks == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) part[wave * 64 + c * 8 + i] = acc[i];
  }
  __syncthreads();
  if (tid < 64) {
    float o = 0.f;
#pragma unroll
    for (int w = 0; w < DF_WAVES; ++w) o += part[w * 64 + tid];
    oval = o / sum;
  }
  if (tid < 64) {
    if (CHAIN && Wo) qkv[tid] = oval;  // q is in registers, the new k / v rows were consumed in front of the barrier above
    else out[((int64_t)b * H + h) * 64 + tid] = oval;
  }
  PM_STAMP(7);
  if constexpr (CHAIN) {
    if (!Wo) return;
    __syncthreads();
    float* my = hparts + (int64_t)blockIdx.x * d;
    if constexpr (!WO_EARLY) {
#pragma unroll
      for (int q = 0; q < NO; ++q)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int n = q * DF_THREADS + wave * 64 + i * 8 + (lane >> 3);
          wo[q][i] = *(const bf16x8*)(Wo + (int64_t)(n < d ? n : d - 1) * inner + h * 64 + (lane & 7) * 8);
        }
    }
    const f32x4 o0 = *(const f32x4*)(qkv + (lane & 7) * 8), o1 = *(const f32x4*)(qkv + (lane & 7) * 8 + 4);
#pragma unroll
    for (int q = 0; q < NO; ++q) {
      float res = 0.f;
#pragma unroll
      for (int i = 0; i < (PM_CHAIN_ABL == 2 ? 1 : 8); ++i) {
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) a = fmaf((float)wo[q][i][e], o0[e], a);
#pragma unroll
        for (int e = 0; e < 4; ++e) a = fmaf((float)wo[q][i][4 + e], o1[e], a);
        a = sum8_dpp(a);
        res = (lane & 7) == i ? a : res;  // lane (g, c) keeps row i = c of its group g
      }
      const int n = q * DF_THREADS + wave * 64 + (lane & 7) * 8 + (lane >> 3);
      if (n < d && (PM_CHAIN_ABL != 3 || res == 123.f)) my[n] = res;
    }
    PM_STAMP(8);
  }
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dec_argmax_reduce_kernel(const float* __restrict__ ws_val, const int* __restrict__ ws_idx,
                                                                int nwg, const int* __restrict__ pos_ptr,
                                                                const int64_t* __restrict__ prompt, int P,
                                                                int64_t* __restrict__ tok_cur, int64_t* __restrict__ tokens_out,
                                                                int Ttot, float* __restrict__ margin_out,
                                                                // fused tail (all null / 0 for the plain reduce):
                                                                const bf16* __restrict__ E, const float* __restrict__ pos_tab,
                                                                float* __restrict__ x, int d, int V, int* ticket,
                                                                int* pos_rw) {
  __shared__ float sv[4];
  __shared__ int si[4];
  __shared__ float s2[4];
  __shared__ int64_t snext;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // what does not depend on the winner is requested up front, in the order of its use (loads return in order): the position,
  // the forced prompt token and the next step's positional row - each used to cost its own round trip after the reduction
  const int t = *pos_ptr;  // the token just consumed sits at position t; this step decides position t + 1
  const int t1 = t + 1;
  const int64_t forced = t1 < P ? prompt[(int64_t)b * P + t1] : -1;
  f32x4 pp0 = {0.f, 0.f, 0.f, 0.f}, pp1 = pp0;
  if (E && tid < d / 8) {
    pp0 = *(const f32x4*)(pos_tab + (int64_t)t1 * d + tid * 8);
    pp1 = *(const f32x4*)(pos_tab + (int64_t)t1 * d + tid * 8 + 4);
  }
  float bv = -INFINITY, second = -INFINITY;
  int bi = 0x7fffffff;
  // four (value, index) pairs per thread requested together, branch-free (clamped index, the repeats dropped below): one
  // request per loop trip made every trip a dependent memory round trip (811 tiles = 4 trips per thread at Whisper's vocabulary)
  for (int i0 = tid; i0 < nwg; i0 += 4 * 256) {
    float v4[4];
    int x4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = i0 + 256 * j;
      const int ic = i < nwg ? i : nwg - 1;
      v4[j] = ws_val[(int64_t)b * nwg + ic];
      x4[j] = ws_idx[(int64_t)b * nwg + ic];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v = i0 + 256 * j < nwg ? v4[j] : -INFINITY;
      const int ix = i0 + 256 * j < nwg ? x4[j] : 0x7fffffff;
      if (v > bv || (v == bv && ix < bi)) { second = fmaxf(second, bv); bv = v; bi = ix; }
      else second = fmaxf(second, v);
    }
  }
  // (winner, runner-up) over the wave: partners 1, 2 lanes away by quad permutes, 4 by the half row's mirror (the quads are
  // uniform by then), 8 by a row rotate, 16 / 32 by v_permlane16 / 32_swap - no LDS round trips (see dpp_mov)
  auto comb = [&](float ov, int oi, float o2) {
    if (ov > bv || (ov == bv && oi < bi)) { second = fmaxf(fmaxf(second, o2), bv); bv = ov; bi = oi; }
    else second = fmaxf(fmaxf(second, o2), ov);
  };
#define PM_AR_DPP(C_) comb(dpp_mov<C_>(bv), __builtin_amdgcn_update_dpp(0, bi, C_, 0xf, 0xf, true), dpp_mov<C_>(second))
  PM_AR_DPP(0xB1);
  PM_AR_DPP(0x4E);
  PM_AR_DPP(0x141);
  PM_AR_DPP(0x128);
#undef PM_AR_DPP
  {
    const auto rv = __builtin_amdgcn_permlane16_swap(__float_as_uint(bv), __float_as_uint(bv), false, false);
    const auto ri = __builtin_amdgcn_permlane16_swap((unsigned)bi, (unsigned)bi, false, false);
    const auto r2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(second), __float_as_uint(second), false, false);
    const int sel = (lane & 16) ? 0 : 1;  // the partner's copy: result 1 in even rows of 16, result 0 in odd rows
    comb(__uint_as_float(rv[sel]), (int)ri[sel], __uint_as_float(r2[sel]));
  }
  {
    const auto rv = __builtin_amdgcn_permlane32_swap(__float_as_uint(bv), __float_as_uint(bv), false, false);
    const auto ri = __builtin_amdgcn_permlane32_swap((unsigned)bi, (unsigned)bi, false, false);
    const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(second), __float_as_uint(second), false, false);
    const int sel = (lane & 32) ? 0 : 1;
    comb(__uint_as_float(rv[sel]), (int)ri[sel], __uint_as_float(r2[sel]));
  }
  if (lane == 0) { sv[wave] = bv; si[wave] = bi; s2[wave] = second; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w) {
      if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { second = fmaxf(fmaxf(second, s2[w]), bv); bv = sv[w]; bi = si[w]; }
      else second = fmaxf(fmaxf(second, s2[w]), sv[w]);
    }
    const int64_t next = (t1 < P) ? forced : (int64_t)bi;
    tok_cur[b] = next;
    if (t + 1 < Ttot) tokens_out[(int64_t)b * Ttot + t + 1] = next;
    // NOTE: `second` is the runner-up among per-tile winners, i.e. a lower bound on the true top1 - top2 margin's
    // complement; it is diagnostic only (tests classify near-ties with it).
    if (margin_out && t + 1 < Ttot) margin_out[(int64_t)b * Ttot + t + 1] = bv - second;
    snext = next;
  }
  if (!E) return;
  // ---- fused tail: the next step's input row x[b] = E[next] + pos[t + 1] (what dec_embed would compute after the
  // position moved), then the LAST workgroup to get here moves the position: every workgroup read it before its ticket
  __syncthreads();
  int64_t id = snext;
  id = id < 0 ? 0 : (id >= V ? V - 1 : id);
  for (int c = tid; c < d / 8; c += 256) {
    const bf16x8 e = *(const bf16x8*)(E + id * d + c * 8);
    f32x4 p0 = pp0, p1 = pp1;
    if (c != tid) {  // d > 2048 only
      p0 = *(const f32x4*)(pos_tab + (int64_t)t1 * d + c * 8);
      p1 = *(const f32x4*)(pos_tab + (int64_t)t1 * d + c * 8 + 4);
    }
    f32x4 o0, o1;
#pragma unroll
    for (int i = 0; i < 4; ++i) { o0[i] = (float)e[i] + p0[i]; o1[i] = (float)e[4 + i] + p1[i]; }
    *(f32x4*)(x + (int64_t)b * d + c * 8) = o0;
    *(f32x4*)(x + (int64_t)b * d + c * 8 + 4) = o1;
  }
  if (tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int n = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n == (int)gridDim.x - 1) {
      __hip_atomic_store(pos_rw, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---- Whisper's decoding-time logit filters, on the device, between the vocabulary projection and the token choice.
// The reference has no Whisper decoding at all (README.md:86-87 lists tokenizer / timestamp handling as TODO); the rules
// restated here are those of OpenAI's whisper (decoding.py: SuppressTokens, SuppressBlank, ApplyTimestampRules), taken from
// its published description - that package is not in the image, so oracle/ref_whisper_rules.py is "parity unpinned".
// One workgroup per sequence; n = *pos_ptr + 1 is the index of the token being chosen, generated tokens are ids[P .. n).
//   always        : logits[suppress[i]] = -inf;  logits[no_timestamps] = -inf
//   n == P        : logits[blank[i]] = -inf (no blank / end-of-text first); text tokens [0, ts_begin) = -inf (a transcript
//                   starts with a timestamp); timestamps beyond ts_begin + max_initial are -inf
//   pairing       : last token a timestamp -> if the one before was too (or there is none): timestamps [ts_begin, V) = -inf,
//                   else (an open segment must be closed): text [0, eot) = -inf
//   monotonic     : timestamps below the last emitted one (below or equal, unless that one still waits for its partner) = -inf
//   probability   : if logsumexp(timestamp logits) > max(text logits): text [0, ts_begin) = -inf
__global__ __launch_bounds__(DF_THREADS) void dec_whisper_rules_kernel(float* __restrict__ logits, int64_t ldl, int V,
                                                               const int64_t* __restrict__ tokens, int Ttot,
                                                               const int* __restrict__ pos_ptr, int P, int eot, int no_ts,
                                                               int ts_begin, int max_initial, const int* __restrict__ suppress,
                                                               int n_suppress, const int* __restrict__ blank, int n_blank) {
  __shared__ float red[4 * DF_WAVES];
  __shared__ int redi[DF_WAVES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = *pos_ptr + 1;
  if (n < P) return;  // the prompt is still being forced
  float* lg = logits + (int64_t)blockIdx.x * ldl;
  const int64_t* ids = tokens + (int64_t)blockIdx.x * Ttot;
  const int ngen = n - P;
  const bool last_ts = ngen >= 1 && ids[n - 1] >= ts_begin;
  const bool penult_ts = ngen < 2 || ids[n - 2] >= ts_begin;
  // index of the last emitted timestamp (-1: none)
  int li = -1;
  for (int i = P + tid; i < n; i += DF_THREADS)
    if (ids[i] >= ts_begin) li = i;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) li = max(li, __shfl_xor(li, o, 64));
  if (lane == 0) redi[wave] = li;
  __syncthreads();
  li = redi[0];
#pragma unroll
  for (int w = 1; w < DF_WAVES; ++w) li = max(li, redi[w]);
  int ts_floor = ts_begin;  // timestamps in [ts_begin, ts_floor) are forbidden
  if (li >= 0) ts_floor = (int)ids[li] + ((last_ts && !penult_ts) ? 0 : 1);
  const bool no_text_pair = last_ts && !penult_ts;  // text [0, eot) forbidden
  const bool no_ts_pair = last_ts && penult_ts;     // every timestamp forbidden
  const bool first = ngen == 0;
  const int ts_cap = (first && max_initial >= 0) ? ts_begin + max_initial : V - 1;  // timestamps above are forbidden
  for (int v = tid; v < V; v += DF_THREADS) {
    bool dead = v == no_ts;
    if (v < ts_begin) dead = dead || first || (no_text_pair && v < eot);
    else dead = dead || no_ts_pair || v < ts_floor || v > ts_cap;
    if (dead) lg[v] = -INFINITY;
  }
  for (int i = tid; i < n_suppress; i += DF_THREADS) lg[suppress[i]] = -INFINITY;
  if (first)
    for (int i = tid; i < n_blank; i += DF_THREADS) lg[blank[i]] = -INFINITY;
  __syncthreads();
  // statistics over the filtered row (the listed suppressions included): V floats from L2
  float m_text = -INFINITY, m_ts = -INFINITY;
  for (int v = tid; v < V; v += DF_THREADS) {
    const float x = lg[v];
    if (v < ts_begin) m_text = fmaxf(m_text, x);
    else m_ts = fmaxf(m_ts, x);
  }
  m_text = block_reduce8(m_text, red, true);
  m_ts = block_reduce8(m_ts, red + DF_WAVES, true);
  if (m_ts == -INFINITY) return;  // no timestamp allowed at all: nothing to compare
  float se = 0.f;
  for (int v = ts_begin + tid; v < V; v += DF_THREADS) se += expf(lg[v] - m_ts);
  se = block_reduce8(se, red + 2 * DF_WAVES, false);
  if (m_ts + logf(se) > m_text)
    for (int v = tid; v < ts_begin; v += DF_THREADS) lg[v] = -INFINITY;
}

// ---------------------------------------------------------------------------------------------------------------
// Top-k sampling of the next token (the reference's text/generator.py:30-32: topk, softmax over the k logits,
// multinomial), one workgroup per sequence: k rounds of a block-wide arg-max in the order (value descending, index
// ascending) - each round takes the best element that comes AFTER the previous winner in that order, so nothing is
// masked or sorted - then a softmax over the k winners and one draw from a counter-based generator keyed by (seed,
// position, sequence): the same seed gives the same text.  The tail is pm_dec_next_token's (prompt forcing, next
// embedding row, ticketed position advance).
constexpr int DS_MAXK = 64;

__device__ __forceinline__ float ds_uniform(uint64_t seed, int t, int b) {  // splitmix64 finaliser -> [0, 1)
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(t + 1) + 0xBF58476D1CE4E5B9ull * (uint64_t)(b + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

__global__ __launch_bounds__(256) void dec_sample_topk_kernel(const float* __restrict__ logits, int ldl, int V, int k,
                                                              uint64_t seed, const int64_t* __restrict__ prompt, int P,
                                                              int64_t* __restrict__ tok_cur, int64_t* __restrict__ tokens_out,
                                                              int Ttot, const bf16* __restrict__ E,
                                                              const float* __restrict__ pos_tab, float* __restrict__ x, int d,
                                                              int* ticket, int* pos_rw) {
  __shared__ float topv[DS_MAXK];
  __shared__ int topi[DS_MAXK];
  __shared__ float wv[4];
  __shared__ int wi[4];
  __shared__ int64_t snext;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = logits + (int64_t)b * ldl;
  float pv = INFINITY;  // previous winner (value, index): the first round accepts everything
  int pi = -1;
  for (int r = 0; r < k; ++r) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < V; i += 256) {
      const float v = row[i];
      const bool after = v < pv || (v == pv && i > pi);
      if (after && (v > bv || (v == bv && i < bi))) { bv = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { wv[wave] = bv; wi[wave] = bi; }
    __syncthreads();
    bv = wv[0];
    bi = wi[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (wv[w] > bv || (wv[w] == bv && wi[w] < bi)) { bv = wv[w]; bi = wi[w]; }
    if (tid == 0) { topv[r] = bv; topi[r] = bi; }
    pv = bv;
    pi = bi;
    __syncthreads();
  }
  const int t = *pos_rw;
  if (tid == 0) {
    float cum[DS_MAXK];
    float sum = 0.f;
    for (int r = 0; r < k; ++r) { sum += expf(topv[r] - topv[0]); cum[r] = sum; }
    const float u = ds_uniform(seed, t, b) * sum;
    int pick = k - 1;
    for (int r = k - 1; r >= 0; --r)
      if (u < cum[r]) pick = r;
    const int64_t next = (t + 1 < P) ? prompt[(int64_t)b * P + t + 1] : (int64_t)topi[pick];
    tok_cur[b] = next;
    if (t + 1 < Ttot) tokens_out[(int64_t)b * Ttot + t + 1] = next;
    snext = next;
  }
  __syncthreads();
  const int t1 = t + 1;
  int64_t id = snext;
  id = id < 0 ? 0 : (id >= V ? V - 1 : id);
  for (int c = tid; c < d / 8; c += 256) {
    const bf16x8 e = *(const bf16x8*)(E + id * d + c * 8);
    const f32x4 p0 = *(const f32x4*)(pos_tab + (int64_t)t1 * d + c * 8), p1 = *(const f32x4*)(pos_tab + (int64_t)t1 * d + c * 8 + 4);
    f32x4 o0, o1;
#pragma unroll
    for (int i = 0; i < 4; ++i) { o0[i] = (float)e[i] + p0[i]; o1[i] = (float)e[4 + i] + p1[i]; }
    *(f32x4*)(x + (int64_t)b * d + c * 8) = o0;
    *(f32x4*)(x + (int64_t)b * d + c * 8 + 4) = o1;
  }
  if (tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int n = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n == (int)gridDim.x - 1) {
      __hip_atomic_store(pos_rw, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
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

#define PM_DL_ARGS                                                                                                   \
  x, ldx, gamma, beta, eps, W, ldw, bias, resid, ldr, out, ldo, M, N, K, mode, kc, vc, inner, H, Tmax, pos_ptr, wv, wi, nwg
template <int ACT, int FT, int NSTEP, bool LN>
static void dl_launch_mt(int mt, dim3 grid, hipStream_t st, const float* x, int ldx, const float* gamma, const float* beta,
                         float eps, const bf16* W, int64_t ldw, const float* bias, const float* resid, int ldr, float* out,
                         int ldo, int M, int N, int K, int mode, bf16* kc, bf16* vc, int inner, int H, int Tmax,
                         const int* pos_ptr, float* wv, int* wi, int nwg) {
  if (mt <= 1) hipLaunchKernelGGL((dec_linear_kernel<ACT, 1, FT, NSTEP, LN>), grid, dim3(256), 0, st, PM_DL_ARGS);
  else if (mt == 2) hipLaunchKernelGGL((dec_linear_kernel<ACT, 2, FT, NSTEP, LN>), grid, dim3(256), 0, st, PM_DL_ARGS);
  else hipLaunchKernelGGL((dec_linear_kernel<ACT, 4, FT, NSTEP, LN>), grid, dim3(256), 0, st, PM_DL_ARGS);
}

template <int ACT, int FT>
static int dl_launch(int mt, dim3 grid, hipStream_t st, const float* x, int ldx, const float* gamma, const float* beta,
                     float eps, const bf16* W, int64_t ldw, const float* bias, const float* resid, int ldr, float* out,
                     int ldo, int M, int N, int K, int mode, bf16* kc, bf16* vc, int inner, int H, int Tmax,
                     const int* pos_ptr, float* wv, int* wi, int nwg) {
  const int per_wave = (K / 32 + 3) / 4;  // K steps each wave owns
  if (gamma) {  // LayerNorm: the wave's whole share of x must sit in registers
    if (mt > 2 && per_wave > 4) return PM_EUNSUPPORTED;
    if (per_wave <= 4) dl_launch_mt<ACT, FT, 4, true>(mt, grid, st, PM_DL_ARGS);
    else if (per_wave <= 8 && FT == 1) dl_launch_mt<ACT, FT, 8, true>(mt, grid, st, PM_DL_ARGS);
    else if (per_wave <= 10 && FT == 1) dl_launch_mt<ACT, FT, 10, true>(mt, grid, st, PM_DL_ARGS);
    else if (per_wave <= 10) dl_launch_mt<ACT, FT == 1 ? 1 : 2, 10, true>(mt, grid, st, PM_DL_ARGS);
    else return PM_EUNSUPPORTED;
  } else {
    if (per_wave <= 4 || mt > 2) dl_launch_mt<ACT, FT, 4, false>(mt, grid, st, PM_DL_ARGS);
    else dl_launch_mt<ACT, FT, 8, false>(mt, grid, st, PM_DL_ARGS);
  }
  return PM_OK;
}
#undef PM_DL_ARGS

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
  if (act != PM_ACT_NONE && act != PM_ACT_GELU && act != PM_ACT_GELU_TANH) return PM_EUNSUPPORTED;
  if (K > 1280 && gamma) return PM_EUNSUPPORTED;
  const int per_wave = (int)((K / 32 + 3) / 4);
  const int ft = mode == DL_ARGMAX ? (gamma && per_wave > 4 ? 2 : 4) : 1;  // see pm_dec_argmax_tile()
  const int nwg = (int)((N + DL_FEATS * ft - 1) / (DL_FEATS * ft));
  int mt = (int)((M + 15) / 16);
  hipStream_t st = (hipStream_t)stream;
  int rc = PM_OK;
  // rows are independent: with few feature tiles (N / 16 workgroups) the 16-row tiles go to separate workgroups, each
  // reading only its rows of x (PM_DEC_ROWSPLIT=0 switches this off)
  static const bool rowsplit = [] { const char* e = getenv("PM_DEC_ROWSPLIT"); return !e || atoi(e) != 0; }();
  unsigned gz = 1;
  if (rowsplit && mode != DL_ARGMAX && mt > 1 && nwg * mt <= 512) { gz = (unsigned)mt; mt = 1; }
  static const bool logits_persist = [] { const char* e = getenv("PM_DEC_LOGITS_PERSIST"); return !e || atoi(e) != 0; }();
  if (mode == DL_ARGMAX && logits_persist && gamma && M <= 32 && per_wave <= 4 && nwg >= 64) {
    // final LayerNorm + vocabulary projection + arg-max tiles as one persistent launch (dec_logits_kernel); the arg-max tiles
    // and their workspace layout are this function's (pm_dec_argmax_tile)
    static const int cus = [] {
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
      return n;
    }();
    // one 512-thread workgroup per CU whose step is a whole 64-feature tile (PM_DEC_LOGITS_WAVES=4: two 256-thread workgroups
    // per CU, 32 features per step)
    static const int lw = [] { const char* e = getenv("PM_DEC_LOGITS_WAVES"); return e ? atoi(e) : 8; }();
    if (lw == 8 && ft == 4) {
      const int grid = nwg < cus ? nwg : cus;
      if (mt <= 1)
        hipLaunchKernelGGL((dec_logits_kernel<1, 8, 4>), dim3(grid), dim3(512), 0, st, x, (int)ldx, gamma, beta, eps, (const bf16*)w, ldw,
                           bias, (int)M, (int)N, (int)K, ws_val, (int*)ws_idx, nwg, 1);
      else
        hipLaunchKernelGGL((dec_logits_kernel<2, 8, 4>), dim3(grid), dim3(512), 0, st, x, (int)ldx, gamma, beta, eps, (const bf16*)w, ldw,
                           bias, (int)M, (int)N, (int)K, ws_val, (int*)ws_idx, nwg, 1);
    } else {
      const int grid = nwg < 2 * cus ? nwg : 2 * cus;
      if (mt <= 1)
        hipLaunchKernelGGL((dec_logits_kernel<1, 4, 2>), dim3(grid), dim3(256), 0, st, x, (int)ldx, gamma, beta, eps, (const bf16*)w, ldw,
                           bias, (int)M, (int)N, (int)K, ws_val, (int*)ws_idx, nwg, ft / 2);
      else
        hipLaunchKernelGGL((dec_logits_kernel<2, 4, 2>), dim3(grid), dim3(256), 0, st, x, (int)ldx, gamma, beta, eps, (const bf16*)w, ldw,
                           bias, (int)M, (int)N, (int)K, ws_val, (int*)ws_idx, nwg, ft / 2);
    }
  } else if (mode == DL_ARGMAX && ft == 2)
    rc = dl_launch<PM_ACT_NONE, 2>(mt, dim3(nwg), st, x, (int)ldx, gamma, beta, eps, (const bf16*)w, ldw, bias, resid, (int)ldr, out,
                              (int)ldo, (int)M, (int)N, (int)K, mode, (bf16*)kcache, (bf16*)vcache, (int)inner, (int)H,
                              (int)Tmax, (const int*)pos_ptr, ws_val, (int*)ws_idx, nwg);
  else if (mode == DL_ARGMAX)
    rc = dl_launch<PM_ACT_NONE, 4>(mt, dim3(nwg), st, x, (int)ldx, gamma, beta, eps, (const bf16*)w, ldw, bias, resid, (int)ldr, out,
                              (int)ldo, (int)M, (int)N, (int)K, mode, (bf16*)kcache, (bf16*)vcache, (int)inner, (int)H,
                              (int)Tmax, (const int*)pos_ptr, ws_val, (int*)ws_idx, nwg);
  else if (act == PM_ACT_GELU_TANH)
    rc = dl_launch<PM_ACT_GELU_TANH, 1>(mt, dim3(nwg, 1, gz), st, x, (int)ldx, gamma, beta, eps, (const bf16*)w, ldw, bias, resid, (int)ldr, out,
                           (int)ldo, (int)M, (int)N, (int)K, mode, (bf16*)kcache, (bf16*)vcache, (int)inner, (int)H,
                           (int)Tmax, (const int*)pos_ptr, ws_val, (int*)ws_idx, nwg);
  else if (act == PM_ACT_GELU)
    rc = dl_launch<PM_ACT_GELU, 1>(mt, dim3(nwg, 1, gz), st, x, (int)ldx, gamma, beta, eps, (const bf16*)w, ldw, bias, resid, (int)ldr, out,
                           (int)ldo, (int)M, (int)N, (int)K, mode, (bf16*)kcache, (bf16*)vcache, (int)inner, (int)H,
                           (int)Tmax, (const int*)pos_ptr, ws_val, (int*)ws_idx, nwg);
  else
    rc = dl_launch<PM_ACT_NONE, 1>(mt, dim3(nwg, 1, gz), st, x, (int)ldx, gamma, beta, eps, (const bf16*)w, ldw, bias, resid, (int)ldr, out,
                           (int)ldo, (int)M, (int)N, (int)K, mode, (bf16*)kcache, (bf16*)vcache, (int)inner, (int)H,
                           (int)Tmax, (const int*)pos_ptr, ws_val, (int*)ws_idx, nwg);
  if (rc != PM_OK) return rc;
  PM_CHECK_LAUNCH();
  return PM_OK;
}

/* pm_dec_linear in plain mode (no LayerNorm) with K split over k_split (2..8) workgroups per 16-feature tile: each part
 * reads 1 / k_split of x and of its weight rows; the last part to finish (agent-scope ticket, no spinning) adds the parts
 * in part order and applies bias / activation / residual.  split_ws: ceil(N / 16) * k_split * mt * 256 floats (mt = ceil(M / 16) rounded up to 1, 2 or 4);
 * split_cnt: ceil(N / 16) * 4 int32 (a ticket per feature tile and row tile), zero before the first launch and zero
 * again after every launch. */
static int dec_linear_ksplit_impl(const float* x, int64_t ldx, const void* w, int64_t ldw, const float* bias,
                                  const float* resid, int64_t ldr, float* out, int64_t ldo, int64_t M, int64_t N, int64_t K,
                                  int act, int64_t k_split, float* split_ws, int32_t* split_cnt, int mode, void* stream) {
  if (!x || !w || !out || M <= 0 || N <= 0 || K <= 0) return PM_EINVAL;
  if (mode == DL_PLAIN && (!split_ws || !split_cnt)) return PM_EINVAL;
  if (M > 64 || K % 32) return PM_EUNSUPPORTED;
  if (k_split < 2 || k_split > 8 || K / 32 < k_split) return PM_EINVAL;
  if (ldx < K || ldw < K || ldx % 4 || ldw % 8) return PM_EALIGN;
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)split_ws) & 15) return PM_EALIGN;
  if (mode == DL_PARTS && ((((uintptr_t)out) & 15) || ldo % 4 || ldr % 4)) return PM_EALIGN;
  if (ldo < N || (mode == DL_PLAIN && resid && ldr < N)) return PM_EINVAL;
  if (act != PM_ACT_NONE && act != PM_ACT_GELU && act != PM_ACT_GELU_TANH) return PM_EUNSUPPORTED;
  const int nwg = (int)((N + DL_FEATS - 1) / DL_FEATS);
  int mt = (int)((M + 15) / 16);
  static const bool rowsplit = [] { const char* e = getenv("PM_DEC_ROWSPLIT"); return !e || atoi(e) != 0; }();
  unsigned gz = 1;
  if (rowsplit && mt > 1 && nwg * mt * k_split <= 1024) { gz = (unsigned)mt; mt = 1; }
  const int k_eff = (int)(((K / 32 + k_split - 1) / k_split) * 32);  // the longest part decides the instantiation
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(nwg, (unsigned)k_split, gz);
  int rc;
  // (dl_launch picks NSTEP from its K argument; the kernel takes the true K from its own parameter list, so pass
  // k_eff only for that choice: both instantiations below receive the real K)
  const int per_wave = (k_eff / 32 + 3) / 4;
#define PM_DLK(ACT_, NS_)                                                                                             \
  do {                                                                                                                \
    if (mt <= 1) hipLaunchKernelGGL((dec_linear_kernel<ACT_, 1, 1, NS_, false>), grid, dim3(256), 0, st, x, (int)ldx,   \
                                    (const float*)nullptr, (const float*)nullptr, 0.f, (const bf16*)w, ldw, bias, resid, \
                                    (int)ldr, out, (int)ldo, (int)M, (int)N, (int)K, mode, (bf16*)nullptr,           \
                                    (bf16*)nullptr, 0, 0, 0, (const int*)nullptr, split_ws, (int*)split_cnt, nwg);       \
    else if (mt == 2) hipLaunchKernelGGL((dec_linear_kernel<ACT_, 2, 1, NS_, false>), grid, dim3(256), 0, st, x,         \
                                         (int)ldx, (const float*)nullptr, (const float*)nullptr, 0.f, (const bf16*)w,    \
                                         ldw, bias, resid, (int)ldr, out, (int)ldo, (int)M, (int)N, (int)K, mode,     \
                                         (bf16*)nullptr, (bf16*)nullptr, 0, 0, 0, (const int*)nullptr, split_ws,          \
                                         (int*)split_cnt, nwg);                                                          \
    else hipLaunchKernelGGL((dec_linear_kernel<ACT_, 4, 1, 4, false>), grid, dim3(256), 0, st, x, (int)ldx,              \
                            (const float*)nullptr, (const float*)nullptr, 0.f, (const bf16*)w, ldw, bias, resid,         \
                            (int)ldr, out, (int)ldo, (int)M, (int)N, (int)K, mode, (bf16*)nullptr, (bf16*)nullptr, 0, \
                            0, 0, (const int*)nullptr, split_ws, (int*)split_cnt, nwg);                                   \
  } while (0)
  rc = PM_OK;
  if (act == PM_ACT_GELU) {
    if (per_wave <= 4 || mt > 2) PM_DLK(PM_ACT_GELU, 4);
    else PM_DLK(PM_ACT_GELU, 8);
  } else if (act == PM_ACT_GELU_TANH) {
    if (per_wave <= 4 || mt > 2) PM_DLK(PM_ACT_GELU_TANH, 4);
    else PM_DLK(PM_ACT_GELU_TANH, 8);
  } else {
    if (per_wave <= 4 || mt > 2) PM_DLK(PM_ACT_NONE, 4);
    else PM_DLK(PM_ACT_NONE, 8);
  }
#undef PM_DLK
  if (rc != PM_OK) return rc;
  PM_CHECK_LAUNCH();
  return PM_OK;
}

extern "C" int pm_dec_linear_ksplit(const float* x, int64_t ldx, const void* w, int64_t ldw, const float* bias,
                                    const float* resid, int64_t ldr, float* out, int64_t ldo, int64_t M, int64_t N, int64_t K,
                                    int act, int64_t k_split, float* split_ws, int32_t* split_cnt, void* stream) {
  return dec_linear_ksplit_impl(x, ldx, w, ldw, bias, resid, ldr, out, ldo, M, N, K, act, k_split, split_ws, split_cnt, DL_PLAIN,
                                stream);
}

extern "C" int pm_dec_linear_kparts(const float* x, int64_t ldx, const void* w, int64_t ldw, float* parts, int64_t ld_parts,
                                    int64_t part_stride, int64_t M, int64_t N, int64_t K, int64_t k_split, void* stream) {
  if (part_stride < M * ld_parts) return PM_EINVAL;
  return dec_linear_ksplit_impl(x, ldx, w, ldw, nullptr, nullptr, part_stride, parts, ld_parts, M, N, K, PM_ACT_NONE, k_split,
                                nullptr, nullptr, DL_PARTS, stream);
}

/* features per argmax tile for a given K (callers size ws_val / ws_idx as ceil(N / tile)) */
extern "C" int pm_dec_argmax_tile(int64_t K) { return ((K / 32 + 3) / 4 > 4) ? 32 : 64; }

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

template <typename KT, bool CHAIN>
static int dec_attention_fused_impl(const float* x, int64_t d, const float* gamma, const float* beta, float eps, const void* w,
                                    const float* bias, void* kc, void* vc, int64_t stride_b, int64_t stride_h, int64_t stride_k,
                                    const int32_t* pos_ptr, int64_t lk_const, int64_t lk_max, float* out, int64_t B, int64_t H,
                                    int self_attn, const float* x_parts, int64_t n_parts, int64_t part_stride,
                                    int64_t part_row_stride, const float* x_bias, float* x_out, const void* w_out,
                                    float* head_parts, void* stream) {
  if (!x || !gamma || !beta || !w || !kc || !vc || B <= 0 || H <= 0 || d <= 0 || lk_max <= 0) return PM_EINVAL;
  if (w_out ? !head_parts : !out) return PM_EINVAL;
  if (n_parts < 0 || n_parts > 64 || (n_parts > 0 && (!x_parts || !x_out || x_out == x))) return PM_EINVAL;
  if (d % 64 || d > 1280 || lk_max > DA_MAXK) return PM_EUNSUPPORTED;
  if (self_attn ? !pos_ptr : lk_const <= 0) return PM_EINVAL;
  if ((stride_b | stride_h | stride_k) % 8) return PM_EALIGN;
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)kc | (uintptr_t)vc | (uintptr_t)out | (uintptr_t)w_out) & 15) return PM_EALIGN;
  if (B * H > 0x7fffffff) return PM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int nch = (int)(d / 64);
#define PM_DF(SELF_, NCH_)                                                                                             \
  hipLaunchKernelGGL((dec_attn_fused_kernel<SELF_, NCH_, KT, CHAIN>), dim3((unsigned)(B * H)), dim3(DF_THREADS), 0, st, x,  \
                     (int)d, gamma, beta, eps, (const bf16*)w, bias, (KT*)kc, (KT*)vc, stride_b, stride_h, stride_k,      \
                     (const int*)pos_ptr, (int)lk_const, out, (int)H, x_parts, (int)n_parts, part_stride, part_row_stride, \
                     x_bias, x_out, (const bf16*)w_out, head_parts)
  if (self_attn) {
    if (nch <= 8) PM_DF(true, 8);
    else if (nch <= 12) PM_DF(true, 12);
    else if (nch <= 16) PM_DF(true, 16);
    else PM_DF(true, 20);
  } else {
    if (nch <= 8) PM_DF(false, 8);
    else if (nch <= 12) PM_DF(false, 12);
    else if (nch <= 16) PM_DF(false, 16);
    else PM_DF(false, 20);
  }
#undef PM_DF
  PM_CHECK_LAUNCH();
  return PM_OK;
}

extern "C" int pm_dec_attention_fused(const float* x, int64_t d, const float* gamma, const float* beta, float eps,
                                      const void* w, const float* bias, void* kc, void* vc, int64_t stride_b,
                                      int64_t stride_h, int64_t stride_k, const int32_t* pos_ptr, int64_t lk_const,
                                      int64_t lk_max, float* out, int64_t B, int64_t H, int self_attn, void* stream) {
  return dec_attention_fused_impl<bf16, false>(x, d, gamma, beta, eps, w, bias, kc, vc, stride_b, stride_h, stride_k, pos_ptr,
                                               lk_const, lk_max, out, B, H, self_attn, nullptr, 0, 0, 0, nullptr, nullptr,
                                               nullptr, nullptr, stream);
}

extern "C" int pm_dec_attention_fused_kv32(const float* x, int64_t d, const float* gamma, const float* beta, float eps,
                                           const void* w, const float* bias, void* kc, void* vc, int64_t stride_b,
                                           int64_t stride_h, int64_t stride_k, const int32_t* pos_ptr, int64_t lk_const,
                                           int64_t lk_max, float* out, int64_t B, int64_t H, int self_attn, void* stream) {
  return dec_attention_fused_impl<float, false>(x, d, gamma, beta, eps, w, bias, kc, vc, stride_b, stride_h, stride_k, pos_ptr,
                                                lk_const, lk_max, out, B, H, self_attn, nullptr, 0, 0, 0, nullptr, nullptr,
                                                nullptr, nullptr, stream);
}

extern "C" int pm_dec_attention_chain(const float* x, int64_t d, const float* gamma, const float* beta, float eps,
                                      const void* w, const float* bias, void* kc, void* vc, int64_t stride_b,
                                      int64_t stride_h, int64_t stride_k, const int32_t* pos_ptr, int64_t lk_const,
                                      int64_t lk_max, int64_t B, int64_t H, int self_attn, int kv_f32, const float* x_parts,
                                      int64_t n_parts, int64_t part_stride, int64_t part_row_stride, const float* x_bias,
                                      float* x_out, const void* w_out, float* head_parts, float* out, void* stream) {
  if (kv_f32)
    return dec_attention_fused_impl<float, true>(x, d, gamma, beta, eps, w, bias, kc, vc, stride_b, stride_h, stride_k, pos_ptr,
                                                 lk_const, lk_max, out, B, H, self_attn, x_parts, n_parts, part_stride,
                                                 part_row_stride, x_bias, x_out, w_out, head_parts, stream);
  return dec_attention_fused_impl<bf16, true>(x, d, gamma, beta, eps, w, bias, kc, vc, stride_b, stride_h, stride_k, pos_ptr,
                                              lk_const, lk_max, out, B, H, self_attn, x_parts, n_parts, part_stride,
                                              part_row_stride, x_bias, x_out, w_out, head_parts, stream);
}

extern "C" int pm_dec_argmax_reduce(const float* ws_val, const int32_t* ws_idx, int64_t n_tiles, const int32_t* pos_ptr,
                                    const int64_t* prompt, int64_t P, int64_t* tok_cur, int64_t* tokens_out,
                                    int64_t Ttot, float* margin_out, int64_t B, void* stream) {
  if (!ws_val || !ws_idx || !pos_ptr || !prompt || !tok_cur || !tokens_out || n_tiles <= 0 || P <= 0 || B <= 0 || Ttot < P)
    return PM_EINVAL;
  hipLaunchKernelGGL(dec_argmax_reduce_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, ws_val,
                     (const int*)ws_idx, (int)n_tiles, (const int*)pos_ptr, prompt, (int)P, tok_cur, tokens_out, (int)Ttot,
                     margin_out, (const bf16*)nullptr, (const float*)nullptr, (float*)nullptr, 0, 0, (int*)nullptr,
                     (int*)nullptr);
  PM_CHECK_LAUNCH();
  return PM_OK;
}

/* pm_dec_argmax_reduce + the next step's pm_dec_embed + pm_dec_advance in ONE launch: each sequence's workgroup picks
 * its token, writes x[b] = emb[token] + pos[t + 1], and the last workgroup to finish (agent-scope ticket, left at 0)
 * stores t + 1 to *pos_ptr.  Before the first step of a run the caller sets *pos_ptr = 0, tok_cur and runs
 * pm_dec_embed once. */
extern "C" int pm_dec_next_token(const float* ws_val, const int32_t* ws_idx, int64_t n_tiles, int32_t* pos_ptr,
                                 const int64_t* prompt, int64_t P, int64_t* tok_cur, int64_t* tokens_out, int64_t Ttot,
                                 float* margin_out, const void* emb, const float* pos, float* x, int64_t d, int64_t V,
                                 int32_t* ticket, int64_t B, void* stream) {
  if (!ws_val || !ws_idx || !pos_ptr || !prompt || !tok_cur || !tokens_out || n_tiles <= 0 || P <= 0 || B <= 0 || Ttot < P)
    return PM_EINVAL;
  if (!emb || !pos || !x || !ticket || d <= 0 || V <= 0) return PM_EINVAL;
  if (d % 8) return PM_EUNSUPPORTED;
  if (((uintptr_t)emb | (uintptr_t)pos | (uintptr_t)x) & 15) return PM_EALIGN;
  hipLaunchKernelGGL(dec_argmax_reduce_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, ws_val,
                     (const int*)ws_idx, (int)n_tiles, (const int*)pos_ptr, prompt, (int)P, tok_cur, tokens_out, (int)Ttot,
                     margin_out, (const bf16*)emb, pos, x, (int)d, (int)V, (int*)ticket, (int*)pos_ptr);
  PM_CHECK_LAUNCH();
  return PM_OK;
}

extern "C" int pm_dec_whisper_rules(float* logits, int64_t ldl, int64_t V, const int64_t* tokens, int64_t Ttot,
                                    const int32_t* pos_ptr, int64_t P, int64_t eot, int64_t no_timestamps,
                                    int64_t timestamp_begin, int64_t max_initial_timestamp, const int32_t* suppress,
                                    int64_t n_suppress, const int32_t* blank, int64_t n_blank, int64_t B, void* stream) {
  if (!logits || !tokens || !pos_ptr || V <= 0 || ldl < V || P <= 0 || Ttot < P || B <= 0) return PM_EINVAL;
  if (timestamp_begin <= 0 || timestamp_begin >= V || eot < 0 || eot >= timestamp_begin || no_timestamps >= V) return PM_EINVAL;
  if ((n_suppress > 0 && !suppress) || (n_blank > 0 && !blank) || n_suppress < 0 || n_blank < 0) return PM_EINVAL;
  hipLaunchKernelGGL(dec_whisper_rules_kernel, dim3((unsigned)B), dim3(DF_THREADS), 0, (hipStream_t)stream, logits, ldl, (int)V, tokens,
                     (int)Ttot, (const int*)pos_ptr, (int)P, (int)eot, (int)no_timestamps, (int)timestamp_begin,
                     (int)max_initial_timestamp, (const int*)suppress, (int)n_suppress, (const int*)blank, (int)n_blank);
  PM_CHECK_LAUNCH();
  return PM_OK;
}

/* Top-k sampling instead of the arg-max of pm_dec_next_token: logits (B, V) f32 (row stride ldl) of the last position ->
 * the k (1..64) largest per row (ties: lowest index first), softmax over them, one draw per sequence from a
 * counter-based generator keyed by (seed, position, sequence); then pm_dec_next_token's tail (prompt forcing, the next
 * step's embedding row, ticketed position advance).  k = 1 is the arg-max. */
extern "C" int pm_dec_sample_topk(const float* logits, int64_t ldl, int64_t V, int64_t k, uint64_t seed, int32_t* pos_ptr,
                                  const int64_t* prompt, int64_t P, int64_t* tok_cur, int64_t* tokens_out, int64_t Ttot,
                                  const void* emb, const float* pos, float* x, int64_t d, int32_t* ticket, int64_t B,
                                  void* stream) {
  if (!logits || !pos_ptr || !prompt || !tok_cur || !tokens_out || !emb || !pos || !x || !ticket) return PM_EINVAL;
  if (V <= 0 || ldl < V || k < 1 || k > DS_MAXK || k > V || P <= 0 || B <= 0 || Ttot < P || d <= 0) return PM_EINVAL;
  if (d % 8) return PM_EUNSUPPORTED;
  if (((uintptr_t)emb | (uintptr_t)pos | (uintptr_t)x) & 15) return PM_EALIGN;
  hipLaunchKernelGGL(dec_sample_topk_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, logits, (int)ldl, (int)V,
                     (int)k, seed, prompt, (int)P, tok_cur, tokens_out, (int)Ttot, (const bf16*)emb, pos, x, (int)d,
                     (int*)ticket, (int*)pos_ptr);
  PM_CHECK_LAUNCH();
  return PM_OK;
}

extern "C" int pm_dec_advance(int32_t* pos_ptr, void* stream) {
  if (!pos_ptr) return PM_EINVAL;
  hipLaunchKernelGGL(dec_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (int*)pos_ptr);
  PM_CHECK_LAUNCH();
  return PM_OK;
}
