// attention_bf16.hip - softmax(q k^T / sqrt(64) [causal]) v for head_dim 64, flash-style on MFMA.
// (reference: F.scaled_dot_product_attention at pytorch_models/transformer.py:52; the head
//  split / merge of transformer.py:47-53 is folded into the addressing.)
//
// Roofline: MFMA-bound for long sequences (4*Lq*Lk*64 flop per head against (Lq + 2*Lk)*128 B... read once
// per 128-query block); at L = 197 it is a small kernel whose K/V (25 KB per head) live in L2.
//
// Workgroup = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries.  K/V tiles of
// 64 keys are register-staged into a double-buffered LDS image (one barrier per tile).
//   S^T = K Q^T   : MFMA 32x32x16, A = K rows from LDS (ds_read_b128, XOR-swizzled), B = Q (registers).
//                   The accumulator then has the QUERY on the lane and 32 of the tile's 64 keys in
//                   registers (the other 32 sit in lane ^ 32): the row max / sum are in-lane + one swap.
//   O^T = V^T P^T : the S^T accumulator, converted to bf16 in place, IS the B operand (k = key); the
//                   A operand V^T comes from the row-major V image through ds_read_b64_tr_b16.
//                   O^T again has the query on the lane, so the online-softmax rescale is per lane.
#include <cstdlib>

#include "common.h"

namespace {
// a lane's value combined with its partner's 32 lanes away: gfx950's v_permlane32_swap (one vector-ALU instruction; hipcc turns
// __shfl_xor into ds_bpermute_b32, an LDS round trip)
__device__ __forceinline__ float ah_max_xor32(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float ah_add_xor32(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}


constexpr int KV_TILE = 64;
constexpr int TILE_B = KV_TILE * 128;  // 8 KiB: 64 rows x 64 bf16

__device__ __forceinline__ bf16x8 tr_read_pair(const char* p0, const char* p1) {
  union { s16x4 h[2]; bf16x8 v; } u;
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((PM_LDS s16x4*)p0);
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((PM_LDS s16x4*)p1);
  return u.v;
}

template <bool CAUSAL, bool BIAS>
__global__ __launch_bounds__(256) void attn_fwd_hd64(const bf16* __restrict__ Q, int64_t qsb, int64_t qst,
                                                     const bf16* __restrict__ K, int64_t ksb, int64_t kst,
                                                     const bf16* __restrict__ V, int64_t vsb, int64_t vst,
                                                     bf16* __restrict__ O, int64_t osb, int64_t ost, int H, int Lq,
                                                     int Lk, int nqb, const float* __restrict__ bias, int64_t bsb,
                                                     int64_t bsh, int64_t bsq) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_B];  // [buf][K, V]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int qb = wg % nqb, bh = wg / nqb;
  const int h = bh % H, b = bh / H;
  const int q0 = qb * 128;
  const int r = lane & 31, hh = lane >> 5;

  const bf16* Qp = Q + (int64_t)b * qsb + h * 64;
  const bf16* Kp = K + (int64_t)b * ksb + h * 64;
  const bf16* Vp = V + (int64_t)b * vsb + h * 64;

  // this lane's query row (clamped for loads; masked at the store)
  const int qi = q0 + wave * 32 + r;
  [[maybe_unused]] const int qi_ld = qi < Lq ? qi : Lq - 1;
  bf16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = *(const bf16x8*)(Qp + (int64_t)qi_ld * qst + s * 16 + hh * 8);

  int nT = (Lk + KV_TILE - 1) / KV_TILE;
  if (CAUSAL) {
    int qlast = q0 + 127;
    if (qlast > Lq - 1) qlast = Lq - 1;
    const int tl = qlast / KV_TILE + 1;  // tiles holding any key <= the block's last query
    if (tl < nT) nT = tl;
  }

  // register staging: thread t moves chunks (t, t + 256) of the 64 x 8 chunk grid of K and of V
  const int srow0 = tid >> 3, sch = tid & 7;
  bf16x8 kreg[2], vreg[2];
  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int key = t * KV_TILE + srow0 + i * 32;
      key = key < Lk ? key : Lk - 1;
      kreg[i] = *(const bf16x8*)(Kp + (int64_t)key * kst + sch * 8);
      vreg[i] = *(const bf16x8*)(Vp + (int64_t)key * vst + sch * 8);
    }
  };
  auto write_tile = [&](char* kbuf) {
    char* vbuf = kbuf + TILE_B;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = srow0 + i * 32;
      *(bf16x8*)(kbuf + row * 128 + swz_pos(row, sch) * 16) = kreg[i];
      *(bf16x8*)(vbuf + row * 128 + ((sch ^ (((row >> 1) & 1) << 2)) * 16)) = vreg[i];
    }
  };

  f32x16 oacc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { oacc[0][i] = 0.f; oacc[1][i] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;
  const float c = 0.125f * 1.4426950408889634f;  // 1/sqrt(64) * log2(e)

  // tr-read lane geometry (see header): 16-lane group g, lane-in-group i = 4*qq + pp
  const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;

  const bool wave_live = q0 + wave * 32 < Lq;
  const float* bias_row = BIAS ? bias + (int64_t)b * bsb + (int64_t)h * bsh + (int64_t)qi_ld * bsq : nullptr;
  const bool bias_vec = BIAS && !(((uintptr_t)bias | (uintptr_t)(bsb * 4) | (uintptr_t)(bsh * 4) | (uintptr_t)(bsq * 4)) & 15);
  load_tile(0);
  write_tile(smem);
  __syncthreads();

  // The tile body exists twice: tiles that need no mask at all (every key valid, below the causal diagonal for this
  // wave, no bias) skip the per-score compare / select work - a third of the VALU instructions of a tile, and this
  // kernel is VALU-bound (SQ counters at L = 1500: vector ALU active 79 % of the time, matrix pipe 31 %) - and only the
  // trailing tiles run the masked form.  Both forms stage K/V and hit the barrier identically.
#define PM_ATT_TILE(MASKED, PAR)                                                                                         \
    const char* kbuf = smem + (PAR) * 2 * TILE_B;                                                                        \
    const char* vbuf = kbuf + TILE_B;                                                                                    \
    if (t + 1 < nT) load_tile(t + 1);                                                                                    \
    /* work that cannot contribute is skipped per wave (the K/V staging and the barrier are not): a wave whose 32 */     \
    /* queries are all past Lq, a causal tile entirely above the wave's last query, and the second 32-key block of a */  \
    /* tail tile when it holds no valid key (L = 197: 5 of the last tile's 64 keys exist) */                             \
    const bool kb1 = !(MASKED) || t * KV_TILE + 32 < Lk;                                                                 \
    const bool skip = !wave_live || (CAUSAL && t * KV_TILE > q0 + wave * 32 + 31);                                       \
    if (!skip) {                                                                                                         \
    /* ---- S^T = K Q^T : two 32-key blocks */                                                                           \
    f32x16 sc[2];                                                                                                        \
_Pragma("unroll")                                                                                                        \
    for (int kb = 0; kb < 2; ++kb) {                                                                                     \
_Pragma("unroll")                                                                                                        \
      for (int i = 0; i < 16; ++i) sc[kb][i] = kb == 1 && !kb1 ? -1e30f : 0.f;                                           \
      if (kb == 1 && !kb1) continue;  /* second 32-key block holds no valid key (tail tile): skip its MFMAs */           \
      const int row = kb * 32 + r;                                                                                       \
_Pragma("unroll")                                                                                                        \
      for (int s = 0; s < 4; ++s) {                                                                                      \
        const bf16x8 kf = *(const bf16x8*)(kbuf + row * 128 + swz_pos(row, 2 * s + hh) * 16);                            \
        sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sc[kb], 0, 0, 0);                                    \
      }                                                                                                                  \
    }                                                                                                                    \
    /* ---- masks (key tail / causal), scale into the log2 domain */                                                     \
    const int key_base = t * KV_TILE + 4 * hh;                                                                           \
    const bool tail = (t + 1) * KV_TILE > Lk;                                                                            \
    const bool diag = CAUSAL && ((t + 1) * KV_TILE - 1 > q0 + wave * 32);                                                \
    float mx = -1e30f;                                                                                                   \
    f32x4 bq[2][4];  /* this lane's 8 groups of 4 consecutive keys: 16-byte loads when the bias rows allow it */          \
    if constexpr (BIAS) {                                                                                                \
_Pragma("unroll")                                                                                                        \
      for (int kb = 0; kb < 2; ++kb)                                                                                     \
_Pragma("unroll")                                                                                                        \
        for (int gq = 0; gq < 4; ++gq) {                                                                                 \
          const int key = key_base + kb * 32 + 8 * gq;                                                                   \
          if (bias_vec && key + 3 < Lk) {                                                                                \
            bq[kb][gq] = *(const f32x4*)(bias_row + key);                                                                \
          } else {                                                                                                       \
_Pragma("unroll")                                                                                                        \
            for (int e = 0; e < 4; ++e) bq[kb][gq][e] = key + e < Lk ? bias_row[key + e] : 0.f;                          \
          }                                                                                                              \
        }                                                                                                                \
    }                                                                                                                    \
_Pragma("unroll")                                                                                                        \
    for (int kb = 0; kb < 2; ++kb)                                                                                       \
_Pragma("unroll")                                                                                                        \
      for (int i = 0; i < 16; ++i) {                                                                                     \
        if (kb == 1 && !kb1) continue;                                                                                   \
        float v = (MASKED) ? sc[kb][i] * c : sc[kb][i];  /* mask-free tiles keep RAW scores: max(c s) = c max(s) */      \
        if constexpr (BIAS) {  /* additive attn_bias[b, h, q, k] (strides may be 0 = broadcast), natural-log units */    \
          v = fmaf(bq[kb][i >> 2][i & 3], 1.4426950408889634f, v);  /* keys past Lk carry 0 and are masked below */      \
        }                                                                                                                \
        if ((MASKED) && (tail || diag)) {                                                                                \
          const int key = key_base + kb * 32 + (i & 3) + 8 * (i >> 2);                                                   \
          if (key >= Lk || (CAUSAL && key > qi)) v = -1e30f;                                                             \
        }                                                                                                                \
        sc[kb][i] = v;                                                                                                   \
        mx = fmaxf(mx, v);                                                                                               \
      }                                                                                                                  \
    mx = ah_max_xor32(mx);                                                                                               \
    if (!(MASKED)) mx *= c;                                                                                              \
    const float m_new = fmaxf(m_run, mx);                                                                                \
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);                                                           \
    m_run = m_new;                                                                                                       \
    f32x2 ps2 = {0.f, 0.f};  /* two chains for the row sum: two scores per packed fma / add (the kernel is VALU-bound) */  \
_Pragma("unroll")                                                                                                        \
    for (int kb = 0; kb < 2; ++kb)                                                                                       \
_Pragma("unroll")                                                                                                        \
      for (int i = 0; i < 16; i += 2) {                                                                                  \
        if (kb == 1 && !kb1) continue;                                                                                   \
        /* mask-free: the scale rides in the exponent's fma, exp2(fma(s, c, -max)) - no multiply per score */            \
        const f32x2 s2 = {sc[kb][i], sc[kb][i + 1]};                                                                     \
        const f32x2 a = (MASKED) ? s2 - f32x2{m_new, m_new} : __builtin_elementwise_fma(s2, f32x2{c, c}, f32x2{-m_new, -m_new}); \
        f32x2 p;                                                                                                         \
        p[0] = __builtin_amdgcn_exp2f(a[0]);                                                                             \
        p[1] = __builtin_amdgcn_exp2f(a[1]);                                                                             \
        sc[kb][i] = p[0];                                                                                                \
        sc[kb][i + 1] = p[1];                                                                                            \
        ps2 += p;                                                                                                        \
      }                                                                                                                  \
    l_run = fmaf(l_run, alpha, ps2[0] + ps2[1]);                                                                         \
_Pragma("unroll")                                                                                                        \
    for (int i = 0; i < 16; ++i) { oacc[0][i] *= alpha; oacc[1][i] *= alpha; }                                           \
    /* ---- O^T += V^T P^T */                                                                                            \
_Pragma("unroll")                                                                                                        \
    for (int kb = 0; kb < 2; ++kb)                                                                                       \
_Pragma("unroll")                                                                                                        \
      for (int s = 0; s < 2; ++s) {                                                                                      \
        if (kb == 1 && !kb1) continue;                                                                                   \
        bf16x8 pf;                                                                                                       \
_Pragma("unroll")                                                                                                        \
        for (int j = 0; j < 8; ++j) pf[j] = (bf16)sc[kb][8 * s + j];                                                     \
        const int row = kb * 32 + 16 * s + 4 * (g >> 1) + qq;                                                            \
        const int flip = ((row >> 1) & 1) << 2;                                                                          \
_Pragma("unroll")                                                                                                        \
        for (int db = 0; db < 2; ++db) {                                                                                 \
          const int ch = db * 4 + 2 * (g & 1) + (pp >> 1);                                                               \
          const char* p0 = vbuf + row * 128 + ((ch ^ flip) * 16) + (pp & 1) * 8;                                         \
          const bf16x8 vf = tr_read_pair(p0, p0 + 8 * 128);                                                              \
          oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[db], 0, 0, 0);                                 \
        }                                                                                                                \
      }                                                                                                                  \
    }                                                                                                                    \
    if (t + 1 < nT) write_tile(smem + ((PAR) ^ 1) * 2 * TILE_B);                                                         \
    __syncthreads();                                                                                                     \
  /* end of PM_ATT_TILE */
  int n_plain = BIAS ? 0 : (Lk / KV_TILE < nT ? Lk / KV_TILE : nT);  // leading tiles whose 64 keys all exist
  if (CAUSAL) {  // ... and lie at or below this wave's FIRST query: (t + 1) * 64 - 1 <= q0 + wave * 32
    const int below = (q0 + wave * 32 + 1) / KV_TILE;
    n_plain = below < n_plain ? below : n_plain;
  }
  // the mask-free loop is unrolled by two so that the LDS buffer of a tile is a compile-time constant (the fragment and
  // transposed-read addresses become base + immediate instead of ~36 integer instructions per tile)
  int t = 0;
  for (; t + 1 < n_plain; ++t) {
    { PM_ATT_TILE(false, 0) }
    ++t;
    { PM_ATT_TILE(false, 1) }
  }
  if (t < n_plain) {  // t is even here
    { PM_ATT_TILE(false, 0) }
    ++t;
  }
  for (; t < nT; ++t) { PM_ATT_TILE(true, t & 1) }
#undef PM_ATT_TILE

  const float l_tot = ah_add_xor32(l_run);
  const float inv = 1.0f / l_tot;
  if (qi < Lq) {
    bf16* op = O + (int64_t)b * osb + (int64_t)qi * ost + h * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16)(oacc[db][4 * gq + j] * inv);
        *(bf16x4*)(op + db * 32 + 8 * gq + 4 * hh) = o;
      }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Short self-attention (Lq, Lk <= 256, no mask, no bias: ViT's 197 tokens).  At this length the tiled kernel above is
// bound by what a CU can pull in (two workgroups per head each stage the head's whole K and V: 467 MB at C2 against
// 309 MB read once, at the ~4.4 TB/s that 256 CUs ingest), not by its arithmetic.  So: ONE persistent workgroup per
// CU walks (batch, head) pairs; 8 waves x 32 queries cover the head, its K and V land in LDS once (LDS-DMA, no
// registers), double-buffered - the next head's 50 KB stream in while this one is computed - and a wave keeps ALL its
// scores in registers (NB blocks of 32 keys x 16 values): the softmax is exact in one pass (no running max, no O
// rescale), padding is per 32 keys (QK^T) and 16 keys (PV).  One barrier per head.  Outputs leave through a wave-private
// 4 KiB transposition so that a store instruction writes 8 full 128-byte rows instead of 32 16-byte fragments.
// Layouts (K image, V image, fragment and transposed reads, S^T / O^T accumulator geometry) are the tiled kernel's.
// Measured at C2 (B = 256, H = 12, L = 197; tools/attn_bench.py, then in the model): 110 -> 86 us standalone, 92 -> 78 us
// per layer inside ViT-B/16 (round 2); 82 -> 73 us standalone on one box in round 3 (profiles/r03/attention_head_stamps.txt).
// With the arithmetic switched off the kernel streams its 309 MB in 60 us (5.2 TB/s), with the loads switched off it computed
// for 65 us (vector ALU ~45 % busy, matrix pipe 19 %: SQ counters).  Round 3's in-kernel stamps (tools/attn_stamps.py) showed
// where the rest went - not into waiting for HBM (0.2 us per head) but into REQUESTING: an LDS-DMA piece, a load or a store costs
// the wave that issues it 100-200 ns in which it issues nothing else, all eight waves did their 15 behind the barrier, and
// QK^T ran as ds_read / wait / MFMA.  Hence, below: the query-less wave requests every K / V piece (PM_AH_SOLO), the partners of
// a SIMD make their remaining requests at different times (PM_AH_LATE), queries come as whole rows through the staging area
// (PM_AH_QLDS), K fragments are read two blocks ahead (PM_AH_KPF).  Tried and dropped: waves 4-7 delayed by the whole QK^T phase
// (91 us); four output stores kept in flight across the barrier (no change); every request spread between the arithmetic's
// blocks (no change: the time moves with the requests); wave 7 requesting with 64-bit lane addresses (it became the critical path).
//
// LDS-DMA that hipcc does not see.  With the builtin form it orders every later ds_read_b64_tr_b16 (an intrinsic without a
// memory operand: "may alias") behind a vmcnt(0) of its own - the next head's prefetch then never overlaps this head's
// arithmetic.  Hidden, the completion is ours to count: the vmcnt wait in
// front of the per-head barrier.  (cdna_hip_programming.md, inline-asm rules: M0 written in the statement that reads it.)
__device__ __forceinline__ void glds16_hidden(const void* gsrc, unsigned lds_dst_wave_base) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst_wave_base);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(dst)
               : "memory");
}

// In-kernel stamps (variant builds only: -DPM_AH_STAMPS=1; tools/attn_stamps.py): per workgroup and wave, its PM_AH_STAMP_IT-th head.
#ifndef PM_AH_STAMPS
#define PM_AH_STAMPS 0
#endif
#ifndef PM_AH_TRIM
#define PM_AH_TRIM 1
#endif
#ifndef PM_AH_SOLO
#define PM_AH_SOLO 1
#endif
#ifndef PM_AH_SOLO4
#define PM_AH_SOLO4 1
#endif
#ifndef PM_AH_LATE
#define PM_AH_LATE 1
#endif
#ifndef PM_AH_KPF
#define PM_AH_KPF 1
#endif
#ifndef PM_AH_QLDS
#define PM_AH_QLDS 1
#endif
#ifndef PM_AH_STAMP_IT
#define PM_AH_STAMP_IT 5
#endif
#if PM_AH_STAMPS
__device__ unsigned long long g_ah_stamps[256 * 8 * 8];
#define PM_AH_STAMP(i_)                                                                                              \
  do {                                                                                                               \
    if (it == PM_AH_STAMP_IT && lane == 0 && blockIdx.x < 256) g_ah_stamps[(blockIdx.x * 8 + wave) * 8 + (i_)] = wall_clock64(); \
  } while (0)
#else
#define PM_AH_STAMP(i_)
#endif

// the same with the address as uniform 64-bit base (SGPR pair) + 32-bit lane offset: no 64-bit vector arithmetic per request
__device__ __forceinline__ void glds16_hidden_s(const void* uniform_base, unsigned lane_off, unsigned lds_dst_wave_base) {
  unsigned keep;
  const uint64_t b = (uint64_t)(uintptr_t)uniform_base;
  // (readfirstlane returns int: without the unsigned casts the low word would be sign-extended over the high one)
  const uint64_t sb = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32)) << 32) |
                      (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst_wave_base);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(lane_off), "s"(sb), "s"(dst)
               : "memory");
}

// four pieces behind ONE write of M0: the instruction's immediate offset moves the LDS address and the global address alike
// (piece i of the group: + 1024 i), so lane offset i is passed less 1024 i (callers: off[i] >= 1024 i)
__device__ __forceinline__ void glds16_hidden_s4(const void* uniform_base, unsigned o0, unsigned o1, unsigned o2, unsigned o3,
                                                 unsigned lds_dst_wave_base) {
  unsigned keep;
  const uint64_t b = (uint64_t)(uintptr_t)uniform_base;
  const uint64_t sb = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32)) << 32) |
                      (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst_wave_base);
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %5\n\tglobal_load_lds_dwordx4 %2, %5 offset:1024\n\t"
      "global_load_lds_dwordx4 %3, %5 offset:2048\n\tglobal_load_lds_dwordx4 %4, %5 offset:3072\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(o0), "v"(o1 - 1024u), "v"(o2 - 2048u), "v"(o3 - 3072u), "s"(sb), "s"(dst)
      : "memory");
}

template <int NB>
__global__ __launch_bounds__(512) void attn_head_hd64(const bf16* __restrict__ Q, int64_t qsb, int64_t qst,
                                                      const bf16* __restrict__ K, int64_t ksb, int64_t kst,
                                                      const bf16* __restrict__ V, int64_t vsb, int64_t vst,
                                                      bf16* __restrict__ O, int64_t osb, int64_t ost, int H, int Lq,
                                                      int Lk, int nheads) {
  constexpr int ROWS = NB * 32, IMG = ROWS * 128, BUF = 2 * IMG;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF + 8 * 4096];  // [buf][K image, V image] + staging per wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int qi = wave * 32 + r;
  [[maybe_unused]] const int qi_ld = qi < Lq ? qi : Lq - 1;
  const bool live = wave * 32 < Lq;
  const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  const float c = 0.125f * 1.4426950408889634f;  // 1/sqrt(64) * log2(e)
  char* const stg = smem + 2 * BUF + wave * 4096;
  const unsigned lds0 = (unsigned)(uintptr_t)(PM_LDS char*)smem;

  // K / V rows of one head straight into LDS: NB pieces (8 rows of 128 bytes) per wave, the swizzles ride on the SOURCE
  // chunk; rows past Lk repeat row Lk - 1 (finite values under probabilities that are exactly 0)
#define PM_AH_PIECE(HEAD, BUFI, PI_)                                                                             \
  {                                                                                                              \
    const int b_ = (HEAD) / H, h_ = (HEAD) - b_ * H;                                                             \
    const bf16* Kp_ = K + (int64_t)b_ * ksb + h_ * 64;                                                           \
    const bf16* Vp_ = V + (int64_t)b_ * vsb + h_ * 64;                                                           \
    const unsigned base_ = lds0 + (BUFI) * BUF;                                                                  \
    const int pi = (PI_);                                                                                        \
    const bool isv = pi >= 4 * NB;                                                                               \
    const int piece = isv ? pi - 4 * NB : pi;                                                                    \
    const int row = piece * 8 + (lane >> 3), pos = lane & 7;                                                     \
    const int src = row < Lk ? row : Lk - 1;                                                                     \
    const bf16* g_ = !isv ? Kp_ + (int64_t)src * kst + swz_pos(row, pos) * 8                                       \
                          : Vp_ + (int64_t)src * vst + (pos ^ (((row >> 1) & 1) << 2)) * 8;                       \
    /* pieces wholly past the keys stay unrequested (PM_AH_TRIM): their K rows only meet scores that are overwritten with \
       -1e30, their V rows (16-key steps past Lk) are never read; whatever the image holds there, NaN patterns included */ \
    if (!PM_AH_TRIM || piece * 8 < (isv ? ((Lk + 15) & ~15) : Lk))                                               \
      glds16_hidden(g_, base_ + (isv ? IMG : 0) + piece * 1024);                                                 \
  }
#define PM_AH_ISSUE(HEAD, BUFI)                                                                                  \
  {                                                                                                              \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) PM_AH_PIECE(HEAD, BUFI, wave + 8 * i)                         \
  }
  // PM_AH_SOLO: with Lq <= 224 (ViT's 197) wave 7 owns no queries and requests EVERY piece of the next head - an LDS-DMA
  // request costs the wave that issues it ~200 cycles in which it issues nothing else (the stamps of tools/attn_stamps.py:
  // the 15 requests per wave behind the barrier held every wave 1.4-2.8 us of a 6.7 us head; with the pieces gone the
  // queries and stores take 0.16 us).  Uniform base + 32-bit lane offset keeps wave 7's own arithmetic per piece at 4 ops.
#define PM_AH_ISSUE_SOLO(HEAD, BUFI)                                                                             \
  {                                                                                                              \
    const int b_ = (HEAD) / H, h_ = (HEAD) - b_ * H;                                                             \
    const bf16* Kp_ = K + (int64_t)b_ * ksb + h_ * 64;                                                           \
    const bf16* Vp_ = V + (int64_t)b_ * vsb + h_ * 64;                                                           \
    const unsigned base_ = lds0 + (BUFI) * BUF;                                                                  \
    const int lrow = lane >> 3, pos = lane & 7;                                                                  \
    const unsigned kso = (unsigned)swz_pos(lrow, pos) * 16, vso = (unsigned)(pos ^ (((lrow >> 1) & 1) << 2)) * 16; \
    const int nk_ = PM_AH_TRIM ? (Lk + 7) >> 3 : 4 * NB, nv_ = PM_AH_TRIM ? ((Lk + 15) & ~15) >> 3 : 4 * NB;     \
    const unsigned ks2 = (unsigned)(kst * 2), vs2 = (unsigned)(vst * 2);                                         \
    int p_ = 0;                                                                                                  \
    /* swz_pos(row, pos) with row = 8 p_ + lrow: the piece's parity flips chunk bit 2 */                         \
    for (; PM_AH_SOLO4 && p_ + 4 <= nk_; p_ += 4)                                                                \
      glds16_hidden_s4(Kp_, (unsigned)min(p_ * 8 + lrow, Lk - 1) * ks2 + kso,                                    \
                       (unsigned)min(p_ * 8 + 8 + lrow, Lk - 1) * ks2 + (kso ^ 64u),                             \
                       (unsigned)min(p_ * 8 + 16 + lrow, Lk - 1) * ks2 + kso,                                    \
                       (unsigned)min(p_ * 8 + 24 + lrow, Lk - 1) * ks2 + (kso ^ 64u), base_ + p_ * 1024);        \
    for (; p_ < nk_; ++p_)                                                                                       \
      glds16_hidden_s(Kp_, (unsigned)min(p_ * 8 + lrow, Lk - 1) * ks2 + (kso ^ ((unsigned)(p_ & 1) << 6)), base_ + p_ * 1024); \
    for (p_ = 0; PM_AH_SOLO4 && p_ + 4 <= nv_; p_ += 4)                                                          \
      glds16_hidden_s4(Vp_, (unsigned)min(p_ * 8 + lrow, Lk - 1) * vs2 + vso,                                    \
                       (unsigned)min(p_ * 8 + 8 + lrow, Lk - 1) * vs2 + vso,                                     \
                       (unsigned)min(p_ * 8 + 16 + lrow, Lk - 1) * vs2 + vso,                                    \
                       (unsigned)min(p_ * 8 + 24 + lrow, Lk - 1) * vs2 + vso, base_ + IMG + p_ * 1024);          \
    for (; p_ < nv_; ++p_)                                                                                       \
      glds16_hidden_s(Vp_, (unsigned)min(p_ * 8 + lrow, Lk - 1) * vs2 + vso, base_ + IMG + p_ * 1024);           \
  }
#define PM_AH_LOADQ(HEAD, DST)                                                                                   \
  {                                                                                                              \
    const int b_ = (HEAD) / H, h_ = (HEAD) - b_ * H;                                                             \
    const bf16* Qp_ = Q + (int64_t)b_ * qsb + h_ * 64 + (int64_t)qi_ld * qst + hh * 8;                           \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) DST[s] = *(const bf16x8*)(Qp_ + s * 16);                       \
  }

#define PM_AH_STORE(HEAD) /* the four read-backs first (one LDS round trip), then the four stores */             \
  {                                                                                                              \
    const int b_ = (HEAD) / H, h_ = (HEAD) - b_ * H;                                                             \
    bf16* op_ = O + (int64_t)b_ * osb + h_ * 64 + (lane & 7) * 8 + (int64_t)(wave * 32 + (lane >> 3)) * ost;     \
    bf16x8 o_[4];                                                                                                \
    _Pragma("unroll") for (int k4 = 0; k4 < 4; ++k4) {                                                           \
      const int row = k4 * 8 + (lane >> 3);                                                                      \
      o_[k4] = *(const bf16x8*)(stg + row * 128 + (((lane & 7) ^ (row & 7)) * 16));                              \
    }                                                                                                            \
    _Pragma("unroll") for (int k4 = 0; k4 < 4; ++k4)                                                             \
      if (wave * 32 + k4 * 8 + (lane >> 3) < Lq) *(bf16x8*)(op_ + (int64_t)(k4 * 8) * ost) = o_[k4];             \
  }

  // Queries.  PM_AH_QLDS 1: a wave's 32 query rows come in as four LDS-DMA pieces of 8 whole rows (full 128-byte lines, 8 per
  // request) into its own 4 KiB staging area - which is free between the previous head's stores (right behind the barrier) and
  // this head's transposition - and are read from there in the K fragment's layout when the head's arithmetic is done.  0: every
  // lane loads its own row's 16-byte chunks (four requests of 32 partial lines each: 128 line accesses per wave and head against
  // 56 for its K / V pieces and 32 for its stores - the stamps showed the CU's memory pipe, not HBM, holding the waves).
#define PM_AH_QISSUE(HEAD)                                                                                       \
  {                                                                                                              \
    const int b_ = (HEAD) / H, h_ = (HEAD) - b_ * H;                                                             \
    const bf16* Qp_ = Q + (int64_t)b_ * qsb + h_ * 64;                                                           \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                              \
      const int row = j * 8 + (lane >> 3), pos = lane & 7;                                                       \
      const int src = wave * 32 + row < Lq ? wave * 32 + row : Lq - 1;                                           \
      glds16_hidden(Qp_ + (int64_t)src * qst + swz_pos(row, pos) * 8, lds0 + 2 * BUF + wave * 4096 + j * 1024);  \
    }                                                                                                            \
  }
#define PM_AH_QREAD(DST)                                                                                         \
  {                                                                                                              \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) DST[s] = *(const bf16x8*)(stg + r * 128 + swz_pos(r, 2 * s + hh) * 16); \
  }

  int head = blockIdx.x;
  if (head >= nheads) return;
  bf16x8 qf[4];
  // a wave requests NB pieces per head, less at most one K and one V piece trimmed past the keys: with QW requests allowed
  // in flight everything older than its pieces has landed
  [[maybe_unused]] constexpr int QW = PM_AH_TRIM ? (NB > 2 ? NB - 2 : 0) : NB;
  // (Lk >= 32: every clamped row offset of a group of four stays >= the 3 KiB the instruction offsets add)
  const bool solo = PM_AH_SOLO && Lq <= 224 && Lk >= 32 && (int64_t)Lk * max(kst, vst) < (1ll << 30);
#define PM_AH_KV(HEAD, BUFI)                                                                                     \
  {                                                                                                              \
    if (solo) {                                                                                                  \
      if (wave == 7) PM_AH_ISSUE_SOLO(HEAD, BUFI)                                                                \
    } else                                                                                                       \
      PM_AH_ISSUE(HEAD, BUFI)                                                                                    \
  }
#if PM_AH_QLDS
  if (live) PM_AH_QISSUE(head)
  PM_AH_KV(head, 0)
  if (live) {
    if (solo) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // a wave with queries has nothing else in flight
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QW) : "memory");  // in order: the queries are older than the K / V pieces
    PM_AH_QREAD(qf)
  }
#else
  bf16x8 qn[4];
  PM_AH_KV(head, 0)
  PM_AH_LOADQ(head, qf)
#pragma unroll
  for (int s = 0; s < 4; ++s) qn[s] = qf[s];
#endif
  for (int it = 0; head < nheads; ++it, head += gridDim.x) {
    const int bufi = it & 1;
    PM_AH_STAMP(0);
    // this wave's pieces of `head` (and its queries) have landed; everything else it has in flight is a head old
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PM_AH_STAMP(1);
    __builtin_amdgcn_s_barrier();  // ... everyone's have; and everyone is done with the other buffer
    PM_AH_STAMP(2);
    const int nxt = head + gridDim.x;
    const bool more = nxt < nheads, prev = live && it > 0;
    // The PREVIOUS head's rows leave from the wave's staging area (their stores drain under this head's arithmetic instead
    // of in front of the next wait) and the next head's queries are requested: by waves 0-3 right here, by waves 4-7 - their
    // partners on the SIMDs - behind their QK^T (PM_AH_LATE 1; 0 = all here).  A request costs its wave ~100 ns in which it
    // issues nothing else, and with both partners in that phase the SIMD stood idle (stamps: waves 4-6 left it 2.0 us after
    // the barrier, waves 0-3 1.2 us).
#if PM_AH_QLDS
#define PM_AH_SIDE()                                                                                             \
  {                                                                                                              \
    if (prev) PM_AH_STORE(head - (int)gridDim.x)                                                                 \
    if (live && more) PM_AH_QISSUE(nxt) /* behind the stores' reads of the staging area */                       \
  }
#else
#define PM_AH_SIDE()                                                                                             \
  {                                                                                                              \
    if (live && more) PM_AH_LOADQ(nxt, qn)                                                                       \
    if (prev) PM_AH_STORE(head - (int)gridDim.x)                                                                 \
  }
#endif
    const bool late = PM_AH_LATE && wave >= 4;
    if (more) PM_AH_KV(nxt, bufi ^ 1)
    if (!late) PM_AH_SIDE()
    PM_AH_STAMP(3);
    if (live) {
      const char* kimg = smem + bufi * BUF;
      const char* vimg = kimg + IMG;
      // ---- S^T = K Q^T, all NB blocks
      f32x16 sc[NB];
#if PM_AH_KPF
      // The K fragments of blocks kb + 1 (and kb + 2's after block kb's MFMAs) are in flight while block kb's MFMAs run.  Left
      // to hipcc the loop was  ds_read; s_waitcnt lgkmcnt(0); v_mfma  28 times over with ONE fragment register - the kernel
      // sits at the register limit and its scheduler trades every read-ahead for pressure - i.e. an LDS round trip per MFMA
      // (1.1 us for 0.37 us of matrix pipe).  So the reads are inline asm it does not track, and each block's MFMAs sit
      // behind a counted wait that takes the fragments as operands (nothing that uses them can be scheduled above it).
      // swz_pos(32 kb + r, c) does not depend on kb: four lane addresses, the block rides in the instruction offset.
      bf16x8 kfr[2][4];
      uint32_t ka[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) ka[s] = (uint32_t)(uintptr_t)(PM_LDS const char*)kimg + (uint32_t)(r * 128 + swz_pos(r, 2 * s + hh) * 16);
#define PM_AH_KREAD(KB_)                                                                                         \
  _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                                  \
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kfr[(KB_) & 1][s]) : "v"(ka[s]), "n"((KB_) * 4096));
#define PM_AH_KWAIT(N_, KB_)                                                                                     \
  asm volatile("s_waitcnt lgkmcnt(%4)"                                                                           \
               : "+v"(kfr[(KB_) & 1][0]), "+v"(kfr[(KB_) & 1][1]), "+v"(kfr[(KB_) & 1][2]), "+v"(kfr[(KB_) & 1][3])      \
               : "n"(N_));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the counter starts from nothing of hipcc's own
      PM_AH_KREAD(0)
      if (NB > 1) PM_AH_KREAD(1)
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) sc[kb][i] = 0.f;
        if (kb + 1 < NB) { PM_AH_KWAIT(4, kb) } else { PM_AH_KWAIT(0, kb) }
#pragma unroll
        for (int s = 0; s < 4; ++s) sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[kb & 1][s], qf[s], sc[kb], 0, 0, 0);
        if (kb + 2 < NB) PM_AH_KREAD(kb + 2)
        __builtin_amdgcn_sched_barrier(0);
      }
#undef PM_AH_KREAD
#undef PM_AH_KWAIT
#else
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) sc[kb][i] = 0.f;
        const int row = kb * 32 + r;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const bf16x8 kf = *(const bf16x8*)(kimg + row * 128 + swz_pos(row, 2 * s + hh) * 16);
          sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sc[kb], 0, 0, 0);
        }
      }
#endif
      if (late) PM_AH_SIDE()
      if (Lk < ROWS) {  // keys past Lk: last block only (the dispatcher picks the smallest NB that covers Lk)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = (NB - 1) * 32 + 4 * hh + (i & 3) + 8 * (i >> 2);
          if (key >= Lk) sc[NB - 1][i] = -1e30f;
        }
      }
      PM_AH_STAMP(4);
      // ---- exact softmax: max of the raw scores, the scale rides in the exponent's fma
      float mx = -1e30f;
#pragma unroll
      for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sc[kb][i]);
      mx = ah_max_xor32(mx);
      const float mc = mx * c;
      PM_AH_STAMP(5);
      f32x2 ps2 = {0.f, 0.f};  // two chains for the row sum
      f32x16 oacc[2];
#pragma unroll
      for (int i = 0; i < 16; ++i) { oacc[0][i] = 0.f; oacc[1][i] = 0.f; }
      // ---- O^T = V^T P^T in 16-key steps, each block's exponentials right in front of its MFMAs
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          // the last block's groups of 4 registers whose keys (8 gq + 0..7 of the block, both lane halves) do not exist
          // carry no exponentials at all (wave-uniform): L = 197 evaluates 4 of that block's 16
          const bool none = kb == NB - 1 && kb * 32 + 8 * gq >= Lk;
#pragma unroll
          for (int e = 0; e < 4; e += 2) {  // two scores per packed fma / add (v_pk_fma_f32, v_pk_add_f32)
            const f32x2 a = __builtin_elementwise_fma(f32x2{sc[kb][4 * gq + e], sc[kb][4 * gq + e + 1]}, f32x2{c, c}, f32x2{-mc, -mc});
            f32x2 p;
            p[0] = none ? 0.f : __builtin_amdgcn_exp2f(a[0]);
            p[1] = none ? 0.f : __builtin_amdgcn_exp2f(a[1]);
            sc[kb][4 * gq + e] = p[0];
            sc[kb][4 * gq + e + 1] = p[1];
            ps2 += p;
          }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (kb == NB - 1 && (kb * 32 + 16 * s) >= Lk) continue;  // 16 keys that do not exist (wave-uniform)
          bf16x8 pf;
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[j] = (bf16)sc[kb][8 * s + j];
          const int row = kb * 32 + 16 * s + 4 * (g >> 1) + qq;
          const int flip = ((row >> 1) & 1) << 2;
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            const int ch = db * 4 + 2 * (g & 1) + (pp >> 1);
            const char* p0 = vimg + row * 128 + ((ch ^ flip) * 16) + (pp & 1) * 8;
            const bf16x8 vf = tr_read_pair(p0, p0 + 8 * 128);
            oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[db], 0, 0, 0);
          }
        }
      }
#if PM_AH_QLDS
      if (more) {  // the next head's queries, requested a head's arithmetic ago
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next head's pieces, the previous head's stores, these queries
        PM_AH_QREAD(qf)
      }
#endif
      PM_AH_STAMP(6);
      const float ps = ps2[0] + ps2[1];
      const float inv = 1.0f / ah_add_xor32(ps);  // (a + b on both sides: the same bits as ps + shfl_xor(ps))
      // ---- O^T (query on the lane) -> row-major rows through the wave's own 4 KiB: chunk c of query row q at c ^ (q & 7)
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          // aligned register pairs spelled out: two packed multiplies and two packed conversions per four values (hipcc's own
          // pairing took (1, 2) and patched 0 and 3 around it: 13 instructions)
          const f32x2 lo = f32x2{oacc[db][4 * gq], oacc[db][4 * gq + 1]} * f32x2{inv, inv};
          const f32x2 hi = f32x2{oacc[db][4 * gq + 2], oacc[db][4 * gq + 3]} * f32x2{inv, inv};
          bf16x4 o;
          o[0] = (bf16)lo[0]; o[1] = (bf16)lo[1]; o[2] = (bf16)hi[0]; o[3] = (bf16)hi[1];
          *(bf16x4*)(stg + r * 128 + (((db * 4 + gq) ^ (r & 7)) * 16) + hh * 8) = o;
        }
      PM_AH_STAMP(7);
    }
#if !PM_AH_QLDS
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = qn[s];
#endif
  }
  if (live) PM_AH_STORE(head - (int)gridDim.x)  // the last head of this workgroup
#undef PM_AH_STORE
#undef PM_AH_PIECE
#undef PM_AH_ISSUE
#undef PM_AH_ISSUE_SOLO
#undef PM_AH_KV
#undef PM_AH_SIDE
#undef PM_AH_LOADQ
#undef PM_AH_QISSUE
#undef PM_AH_QREAD
}

}  // namespace

#if PM_AH_STAMPS
extern "C" int pm_debug_ah_stamps(void* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ah_stamps), sizeof(unsigned long long) * 256 * 8 * 8) == hipSuccess ? 0 : 1;
}
#endif

static int attention_impl(const void* q, int64_t q_stride_b, int64_t q_stride_t, const void* k, int64_t k_stride_b,
                          int64_t k_stride_t, const void* v, int64_t v_stride_b, int64_t v_stride_t, void* o,
                          int64_t o_stride_b, int64_t o_stride_t, int64_t B, int64_t H, int64_t Lq, int64_t Lk, int causal,
                          const float* bias, int64_t bsb, int64_t bsh, int64_t bsq, void* stream) {
  if (!q || !k || !v || !o || B < 0 || H <= 0 || Lq < 0 || Lk <= 0) return PM_EINVAL;
  if (B == 0 || Lq == 0) return PM_OK;
  if ((q_stride_t | k_stride_t | v_stride_t | q_stride_b | k_stride_b | v_stride_b) % 8) return PM_EALIGN;
  if ((o_stride_t | o_stride_b) % 4) return PM_EALIGN;
  if (q_stride_t < H * 64 || k_stride_t < H * 64 || v_stride_t < H * 64 || o_stride_t < H * 64) return PM_EINVAL;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) return PM_EALIGN;
  if ((uintptr_t)o & 7) return PM_EALIGN;
  if (Lq > (1 << 24) || Lk > (1 << 24)) return PM_EINVAL;
  if (bias && (bsb < 0 || bsh < 0 || bsq < Lk)) return PM_EINVAL;
  const int nqb = (int)((Lq + 127) / 128);
  const int64_t nblk = B * H * nqb;
  if (nblk > 0x7fffffff) return PM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
#define PM_ATT(C_, B_)                                                                                                   \
  hipLaunchKernelGGL((attn_fwd_hd64<C_, B_>), dim3((unsigned)nblk), dim3(256), 0, st, (const bf16*)q, q_stride_b, q_stride_t, \
                     (const bf16*)k, k_stride_b, k_stride_t, (const bf16*)v, v_stride_b, v_stride_t, (bf16*)o, o_stride_b,  \
                     o_stride_t, (int)H, (int)Lq, (int)Lk, nqb, bias, bsb, bsh, bsq)
  static const bool short_ok = [] { const char* e = getenv("PM_ATTN_SHORT"); return !e || atoi(e) != 0; }();
  if (short_ok && !causal && !bias && Lk <= 256 && Lq <= 256 && !((o_stride_t | o_stride_b) % 8) && !((uintptr_t)o & 15)) {
    // short self-attention: one persistent workgroup per CU walks the heads (K / V of a head in LDS once, one-pass softmax)
    static const int cus = [] {
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
      return n;
    }();
    const int64_t nheads = B * H;
    if (nheads > 0x7fffffff) return PM_EINVAL;
    const unsigned grid = (unsigned)(nheads < cus ? nheads : cus);
    const int nb = (int)((Lk + 31) / 32);
#define PM_ATTS(NB_)                                                                                                     \
  hipLaunchKernelGGL((attn_head_hd64<NB_>), dim3(grid), dim3(512), 0, st, (const bf16*)q, q_stride_b, q_stride_t,          \
                     (const bf16*)k, k_stride_b, k_stride_t, (const bf16*)v, v_stride_b, v_stride_t, (bf16*)o, o_stride_b, \
                     o_stride_t, (int)H, (int)Lq, (int)Lk, (int)nheads)
    switch (nb) {
      case 1: PM_ATTS(1); break;
      case 2: PM_ATTS(2); break;
      case 3: PM_ATTS(3); break;
      case 4: PM_ATTS(4); break;
      case 5: PM_ATTS(5); break;
      case 6: PM_ATTS(6); break;
      case 7: PM_ATTS(7); break;
      default: PM_ATTS(8); break;
    }
#undef PM_ATTS
    PM_CHECK_LAUNCH();
    return PM_OK;
  }
  if (causal && bias) PM_ATT(true, true);
  else if (causal) PM_ATT(true, false);
  else if (bias) PM_ATT(false, true);
  else PM_ATT(false, false);
#undef PM_ATT
  PM_CHECK_LAUNCH();
  return PM_OK;
}

extern "C" int pm_attention_bf16(const void* q, int64_t q_stride_b, int64_t q_stride_t, const void* k,
                                 int64_t k_stride_b, int64_t k_stride_t, const void* v, int64_t v_stride_b,
                                 int64_t v_stride_t, void* o, int64_t o_stride_b, int64_t o_stride_t, int64_t B,
                                 int64_t H, int64_t Lq, int64_t Lk, int causal, void* stream) {
  return attention_impl(q, q_stride_b, q_stride_t, k, k_stride_b, k_stride_t, v, v_stride_b, v_stride_t, o, o_stride_b,
                        o_stride_t, B, H, Lq, Lk, causal, nullptr, 0, 0, 0, stream);
}

extern "C" int pm_attention_bias_bf16(const void* q, int64_t q_stride_b, int64_t q_stride_t, const void* k,
                                      int64_t k_stride_b, int64_t k_stride_t, const void* v, int64_t v_stride_b,
                                      int64_t v_stride_t, void* o, int64_t o_stride_b, int64_t o_stride_t, int64_t B,
                                      int64_t H, int64_t Lq, int64_t Lk, int causal, const float* bias,
                                      int64_t bias_stride_b, int64_t bias_stride_h, int64_t bias_stride_q, void* stream) {
  if (!bias) return PM_EINVAL;
  return attention_impl(q, q_stride_b, q_stride_t, k, k_stride_b, k_stride_t, v, v_stride_b, v_stride_t, o, o_stride_b,
                        o_stride_t, B, H, Lq, Lk, causal, bias, bias_stride_b, bias_stride_h, bias_stride_q, stream);
}
