// common.h - shared device helpers for the gfx950 kernels of libpm_mi355x.so.
// CDNA4 only: 64-wide wavefronts, MFMA 16x16x32 / 32x32x16 bf16, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pm_mi355x.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define PM_LDS __attribute__((address_space(3)))
#define PM_GLOBAL __attribute__((address_space(1)))

// LayerNorm folded into the GEMMs around it (see linear_bf16.hip, "LayerNorm fold"): all three pointers may be null.
// Sum over the 8 consecutive lanes of an aligned 8-lane group, result in every lane: two quad_perm DPP steps
// (xor 1, xor 2) and row_half_mirror (lane i <-> 7 - i crosses the two quads).  VALU only.
__device__ __forceinline__ float sum8_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  return v;
}

struct PmLnFold {
  const float* stats;  // (M, 2): mean, rstd of each input row -> the epilogue computes rstd * (acc - mean * s[n]) + bias[n]
  const float* s;      // (N): column sums of the gamma-folded weight
  float* row_out;      // (M, N / 64, 2): per 64-feature block (sum, sum of squares) of the bf16-rounded OUTPUT rows
};

#define PM_CHECK_LAUNCH()                                  \
  do {                                                     \
    if (hipGetLastError() != hipSuccess) return PM_ELAUNCH; \
  } while (0)

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }

// async global -> LDS copy of 16 B per lane; the LDS destination is wave-uniform base + lane*16
#ifndef PM_GLDS_AUX
#define PM_GLDS_AUX 0  // cache-policy bits of the LDS-DMA loads: 1 = sc0, 2 = nt, 16 = sc1 (experiments; 0 in the product)
#endif
// Output rows of the large-M GEMMs: 1 = non-temporal stores (the L2 does not keep what the next kernel re-reads from HBM / the
// Infinity Cache anyway, so the operand panels keep their lines)
#ifndef PM_Y_NT
#define PM_Y_NT 1
#endif
template <typename V>
__device__ __forceinline__ void store_y(V* p, V v) {
#if PM_Y_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((const PM_GLOBAL void*)gsrc, (PM_LDS void*)lds_dst_wave_base, 16, 0, PM_GLDS_AUX);
}
// per-operand cache policy of the large-M GEMMs' LDS-DMA loads (token rows / weight rows)
#ifndef PM_GLDS_X_AUX
#define PM_GLDS_X_AUX 0
#endif
#ifndef PM_GLDS_W_AUX
#define PM_GLDS_W_AUX 0
#endif
template <int AUX>
__device__ __forceinline__ void glds16_aux(const void* gsrc, void* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((const PM_GLOBAL void*)gsrc, (PM_LDS void*)lds_dst_wave_base, 16, 0, AUX);
}

// 16-byte global store / load with sc1 (L1 bypassed, agent-coherent): the payload side of a workgroup-to-workgroup
// hand-off that uses an agent-scope ticket instead of cache-wide fences (MI355X_MICROARCH.md, inter-workgroup visibility).
// The load returns after its own s_waitcnt vmcnt(0) - the compiler does not track asm loads.
__device__ __forceinline__ void store_sc1_x4(f32x4* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 load_sc1_x4(const f32x4* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// The same load WITHOUT its wait, and the wait that publishes up to four of them: a combine loop that waits per load pays one L2
// round trip per partial (the K-split combine: kparts of them in series), this pays one.  The registers pass through the wait
// as in/out operands so that nothing that consumes them can be scheduled ahead of it.
__device__ __forceinline__ f32x4 load_sc1_x4_async(const f32x4* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void wait_loads(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}

__device__ __forceinline__ void wait_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// erf with |error| <= 1.5e-7 (Abramowitz & Stegun 7.1.26): one rcp, one exp, five fma.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __expf(-ax * ax);
  const float r = fmaf(-p * t, e, 1.0f);
  return copysignf(r, x);
}

// erf-GELU without transcendentals for bf16 outputs: Phi(x) - 1/2 = x q(x^2) with q a degree-8 minimax polynomial
// on |x| <= 4.5 (x clamped beyond: Phi(4.5) = 1 - 3.4e-6).  |gelu_poly(x) - x Phi(x)| <= 3.7e-5 for all x, i.e.
// >= 100x below the bf16 rounding of the result wherever |result| > 1e-2.  11 VALU ops, all packable; the erf_fast
// form costs ~16 including an exp and an rcp at quarter rate - the fc1 epilogue is VALU-bound, so this matters.
__device__ __forceinline__ float gelu_poly(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -4.5f, 4.5f);
  const float t = fmaf(xc * xc, 2.0f / 20.25f, -1.0f);
  float q = 3.353692146e-03f;
  q = fmaf(q, t, -9.328538250e-03f);
  q = fmaf(q, t, 1.220852128e-02f);
  q = fmaf(q, t, -1.674404426e-02f);
  q = fmaf(q, t, 2.762940359e-02f);
  q = fmaf(q, t, -4.055576763e-02f);
  q = fmaf(q, t, 5.481856801e-02f);
  q = fmaf(q, t, -7.717196008e-02f);
  q = fmaf(q, t, 1.569021127e-01f);
  return x * fmaf(xc, q, 0.5f);
}

// The same polynomial on two values at once: v_pk_mul_f32 / v_pk_fma_f32 halve the instruction count of the fc1
// epilogue's activation (only the clamp has no packed form).  Bit-identical to gelu_poly per element.
__device__ __forceinline__ f32x2 gelu_poly2(f32x2 x) {
  f32x2 xc;
  xc[0] = __builtin_amdgcn_fmed3f(x[0], -4.5f, 4.5f);
  xc[1] = __builtin_amdgcn_fmed3f(x[1], -4.5f, 4.5f);
  const f32x2 k1 = {2.0f / 20.25f, 2.0f / 20.25f}, m1 = {-1.0f, -1.0f}, half = {0.5f, 0.5f};
  const f32x2 t = __builtin_elementwise_fma(xc * xc, k1, m1);
  f32x2 q = {3.353692146e-03f, 3.353692146e-03f};
  q = __builtin_elementwise_fma(q, t, f32x2{-9.328538250e-03f, -9.328538250e-03f});
  q = __builtin_elementwise_fma(q, t, f32x2{1.220852128e-02f, 1.220852128e-02f});
  q = __builtin_elementwise_fma(q, t, f32x2{-1.674404426e-02f, -1.674404426e-02f});
  q = __builtin_elementwise_fma(q, t, f32x2{2.762940359e-02f, 2.762940359e-02f});
  q = __builtin_elementwise_fma(q, t, f32x2{-4.055576763e-02f, -4.055576763e-02f});
  q = __builtin_elementwise_fma(q, t, f32x2{5.481856801e-02f, 5.481856801e-02f});
  q = __builtin_elementwise_fma(q, t, f32x2{-7.717196008e-02f, -7.717196008e-02f});
  q = __builtin_elementwise_fma(q, t, f32x2{1.569021127e-01f, 1.569021127e-01f});
  return x * __builtin_elementwise_fma(xc, q, half);
}

// activation of four values (the GEMM epilogues' unit): packed for GELU, element-wise otherwise
template <int ACT>
__device__ __forceinline__ f32x4 apply_act4(f32x4 v);

// activation table of MLP (pytorch_models/transformer.py:60-65); PRECISE selects libm erff/tanhf.
template <int ACT, bool PRECISE>
__device__ __forceinline__ float apply_act(float x) {
  if constexpr (ACT == PM_ACT_GELU) {
    if constexpr (PRECISE) return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
    else return gelu_poly(x);
  } else if constexpr (ACT == PM_ACT_GELU_TANH) {
    const float u = 0.7978845608028654f * fmaf(0.044715f * x * x, x, x);
    return 0.5f * x * (1.0f + tanhf(u));
  } else if constexpr (ACT == PM_ACT_RELU) {
    return fmaxf(x, 0.0f);
  } else if constexpr (ACT == PM_ACT_SILU) {
    if constexpr (PRECISE) return x / (1.0f + expf(-x));
    else return x / (1.0f + __expf(-x));
  } else {
    return x;
  }
}

template <int ACT>
__device__ __forceinline__ f32x4 apply_act4(f32x4 v) {
  if constexpr (ACT == PM_ACT_GELU) {
    const f32x2 a = gelu_poly2(f32x2{v[0], v[1]}), b = gelu_poly2(f32x2{v[2], v[3]});
    return f32x4{a[0], a[1], b[0], b[1]};
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = apply_act<ACT, false>(v[r]);
    return v;
  }
}

// Workgroup-id remap so that each XCD (blocks b, b+8, b+16, ... share one) walks a CONTIGUOUS range
// of tile ids: neighbouring tiles then hit the same per-XCD L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// LDS tile with 128-byte rows (64 bf16) in 16-byte chunks; chunk c of row r lives at position
// c ^ ((r >> 1) & 7): conflict-free for ds_read_b128 fragment reads of both the 16-row x 4-chunk
// (MFMA 16x16x32) and 32-row x 2-chunk (MFMA 32x32x16) shapes.
__device__ __forceinline__ int swz_pos(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
