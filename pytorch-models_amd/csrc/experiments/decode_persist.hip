// decode_persist.hip - every layer of one KV-cached decode step in ONE persistent launch (pm_dec_layers).
//
// The launch-per-stage form of the step (decode.hip, generate.py "launches") is a chain of ~6 dependent launches per layer:
// at batch 32 each is 4-10 us of which ~1.5 us is the kernel boundary and ~2 us the first round trip for weights that do
// not depend on anything, and the HBM-bound cross-attention blocks (0.8 GB of K/V per step) cannot start their stream
// before the stage in front of them has drained.  Here one 512-thread workgroup per CU walks the stages of all layers
//   S0 self block (b, h)      LayerNorm -> q, k, v of head h -> cache append -> softmax(q K^T / 8) V      (transformer.py:98)
//   S1 out_proj + residual    16-feature x 16-row tiles, K split over the 4 waves of a half workgroup      (transformer.py:53)
//   S2 cross block (b, h)     LayerNorm -> q of head h -> attention over the packed cross K/V              (transformer.py:99)
//   S3 out_proj + residual
//   S4 linear1 + GELU         LayerNorm inside                                                             (transformer.py:59-66)
//   S5 linear2 + residual     K split over workgroups, combined by the last part to finish (ticket)
// and the stages hand their activations (fp32, <= 256 KB per stage) to each other through L2: every handed-off byte is
// stored and loaded with sc1, a producer's waves drain (vmcnt(0)), meet at a workgroup barrier, then ONE lane adds to the
// agent-scope arrival counter of (stage, 16-row tile); a consumer polls that counter with sc1 loads, meets its
// workgroup at a barrier and only then loads (MI355X_MICROARCH.md, inter-workgroup visibility, first row of the table of
// measured hand-offs; cdna_hip_programming.md Guideline 16).  No grid barrier, no fence, nothing placement-dependent.
// What a stage can request before its inputs exist - its weights, LayerNorm vectors, and for the attention blocks the
// K (and the head of the V) stream of its (sequence, head) - is requested BEFORE the poll, so it travels during the
// stage in front.  Waits are monotone (counter >= epoch * producers, epoch = position + 1: counters are zeroed by the
// caller at reset only) and every spin is bounded: on a timeout the workgroup sets *err and goes on, so a launch always
// drains; later launches return at once while *err is set and the host raises.
// The arithmetic is the launch form's: activations fp32 (x split into three bf16 terms for the MFMA: exact), weights and
// K/V caches bf16, fixed summation orders (no atomics on data): graph replay == eager == second run, bit for bit.
#include <mutex>

#include "../common.h"
#include "../../../include/pm_mi355x_experiments.h"

namespace {

constexpr int PS_THREADS = 512;
constexpr int PS_MAXK = 2048;  // keys an attention task can park scores for (S and Tmax)
constexpr int PS_MAXL = 48;    // layers (the table is staged in LDS)
constexpr unsigned PS_SPIN_MAX = 1u << 17;

struct PsArgs {
  const pm_dec_layer_t* tab;
  int n_layers, B, d, H, S, Tmax, ksplit, ldh, hid, act;
  const int* pos_ptr;
  float *x, *att, *h;
  int* cnt;  // (n_layers * 6 stages) x 4 row tiles arrival counters
  float* ks_ws;
  int* ks_tick;
  int* err;
};

__device__ __forceinline__ void lds_barrier() {  // LDS traffic only: global loads stay in flight across it
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}
__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ldf_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stf_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
struct Buf {  // raw buffer over an activation array: 16-byte sc1 loads / stores the compiler counts in vmcnt
  __amdgpu_buffer_rsrc_t r;
  __device__ __forceinline__ Buf(const void* p, int64_t bytes)
      : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000)) {}
  __device__ __forceinline__ f32x4 ld(int byte_off) const {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
  }
  __device__ __forceinline__ void st(int byte_off, f32x4 v) const {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, 16);
  }
};

// one lane polls; monotone counter.  false = gave up (err set): the caller goes on with whatever is there
__device__ __forceinline__ bool ps_wait(const int* c, int target, int* err) {
  unsigned spins = 0;
  while (ld_agent(c) < target) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > PS_SPIN_MAX) {
      __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
  }
  return true;
}

__device__ __forceinline__ float block_reduce8(float v, float* slot, bool is_max) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) slot[wave] = v;
  lds_barrier();
  float r = slot[0];
#pragma unroll
  for (int w = 1; w < 8; ++w) r = is_max ? fmaxf(r, slot[w]) : r + slot[w];
  return r;
}

__device__ __forceinline__ void split3(const float (&v)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const bf16 h = (bf16)v[i];
    const float r1 = v[i] - (float)h;
    const bf16 m = (bf16)r1;
    hi[i] = h;
    mid[i] = m;
    lo[i] = (bf16)(r1 - (float)m);
  }
}

__device__ __forceinline__ float act_rt(float x, int act) {
  if (act == PM_ACT_GELU) return apply_act<PM_ACT_GELU, true>(x);
  if (act == PM_ACT_GELU_TANH) return apply_act<PM_ACT_GELU_TANH, true>(x);
  return x;
}

struct PsLds {
  float sc[PS_MAXK];
  float xn[1280];
  float qkv[192];
  float scratch[4 * 8];
  float part[8 * 64];
  float lpart[2 * 256];
  f32x4 red[2 * 4 * 64];
};

// -------------------------------------------------------------------------------------------------------------------
// Attention block of one (sequence b, head h): wave 0 is the chain wave (it alone loads the row and normalises it, so
// the row does not queue behind the K/V stream: a wave's loads return in issue order), waves 1-7 stream K and V
// (7 x 8 = 56 keys per pass, 8 lanes x 16 B per key).  KREG / VREG passes of K / V are requested before the poll.
template <bool SELF, int NCH, int KREG, int VREG>
__device__ __forceinline__ void ps_attn_task(const PsArgs& p, PsLds& s, int b, int h, int tpos, const float* gamma,
                                             const float* beta, float eps, const bf16* Wp, const float* bp, bf16* Kc,
                                             bf16* Vc, int64_t sb, int64_t sh, int64_t sk, const int* wait_c,
                                             int wait_target, int* sig_c) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int d = p.d, inner = p.H * 64;
  const int Lk = SELF ? tpos + 1 : p.S;
  const int Lc = SELF ? tpos : p.S;  // keys read from memory (self: the newest one comes from LDS)
  const int nch = d >> 6;
  constexpr int NP = SELF ? 3 : 1;
  constexpr bool UPFRONT = !(SELF && NCH > 8);  // all projections' weights up front only while they fit the register file

  // ---- before the poll: everything that does not depend on the row
  // (wave 0 keeps gamma / beta in the registers the streaming waves use for K: the two roles never meet in one wave)
  constexpr int GREG = (NCH + 1) / 2;  // bf16x8 = 4 floats: NCH gamma + NCH beta values
  constexpr int KR = KREG > GREG ? KREG : GREG;
  bf16x8 kreg[KR], vreg[VREG];
  if (wave == 0) {
#pragma unroll
    for (int j = 0; j < GREG; ++j) {
      f32x4 gbv;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = (j * 4 + e) >> 1;  // element pairs: (gamma_i, beta_i)
        const int k = lane + 64 * i;
        gbv[e] = (i < nch) ? (((j * 4 + e) & 1) ? beta[k] : gamma[k]) : 0.f;
      }
      kreg[j] = __builtin_bit_cast(bf16x8, gbv);
    }
  }
  const int prow = tid >> 3, pl = tid & 7;
  float bpe[NP];
#pragma unroll
  for (int o = 0; o < NP; ++o) bpe[o] = bp ? bp[o * inner + h * 64 + prow] : 0.f;
  bf16x8 wv[UPFRONT ? NP : 1][NCH];
  if constexpr (UPFRONT) {
#pragma unroll
    for (int o = 0; o < NP; ++o) {
      const bf16* wr = Wp + ((int64_t)o * inner + h * 64 + prow) * d + pl * 8;
#pragma unroll
      for (int i = 0; i < NCH; ++i)
        if (i < nch) wv[o][i] = *(const bf16x8*)(wr + i * 64);
    }
  }
  const int c = lane & 7, ks = lane >> 3;
  const int sw = wave - 1;  // streaming wave 0..6 (wave 0: none)
  const bf16* kb = Kc + b * sb + h * sh + c * 8;
  const bf16* vb = Vc + b * sb + h * sh + c * 8;
  if (wave != 0) {
#pragma unroll
    for (int u = 0; u < KREG; ++u) {
      int key = u * 56 + sw * 8 + ks;
      key = key < Lc ? key : (Lc > 0 ? Lc - 1 : 0);
      kreg[u] = SELF ? *(const bf16x8*)(kb + key * sk) : __builtin_nontemporal_load((const bf16x8*)(kb + key * sk));
    }
#pragma unroll
    for (int u = 0; u < VREG; ++u) {
      int key = u * 56 + sw * 8 + ks;
      key = key < Lc ? key : (Lc > 0 ? Lc - 1 : 0);
      vreg[u] = SELF ? *(const bf16x8*)(vb + key * sk) : __builtin_nontemporal_load((const bf16x8*)(vb + key * sk));
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if (tid == 0 && wait_c) ps_wait(wait_c, wait_target, p.err);
  lds_barrier();

  // ---- LayerNorm of row b by wave 0 alone (two-pass, wave-local reductions)
  if (wave == 0) {
    const float* xr = p.x + (int64_t)b * d;
    float xe[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) xe[i] = (i < nch) ? ldf_agent(xr + lane + 64 * i) : 0.f;
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) sm += xe[i];
    const float mean = wave_sum(sm) / (float)d;
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (i < nch) q2 = fmaf(xe[i] - mean, xe[i] - mean, q2);
    const float rstd = rsqrtf(wave_sum(q2) / (float)d + eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (i < nch) {
        const f32x4 gbv = __builtin_bit_cast(f32x4, kreg[i >> 1]);
        s.xn[lane + 64 * i] = (xe[i] - mean) * rstd * gbv[(i & 1) * 2] + gbv[(i & 1) * 2 + 1];
      }
  }
  lds_barrier();
  // ---- q (k, v) = W xn + b : fp32 FMA over this lane's chunks, then across the 8 lanes of the row
#pragma unroll
  for (int o = 0; o < NP; ++o) {
    float acc = 0.f;
    if constexpr (!UPFRONT) {
      const bf16* wr = Wp + ((int64_t)o * inner + h * 64 + prow) * d + pl * 8;
#pragma unroll
      for (int i = 0; i < NCH; ++i)
        if (i < nch) wv[0][i] = *(const bf16x8*)(wr + i * 64);
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (i < nch) {
        const float* xp = s.xn + i * 64 + pl * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = fmaf((float)wv[UPFRONT ? o : 0][i][e], xp[e], acc);
      }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (pl == 0) {
      float v = acc + bpe[o];
      if (SELF && o > 0) {  // cached k / v are bf16: round once, store, and use the rounded value for this step too
        const bf16 r = (bf16)v;
        (o == 1 ? Kc : Vc)[b * sb + h * sh + (int64_t)tpos * sk + prow] = r;
        v = (float)r;
      }
      s.qkv[o * 64 + prow] = v;
    }
  }
  lds_barrier();

  // ---- scores
  const f32x4 q0 = *(const f32x4*)(s.qkv + c * 8), q1 = *(const f32x4*)(s.qkv + c * 8 + 4);
  auto score = [&](const bf16x8& kv, int key) {
    float sv = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) sv = fmaf(q0[i], (float)kv[i], sv);
#pragma unroll
    for (int i = 0; i < 4; ++i) sv = fmaf(q1[i], (float)kv[4 + i], sv);
    sv += __shfl_xor(sv, 1, 64);
    sv += __shfl_xor(sv, 2, 64);
    sv += __shfl_xor(sv, 4, 64);
    if (c == 0 && key < Lc) s.sc[key] = sv * 0.125f;
  };
  if (wave != 0) {
#pragma unroll
    for (int u = 0; u < KREG; ++u) score(kreg[u], u * 56 + sw * 8 + ks);
    for (int k0 = KREG * 56; k0 < Lc; k0 += 56 * 4) {  // keys beyond the register-resident passes: streamed
      bf16x8 kv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int key = k0 + u * 56 + sw * 8 + ks;
        key = key < Lc ? key : Lc - 1;
        kv[u] = SELF ? *(const bf16x8*)(kb + key * sk) : __builtin_nontemporal_load((const bf16x8*)(kb + key * sk));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) score(kv[u], k0 + u * 56 + sw * 8 + ks);
    }
  } else if (SELF && tid < 8) {  // the new key (position t), same summation shape as above
    float sv = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) sv = fmaf(q0[i], s.qkv[64 + c * 8 + i], sv);
#pragma unroll
    for (int i = 0; i < 4; ++i) sv = fmaf(q1[i], s.qkv[64 + c * 8 + 4 + i], sv);
    sv += __shfl_xor(sv, 1, 64);
    sv += __shfl_xor(sv, 2, 64);
    sv += __shfl_xor(sv, 4, 64);
    if (c == 0) s.sc[Lk - 1] = sv * 0.125f;
  }
  lds_barrier();
  float mx = -INFINITY;
  for (int k = tid; k < Lk; k += PS_THREADS) mx = fmaxf(mx, s.sc[k]);
  mx = block_reduce8(mx, s.scratch, true);
  float sum = 0.f;
  for (int k = tid; k < Lk; k += PS_THREADS) {
    const float pe = expf(s.sc[k] - mx);
    s.sc[k] = pe;
    sum += pe;
  }
  sum = block_reduce8(sum, s.scratch + 8, false);  // its barrier also publishes the p values

  // ---- P.V
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  auto pv = [&](const bf16x8& vv, int key) {
    // a row that does not exist contributes NOTHING - not 0 * v: at the first self-attention step (Lc = 0) the clamped loads
    // above read cache row 0 before anybody has written it, and 0 * NaN of recycled memory poisoned every later step
    const bool ok = key < Lc;
    const float pe = ok ? s.sc[key] : 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = ok ? fmaf(pe, (float)vv[i], acc[i]) : acc[i];
  };
  if (wave != 0) {
#pragma unroll
    for (int u = 0; u < VREG; ++u) pv(vreg[u], u * 56 + sw * 8 + ks);
    for (int k0 = VREG * 56; k0 < Lc; k0 += 56 * 4) {
      bf16x8 vv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int key = k0 + u * 56 + sw * 8 + ks;
        key = key < Lc ? key : Lc - 1;
        vv[u] = SELF ? *(const bf16x8*)(vb + key * sk) : __builtin_nontemporal_load((const bf16x8*)(vb + key * sk));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) pv(vv[u], k0 + u * 56 + sw * 8 + ks);
    }
  } else if (SELF && ks == 0) {
    const float pe = s.sc[Lk - 1];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = fmaf(pe, s.qkv[128 + c * 8 + i], acc[i]);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    acc[i] += __shfl_xor(acc[i], 8, 64);
    acc[i] += __shfl_xor(acc[i], 16, 64);
    acc[i] += __shfl_xor(acc[i], 32, 64);
  }
  if (ks == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) s.part[wave * 64 + c * 8 + i] = acc[i];
  }
  lds_barrier();
  if (tid < 64) {
    float o = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) o += s.part[w * 64 + tid];
    stf_agent(p.att + ((int64_t)b * p.H + h) * 64 + tid, o / sum);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  lds_barrier();  // the stores above are drained before anyone signals; also fences s.* reuse by the next task
  if (tid == 0 && sig_c) __hip_atomic_fetch_add(sig_c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The same block as its own launch (the launch-per-stage step form): one workgroup per (sequence, head), nothing to wait
// for, nobody to signal.  Against decode.hip's dec_attn_fused_kernel it keeps the whole K stream (and the head of V) in
// flight from the first instruction - that kernel had 4 passes (32 KB per CU) in flight per round trip, which is what
// held a CU at ~17 GB/s - without putting the row behind it: wave 0, which loads and normalises the row, requests no K / V.
template <bool SELF, int NCH, int KREG, int VREG>
__global__ __launch_bounds__(PS_THREADS) void dec_attn_v2_kernel(PsArgs p, const float* gamma, const float* beta, float eps,
                                                                 const bf16* Wp, const float* bp, bf16* Kc, bf16* Vc, int64_t sb,
                                                                 int64_t sh, int64_t sk) {
  __shared__ PsLds s;
  const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
  const int tpos = SELF ? *p.pos_ptr : 0;
  ps_attn_task<SELF, NCH, KREG, VREG>(p, s, b, h, tpos, gamma, beta, eps, Wp, bp, Kc, Vc, sb, sh, sk, nullptr, 0, nullptr);
}

// -------------------------------------------------------------------------------------------------------------------
// One linear stage: y[rows of a 16-row tile][16 features] = [LayerNorm](x) W^T + b [act] [+ resid], a tile per HALF
// workgroup (4 waves split K, fixed-order LDS reduction; decode.hip's dec_linear_kernel at MT = FT = 1), both halves of
// a workgroup run a tile each in lockstep.  kparts > 1: K split over tasks, combined by the last part to finish.
struct PsLin {
  const float* x;  // handed-off input (sc1 loads)
  int ldx;
  const float *gamma, *beta;
  float eps;
  const bf16* W;
  int64_t ldw;
  const float* bias;
  const float* resid;  // handed-off (sc1) or null
  int ldr;
  float* out;
  int ldo;
  int N, K, act, kparts;
  const int* wait_c;  // + row tile
  int wait_per_row;   // target = epoch * (wait_per_row ? rows(rt) * wait_n : wait_n)
  int wait_n;
  int* sig_c;         // + row tile
};

template <int NSTEP, bool LN>
__device__ __forceinline__ void ps_lin_stage(const PsArgs& p, PsLds& s, float* gb, const PsLin& a, int epoch) {
  constexpr int GBK = 128 * NSTEP;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2, hw = wave & 3;
  const int G = gridDim.x, w = blockIdx.x;
  const int fi = lane & 15, kq = lane >> 4;
  const int M = p.B, mt = (M + 15) >> 4;
  const int nft = a.N >> 4;
  const int ntasks = nft * mt * a.kparts;
  const int ksteps = a.K >> 5;
  const Buf xb(a.x, (int64_t)M * a.ldx * 4), ob(a.out, (int64_t)M * a.ldo * 4);
  const Buf rb(a.resid ? a.resid : a.x, a.resid ? (int64_t)M * a.ldr * 4 : 0);
  float* lpart = s.lpart + half * 256;
  f32x4* red = s.red + half * 256;

  if constexpr (LN) {  // both halves normalise with the same vectors: staged once per stage (weights: plain loads)
    for (int k = tid; k < a.K; k += PS_THREADS) {
      gb[k] = a.gamma[k];
      gb[GBK + k] = a.beta[k];
    }
  }
  for (int base = w; base < ntasks; base += 2 * G) {  // half 0 takes task base, half 1 task base + G
    const int task_raw = base + half * G;
    const bool active = task_raw < ntasks;
    const int task = active ? task_raw : base;
    const int kp = task % a.kparts, tile = task / a.kparts;
    const int ft = tile % nft, rt = tile / nft;
    const int n0 = ft * 16, rbase = rt * 16;
    const int ks0 = ksteps * kp / a.kparts, ks1 = ksteps * (kp + 1) / a.kparts;
    int row = rbase + fi;
    row = row < M ? row : M - 1;

    // ---- before the poll: weights and bias
    bf16x8 wreg[NSTEP];
    int wrow = n0 + fi;
    const bf16* wp = a.W + (int64_t)wrow * a.ldw + kq * 8;
#pragma unroll
    for (int u = 0; u < NSTEP; ++u) {
      int st = ks0 + hw + 4 * u;
      st = st < ks1 ? st : ks1 - 1;
      wreg[u] = *(const bf16x8*)(wp + st * 32);
    }
    f32x4 bpre = {0.f, 0.f, 0.f, 0.f};
    if (hw == 0 && a.bias) bpre = *(const f32x4*)(a.bias + n0 + kq * 4);
    __builtin_amdgcn_sched_barrier(0);
    if (hw == 0 && lane == 0 && active && a.wait_c) {
      const int rows = (M - rbase) < 16 ? (M - rbase) : 16;
      ps_wait(a.wait_c + rt, epoch * (a.wait_per_row ? rows * a.wait_n : a.wait_n), p.err);
    }
    lds_barrier();  // also publishes gb

    // ---- the handed-off operands
    f32x4 xv[NSTEP][2];
#pragma unroll
    for (int u = 0; u < NSTEP; ++u) {
      int st = ks0 + hw + 4 * u;
      st = st < ks1 ? st : ks1 - 1;
      const int off = (row * a.ldx + st * 32 + kq * 8) * 4;
      xv[u][0] = xb.ld(off);
      xv[u][1] = xb.ld(off + 16);
    }
    f32x4 rpre = {0.f, 0.f, 0.f, 0.f};
    if (hw == 0 && a.resid) rpre = rb.ld((row * a.ldr + n0 + kq * 4) * 4);

    float mean = 0.f, rstd = 1.f;
    if constexpr (LN) {  // two-pass statistics from the registers that hold the row (decode.hip)
      float sm = 0.f;
#pragma unroll
      for (int u = 0; u < NSTEP; ++u)
        if (ks0 + hw + 4 * u < ks1) {
#pragma unroll
          for (int i = 0; i < 4; ++i) sm += xv[u][0][i] + xv[u][1][i];
        }
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      if (kq == 0) lpart[hw * 64 + fi] = sm;
      lds_barrier();
      mean = ((lpart[fi] + lpart[64 + fi]) + (lpart[128 + fi] + lpart[192 + fi])) / (float)a.K;
      lds_barrier();
      float q = 0.f;
#pragma unroll
      for (int u = 0; u < NSTEP; ++u)
        if (ks0 + hw + 4 * u < ks1) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float d0 = xv[u][0][i] - mean, d1 = xv[u][1][i] - mean;
            q = fmaf(d0, d0, q);
            q = fmaf(d1, d1, q);
          }
        }
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      if (kq == 0) lpart[hw * 64 + fi] = q;
      lds_barrier();
      rstd = rsqrtf(((lpart[fi] + lpart[64 + fi]) + (lpart[128 + fi] + lpart[192 + fi])) / (float)a.K + a.eps);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NSTEP; ++u) {
      if (ks0 + hw + 4 * u >= ks1) break;
      const int k0 = (ks0 + hw + 4 * u) * 32 + kq * 8;
      float v[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = xv[u][0][i]; v[4 + i] = xv[u][1][i]; }
      if constexpr (LN) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (v[i] - mean) * rstd * gb[k0 + i] + gb[GBK + k0 + i];
      }
      bf16x8 hi, mid, lo;
      split3(v, hi, mid, lo);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[u], hi, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[u], mid, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[u], lo, acc, 0, 0, 0);
    }
    red[hw * 64 + lane] = acc;
    lds_barrier();
    bool final_tile = false;
    if (hw == 0) {
      // D[row = feature 4*kq + r][col = sequence fi]; partial sums added in wave order 0..3
      f32x4 v = red[lane];
#pragma unroll
      for (int q = 1; q < 4; ++q) v += red[q * 64 + lane];
      final_tile = active;
      if (a.kparts > 1) {
        // K split: publish this part (sc1), take a ticket; the last part to finish adds the parts in part order
        f32x4* my = (f32x4*)p.ks_ws + (int64_t)(tile * a.kparts + kp) * 64 + lane;
        if (active) store_sc1_x4(my, v);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int ticket = 0;
        if (lane == 0 && active) ticket = __hip_atomic_fetch_add(p.ks_tick + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        final_tile = active && ticket == a.kparts - 1;
        if (final_tile) {
          v = f32x4{0.f, 0.f, 0.f, 0.f};
          for (int q0 = 0; q0 < a.kparts; q0 += 4) {
            f32x4 pq[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int q = q0 + j < a.kparts ? q0 + j : a.kparts - 1;
              pq[j] = load_sc1_x4_async((const f32x4*)p.ks_ws + (int64_t)(tile * a.kparts + q) * 64 + lane);
            }
            wait_loads(pq[0], pq[1], pq[2], pq[3]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (q0 + j < a.kparts) v += pq[j];
          }
          if (lane == 0) __hip_atomic_store(p.ks_tick + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (final_tile) {
        f32x4 e;
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = act_rt(v[r] + bpre[r], a.act) + rpre[r];
        if (rbase + fi < M) ob.st(((rbase + fi) * a.ldo + n0 + kq * 4) * 4, e);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    lds_barrier();
    if (hw == 0 && lane == 0 && final_tile) __hip_atomic_fetch_add(a.sig_c + rt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// -------------------------------------------------------------------------------------------------------------------
template <int NCH, int NSTEP>
__global__ __launch_bounds__(PS_THREADS) void dec_layers_kernel(PsArgs p) {
  __shared__ PsLds s;
  __shared__ float gb[2 * 128 * NSTEP];
  __shared__ pm_dec_layer_t tabL[PS_MAXL];
  if (ld_agent(p.err) != 0) return;  // a hand-off of an earlier launch gave up: drain at once, the host raises
  // the layer table goes to LDS once: read through the kernel-argument pointer where it is used, every stage paid the
  // scalar loads' round trips in its prologue
  for (int i = threadIdx.x; i < p.n_layers * (int)(sizeof(pm_dec_layer_t) / 4); i += PS_THREADS)
    ((int*)tabL)[i] = ((const int*)p.tab)[i];
  lds_barrier();
  const int t = *p.pos_ptr;
  const int epoch = t + 1;
  const int G = gridDim.x, w = blockIdx.x;
  const int B = p.B, H = p.H, d = p.d, inner = H * 64;
  const int nft_d = d >> 4;
  // K / V passes (56 keys each) an attention task requests BEFORE it polls for its row: what the register file allows
  // beside the projection weights (NCH 16-byte chunks per lane and projection)
  constexpr int KREG_X = NCH <= 8 ? 24 : NCH <= 12 ? 16 : 8;
  constexpr int VREG_X = NCH <= 8 ? 4 : 2;
  constexpr int KREG_S = NCH <= 8 ? 8 : 4;

  for (int l = 0; l < p.n_layers; ++l) {
    const pm_dec_layer_t& L = tabL[l];
    int* c = p.cnt + l * 24;
    const int* prev = l > 0 ? c - 4 : nullptr;  // stage 5 of the layer before
    // ---- S0: self-attention block
    for (int task = w; task < B * H; task += G) {
      const int b = task / H, h = task - b * H;
      ps_attn_task<true, NCH, KREG_S, KREG_S>(p, s, b, h, t, L.sa_g, L.sa_b, L.sa_eps, (const bf16*)L.w_qkv, L.b_qkv, (bf16*)L.kc,
                                              (bf16*)L.vc, (int64_t)H * p.Tmax * 64, (int64_t)p.Tmax * 64, 64,
                                              prev ? prev + (b >> 4) : nullptr, epoch * nft_d, c + 0 + (b >> 4));
    }
    // ---- S1: x += att Wo^T + bo
    {
      PsLin a{p.att, inner, nullptr, nullptr, 0.f, (const bf16*)L.w_so, inner, L.b_so, p.x, d, p.x, d, d, inner, PM_ACT_NONE, 1,
              c + 0, 1, H, c + 4};
      ps_lin_stage<NSTEP, false>(p, s, gb, a, epoch);
    }
    const int* mlp_wait = c + 4;
    if (L.w_q) {
      // ---- S2: cross-attention block over the packed (B, S, [k | v]) projection of the memory
      for (int task = w; task < B * H; task += G) {
        const int b = task / H, h = task - b * H;
        bf16* kv = (bf16*)const_cast<void*>(L.cross_kv);
        ps_attn_task<false, NCH, KREG_X, VREG_X>(p, s, b, h, 0, L.ca_g, L.ca_b, L.ca_eps, (const bf16*)L.w_q, L.b_q, kv, kv + inner,
                                                 (int64_t)p.S * 2 * inner, 64, 2 * inner, c + 4 + (b >> 4), epoch * nft_d,
                                                 c + 8 + (b >> 4));
      }
      // ---- S3
      PsLin a{p.att, inner, nullptr, nullptr, 0.f, (const bf16*)L.w_co, inner, L.b_co, p.x, d, p.x, d, d, inner, PM_ACT_NONE, 1,
              c + 8, 1, H, c + 12};
      ps_lin_stage<NSTEP, false>(p, s, gb, a, epoch);
      mlp_wait = c + 12;
    }
    // ---- S4: h = act(LN(x) W1^T + b1)
    {
      PsLin a{p.x, d, L.mlp_g, L.mlp_b, L.mlp_eps, (const bf16*)L.w1, d, L.b1, nullptr, 0, p.h, p.ldh, p.hid, d, p.act, 1,
              mlp_wait, 0, nft_d, c + 16};
      ps_lin_stage<NSTEP, true>(p, s, gb, a, epoch);
    }
    // ---- S5: x += h W2^T + b2 (K = hid split over workgroups)
    {
      PsLin a{p.h, p.ldh, nullptr, nullptr, 0.f, (const bf16*)L.w2, p.hid, L.b2, p.x, d, p.x, d, d, p.hid, PM_ACT_NONE, p.ksplit,
              c + 16, 0, p.hid >> 4, c + 20};
      ps_lin_stage<NSTEP, false>(p, s, gb, a, epoch);
    }
  }
}

struct DevInfo {
  int cus = 0;
};
DevInfo& dev_info() {
  static DevInfo di;
  static std::once_flag once;
  std::call_once(once, [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) di.cus = prop.multiProcessorCount;
  });
  return di;
}

template <int NCH, int NSTEP>
int launch_layers(const PsArgs& a, int grid, hipStream_t st) {
  static int resident = -1;  // workgroups of this instantiation one CU holds (0: does not fit)
  if (resident < 0) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, dec_layers_kernel<NCH, NSTEP>, PS_THREADS, 0) != hipSuccess) n = 0;
    resident = n;
  }
  if (resident < 1) return PM_EUNSUPPORTED;
  hipLaunchKernelGGL((dec_layers_kernel<NCH, NSTEP>), dim3((unsigned)grid), dim3(PS_THREADS), 0, st, a);
  return PM_OK;
}

}  // namespace

/* pm_dec_attention_fused (same arguments, same result up to the summation order of the row's LayerNorm statistics) on the
 * persistent step's attention block: lk_max <= 2048. */
extern "C" int pm_dec_attention_fused_v2(const float* x, int64_t d, const float* gamma, const float* beta, float eps,
                                         const void* w, const float* bias, void* kc, void* vc, int64_t stride_b,
                                         int64_t stride_h, int64_t stride_k, const int32_t* pos_ptr, int64_t lk_const,
                                         int64_t lk_max, float* out, int64_t B, int64_t H, int self_attn, void* stream) {
  if (!x || !gamma || !beta || !w || !kc || !vc || !out || B <= 0 || H <= 0 || d <= 0 || lk_max <= 0) return PM_EINVAL;
  if (d % 64 || d > 1280 || lk_max > PS_MAXK) return PM_EUNSUPPORTED;
  if (self_attn ? !pos_ptr : lk_const <= 0) return PM_EINVAL;
  if ((stride_b | stride_h | stride_k) % 8) return PM_EALIGN;
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)kc | (uintptr_t)vc | (uintptr_t)out) & 15) return PM_EALIGN;
  if (B * H > 0x7fffffff) return PM_EINVAL;
  PsArgs a{};
  a.B = (int)B; a.d = (int)d; a.H = (int)H; a.S = (int)lk_const; a.pos_ptr = (const int*)pos_ptr;
  a.x = const_cast<float*>(x); a.att = out;
  hipStream_t st = (hipStream_t)stream;
  const int nch = (int)(d / 64);
#define PM_AV2(SELF_, NCH_, KR_, VR_)                                                                                       \
  hipLaunchKernelGGL((dec_attn_v2_kernel<SELF_, NCH_, KR_, VR_>), dim3((unsigned)(B * H)), dim3(PS_THREADS), 0, st, a, gamma, \
                     beta, eps, (const bf16*)w, bias, (bf16*)kc, (bf16*)vc, stride_b, stride_h, stride_k)
  if (self_attn) {
    if (nch <= 8) PM_AV2(true, 8, 8, 8);
    else if (nch <= 12) PM_AV2(true, 12, 8, 8);
    else if (nch <= 16) PM_AV2(true, 16, 8, 8);
    else PM_AV2(true, 20, 8, 8);
  } else {
    if (nch <= 8) PM_AV2(false, 8, 27, 14);
    else if (nch <= 12) PM_AV2(false, 12, 27, 12);
    else if (nch <= 16) PM_AV2(false, 16, 27, 8);
    else PM_AV2(false, 20, 27, 4);
  }
#undef PM_AV2
  PM_CHECK_LAUNCH();
  return PM_OK;
}

/* number of workgroups pm_dec_layers launches (one per CU): sizes nothing the caller allocates, exposed for tests */
extern "C" int pm_dec_layers_grid(void) { return dev_info().cus; }

extern "C" int pm_dec_layers(const pm_dec_layer_t* layers, int64_t n_layers, int64_t B, int64_t d, int64_t H, int64_t S,
                             int64_t Tmax, int64_t hid, int act, int64_t k_split, const int32_t* pos_ptr, float* x, float* att,
                             float* h, int64_t ldh, int32_t* counters, float* split_ws, int32_t* split_cnt, int32_t* err,
                             void* stream) {
  if (!layers || !pos_ptr || !x || !att || !h || !counters || !split_ws || !split_cnt || !err) return PM_EINVAL;
  if (n_layers <= 0 || B <= 0 || d <= 0 || H <= 0 || S < 0 || Tmax <= 0 || hid <= 0 || ldh < hid) return PM_EINVAL;
  if (n_layers > PS_MAXL) return PM_EUNSUPPORTED;
  if (B > 64 || d % 64 || d > 1280 || H * 64 != d || S > PS_MAXK || Tmax > PS_MAXK || ldh % 4 || hid % 32) return PM_EUNSUPPORTED;
  if (act != PM_ACT_NONE && act != PM_ACT_GELU && act != PM_ACT_GELU_TANH) return PM_EUNSUPPORTED;
  if (k_split < 1 || k_split > 8 || hid / 32 < k_split) return PM_EINVAL;
  {  // every K part of linear2 must fit the registers of the instantiation picked below: ceil(steps / 4 waves) <= NSTEP
    const int nstep = d <= 512 ? 4 : d <= 1024 ? 8 : 10;
    const int64_t steps = (hid / 32 + k_split - 1) / k_split;
    if ((steps + 3) / 4 > nstep) return PM_EUNSUPPORTED;
  }
  if (((uintptr_t)x | (uintptr_t)att | (uintptr_t)h | (uintptr_t)split_ws | (uintptr_t)layers) & 15) return PM_EALIGN;
  const int cus = dev_info().cus;
  if (cus <= 0) return PM_ELAUNCH;
  PsArgs a{layers, (int)n_layers, (int)B, (int)d, (int)H, (int)S, (int)Tmax, (int)k_split, (int)ldh, (int)hid, act, (const int*)pos_ptr,
           x, att, h, (int*)counters, split_ws, (int*)split_cnt, (int*)err};
  hipStream_t st = (hipStream_t)stream;
  const int nch = (int)(d / 64);
  int rc;
  if (nch <= 8) rc = launch_layers<8, 4>(a, cus, st);
  else if (nch <= 12) rc = launch_layers<12, 8>(a, cus, st);
  else if (nch <= 16) rc = launch_layers<16, 8>(a, cus, st);
  else rc = launch_layers<20, 10>(a, cus, st);
  if (rc != PM_OK) return rc;
  PM_CHECK_LAUNCH();
  return PM_OK;
}
