// linear_bf16_tile.hip - the large-M kernel of pm_linear_bf16: persistent (64 MI) x 256 x 64 tiles, MI = 4 (256 token rows)
// or 5 (320 token rows).  (same operation and reference sites as linear_bf16.hip: pytorch_models/transformer.py:28-31,47-53,59-66)
//
// Why two heights.  One workgroup per CU walks whole tiles; M = 50432 token rows (ViT-B/16 at batch 256) by N = 768
// features is 591 tiles of 256 x 256 = 2.31 rounds of 256 workgroups, run as 3; as 320 x 256 tiles it is 474 = 1.85
// rounds run as 2, each 1.25 x the work: 2.5 instead of 3 (and instead of the 5 x 0.5 of 256 x 128 tiles, whose K loop
// pulls 48 KB per 1024 matrix-pipe clocks through the CU's vector-memory path; this one pulls 72 KB per 2560).  No tile
// is cut along K, so a row's sum does not depend on where the row sits in the batch (batch-permutation invariance stays
// bit-exact).  The dispatcher (linear_bf16.hip, pm_linear_pick_kernel) prices both heights per shape.
//
// K loop: as the 256 x 256 x 64 kernel it replaces - 8 waves as 4 (tokens) x 2 (features), each wave (16 MI) x 128 =
// MI x 8 MFMA 16x16x32 tiles (32 MI accumulator VGPRs), two LDS stages of (64 MI + 256) 128-byte rows filled by
// 16-byte LDS-DMA with the XOR swizzle on the SOURCE address, one raw barrier per 64 of K, one stream of K steps across
// all of a workgroup's tiles, waves 4-7 at s_setprio 1.
//
// Epilogue (new): everything is finished IN THE ACCUMULATOR LAYOUT - a lane holds 4 consecutive features of one token:
// LayerNorm fold / bias / activation / residual (8-byte loads, requested two row blocks ahead) / ONE rounding to bf16 -
// and only the rounded 8 bytes per lane go through a wave-private 2 KiB transposition (16 tokens x 64 features of bf16)
// to leave as 16-byte stores of full 128-byte row segments.  No barrier, half the LDS traffic of an fp32 transposition,
// no second conversion pass.  The LayerNorm fold's row partials (sum and sum of squares of the ROUNDED outputs per row
// and 64-feature block) come from the matrix pipe: the packed outputs are already a B fragment (token = lane & 15), so
// ones x F^T gives the row sums and the diagonal of F x F^T the sums of squares (bf16 products are exact in fp32):
// 4 MFMAs per 16 x 64 block instead of ~60 vector instructions per lane.
#include <cstdlib>

#include "common.h"

namespace {
// In-kernel stamps of the epilogue (variant builds only: -DPM_TILE_STAMPS=1; tools/tile_stamps.py): per workgroup, its first tile:
// slot 0 = the last K step's barrier, 1 = epilogue start, 2 + q = behind block q, 15 = wave 0's view only.
#ifndef PM_TILE_RESID16  // experiment (tools/build_variant.sh): the residual as two 16-byte requests per block + lane swaps
#define PM_TILE_RESID16 0  // instead of four 8-byte ones - same results, fc2 half batch 93.5 us against 92.7: not the request count
#endif
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#ifndef PM_TILE_STAMPS
#define PM_TILE_STAMPS 0
#endif
#if PM_TILE_STAMPS
__device__ unsigned long long g_tile_stamps[256 * 16];
#define PM_TSTAMP(i_, cond_)                                                                       \
  do {                                                                                             \
    if ((cond_) && threadIdx.x == 0) g_tile_stamps[blockIdx.x * 16 + (i_)] = wall_clock64();        \
  } while (0)
#else
#define PM_TSTAMP(i_, cond_)
#endif

constexpr int TBN = 256, TBK = 64;
#ifndef PM_TGROUP_M
#define PM_TGROUP_M 4
#endif
constexpr int TGROUP_M = PM_TGROUP_M;

__device__ __forceinline__ bf16x8 tread(const char* tile, int row, int chunk) {
  return *(const bf16x8*)(tile + row * 128 + swz_pos(row, chunk) * 16);
}

__device__ __forceinline__ void ttile_coords(int t, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int per = TGROUP_M * tiles_n;
  const int sr = t / per, r = t - sr * per;
  const int left = tiles_m - sr * TGROUP_M;
  const int gm = left < TGROUP_M ? left : TGROUP_M;
  tn = r / gm;
  tm = sr * TGROUP_M + (r - tn * gm);
}

// EPI: 0 = bias + activation; 1 = the same behind the LayerNorm fold (ln.stats, ln.s); 2 = bias + activation + residual.
// Modes 0 and 2 also emit the fold's row partials when ln.row_out is set.
enum { EPI_PLAIN = 0, EPI_LNC = 1, EPI_RES = 2 };

template <int MI, int ACT, int EPI>
__global__ __launch_bounds__(512, 2) void linear_bf16_tile_kernel(
    const bf16* __restrict__ X, int64_t ldx, const bf16* __restrict__ W, int64_t ldw, const float* __restrict__ bias,
    const bf16* resid, int64_t ldr, int resid_period, bf16* Y, int64_t ldy, int M, int N, int K, int tiles_m, int tiles_n,
    int x_rows_per_batch, int64_t x_batch_stride, PmLnFold ln) {
  constexpr bool LNC = EPI == EPI_LNC, RES = EPI == EPI_RES;
  constexpr int RPF = MI > 4 ? 1 : 2;  // residual row blocks requested ahead of their use in the epilogue (registers decide)
  constexpr int BM = 64 * MI, WR = 16 * MI;      // tile rows, rows per wave
  constexpr int STAGE = (BM + TBN) * 128;        // 64 KiB (MI 4) / 72 KiB (MI 5)
  constexpr int XP = MI, WP = 4;                 // LDS-DMA pieces (8 rows x 128 B) per wave and step: token rows / weight rows
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE + 8 * 2048];  // two stages + 2 KiB of bf16 epilogue staging per wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntiles = tiles_m * tiles_n;

  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, nloc = gridDim.x >> 3;
  const int cq = ntiles >> 3, cr = ntiles & 7;
  const int tbase = xcd * cq + (xcd < cr ? xcd : cr), tcount = cq + (xcd < cr ? 1 : 0);
  const int my_tiles = local < tcount ? (tcount - local + nloc - 1) / nloc : 0;
  const int nk = K / TBK;
  const int P = my_tiles * nk;

  // staging side of the stream (a macro over plain locals: see linear_bf16.hip).  An LDS-DMA piece is 8 consecutive rows
  // x 128 bytes; M and N are multiples of 8 on this path, so a piece never straddles the operand's last row and its first
  // row is clamped as a whole (rows beyond M / N are never stored).  The address of a piece is then a UNIFORM 64-bit row
  // base (SGPR pair, once per tile) plus a 32-bit lane offset that is the same for every piece of an operand up to the
  // swizzle phase (chunk (lane & 7) ^ (lane >> 4) ^ 0 or 4): two VGPRs per operand for the whole kernel instead of one
  // 64-bit vector address per piece (the dispatcher keeps a row's byte offset inside a piece far below 2^31).
  const char* xbase[XP];
  const char* wbase[WP];
  const uint32_t lsw = (uint32_t)(((lane & 7) ^ (lane >> 4)) * 16);  // < 128: the phase flips its bit 6, nothing carries
  const uint32_t xl0 = (uint32_t)(lane >> 3) * (uint32_t)(ldx * 2), wl0 = (uint32_t)(lane >> 3) * (uint32_t)(ldw * 2);
  int pp = 0, pp_kt = 0, pp_tile = 0, pp_buf = 0;
#define PM_TSTAGE_NEXT()                                                                                             \
  if (pp < P) {                                                                                                      \
    if (pp_kt == 0) {                                                                                                \
      int tm_, tn_;                                                                                                  \
      ttile_coords(tbase + local + pp_tile * nloc, tiles_m, tiles_n, tm_, tn_);                                      \
      _Pragma("unroll") for (int i = 0; i < XP; ++i) {                                                               \
        int r_ = tm_ * BM + wave * (8 * MI) + i * 8;                                                                 \
        r_ = r_ < M - 8 ? r_ : M - 8;                                                                                \
        xbase[i] = (const char*)X + (int64_t)r_ * ldx * 2;                                                           \
      }                                                                                                              \
      _Pragma("unroll") for (int i = 0; i < WP; ++i) {                                                               \
        int r_ = tn_ * TBN + wave * 32 + i * 8;                                                                      \
        r_ = r_ < N - 8 ? r_ : N - 8;                                                                                \
        wbase[i] = (const char*)W + (int64_t)r_ * ldw * 2;                                                           \
      }                                                                                                              \
    }                                                                                                                \
    char* xs_ = smem + pp_buf * STAGE;                                                                               \
    const uint32_t kb_ = (uint32_t)(pp_kt * (TBK * 2));                                                              \
    /* swizzle phase of piece i: bit 2 of (first row of the piece) >> 1, i.e. (wave * 4 MI + 4 i) & 4.  The phase is   \
       applied per piece from a scalar the compiler cannot hoist (one v_xor per piece, no second pair of lane offsets  \
       held - and at MI = 5 spilled - across the K loop) */                                                           \
    _Pragma("unroll") for (int i = 0; i < XP; ++i) {                                                                 \
      uint32_t ph_ = (uint32_t)(((wave * MI + i) & 1) * 64);                                                         \
      asm volatile("" : "+s"(ph_));                                                                                  \
      glds16_aux<PM_GLDS_X_AUX>(xbase[i] + (uint32_t)(xl0 + kb_ + (lsw ^ ph_)), xs_ + (wave * (8 * MI) + i * 8) * 128); \
    }                                                                                                                \
    _Pragma("unroll") for (int i = 0; i < WP; ++i) {                                                                 \
      uint32_t ph_ = (uint32_t)((i & 1) * 64);                                                                       \
      asm volatile("" : "+s"(ph_));                                                                                  \
      glds16_aux<PM_GLDS_W_AUX>(wbase[i] + (uint32_t)(wl0 + kb_ + (lsw ^ ph_)), xs_ + BM * 128 + (wave * 32 + i * 8) * 128); \
    }                                                                                                                \
    ++pp;                                                                                                            \
    pp_buf ^= 1;                                                                                                     \
    if (++pp_kt == nk) { pp_kt = 0; ++pp_tile; }                                                                     \
  }

  f32x4 acc[8][MI];  // [feature subtile j][token subtile i]
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (wave >= 4) __builtin_amdgcn_s_setprio(1);  // the younger wave of each SIMD pair loses every arbitration otherwise
  PM_TSTAGE_NEXT();
  const int fr = lane & 15, fq = lane >> 4;
  int buf = 0, kt = 0, ti = 0;
  f32x2 lnst[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) lnst[i] = f32x2{0.f, 1.f};
  bf16x4 rv[RES ? RPF + 1 : 1][4];  // residual values of row blocks q .. q + RPF (accumulator layout)

#ifdef PM_TILE_ABLATE_RESID  // experiment only (tools/build_variant.sh): the address arithmetic stays, the load does not
#define PM_TILE_RESID_LOAD(dst_, ptr_) { asm volatile("" ::"v"(ptr_)); dst_ = bf16x4{}; }
#else
#define PM_TILE_RESID_LOAD(dst_, ptr_) dst_ = *(const bf16x4*)(ptr_)
#endif
  // residual rows of epilogue block q = hf * MI + i (16 tokens x 64 features of this wave): 8 bytes per lane and feature
  // subtile, addressed as uniform 64-bit base + 32-bit lane offset (SGPR pair + one VGPR, no 64-bit vector arithmetic)
#define PM_TLOAD_RESID(q_, slot_, m0_, n0_, fr_, fq_)                                                                \
  {                                                                                                                  \
    const int hf_ = (q_) / MI, i_ = (q_) - hf_ * MI;                                                                 \
    int r0_ = (m0_) + i_ * 16;                                                                                       \
    int mm_ = r0_ + (fr_);                                                                                           \
    mm_ = mm_ < M ? mm_ : M - 1;                                                                                     \
    r0_ = r0_ < M - 16 ? r0_ : M - 16; /* a block beyond the last row: base inside the operand, offset >= 0 */       \
    const char* rb_;                                                                                                 \
    uint32_t ro_;                                                                                                    \
    if (resid_period) {                                                                                              \
      rb_ = (const char*)resid;                                                                                      \
      ro_ = (uint32_t)(mm_ % resid_period) * (uint32_t)(ldr * 2);                                                    \
    } else {                                                                                                         \
      rb_ = (const char*)resid + (int64_t)r0_ * ldr * 2;                                                             \
      ro_ = (uint32_t)(mm_ - r0_) * (uint32_t)(ldr * 2);                                                             \
    }                                                                                                                \
    if (PM_TILE_RESID16) {                                                                                           \
      /* two 16-byte requests instead of four of 8 bytes: lanes fq and fq ^ 1 (16 lanes apart, the same row) fetch the 8    \
         features they share of subtiles 2 k + (fq & 1); PM_TRESID_UNSWAP trades halves when the block is used */     \
      _Pragma("unroll") for (int k_ = 0; k_ < 2; ++k_) {                                                             \
        int nn_ = (n0_) + hf_ * 64 + (2 * k_ + ((fq_) & 1)) * 16 + ((fq_) & ~1) * 4;                                 \
        nn_ = nn_ < N - 8 ? nn_ : N - 8; /* features beyond N are never stored */                                    \
        const bf16x8 l_ = *(const bf16x8*)(rb_ + (uint32_t)(ro_ + (uint32_t)nn_ * 2));                               \
        rv[slot_][2 * k_] = __builtin_shufflevector(l_, l_, 0, 1, 2, 3);                                             \
        rv[slot_][2 * k_ + 1] = __builtin_shufflevector(l_, l_, 4, 5, 6, 7);                                         \
      }                                                                                                              \
    } else {                                                                                                         \
      _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                                             \
        int nn_ = (n0_) + hf_ * 64 + jj * 16 + (fq_) * 4;                                                            \
        nn_ = nn_ < N ? nn_ : N - 4; /* features beyond N are never stored */                                        \
        PM_TILE_RESID_LOAD(rv[slot_][jj], rb_ + (uint32_t)(ro_ + (uint32_t)nn_ * 2));                                \
      }                                                                                                              \
    }                                                                                                                \
  }
  // even rows of 16 lanes (fq even) hold (own, partner's) halves of subtile 2 k, odd rows of subtile 2 k + 1: v_permlane16_swap
  // trades the odd rows of its first operand with the even rows of its second - afterwards rv[.][2 k] and rv[.][2 k + 1] are
  // this lane's four features of both subtiles
#define PM_TRESID_UNSWAP(slot_)                                                                                      \
  if (PM_TILE_RESID16) {                                                                                             \
    _Pragma("unroll") for (int k_ = 0; k_ < 2; ++k_) {                                                               \
      u32x2 p_ = __builtin_bit_cast(u32x2, rv[slot_][2 * k_]), q_ = __builtin_bit_cast(u32x2, rv[slot_][2 * k_ + 1]); \
      _Pragma("unroll") for (int w_ = 0; w_ < 2; ++w_) {                                                             \
        const auto r_ = __builtin_amdgcn_permlane16_swap(p_[w_], q_[w_], false, false);                              \
        p_[w_] = r_[0];                                                                                              \
        q_[w_] = r_[1];                                                                                              \
      }                                                                                                              \
      rv[slot_][2 * k_] = __builtin_bit_cast(bf16x4, p_);                                                            \
      rv[slot_][2 * k_ + 1] = __builtin_bit_cast(bf16x4, q_);                                                        \
    }                                                                                                                \
  }

  for (int pc = 0; pc < P; ++pc) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // two stages: only step pc itself was in flight
    __builtin_amdgcn_s_barrier();                     // every wave's part landed; every wave is past step pc-1: its buffer is free
    // The two waves of a SIMD never request at the same time (PM_TILE_ISSUE_SPLIT 1; 0 = both right here): waves 0-3 request
    // their pieces of the next stage here, waves 4-7 theirs behind the MFMAs of the first half step (below).  An LDS-DMA
    // piece costs its wave ~60-100 cycles of issue during which the wave issues nothing else; with both partners in that
    // phase right behind the barrier the SIMD's matrix pipe stood idle for ~500-900 of a step's ~4700 cycles.  Now one
    // partner's MFMAs run under the other's requests: fc2 234 -> 208 us, 8192^3 802 -> 741 (1.48 PFLOP/s), QKV 165 -> 154.
#ifndef PM_TILE_ISSUE_SPLIT
#define PM_TILE_ISSUE_SPLIT 1
#endif
    if (!PM_TILE_ISSUE_SPLIT || wave < 4) PM_TSTAGE_NEXT();
    PM_TSTAMP(0, ti == 0 && kt == nk - 1);
    PM_TSTAMP(14, ti == 0 && kt == 0);
    if (kt == nk - 1) {
      // last K step of the tile: request what the epilogue needs first now, so that the latency hides under this step's MFMAs
      int tm_r, tn_r;
      ttile_coords(tbase + local + ti * nloc, tiles_m, tiles_n, tm_r, tn_r);
      int pl = lane;
      asm volatile("" : "+v"(pl));  // once per tile: lane constants recomputed, not kept across the K loop
      const int pfr = pl & 15, pfq = pl >> 4;
      if constexpr (LNC) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          int mr = tm_r * BM + wm * WR + i * 16 + pfr;
          mr = mr < M ? mr : M - 1;
          lnst[i] = *(const f32x2*)(ln.stats + 2 * (int64_t)mr);
        }
      }
      if constexpr (RES && MI <= 4) {  // (at MI = 5 the K loop has no registers to spare: the epilogue requests its first rows itself)
#pragma unroll
        for (int q = 0; q < RPF; ++q) PM_TLOAD_RESID(q, q, tm_r * BM + wm * WR, tn_r * TBN + wn * 128, pfr, pfq);
      }
    }
    const char* xcur = smem + buf * STAGE;
    [[maybe_unused]] const char* wcur = xcur + BM * 128;  // (the asm-read form addresses the weight rows off xcur)
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      bf16x8 a[8], b[MI];
      // Fragment reads in the order of their first use and the weight subtiles' MFMAs in that order, both pinned, in ONE basic
      // block per case (a tile's first half step starts from the constant 0: nobody clears accumulators): hipcc then waits
      // lgkmcnt(7) for the first 5 (MI) MFMAs and one fragment more per subtile - left to itself it issued the weight fragments
      // in reverse and waited lgkmcnt(0) for all MI + 8 in front of the first MFMA; and a branch between the reads and the MFMAs
      // makes it wait for everything at the block's entry.
#ifndef PM_TILE_ASM_READS
#define PM_TILE_ASM_READS 1
#endif
      // (hipcc's own waits in front of these MFMAs are lgkmcnt(0) whatever the order - it waits for all MI + 8 fragments before
      // the first MFMA -, so the fragment reads are inline asm it does not track, and each subtile's MFMAs sit behind a counted
      // wait that takes the fragments as operands: nothing that uses them can be scheduled above it.)
#if PM_TILE_ASM_READS
#define PM_TREAD1(dst_, base_, off_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(base_), "n"(off_))
#define PM_TWAIT(n_, j_)                                                                                             \
  if constexpr (MI == 5)                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a[j_]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[MI - 1]) : "n"(n_)); \
  else                                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a[j_]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(n_));
#define PM_TREADS()                                                                                                  \
  {                                                                                                                  \
    const uint32_t sw_ = (uint32_t)(((ss * 4 + fq) ^ ((fr >> 1) & 7)) * 16 + fr * 128);                              \
    const uint32_t xb_ = (uint32_t)(uintptr_t)(PM_LDS const char*)xcur + (uint32_t)(wm * WR * 128) + sw_;            \
    const uint32_t wb_ = (uint32_t)(uintptr_t)(PM_LDS const char*)xcur + (uint32_t)(wn * 128 * 128) + sw_;           \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) PM_TREAD1(b[i], xb_, i * 2048);                                   \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) PM_TREAD1(a[j], wb_, BM * 128 + j * 2048);                         \
  }
#else
#define PM_TWAIT(n_, j_)
#define PM_TREADS()                                                                                                  \
  {                                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) b[i] = tread(xcur, wm * WR + i * 16 + fr, ss * 4 + fq);           \
    a[0] = tread(wcur, wn * 128 + fr, ss * 4 + fq);                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    _Pragma("unroll") for (int j = 1; j < 8; ++j) {                                                                  \
      a[j] = tread(wcur, wn * 128 + j * 16 + fr, ss * 4 + fq);                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                             \
    }                                                                                                                \
  }
#endif
      // subtile j's MFMAs: its own fragment and everything older have landed (7 - j younger reads may still be in flight)
#define PM_TWAIT_J(j_)                                                                                               \
  {                                                                                                                  \
    if ((j_) == 0) { PM_TWAIT(7, 0) } else if ((j_) == 1) { PM_TWAIT(6, 1) } else if ((j_) == 2) { PM_TWAIT(5, 2) }  \
    else if ((j_) == 3) { PM_TWAIT(4, 3) } else if ((j_) == 4) { PM_TWAIT(3, 4) } else if ((j_) == 5) { PM_TWAIT(2, 5) } \
    else if ((j_) == 6) { PM_TWAIT(1, 6) } else { PM_TWAIT(0, 7) }                                                   \
  }
      if (kt == 0 && ss == 0) {
        PM_TREADS();
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          PM_TWAIT_J(j);
#pragma unroll
          for (int i = 0; i < MI; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], b[i], zero, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        PM_TREADS();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          PM_TWAIT_J(j);
#pragma unroll
          for (int i = 0; i < MI; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], b[i], acc[j][i], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#undef PM_TREADS
#undef PM_TWAIT_J
#undef PM_TWAIT
      if (MI > 4) __builtin_amdgcn_sched_barrier(0);  // 212 of 256 registers are accumulators + fragments: no hoisting of the next reads
      // (other placements measured equal or slower: both partners between a half step's fragment reads and its MFMAs; waves
      // 0-3 there and waves 4-7 here; four request points by wave pair)
      if (PM_TILE_ISSUE_SPLIT && ss == 0 && wave >= 4) {
        PM_TSTAGE_NEXT();  // waves 4-7: half a step behind their SIMD partners (see the top of the step)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    buf ^= 1;
    if (++kt < nk) continue;

    // ---------------- tile finished: epilogue (the next tile's first K step is already in flight)
    kt = 0;
#ifdef PM_TILE_ABLATE_EPI  // experiment only: keep the accumulators live, skip the epilogue
    {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(acc[j][i]));
      ++ti;
      continue;
    }
#endif
    int tm, tn;
    ttile_coords(tbase + local + ti * nloc, tiles_m, tiles_n, tm, tn);
    ++ti;
    const int m0 = tm * BM + wm * WR, n0 = tn * TBN + wn * 128;
    PM_TSTAMP(1, ti == 1);
    char* stg = smem + 2 * STAGE + wave * 2048;  // wave-private: no barrier in front of the epilogue
    int el = lane;
    asm volatile("" : "+v"(el));  // the epilogue's lane constants are recomputed per tile, not kept across the K loop (registers)
    const int srow = el >> 3, sch = el & 7, efr = el & 15, efq = el >> 4;
    // Every global address below is a UNIFORM 64-bit base (tile, wave, block: scalar registers) plus a 32-bit lane offset.
    // row-wise side: this lane stores rows m0 + 16 i + srow + 8 p, 8 features at n0 + 64 hf + 8 sch
    const uint32_t ylane = (uint32_t)srow * (uint32_t)(ldy * 2) + (uint32_t)sch * 16;
    const int mleft = M - m0 - srow;  // row 16 i + 8 p of this lane exists iff 16 i + 8 p < mleft
    char* const wr_ptr = stg + efr * 128 + (efq & 1) * 8;  // accumulator-layout side of the transposition
    const int wr_sw = (efr >> 1) & 7;
    const char* const rd_ptr0 = stg + srow * 128 + ((sch ^ ((srow >> 1) & 7)) * 16);
    const char* const rd_ptr1 = stg + (srow + 8) * 128 + ((sch ^ (((srow + 8) >> 1) & 7)) * 16);
    if constexpr (RES && MI > 4) {
#pragma unroll
      for (int q = 0; q < RPF; ++q) PM_TLOAD_RESID(q, q, m0, n0, efr, efq);
    }
    // bias (and the LayerNorm fold's column sums) of a 64-feature half: the first half's at the start, the second half's
    // requested in front of the first half's LAST block - the finished blocks' accumulators are dead by then (see the end of the
    // epilogue) - instead of at the top of the second half, where they were waited for with the whole workgroup idle
    f32x4 bvec[4], svec[LNC ? 4 : 1], bnxt[4], snxt[LNC ? 4 : 1];
#define PM_TLOAD_BIAS(bd_, sd_, hf_)                                                                                  \
  _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                                                   \
    int n = n0 + (hf_) * 64 + jj * 16 + efq * 4;                                                                       \
    n = n < N ? n : N - 4; /* features beyond N are never stored */                                                    \
    bd_[jj] = bias ? *(const f32x4*)((const char*)bias + (uint32_t)n * 4) : f32x4{0.f, 0.f, 0.f, 0.f};                 \
    if constexpr (LNC) sd_[jj] = *(const f32x4*)((const char*)ln.s + (uint32_t)n * 4);                                 \
  }
    PM_TLOAD_BIAS(bvec, svec, 0);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {  // 64-feature halves of the wave's 128 features
      if (hf == 1) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          bvec[jj] = bnxt[jj];
          if constexpr (LNC) svec[jj] = snxt[jj];
        }
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        if (hf == 0 && i == MI - 1) PM_TLOAD_BIAS(bnxt, snxt, 1);
        const int q = hf * MI + i;
        if constexpr (RES) {
          if (q + RPF < 2 * MI) PM_TLOAD_RESID(q + RPF, (q + RPF) % (RPF + 1), m0, n0, efr, efq);
          PM_TRESID_UNSWAP(q % (RPF + 1))
        }
        bf16x4 o[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          f32x4 v;
          if constexpr (LNC) {
            const float mu = lnst[i][0], rstd = lnst[i][1];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaf(rstd, acc[hf * 4 + jj][i][r] - mu * svec[jj][r], bvec[jj][r]);
          } else {
            v = acc[hf * 4 + jj][i] + bvec[jj];
          }
          v = apply_act4<ACT>(v);
          if constexpr (RES) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)rv[q % (RPF + 1)][jj][r];
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) o[jj][r] = (bf16)v[r];
          *(bf16x4*)(wr_ptr + (((jj * 2 + (efq >> 1)) ^ wr_sw) * 16)) = o[jj];
        }
        if (!LNC && ln.row_out) {
          // (sum, sum of squares) of this block's ROUNDED outputs per token: the next LayerNorm's partials
          bf16x8 ones;
#pragma unroll
          for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
          const bf16x8 f0 = __builtin_shufflevector(o[0], o[1], 0, 1, 2, 3, 4, 5, 6, 7);
          const bf16x8 f1 = __builtin_shufflevector(o[2], o[3], 0, 1, 2, 3, 4, 5, 6, 7);
          const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
          f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, f0, zero, 0, 0, 0);
          d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, f1, d1, 0, 0, 0);
          f32x4 d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f0, f0, zero, 0, 0, 0);
          d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1, f1, d2, 0, 0, 0);
          // d2[r] = G[4 fq + r][fr]: the diagonal element of token fr sits in the lane with fq == fr >> 2, component fr & 3
          const int rr = efr & 3;
          const float s2 = rr == 0 ? d2[0] : rr == 1 ? d2[1] : rr == 2 ? d2[2] : d2[3];
          const int np = N >> 6;
          char* const sb = (char*)ln.row_out + ((int64_t)(m0 + i * 16) * np + ((n0 + hf * 64) >> 6)) * 8;
          if (efq == (efr >> 2) && m0 + i * 16 + efr < M && n0 + hf * 64 < N)
            *(f32x2*)(sb + (uint32_t)efr * (uint32_t)(np * 8)) = f32x2{d1[0], s2};
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const bf16x8 ov = *(const bf16x8*)(p ? rd_ptr1 : rd_ptr0);
          char* const yb = (char*)Y + ((int64_t)(m0 + i * 16 + p * 8) * ldy + n0 + hf * 64) * 2;
#ifdef PM_TILE_ABLATE_STORE  // experiment only
          asm volatile("" ::"v"(ov), "v"(yb + ylane));
#else
          if (i * 16 + p * 8 < mleft && n0 + hf * 64 + sch * 8 < N)  // N % 8 == 0 on this path
            store_y((bf16x8*)(yb + ylane), ov);
#endif
        }
        PM_TSTAMP(2 + q, ti == 1);
      }
    }
#ifndef PM_TILE_NO_ACC_KILL
    // The accumulators END here: the next tile's first MFMAs start from zero, but in the flat loop hipcc cannot see that and keeps
    // all 32 MI accumulator registers live through the whole epilogue.  An empty asm that DEFINES them closes their live ranges at
    // each block's last use - the epilogue's own temporaries then fit where the finished blocks' accumulators were.
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i) asm volatile("" : "=v"(acc[j][i]));
#endif
  }
#undef PM_TSTAGE_NEXT
#undef PM_TLOAD_RESID
#undef PM_TLOAD_BIAS
}

}  // namespace

#if PM_TILE_STAMPS
extern "C" int pm_debug_tile_stamps(void* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_stamps), sizeof(unsigned long long) * 256 * 16) == hipSuccess ? 0 : 1;
}
#endif

// Internal entry (called from linear_bf16.hip's dispatcher; arguments already validated there).  mi = 4 or 5.
int pm_linear_bf16_tile_launch(int mi, const void* x, int64_t ldx, int64_t x_rows_per_batch, int64_t x_batch_stride,
                               const void* w, int64_t ldw, const float* bias, const void* resid, int64_t ldr,
                               int64_t resid_period, void* y, int64_t ldy, int64_t M, int64_t N, int64_t K, int act, PmLnFold ln,
                               hipStream_t st) {
  const int bm = 64 * mi;
  const int tiles_m = (int)((M + bm - 1) / bm), tiles_n = (int)((N + TBN - 1) / TBN);
  if (ln.stats && (resid || ln.row_out)) return PM_EUNSUPPORTED;  // the dispatcher keeps such calls on the 256 x 128 kernel
  const int epi = ln.stats ? EPI_LNC : resid ? EPI_RES : EPI_PLAIN;
#define PM_TGO(MI_, A, L)                                                                                              \
  hipLaunchKernelGGL((linear_bf16_tile_kernel<MI_, A, L>), dim3(256), dim3(512), 0, st, (const bf16*)x, ldx, (const bf16*)w, \
                     ldw, bias, (const bf16*)resid, ldr, (int)resid_period, (bf16*)y, ldy, (int)M, (int)N, (int)K, tiles_m, \
                     tiles_n, (int)x_rows_per_batch, x_batch_stride, ln)
#define PM_TGO_MI(MI_)                                             \
  if (act == PM_ACT_NONE) {                                        \
    if (epi == EPI_LNC) PM_TGO(MI_, PM_ACT_NONE, EPI_LNC);         \
    else if (epi == EPI_RES) PM_TGO(MI_, PM_ACT_NONE, EPI_RES);    \
    else PM_TGO(MI_, PM_ACT_NONE, EPI_PLAIN);                      \
  } else if (act == PM_ACT_GELU) {                                 \
    if (epi == EPI_LNC) PM_TGO(MI_, PM_ACT_GELU, EPI_LNC);         \
    else if (epi == EPI_RES) PM_TGO(MI_, PM_ACT_GELU, EPI_RES);    \
    else PM_TGO(MI_, PM_ACT_GELU, EPI_PLAIN);                      \
  } else {                                                         \
    return PM_EUNSUPPORTED;                                        \
  }
  if (mi == 4) { PM_TGO_MI(4) }
  else if (mi == 5) { PM_TGO_MI(5) }
  else return PM_EINVAL;
#undef PM_TGO_MI
#undef PM_TGO
  return PM_OK;
}

bool pm_linear_bf16_tile_applies(int64_t M, int64_t N, int64_t K, int act) {
  // shape eligibility only; which tile height is the FASTEST is decided by the cost model of linear_impl
  if (act != PM_ACT_NONE && act != PM_ACT_GELU) return false;
  if (K % TBK || N % 8) return false;
  return M >= 4096;
}
