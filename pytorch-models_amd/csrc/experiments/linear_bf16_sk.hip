// linear_bf16_sk.hip - the 256 x 256 x 64 GEMM of linear_bf16_wide.hip with its K steps dealt out as ONE stream ("stream-K").
// (same operation and reference sites as linear_bf16.hip: pytorch_models/transformer.py:28-31,47-53,59-66)
//
// Why.  ViT-B/16 at batch 256 has M = 50432 = 197 panels of 256 tokens: N = 768 gives 591 tiles of 256 x 256 = 2.31 rounds
// over 256 persistent workgroups, run as 3 (out_proj, linear2: a third of the chip idles in the last round - which is why
// those layers sat on the 256 x 128 kernel, 4.62 rounds run as 5); N = 3072 gives 9.23 run as 10.  Here a tile's K steps
// are the unit: each XCD's contiguous chunk of tiles is a stream of tiles * K / 64 steps cut into equal contiguous ranges,
// one per workgroup, so every workgroup issues the same number of MFMAs whatever the tile count.  A range starts and
// ends inside a tile in general: the two workgroups that share a tile each hold a partial accumulator, and the one that
// finds the other's already published adds it to its own and runs the epilogue:
//   - the part computed at the START of a range (the tile's last K steps) is always published: the workgroup stores its
//     128 accumulator registers per lane lane-linearly (1 KiB per store instruction, sc1), every wave drains (vmcnt(0)),
//     the workgroup meets at a barrier, one lane adds 1 to the boundary's ticket;
//   - the part computed at the END of a range (the tile's first K steps, typically 50-250 us later) first LOADS the
//     ticket: 1 = the other part is there - no store at all, it reads that part (sc1) and finishes the tile.  Only if
//     the other side is late does it publish as well and add to the ticket; whichever add returns 1 finishes.
// No spinning anywhere, so no residency assumption and no deadlock; both orders give a + b in fp32, so the result does not
// depend on who finishes; the finisher zeroes the ticket for the next launch (MI355X_MICROARCH.md, inter-workgroup
// visibility: counter told by the value an add returned / an sc1 load of it, workgroup barrier between that and every
// sc1 load of the bytes).  Workspace = caller-owned: pm_linear_sk_workspace_bytes().
// HYBRID mode (mode = 1; opt-in, PM_GEMM_HYBRID=1: see pm_linear_pick_kernel for why it is not the default): whole tiles dealt out exactly as linear_bf16_wide.hip deals them
// (workgroup l of an XCD takes tiles l, l + 32, ... of the XCD's chunk: neighbours share panels in L2) and only the
// rem = tiles mod 32 tiles of the last, partly filled round are cut - in two K halves, for workgroups 2u and 2u + 1:
// 2u + 1 computes the tile's last K steps FIRST (and publishes them), 2u its first K steps LAST (and finds them there): the
// same two-party hand-off as above, one 256 KiB partial per tail tile and direction instead of one per workgroup, and a
// last round that costs half a tile: N = 768 at M = 50432 runs 2.5 rounds instead of 3.  Needs 2 rem <= 32 per XCD.
// The epilogue is the wide kernel's plus, in the RS instantiation, the LayerNorm fold's row partials (sum, sum of squares
// of each row's ROUNDED outputs per 64-feature block) so that out_proj / linear2 of a fold chain can run here.
#include <cstdlib>

#include "../common.h"
#include "../../../include/pm_mi355x_experiments.h"

namespace {

constexpr int SBM = 256, SBN = 256, SBK = 64;
constexpr int SSTAGE = (SBM + SBN) * SBK * 2;  // 64 KiB
constexpr int SGROUP_M = 4;
constexpr int SK_GRID = 256;
constexpr int64_t SK_AREA = 8 * 32 * 1024;          // one partial accumulator tile: 8 waves x 32 registers x 64 lanes x 16 B
constexpr int64_t SK_TICK_BYTES = 4096;              // SK_GRID + 1 tickets, padded
constexpr int64_t SK_WS_BYTES = SK_TICK_BYTES + (int64_t)(SK_GRID + 1) * 2 * SK_AREA;

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ bf16x8 sread(const char* tile, int row, int chunk) {
  return *(const bf16x8*)(tile + row * 128 + swz_pos(row, chunk) * 16);
}

__device__ __forceinline__ void stile_coords(int t, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int per = SGROUP_M * tiles_n;
  const int sr = t / per, r = t - sr * per;
  const int left = tiles_m - sr * SGROUP_M;
  const int gm = left < SGROUP_M ? left : SGROUP_M;
  tn = r / gm;
  tm = sr * SGROUP_M + (r - tn * gm);
}

template <int ACT, bool RS>
__global__ __launch_bounds__(512, 2) void linear_bf16_sk_kernel(
    const bf16* __restrict__ X, int64_t ldx, const bf16* __restrict__ W, int64_t ldw, const float* __restrict__ bias,
    const bf16* resid, int64_t ldr, int resid_period, bf16* Y, int64_t ldy, int M, int N, int K, int tiles_m, int tiles_n,
    int x_rows_per_batch, int64_t x_batch_stride, PmLnFold ln, char* ws, int mode) {
  __shared__ __attribute__((aligned(16))) char smem[2 * SSTAGE + 8 * 4096];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntiles = tiles_m * tiles_n;
  int* const flag = (int*)(smem + 2 * SSTAGE);  // wave 0's epilogue staging: idle during the partial-tile exchange, every read of the flag sits between two barriers
  int* const tick = (int*)ws;
  const __amdgpu_buffer_rsrc_t area = __builtin_amdgcn_make_buffer_rsrc(ws + SK_TICK_BYTES, 0, (int)(SK_WS_BYTES - SK_TICK_BYTES), 0x00020000);

  // this XCD's contiguous chunk of tiles as a stream of K steps; this workgroup's contiguous range of it
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, nloc = gridDim.x >> 3;
  const int cq = ntiles >> 3, cr = ntiles & 7;
  const int tbase = xcd * cq + (xcd < cr ? xcd : cr), tcount = cq + (xcd < cr ? 1 : 0);
  const int nk = K / SBK;
  const int gid = xcd * nloc + local;  // boundary gid = start of this range, gid + 1 = its end
  // this workgroup's sequence of tiles: sequence index -> tile id = tbase + (seq_mul * index + seq_add), except index
  // tail_pos -> tail_id; the range = P steps starting at step kt0 of sequence index 0; late_start = the step at which the
  // range's last, partial tile begins (-1: the range ends on a tile boundary)
  int P, kt0, late_start, seq_mul, seq_add, tail_pos = -1, tail_id = 0;
  if (mode == 0) {
    const int CS = tcount * nk;
    const int s0 = (int)((int64_t)CS * local / nloc), s1 = (int)((int64_t)CS * (local + 1) / nloc);
    P = s1 - s0;
    kt0 = s0 % nk;
    late_start = (s1 % nk) ? P - (s1 % nk) : -1;
    seq_mul = 1;
    seq_add = s0 / nk;
  } else {
    const int R = tcount / nloc, rem = tcount - R * nloc, h = nk >> 1;
    const int u = local >> 1;
    const bool has_tail = u < rem;  // 2 rem <= nloc (checked on the host)
    seq_mul = nloc;
    seq_add = local;
    P = R * nk;
    kt0 = 0;
    late_start = -1;
    if (has_tail) {
      tail_id = tbase + R * nloc + u;
      if (local & 1) {  // the tile's steps [h, nk) at the START of the range
        tail_pos = 0;
        seq_add = local - nloc;  // whole tiles at sequence indices 1 .. R
        kt0 = h;
        P += nk - h;
      } else {          // steps [0, h) at the END
        tail_pos = R;
        late_start = P;
        P += h;
      }
    }
  }
#define PM_STILE(IDX) ((IDX) == tail_pos ? tail_id : tbase + seq_mul * (IDX) + seq_add)

  uint32_t xoff[4], woff[4];
  int pp = 0, pp_kt = kt0, pp_tile = 0, pp_buf = 0;
#define PM_SSTAGE_NEXT()                                                                                             \
  if (pp < P) {                                                                                                      \
    if (pp_kt == 0 || pp == 0) {                                                                                     \
      int tm_, tn_;                                                                                                  \
      stile_coords(PM_STILE(pp_tile), tiles_m, tiles_n, tm_, tn_);                                                     \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                \
        const int rt = wave * 32 + i * 8 + (lane >> 3);                                                              \
        const int chunk = swz_pos(rt, lane & 7);                                                                     \
        int gm = tm_ * SBM + rt;                                                                                     \
        gm = gm < M ? gm : M - 1;                                                                                    \
        if (x_rows_per_batch > 0) {                                                                                  \
          const int bb = gm / x_rows_per_batch;                                                                      \
          xoff[i] = (uint32_t)((int64_t)bb * x_batch_stride + (int64_t)(gm - bb * x_rows_per_batch) * ldx + chunk * 8); \
        } else {                                                                                                     \
          xoff[i] = (uint32_t)((int64_t)gm * ldx + chunk * 8);                                                        \
        }                                                                                                            \
        int gn = tn_ * SBN + rt;                                                                                     \
        gn = gn < N ? gn : N - 1;                                                                                    \
        woff[i] = (uint32_t)((int64_t)gn * ldw + chunk * 8);                                                          \
      }                                                                                                              \
    }                                                                                                                \
    char* xs_ = smem + pp_buf * SSTAGE;                                                                              \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) glds16(X + xoff[i] + pp_kt * SBK, xs_ + (wave * 32 + i * 8) * 128); \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                    \
        glds16(W + woff[i] + pp_kt * SBK, xs_ + SBM * 128 + (wave * 32 + i * 8) * 128);                               \
    ++pp;                                                                                                            \
    pp_buf ^= 1;                                                                                                     \
    if (++pp_kt == nk) { pp_kt = 0; ++pp_tile; }                                                                     \
  }

  f32x4 acc[8][4];  // [feature subtile j][token subtile i]
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (wave >= 4) __builtin_amdgcn_s_setprio(1);  // the younger wave of each SIMD pair loses every arbitration otherwise
  PM_SSTAGE_NEXT();
  const int fr = lane & 15, fq = lane >> 4;
  int buf = 0, kt = kt0, ti = 0, seg_k0 = kt;
  bool seeded = false;
  f32x2 lnst[4] = {{0.f, 1.f}, {0.f, 1.f}, {0.f, 1.f}, {0.f, 1.f}};
  for (int pc = 0; pc < P; ++pc) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // two stages: only step pc itself was in flight
    __builtin_amdgcn_s_barrier();                     // every wave's part landed; every wave is past step pc - 1
    if (pc == late_start) {
      // first step of this range's last, partial tile: the other part of that tile was computed at the START of the next
      // workgroup's range, i.e. at least one whole tile of K steps ago - look for it (a short, bounded wait: the partner is
      // resident and runs at the same pace) and continue FROM it: the accumulators start as its values instead of zero,
      // so that nothing is added later and no temporaries are needed
      if (tid == 0) {
        int v = 0;
        for (int spin = 0; spin < 4096; ++spin) {
          v = __hip_atomic_load(tick + gid + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (v) break;
          __builtin_amdgcn_s_sleep(4);
        }
        *flag = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      seeded = *flag != 0;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (seeded) {
        const int other = ((gid + 1) * 2) * (int)SK_AREA;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[j][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(area, other + lane * 16,
                                                                                       (wave * 32 + j * 4 + i) * 1024, 16));
        // wait for them HERE with the builtin (which the compiler's wait-count pass models): left pending, the accumulators'
        // loads force a vmcnt(0) in front of the MFMAs of EVERY step - the merge of this rare path with the common one -
        // and that wait also covers the LDS-DMA of the next step: -30 % at 8192^3 before this line
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
      }
    }
    PM_SSTAGE_NEXT();
    bool have_ln = false;
    if (kt == nk - 1 && ln.stats) {  // LayerNorm fold, last K step of the tile: (mean, rstd) of this lane's four token rows
      int tm_r, tn_r;
      stile_coords(PM_STILE(ti), tiles_m, tiles_n, tm_r, tn_r);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int mr = tm_r * SBM + wm * 64 + i * 16 + fr;
        mr = mr < M ? mr : M - 1;
        lnst[i] = *(const f32x2*)(ln.stats + 2 * (int64_t)mr);
      }
      have_ln = true;
    }
    const char* xcur = smem + buf * SSTAGE;
    const char* wcur = xcur + SBM * 128;
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      bf16x8 a[8], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = sread(xcur, wm * 64 + i * 16 + fr, ss * 4 + fq);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = sread(wcur, wn * 128 + j * 16 + fr, ss * 4 + fq);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], b[i], acc[j][i], 0, 0, 0);
    }
    buf ^= 1;
    const bool tile_end = kt == nk - 1, range_end = pc == P - 1;
    if (!tile_end && !range_end) { ++kt; continue; }

    // ---------------- a segment [seg_k0, kt] of tile ti is complete
    int tm, tn;
    stile_coords(PM_STILE(ti), tiles_m, tiles_n, tm, tn);
    bool do_epi = seg_k0 == 0 && tile_end;
    if (!do_epi) {
      const bool early = seg_k0 > 0;  // the tile's last K steps, computed at the start of this range
      const int bnd = early ? gid : gid + 1;
      const int mine = (bnd * 2 + (early ? 0 : 1)) * (int)SK_AREA, other = (bnd * 2 + (early ? 1 : 0)) * (int)SK_AREA;
      int fin = 0;
      if (!early && seeded) {  // the accumulators started from the other part: the tile is complete
        fin = 2;
      } else {  // publish this part
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[j][i]), area, mine + lane * 16,
                                                   (wave * 32 + j * 4 + i) * 1024, 16);  // register index in the scalar offset
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid == 0) *flag = __hip_atomic_fetch_add(tick + bnd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        fin = *flag;  // 1: the other part was published before this add (only when its owner did not find this one in time)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (fin) {
          // rare path: add the other part one register at a time (each load waits for itself: four temporaries in all;
          // with the loads batched the compiler spilled 137 registers of the K loop)
          const f32x4* op = (const f32x4*)(ws + SK_TICK_BYTES + other) + lane;
#pragma unroll
          for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] += load_sc1_x4(op + (wave * 32 + j * 4 + i) * 64);
        }
      }
      if (fin) {
        if (tid == 0) __hip_atomic_store(tick + bnd, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        do_epi = true;
      }
    }
    if (do_epi) {
      if (ln.stats && !have_ln) {  // finishing from the middle of a tile: the rows' (mean, rstd) were not requested yet
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int mr = tm * SBM + wm * 64 + i * 16 + fr;
          mr = mr < M ? mr : M - 1;
          lnst[i] = *(const f32x2*)(ln.stats + 2 * (int64_t)mr);
        }
      }
      const int m0 = tm * SBM + wm * 64, n0 = tn * SBN + wn * 128;
      char* stg = smem + 2 * SSTAGE + wave * 4096;  // every wave stages through its own 4 KiB: no barrier in the epilogue
      const int srow = lane >> 3, sch = lane & 7;
      bf16* const ybase = Y + (int64_t)(m0 + srow) * ldy + n0 + sch * 8;
      const bf16* const rbase = resid && !resid_period ? resid + (int64_t)(m0 + srow) * ldr + n0 + sch * 8 : nullptr;
      const int mleft = M - m0 - srow;  // row k of this lane exists iff 8 k < mleft
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {  // 64-feature halves of the wave's 128 features
        f32x4 bvec[4], svec[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int n = n0 + hf * 64 + jj * 16 + fq * 4;
          bvec[jj] = (bias && n < N) ? *(const f32x4*)(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
          svec[jj] = (ln.stats && n < N) ? *(const f32x4*)(ln.s + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float* const sbase = RS ? ln.row_out + ((int64_t)(m0 + srow) * (N >> 6) + ((n0 + hf * 64) >> 6)) * 2 : nullptr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float mu = lnst[i][0], rstd = lnst[i][1];  // (0, 1) without the LayerNorm fold
          bf16x8 rv[2];
          if (rbase) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
              const bool ok = i * 16 + p * 8 < mleft && n0 + hf * 64 + sch * 8 < N;
              rv[p] = ok ? *(const bf16x8*)(rbase + (int64_t)(i * 16 + p * 8) * ldr + hf * 64) : bf16x8{};
            }
          } else if (resid) {  // periodic residual (row m reads row m % resid_period: a position table)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
              int mm = m0 + i * 16 + srow + p * 8;
              mm = mm < M ? mm : M - 1;
              int nn = n0 + hf * 64 + sch * 8;
              nn = nn < N ? nn : N - 8;
              rv[p] = *(const bf16x8*)(resid + (int64_t)(mm % resid_period) * ldr + nn);
            }
          }
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaf(rstd, acc[hf * 4 + jj][i][r] - mu * svec[jj][r], bvec[jj][r]);
            v = apply_act4<ACT>(v);
            *(f32x4*)(stg + fr * 256 + (((4 * jj + fq) ^ fr) * 16)) = v;
          }
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const int row = srow + p * 8;
            const f32x4 lo = *(const f32x4*)(stg + row * 256 + (((2 * sch) ^ row) * 16));
            const f32x4 hi = *(const f32x4*)(stg + row * 256 + (((2 * sch + 1) ^ row) * 16));
            bf16x8 o;
            if (resid) {
#pragma unroll
              for (int r = 0; r < 4; ++r) { o[r] = (bf16)(lo[r] + (float)rv[p][r]); o[4 + r] = (bf16)(hi[r] + (float)rv[p][4 + r]); }
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) { o[r] = (bf16)lo[r]; o[4 + r] = (bf16)hi[r]; }
            }
            const bool ok = i * 16 + p * 8 < mleft && n0 + hf * 64 + sch * 8 < N;  // N % 8 == 0 on this path
            if constexpr (RS) {  // (sum, sum of squares) of this row's 64 ROUNDED outputs: the next LayerNorm's partials
              float s1 = 0.f, s2 = 0.f;
#pragma unroll
              for (int r = 0; r < 8; ++r) { const float f = (float)o[r]; s1 += f; s2 = fmaf(f, f, s2); }
              s1 = sum8_dpp(s1);
              s2 = sum8_dpp(s2);
              if (sch == 0 && ok) *(f32x2*)(sbase + (int64_t)(i * 16 + p * 8) * (N >> 6) * 2) = f32x2{s1, s2};
            }
            if (ok) *(bf16x8*)(ybase + (int64_t)(i * 16 + p * 8) * ldy + hf * 64) = o;
          }
        }
      }
    }
    if (tile_end) { kt = 0; ++ti; } else { ++kt; }
    seg_k0 = kt;
    // the next segment starts from zero.  (Starting it with MFMAs on a constant-zero C instead - the wide kernel's trick -
    // needs a second copy of the 64-MFMA block under a loop-carried flag: 269 spilled registers here.)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#undef PM_SSTAGE_NEXT
#undef PM_STILE
}

}  // namespace

int64_t pm_linear_sk_ws_bytes() { return SK_WS_BYTES; }

// Shape eligibility (the dispatcher of linear_bf16.hip decides whether it is the fastest kernel): every workgroup's
// range must hold at least one whole tile's worth of K steps, so that a tile is shared by at most two workgroups.
bool pm_linear_bf16_sk_applies(int64_t M, int64_t N, int64_t K, int act) {
  if (act != PM_ACT_NONE && act != PM_ACT_GELU) return false;
  if (K % SBK || N % 8 || M < 4096) return false;
  const int64_t ntiles = ((M + SBM - 1) / SBM) * ((N + SBN - 1) / SBN);
  return ntiles / 8 >= SK_GRID / 8;  // the smallest XCD chunk has at least one tile per workgroup of the XCD
}

// Hybrid mode: whole tiles as the wide kernel + the last round's tiles in two K halves.  Eligible when every XCD's chunk
// leaves at most 16 tiles over (a pair of workgroups per tail tile) and a K half exists.
bool pm_linear_bf16_hyb_applies(int64_t M, int64_t N, int64_t K, int act) {
  if (!pm_linear_bf16_sk_applies(M, N, K, act) || K / SBK < 2) return false;
  const int64_t ntiles = ((M + SBM - 1) / SBM) * ((N + SBN - 1) / SBN);
  const int64_t cq = ntiles >> 3, cr = ntiles & 7;
  const int64_t rem_hi = (cq + (cr ? 1 : 0)) % (SK_GRID / 8), rem_lo = cq % (SK_GRID / 8);
  return 2 * rem_hi <= SK_GRID / 8 && 2 * rem_lo <= SK_GRID / 8 && (rem_hi > 0 || rem_lo > 0);
}

int pm_linear_bf16_sk_launch(const void* x, int64_t ldx, int64_t x_rows_per_batch, int64_t x_batch_stride, const void* w,
                             int64_t ldw, const float* bias, const void* resid, int64_t ldr, int64_t resid_period, void* y,
                             int64_t ldy, int64_t M, int64_t N, int64_t K, int act, PmLnFold ln, void* ws, hipStream_t st,
                             int mode) {
  const int tiles_m = (int)((M + SBM - 1) / SBM), tiles_n = (int)((N + SBN - 1) / SBN);
#define PM_SGO(A, R)                                                                                                     \
  hipLaunchKernelGGL((linear_bf16_sk_kernel<A, R>), dim3(SK_GRID), dim3(512), 0, st, (const bf16*)x, ldx, (const bf16*)w, ldw, \
                     bias, (const bf16*)resid, ldr, (int)resid_period, (bf16*)y, ldy, (int)M, (int)N, (int)K, tiles_m,     \
                     tiles_n, (int)x_rows_per_batch, x_batch_stride, ln, (char*)ws, mode)
  if (ln.row_out) {
    if (act != PM_ACT_NONE) return PM_EUNSUPPORTED;
    PM_SGO(PM_ACT_NONE, true);
  } else if (act == PM_ACT_NONE) PM_SGO(PM_ACT_NONE, false);
  else if (act == PM_ACT_GELU) PM_SGO(PM_ACT_GELU, false);
  else return PM_EUNSUPPORTED;
#undef PM_SGO
  return PM_OK;
}
