// logmel.hip - framed real DFT (torch.stft semantics) -> power -> mel filterbank -> Whisper log-mel.
// (reference: pytorch_models/audio/spectrogram.py:15-16 (stft, center=True, reflect, hann, |.|^2),
//  :44-45 (filters @ power), pytorch_models/audio2text/whisper.py:143-148 (drop last frame, log10,
//  per-sample max - 8 floor, (x + 4) / 4).)
//
// Roofline: HBM-bound by bytes (a 30 s clip is 1.92 MB in, 0.96 MB out) but its arithmetic is an exact-fp32
// contraction: 3000 frames x 400 samples x 402 twiddles = 0.97 GFLOP per clip on the f32 MFMA
// (v_mfma_f32_32x32x2_f32, bitwise an fp32 FMA chain, 157 TFLOP/s peak), i.e. ~6 us per clip of matrix time
// against ~0.5 us of HBM time: priced against BOTH in DESIGN.md.
//
// Workgroup = 32 consecutive frames of one clip, 4 waves.  The 31*hop + n_fft samples the frames cover are
// loaded once (reflect padding resolved at load) into LDS with a one-float skew per hop so that the 32 lanes
// of an MFMA B-operand read (same k, frames j = 0..31 -> addresses (hop+1)*j + const) hit 32 different banks.
// Wave w owns frequency blocks w, w+4, ... (32 bins each); for each block it runs n_fft/2 steps of two
// MFMAs (cos and sin twiddles as the A operand, window folded in, streamed coalesced from an L2-resident
// table; the frame samples as the B operand), squares and adds the two accumulators into power[bin][frame]
// and either stores it (spectrogram mode) or parks it in an LDS tile that the mel phase contracts with the
// CSR form of the (sparse, banded) filterbank.  The log-mel mode also reduces the per-clip maximum with one
// atomicMax per wave; a second tiny kernel applies max(x, peak - 8), (x + 4) / 4.
#include "common.h"

namespace {

constexpr int FT = 32;  // frames per workgroup

__device__ __forceinline__ int enc_max(float f) {  // order-preserving float -> int for atomicMax
  const int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float dec_max(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

// FOLD: the window is symmetric (w[n] == w[N - n], every Hann / Hamming / ...), so w[n] cos is even and w[n] sin odd about N / 2:
//   Re[k] = sum_{n=0}^{N/2} c[k][n] (x[n] + x[N - n]),  Im[k] = sum_{n=1}^{N/2-1} s[k][n] (x[n] - x[N - n])
// with the n = N / 2 cosine entry halved and x[N] := 0 - half the steps of the plain form for two more LDS reads and two adds
// per step (the tables then hold n = 0 .. N/2 (+ zero padding to an even count) instead of n = 0 .. N - 1).
template <bool FOLD>
__global__ __launch_bounds__(256) void stft_mel_kernel(const float* __restrict__ x, int64_t x_stride, int T,
                                                       const float* __restrict__ tw_cos, const float* __restrict__ tw_sin,
                                                       int n_fft, int hop, int nbins, int nblk, int n_frames,
                                                       int tiles_per_clip, int mode, const int* __restrict__ mel_ptr,
                                                       const int* __restrict__ mel_col, const float* __restrict__ mel_val,
                                                       int n_mels, float* __restrict__ out, int* __restrict__ peak,
                                                       int nsamp_skewed) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* smp = lds;                  // skewed samples
  float* pw = lds + nsamp_skewed;    // power tile [bin][32 frames]
  // FOLD: skewed offsets of sample n and of its mirror N - n inside a frame, n = 0 .. 2 * steps - 1 (behind the power tile)
  int* fpos = (int*)(lds + nsamp_skewed + (mode ? nbins * FT : 0));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / tiles_per_clip;
  const int t0 = (blockIdx.x - b * tiles_per_clip) * FT;
  const float* xb = x + (int64_t)b * x_stride;

  // ---- samples of frames t0 .. t0+31: padded index p = t*hop + k maps to sample p - n_fft/2, reflected
  const int nsamp = hop * (FT - 1) + n_fft;
  const int s_base = t0 * hop - n_fft / 2;
  for (int i = tid; i < nsamp; i += 256) {
    int s = s_base + i;
    if (s < 0) s = -s;
    if (s >= T) s = 2 * (T - 1) - s;
    s = s < 0 ? 0 : (s >= T ? T - 1 : s);  // frames past the clip end (masked at the store) may reflect twice
    smp[i + i / hop] = xb[s];
  }
  const int steps = FOLD ? ((n_fft >> 1) + 2) >> 1 : n_fft >> 1;
  if constexpr (FOLD) {
    for (int n = tid; n < 2 * steps; n += 256) {
      const int nn = n <= n_fft / 2 ? n : n_fft / 2;  // zero-weighted padding entries: any valid sample
      const int m = nn == 0 ? 0 : n_fft - nn;         // x[N] does not belong to the frame: n = 0 reads x[0] and drops the mirror
      fpos[2 * n] = nn + nn / hop;
      fpos[2 * n + 1] = m + m / hop;
    }
  }
  __syncthreads();

  const int j = lane & 31, kk = lane >> 5;
  for (int blk = wave; blk < nblk; blk += 4) {
    f32x16 re, im;
#pragma unroll
    for (int r = 0; r < 16; ++r) { re[r] = 0.f; im[r] = 0.f; }
    const float* pc = tw_cos + (int64_t)blk * steps * 64 + lane;
    const float* ps = tw_sin + (int64_t)blk * steps * 64 + lane;
    const float* sj = smp + (hop + 1) * j + kk;
    if constexpr (FOLD) {
      const float* fj = smp + (hop + 1) * j;
#pragma unroll 4
      for (int u = 0; u < steps; ++u) {
        const int n = 2 * u + kk;
        const int2 pp = *(const int2*)(fpos + 2 * n);
        const float a = fj[pp.x];
        const float m = n ? fj[pp.y] : 0.f;
        re = __builtin_amdgcn_mfma_f32_32x32x2f32(pc[u * 64], a + m, re, 0, 0, 0);
        im = __builtin_amdgcn_mfma_f32_32x32x2f32(ps[u * 64], a - m, im, 0, 0, 0);
      }
    }
    // k = 2*s + kk; its skew is k / hop: walk hop-sized segments so the skew is a loop constant
    int s = FOLD ? steps : 0;
    for (int seg = 0; s < steps; ++seg) {
      int s_end = ((seg + 1) * hop) >> 1;  // hop is even: segment boundaries fall between steps
      if (s_end > steps) s_end = steps;
      const float* sp = sj + seg;
      const int cnt = s_end - s;
      const float* spp = sp + 2 * s;
      const float* pcc = pc + (int64_t)s * 64;
      const float* pss = ps + (int64_t)s * 64;
#pragma unroll 8
      for (int u = 0; u < cnt; ++u) {
        const float bv = spp[2 * u];
        const float ac = pcc[u * 64];
        const float as = pss[u * 64];
        re = __builtin_amdgcn_mfma_f32_32x32x2f32(ac, bv, re, 0, 0, 0);
        im = __builtin_amdgcn_mfma_f32_32x32x2f32(as, bv, im, 0, 0, 0);
      }
      s = s_end;
    }
    // D[row = bin][col = frame j]; rows (r & 3) + 8 * (r >> 2) + 4 * kk
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int bin = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
      const float p = re[r] * re[r] + im[r] * im[r];
      if (bin < nbins) {
        if (mode == 0) {
          if (t0 + j < n_frames) out[((int64_t)b * nbins + bin) * n_frames + t0 + j] = p;
        } else {
          pw[bin * FT + j] = p;
        }
      }
    }
  }
  if (mode == 0) return;
  __syncthreads();

  // ---- mel phase: thread -> (frame j, mel rows m = tid/32, +8, ...)
  float mx = -INFINITY;
  const int fj = tid & 31;
  for (int m = tid >> 5; m < n_mels; m += 8) {
    float acc = 0.f;
    const int e1 = mel_ptr[m + 1];
    for (int e = mel_ptr[m]; e < e1; ++e) acc = fmaf(mel_val[e], pw[mel_col[e] * FT + fj], acc);
    const float v = mode == 2 ? log10f(acc) : acc;
    if (t0 + fj < n_frames) {
      out[((int64_t)b * n_mels + m) * n_frames + t0 + fj] = v;
      mx = fmaxf(mx, v);
    }
  }
  if (mode == 2) {
    mx = wave_max(mx);
    if (lane == 0) atomicMax(peak + b, enc_max(mx));
  }
}

__global__ __launch_bounds__(256) void logmel_finalize_kernel(float* __restrict__ out, const int* __restrict__ peak,
                                                              int64_t per_clip, int64_t total) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= total) return;
  // per_clip % 4 == 0 is checked on the host, so the 4 elements share a clip
  const float floor_v = dec_max(peak[i / per_clip]) - 8.0f;
  f32x4 v = *(f32x4*)(out + i);
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = (fmaxf(v[r], floor_v) + 4.0f) * 0.25f;
  *(f32x4*)(out + i) = v;
}

__global__ void fill_int_kernel(int* p, int v, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}

}  // namespace

static int stft_mel_impl(const float* x, int64_t x_stride, int64_t B, int64_t T, const float* tw_cos, const float* tw_sin,
                         int64_t n_fft, int64_t hop, int64_t n_frames, int mode, const int32_t* mel_ptr, const int32_t* mel_col,
                         const float* mel_val, int64_t n_mels, float* out, int32_t* peak, bool fold, void* stream) {
  if (!x || !tw_cos || !tw_sin || !out || B < 0 || T <= 0 || n_frames < 0) return PM_EINVAL;
  if (mode < 0 || mode > 2) return PM_EINVAL;
  if (mode != 0 && (!mel_ptr || !mel_col || !mel_val || n_mels <= 0)) return PM_EINVAL;
  if (mode == 2 && !peak) return PM_EINVAL;
  if (B == 0 || n_frames == 0) return PM_OK;
  if (n_fft < 2 || n_fft % 2 || hop < 2 || hop % 2 || n_fft > 2048 || T <= n_fft / 2) return PM_EUNSUPPORTED;
  if (n_frames > 1 + T / hop) return PM_EINVAL;
  const int nbins = (int)(n_fft / 2 + 1), nblk = (nbins + 31) / 32;
  const int nsamp = (int)(hop * (FT - 1) + n_fft);
  const int nsamp_skewed = ((nsamp + nsamp / (int)hop + 1) + 3) & ~3;
  const int steps_f = ((int)(n_fft / 2) + 2) / 2;
  const int fold_ints = fold ? 4 * steps_f : 0;  // two offsets (sample, mirror) for each of the 2 * steps_f table positions
  const size_t lds_bytes = (size_t)(nsamp_skewed + (mode ? nbins * FT : 0) + fold_ints) * 4;
  if (lds_bytes > 64 * 1024) return PM_EUNSUPPORTED;
  const int tiles_per_clip = (int)((n_frames + FT - 1) / FT);
  if (B * tiles_per_clip > 0x7fffffff) return PM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (mode == 2)
    hipLaunchKernelGGL(fill_int_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, (int*)peak, (int)0x80000000,
                       (int)B);
  if (fold)
    hipLaunchKernelGGL(stft_mel_kernel<true>, dim3((unsigned)(B * tiles_per_clip)), dim3(256), lds_bytes, st, x, x_stride, (int)T,
                       tw_cos, tw_sin, (int)n_fft, (int)hop, nbins, nblk, (int)n_frames, tiles_per_clip, mode, mel_ptr,
                       mel_col, mel_val, (int)n_mels, out, (int*)peak, nsamp_skewed);
  else
    hipLaunchKernelGGL(stft_mel_kernel<false>, dim3((unsigned)(B * tiles_per_clip)), dim3(256), lds_bytes, st, x, x_stride, (int)T,
                       tw_cos, tw_sin, (int)n_fft, (int)hop, nbins, nblk, (int)n_frames, tiles_per_clip, mode, mel_ptr,
                       mel_col, mel_val, (int)n_mels, out, (int*)peak, nsamp_skewed);
  PM_CHECK_LAUNCH();
  return PM_OK;
}

extern "C" int pm_stft_mel(const float* x, int64_t x_stride, int64_t B, int64_t T, const float* tw_cos,
                           const float* tw_sin, int64_t n_fft, int64_t hop, int64_t n_frames, int mode,
                           const int32_t* mel_ptr, const int32_t* mel_col, const float* mel_val, int64_t n_mels,
                           float* out, int32_t* peak, void* stream) {
  return stft_mel_impl(x, x_stride, B, T, tw_cos, tw_sin, n_fft, hop, n_frames, mode, mel_ptr, mel_col, mel_val, n_mels, out, peak,
                       false, stream);
}

extern "C" int pm_stft_mel_folded(const float* x, int64_t x_stride, int64_t B, int64_t T, const float* tw_cos,
                                  const float* tw_sin, int64_t n_fft, int64_t hop, int64_t n_frames, int mode,
                                  const int32_t* mel_ptr, const int32_t* mel_col, const float* mel_val, int64_t n_mels,
                                  float* out, int32_t* peak, void* stream) {
  return stft_mel_impl(x, x_stride, B, T, tw_cos, tw_sin, n_fft, hop, n_frames, mode, mel_ptr, mel_col, mel_val, n_mels, out, peak,
                       true, stream);
}

extern "C" int pm_logmel_finalize(float* out, const int32_t* peak, int64_t B, int64_t per_clip, void* stream) {
  if (!out || !peak || B < 0 || per_clip <= 0) return PM_EINVAL;
  if (B == 0) return PM_OK;
  if (per_clip % 4 || ((uintptr_t)out & 15)) return PM_EALIGN;
  const int64_t total = B * per_clip;
  const int64_t nblk = (total / 4 + 255) / 256;
  if (nblk > 0x7fffffff) return PM_EINVAL;
  hipLaunchKernelGGL(logmel_finalize_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, out, (const int*)peak,
                     per_clip, total);
  PM_CHECK_LAUNCH();
  return PM_OK;
}
