/* pm_mi355x.h - C ABI of libpm_mi355x.so: the MI355X (gfx950) kernels underneath the
 * pytorch_models hot path (shared transformer forward, ViT patch-embed, Whisper front end,
 * encoder and KV-cached decoder).
 *
 * The reference (gau-nernst/pytorch-models) has no FFI: its boundary is its Python class API
 * (SURVEY.md section 8(b)).  Each entry point below replaces the stock-PyTorch primitive named in
 * its comment (reference file:line) and is what the reference-side binding in INTEGRATION.md
 * (a ctypes stub) would bind.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless marked host;
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream); no call synchronises,
 *    allocates or frees device memory; workspaces are caller-owned;
 *  - leading dimensions (ld*) are in ELEMENTS of the tensor's dtype;
 *  - return value: 0 = OK, otherwise a PM_E* code (pm_strerror gives text).  Arguments are
 *    validated on the host before any launch; a rejected call launches nothing;
 *  - dtypes: PM_BF16 / PM_F32 below.
 */
#ifndef PM_MI355X_H
#define PM_MI355X_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PM_ABI_VERSION 1

enum { PM_OK = 0, PM_EINVAL = 1, PM_EUNSUPPORTED = 2, PM_ELAUNCH = 3, PM_EALIGN = 4 };
enum { PM_BF16 = 0, PM_F32 = 1 };
/* activation table of MLP - pytorch_models/transformer.py:60-65 */
enum { PM_ACT_NONE = 0, PM_ACT_GELU = 1, PM_ACT_GELU_TANH = 2, PM_ACT_RELU = 3, PM_ACT_SILU = 4 };

int pm_abi_version(void);
const char* pm_strerror(int code);

/* nn.Linear (+ fused activation / residual): y[M,N] = act(x[M,K] w[N,K]^T + bias[N]) + resid[M,N]
 * Replaces the four nn.Linear of MHA (transformer.py:28-31,47-49,53), MLP's linear1 -> act ->
 * linear2 (transformer.py:59-66) and the residual adds of Encoder/DecoderLayer (transformer.py:98-100,
 * 124-125).  x, w: bf16, K-contiguous; bias: f32 or NULL; resid: resid_dtype or NULL; y: y_dtype.
 * Requires K % 64 == 0 and 16-byte aligned x / w rows (ld % 8 == 0).  N, ldy, ldr multiples of 4 take the
 * vector epilogue; ragged N (a 51865-row vocabulary: the tied-embedding logits of whisper.py:52) stores element-wise. */
int pm_linear_bf16(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias,
                   const void* resid, int64_t ldr, int resid_dtype, void* y, int64_t ldy, int y_dtype,
                   int64_t M, int64_t N, int64_t K, int act, void* stream);

/* pm_linear_bf16 with two extra addressing features (everything else identical):
 *  - x rows in two levels: row m lives at x + (m / x_rows_per_batch) * x_batch_stride + (m % x_rows_per_batch) * ldx
 *    (x_rows_per_batch == 0: plain m * ldx).  With ldx = 2*d and K = 3*d over a zero-padded time-major buffer this
 *    IS Conv1d(d, d, 3, stride 2, padding 1) (whisper.py:19): each output row reads 3 consecutive input rows;
 *  - resid rows repeat every resid_period rows (0: no repeat), e.g. "+ pos_embs[:L]" broadcast over the batch
 *    (whisper.py:31).  Order: y = act(x w^T + bias) + resid. */
int pm_linear_bf16_ex(const void* x, int64_t ldx, int64_t x_rows_per_batch, int64_t x_batch_stride, const void* w,
                      int64_t ldw, const float* bias, const void* resid, int64_t ldr, int resid_dtype,
                      int64_t resid_period, void* y, int64_t ldy, int y_dtype, int64_t M, int64_t N, int64_t K, int act,
                      void* stream);

/* nn.Linear in fp32, for modules whose parameters are fp32 (the reference's default dtype: tests/image/test_vit.py:45 asks
 * 2e-5, tests/audio2text/test_whisper.py:45 5e-5) and for the exact Whisper pipeline: y = act(x w^T + bias) + resid, all
 * operands f32, on the f32-input MFMA (exact fp32 products and accumulation; csrc/linear_f32.hip).  Addressing as
 * pm_linear_bf16_ex (two-level x rows: Conv1d / Conv2d windows as GEMM rows; periodic resid rows: a position table).
 * ldx, ldw multiples of 4, x and w 16-byte aligned; every activation of the table above. */
int pm_linear_f32(const float* x, int64_t ldx, int64_t x_rows_per_batch, int64_t x_batch_stride, const float* w, int64_t ldw,
                  const float* bias, const float* resid, int64_t ldr, int64_t resid_period, float* y, int64_t ldy, int64_t M,
                  int64_t N, int64_t K, int act, void* stream);

/* pm_linear_bf16_ln with a caller-owned workspace: pm_linear_ws_bytes() bytes, 16-byte aligned, its first 4096 bytes zero
 * before the first call and left alone afterwards.  With it the dispatcher may deal a GEMM's K steps out as ONE stream over
 * the persistent workgroups ("stream-K", csrc/linear_bf16_sk.hip) where whole 256 x 256 tiles would leave the last round
 * badly filled (M = 50432: N = 768 is 2.31 rounds, N = 3072 9.23); tiles shared by two workgroups are combined in fp32
 * through the workspace, deterministically.  One workspace serves one stream at a time.  ws == NULL: pm_linear_bf16_ln. */
int pm_linear_bf16_ws(const void* x, int64_t ldx, int64_t x_rows_per_batch, int64_t x_batch_stride, const void* w, int64_t ldw,
                      const float* bias, const void* resid, int64_t ldr, int resid_dtype, int64_t resid_period, void* y,
                      int64_t ldy, int y_dtype, int64_t M, int64_t N, int64_t K, int act, const float* ln_stats,
                      const float* ln_s, float* ln_row_out, void* ws, int64_t ws_bytes, void* stream);
int64_t pm_linear_ws_bytes(void);

/* LayerNorm folded into the GEMMs around it (pre-norm blocks, transformer.py:124-125: x + f(LN(x))).
 * pm_linear_bf16_ln = pm_linear_bf16_ex plus
 *  - ln_stats (M, 2) f32 [mean, rstd per input row] and ln_s (N) f32: y = act(rstd*(x w'^T - mean*ln_s) + bias) + resid,
 *    where the caller passes w' = bf16(gamma (.) w), ln_s[n] = sum_k w'[n][k], bias[n] = b[n] + sum_k beta[k] w[n][k];
 *  - ln_row_out (M, N/64, 2) f32: per row and 64-feature block, (sum, sum of squares) of the bf16-rounded outputs -
 *    the partial statistics of the NEXT LayerNorm, reduced by pm_ln_stats_finalize into (mean, rstd).
 * Served by the persistent kernels only: check pm_linear_ln_supported(M, N, K, act, produce) first (1 = yes). */
int pm_linear_bf16_ln(const void* x, int64_t ldx, int64_t x_rows_per_batch, int64_t x_batch_stride, const void* w,
                      int64_t ldw, const float* bias, const void* resid, int64_t ldr, int resid_dtype,
                      int64_t resid_period, void* y, int64_t ldy, int y_dtype, int64_t M, int64_t N, int64_t K, int act,
                      const float* ln_stats, const float* ln_s, float* ln_row_out, void* stream);
int pm_ln_stats_finalize(const float* row_partials, float* stats, int64_t M, int64_t N, float eps, void* stream);
int pm_linear_ln_supported(int64_t M, int64_t N, int64_t K, int act, int produce);

/* nn.LayerNorm over the last dim (transformer.py:87,90,93; vit.py:69; whisper.py:27,45):
 * y[r,:] = (x[r,:] - mean) * rsqrt(var + eps) * gamma + beta, fp32 statistics, biased variance.
 * x: x_dtype (M, d) with row stride ldx; gamma/beta: f32 (d); y: y_dtype.  d % 8 == 0. */
int pm_layernorm(const void* x, int64_t ldx, int x_dtype, const float* gamma, const float* beta, float eps,
                 void* y, int64_t ldy, int y_dtype, int64_t M, int64_t d, void* stream);

/* pm_layernorm with the two things the audio encoders put right behind a norm fused in:
 * y = act(LayerNorm(x)) + resid.  gamma and beta both NULL = no affine (LayerNorm1d(elementwise_affine=False),
 * audio/data2vec_audio.py:27); act: PM_ACT_NONE | PM_ACT_GELU (erff; audio/wav2vec2.py:38 norm -> GELU);
 * resid: NULL or resid_dtype (M, d) with row stride ldr. */
int pm_layernorm_ex(const void* x, int64_t ldx, int x_dtype, const float* gamma, const float* beta, float eps, int act,
                    const void* resid, int64_t ldr, int resid_dtype, void* y, int64_t ldy, int y_dtype, int64_t M,
                    int64_t d, void* stream);

/* T5's LayerNorm (text/t5.py:15-25): y = x * rsqrt(mean(x^2) + eps) * gamma - no centring, no bias; fp32 statistics. */
int pm_rmsnorm(const void* x, int64_t ldx, int x_dtype, const float* gamma, float eps, void* y, int64_t ldy, int y_dtype,
               int64_t M, int64_t d, void* stream);

/* The gate of GEGLU (text/t5.py:29-38) over a packed projection h = x [w; v]^T, bf16 (M, 2F) with row stride ldh:
 * out[m, f] = gelu_tanh(h[m, f]) * h[m, F + f], bf16 (M, F).  F % 8 == 0. */
int pm_geglu(const void* h, int64_t ldh, void* out, int64_t ldo, int64_t M, int64_t F, void* stream);

/* ---- Wav2Vec2 / Data2VecAudio / SEW (audio/wav2vec2.py, audio/data2vec_audio.py, audio/sew.py).
 * All feature-encoder activations are (clip, time, channel) bf16, so Conv1d(C, C', k, stride s) over them is
 * pm_linear_bf16_ex with row stride s*C and K = k*C (weight K order (tap, channel)).
 *
 * Layer 0 (audio/wav2vec2.py:32-38 with in_dim 1): out[b, t, c] = GELU(norm(bias[c] + sum_j w[c, j] x[b, t*stride + j])),
 * T0 = (L - k) / stride + 1.  x: f32 (B, L); w: f32 (C0, k) = the Conv1d weight as stored; bias: f32 (C0) or NULL;
 * out: bf16 (B, T0, C0).  norm: PM_W2V_NORM_NONE | PM_W2V_NORM_LAYER (LayerNorm1d over channels, gamma / beta f32 (C0)) |
 * PM_W2V_NORM_INSTANCE (InstanceNorm1d(affine) over time per clip and channel: the statistics come from the waveform's
 * 10 x 10 lag products, see csrc/wav2vec2.hip; the caller supplies scratch partials f32 (pm_w2v_stem0_scratch_floats(B, T0))
 * and stats f32 (B, C0, 2); fixed reduction order).  k == 10, C0 % 8 == 0, C0 <= 512. */
enum { PM_W2V_NORM_NONE = 0, PM_W2V_NORM_LAYER = 1, PM_W2V_NORM_INSTANCE = 2 };
int64_t pm_w2v_stem0_scratch_floats(int64_t B, int64_t T0);
int pm_w2v_stem0(const float* x, const float* w, const float* bias, int norm, const float* gamma, const float* beta, float eps,
                 float* partials, float* stats, void* out, int64_t B, int64_t L, int64_t C0, int64_t k, int64_t stride,
                 void* stream);

/* Regrouping for the grouped positional conv (audio/wav2vec2.py:70-74: ConstantPad1d + Conv1d(d, d, k, groups=G)):
 * out[b, g, pad_left + t, c] = bf16(x[b, t, g*cg + c]), zero in the pad rows and for c in [cg, cgp).  x: x_dtype
 * (B, T, >= G*cg) with row stride ldx; out: bf16 (B, G, pad_left + T + pad_right, cgp), cgp % 8 == 0.  Group g's conv
 * window at step t is then the k*cgp contiguous values at out[b, g, t*stride], i.e. one pm_linear_bf16_ex per group. */
int pm_group_windows(const void* x, int64_t ldx, int x_dtype, void* out, int64_t B, int64_t T, int64_t G, int64_t cg,
                     int64_t cgp, int64_t pad_left, int64_t pad_right, void* stream);

/* Grouped Conv1d over time with few channels per group, fused bias / activation / residual (the positional conv of
 * audio/wav2vec2.py:70-74, audio/data2vec_audio.py:25, audio/sew.py:24), on the buffer pm_group_windows builds:
 *   y[b*To + t, g*cg + n] = act(bias[g*cg + n] + sum_{tap < k, c < cg} xg[b, g, t*stride + tap, c] * w[g, n, tap*cgp + c])
 *                           + resid[b*To + t, g*cg + n],          To = (Tp - k) / stride + 1.
 * xg: bf16 (B, G, Tp, cgp); w: bf16 (G, cg, Kp), K order (tap, channel), zero-padded from k*cgp to Kp (% 64 == 0,
 * < 64 of padding); bias: f32 (G*cg) or NULL; resid: bf16 rows of stride ldr, or NULL; y: bf16 rows of stride ldy;
 * act: PM_ACT_NONE | PM_ACT_GELU.  The distinct input rows of a tile stay resident in LDS (see csrc/grouped_conv.hip).
 * pm_grouped_conv_supported(cg, cgp) == 1 for cg % 4 == 0 and cgp in {8, 16, 32, 48, 64} with cgp - cg < 8. */
int pm_grouped_conv_supported(int64_t cg, int64_t cgp);
int pm_grouped_conv_bf16(const void* xg, const void* w, const float* bias, const void* resid, int64_t ldr, void* y,
                         int64_t ldy, int64_t B, int64_t G, int64_t Tp, int64_t cg, int64_t cgp, int64_t k, int64_t stride,
                         int64_t Kp, int act, void* stream);

/* F.avg_pool1d(x, 2) over time on (B, T, d) bf16 -> (B, T / 2, d) bf16 (audio/sew.py:33; a trailing odd row is dropped). */
int pm_avgpool_time2(const void* x, void* out, int64_t B, int64_t T, int64_t d, void* stream);

/* F.scaled_dot_product_attention (transformer.py:52) for head_dim 64, bf16 operands:
 * o[b,i,h,:] = softmax_j(q[b,i,h,:] . k[b,j,h,:] / 8 [j <= i if causal]) v[b,j,h,:].
 * q/k/v/o are addressed as base + b*stride_b + token*stride_t + h*64 (elements), so the packed
 * (tokens, 3*h*64) output of a fused QKV projection is consumed in place and the result is written
 * already merged as (tokens, h*64) - the unflatten/transpose/flatten of transformer.py:47-53 never
 * materialise.  causal is top-left aligned like torch's is_causal. */
int pm_attention_bf16(const void* q, int64_t q_stride_b, int64_t q_stride_t,
                      const void* k, int64_t k_stride_b, int64_t k_stride_t,
                      const void* v, int64_t v_stride_b, int64_t v_stride_t,
                      void* o, int64_t o_stride_b, int64_t o_stride_t,
                      int64_t B, int64_t H, int64_t Lq, int64_t Lk, int causal, void* stream);

/* pm_attention_bf16 with the additive attn_mask of transformer.py:52 (`attn_bias`: T5 / MaxViT style relative position
 * bias): scores + bias[b, h, i, j] before the softmax.  bias: f32, addressed bias + b*stride_b + h*stride_h + i*stride_q + j
 * (stride 0 = broadcast over batch / heads); -inf entries mask keys (a fully masked row yields NaN, like torch). */
int pm_attention_bias_bf16(const void* q, int64_t q_stride_b, int64_t q_stride_t,
                           const void* k, int64_t k_stride_b, int64_t k_stride_t,
                           const void* v, int64_t v_stride_b, int64_t v_stride_t,
                           void* o, int64_t o_stride_b, int64_t o_stride_t,
                           int64_t B, int64_t H, int64_t Lq, int64_t Lk, int causal, const float* bias,
                           int64_t bias_stride_b, int64_t bias_stride_h, int64_t bias_stride_q, void* stream);

/* Small 2-D convolution on NHWC bf16 (reference: conv_norm_act / MBConv / MobileViTBlock, pytorch_models/image/mobile_vit.py:10-69:
 * nn.Conv2d(bias=False) + eval-mode nn.BatchNorm2d [+ nn.SiLU], the norm folded into w and bias by the caller).
 * x: bf16 (N, H, W, Cin); w: bf16 (Cout, kh, kw, Cin / groups); bias: f32 (Cout) or NULL; resid: bf16 like y or NULL (added
 * after the activation); y: bf16 (N, Ho, Wo, Cout), Ho = (H + 2 pad - kh) / stride + 1.  groups divides Cin and Cout (1 = dense,
 * Cin = depthwise).  act: PM_ACT_NONE | PM_ACT_SILU | PM_ACT_RELU.  Direct convolution, fp32 accumulation; off the benchmark path. */
int pm_conv2d_nhwc_bf16(const void* x, int64_t N, int64_t H, int64_t W, int64_t Cin, const void* w, const float* bias,
                        const void* resid, void* y, int64_t Cout, int64_t kh, int64_t kw, int64_t stride, int64_t pad,
                        int64_t groups, int act, void* stream);

/* y[n, c] = mean over r of x[n, r, c]: nn.AdaptiveAvgPool2d(1) + Flatten on NHWC rows (mobile_vit.py:100).  bf16 in / out. */
int pm_mean_rows_bf16(const void* x, void* y, int64_t N, int64_t HW, int64_t C, void* stream);

/* Attention for head dims other than 64 (% 4 == 0 up to 128, % 2 == 0 up to 64; e.g. ViT-H's 80, MobileViT's 16 .. 60): same addressing with h*head_dim head
 * offsets, optional additive bias (NULL = none), Lk <= 2048.  fp32 VALU, correctness-first, off the benchmark path. */
int pm_attention_generic_bf16(const void* q, int64_t q_stride_b, int64_t q_stride_t,
                              const void* k, int64_t k_stride_b, int64_t k_stride_t,
                              const void* v, int64_t v_stride_b, int64_t v_stride_t,
                              void* o, int64_t o_stride_b, int64_t o_stride_t,
                              int64_t B, int64_t H, int64_t Lq, int64_t Lk, int64_t head_dim, int causal,
                              const float* bias, int64_t bias_stride_b, int64_t bias_stride_h, int64_t bias_stride_q,
                              void* stream);

/* pm_attention_generic_bf16 on fp32 operands (strides in elements, multiples of 4; o 16-byte aligned): the attention of
 * fp32 modules and of the exact Whisper pipeline - fp32 in, fp32 arithmetic, fp32 out. */
int pm_attention_generic_f32(const float* q, int64_t q_stride_b, int64_t q_stride_t, const float* k, int64_t k_stride_b,
                             int64_t k_stride_t, const float* v, int64_t v_stride_b, int64_t v_stride_t, float* o,
                             int64_t o_stride_b, int64_t o_stride_t, int64_t B, int64_t H, int64_t Lq, int64_t Lk,
                             int64_t head_dim, int causal, const float* bias, int64_t bias_stride_b, int64_t bias_stride_h,
                             int64_t bias_stride_q, void* stream);

/* ViT token assembly (vit.py:78-81): Conv2d(3, d, P, stride P) on fp32 NCHW images, flatten,
 * transpose, + pe, prepend cls - im2col-free.  imgs: f32 (N,3,Himg,Wimg); w: bf16 (d, 3*P*P)
 * (the Conv2d weight viewed 2-D); bias: f32 (d); pe: f32 (L, d); cls: f32 (d) or NULL;
 * out: bf16 (N, L + (cls != NULL), d) contiguous.  The cls row is broadcast over the batch
 * (SURVEY.md F1).  Requires P % 2 == 0... see pm_strerror for the exact supported set. */
int pm_vit_tokens(const float* imgs, const void* w, const float* bias, const float* pe, const float* cls,
                  void* out, int64_t N, int64_t Himg, int64_t Wimg, int64_t P, int64_t d, void* stream);

/* pm_vit_tokens for any patch size P <= 64 (14: DINOv2; 32; 8): w is bf16 (d, ldw) with the 3*P*P real columns
 * followed by zeros up to ldw >= ceil(3*P*P / 64) * 64.  Scalar gather; not on the benchmark path. */
int pm_vit_tokens_generic(const float* imgs, const void* w, int64_t ldw, const float* bias, const float* pe,
                          const float* cls, void* out, int64_t N, int64_t Himg, int64_t Wimg, int64_t P, int64_t d,
                          void* stream);

/* torch.stft(n_fft, hop, hann window, center=True, reflect, onesided).abs().square() (spectrogram.py:16),
 * optionally followed by the mel filterbank (spectrogram.py:45) and Whisper's log10 (whisper.py:144-145).
 * x: f32, clip b at x + b * x_stride, T samples.  tw_cos / tw_sin: f32 twiddle tables with the window folded in,
 * laid out [ceil(nbins / 32)][n_fft / 2][64]: entry (blk, s, lane) = w[k] * cos|sin(2 pi k bin / n_fft) with
 * k = 2 s + (lane >> 5), bin = 32 blk + (lane & 31) (0 for bin >= nbins).  n_frames <= 1 + T / hop frames are
 * produced (Whisper drops the last one).  mode 0: out (B, n_fft/2+1, n_frames) power; mode 1: out
 * (B, n_mels, n_frames) mel power, the filterbank given in CSR form (mel_ptr[n_mels+1], mel_col, mel_val);
 * mode 2: log10 of mode 1 and peak[b] = order-preserving int encoding of the per-clip maximum (for
 * pm_logmel_finalize).  n_fft, hop even; T > n_fft / 2. */
int pm_stft_mel(const float* x, int64_t x_stride, int64_t B, int64_t T, const float* tw_cos, const float* tw_sin,
                int64_t n_fft, int64_t hop, int64_t n_frames, int mode, const int32_t* mel_ptr, const int32_t* mel_col,
                const float* mel_val, int64_t n_mels, float* out, int32_t* peak, void* stream);

/* pm_stft_mel for a SYMMETRIC window (w[k] == w[n_fft - k], k = 1 .. n_fft - 1: every Hann / Hamming ...): the frame is folded
 * about n_fft / 2 - cosine terms on x[k] + x[n_fft - k], sine terms on x[k] - x[n_fft - k] - which halves the contraction.
 * Tables laid out [ceil(nbins / 32)][S][64] with S = (n_fft / 2 + 2) / 2 and k = 2 s + (lane >> 5): cosine entries for
 * k = 0 .. n_fft / 2 (the k = n_fft / 2 entry HALVED, it meets its own mirror), sine entries for k = 1 .. n_fft / 2 - 1, zero
 * elsewhere.  Everything else as pm_stft_mel. */
int pm_stft_mel_folded(const float* x, int64_t x_stride, int64_t B, int64_t T, const float* tw_cos, const float* tw_sin,
                       int64_t n_fft, int64_t hop, int64_t n_frames, int mode, const int32_t* mel_ptr, const int32_t* mel_col,
                       const float* mel_val, int64_t n_mels, float* out, int32_t* peak, void* stream);

/* whisper.py:146-147 in place: out = (max(out, per-clip max - 8) + 4) / 4; per_clip = n_mels * n_frames (% 4 == 0). */
int pm_logmel_finalize(float* out, const int32_t* peak, int64_t B, int64_t per_clip, void* stream);

/* First stem conv of WhisperEncoder (whisper.py:17-18): Conv1d(C, d, 3, 1, 1) + GELU on channel-major f32
 * x (B, C, T), written time-major bf16 as out (B, T + 2, d) with zero rows 0 and T + 1 (the padding the second,
 * stride-2 conv needs: see pm_linear_bf16_ex).  w: bf16 (d, 3 * Cpad), K order (tap, channel), channels
 * zero-padded to Cpad (% 64 == 0); bias f32 (d). */
int pm_whisper_stem1(const float* x, const void* w, const float* bias, void* out, int64_t B, int64_t C, int64_t Cpad,
                     int64_t T, int64_t d, void* stream);

/* nn.Embedding + positional add (whisper.py:48-49): out[b, l, :] = emb[tokens[b, l], :] + pos[pos0 + l, :].
 * tokens: int64 (B, L); emb: bf16 (V, d); pos: f32 (>= pos0 + L, d) or NULL (no absolute positions: text/t5.py:145);
 * out: bf16 | f32 (B, L, d). */
int pm_embed_tokens(const int64_t* tokens, const void* emb, const float* pos, void* out, int out_dtype, int64_t B,
                    int64_t L, int64_t pos0, int64_t d, int64_t V, void* stream);
/* the same from an fp32 table into fp32 rows (modules whose parameters are fp32) */
int pm_embed_tokens_f32(const int64_t* tokens, const float* emb, const float* pos, float* out, int64_t B, int64_t L,
                        int64_t pos0, int64_t d, int64_t V, void* stream);

/* ---- KV-cached greedy decode step (no reference counterpart: README.md:86 lists Whisper decoding as TODO; the
 * semantics follow pytorch_models/text/generator.py:23-35 over transformer.py:96-100 and whisper.py:47-53).
 * All activations f32, weights / caches bf16.  `pos_ptr` points at ONE device int: the position t of the token being
 * consumed; every kernel of a step reads it, pm_dec_advance increments it, so one captured graph replays all steps. */

/* x[b, :] = emb[tok_cur[b], :] + pos[t, :]  (whisper.py:48-49 for one position). */
int pm_dec_embed(const int64_t* tok_cur, const void* emb, const float* pos, const int32_t* pos_ptr, float* x, int64_t B,
                 int64_t d, int64_t V, void* stream);

/* y = act([LayerNorm_{gamma,beta,eps}](x) w^T + bias) [+ resid] for M <= 64 rows of f32 x; w bf16 (N, K).
 * gamma == NULL: no LayerNorm.  mode 0: out f32 (M, N) (+ resid f32, may alias out);
 * mode 1 (N = 3*inner, [q|k|v] blocks): q -> out f32 (M, inner); k, v -> bf16 caches (M, H, Tmax, 64) at position t;
 * mode 2: no store - per row, the tile's (max logit, lowest index) go to ws_val / ws_idx (M, ceil(N / tile)),
 *         tile = pm_dec_argmax_tile(K) features.  With LayerNorm: K <= 1280.
 * act: PM_ACT_NONE | PM_ACT_GELU (erff) | PM_ACT_GELU_TANH (tanhf).  K % 32 == 0. */
int pm_dec_linear(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, const void* w,
                  int64_t ldw, const float* bias, const float* resid, int64_t ldr, float* out, int64_t ldo, int64_t M,
                  int64_t N, int64_t K, int act, int mode, void* kcache, void* vcache, int64_t inner, int64_t H,
                  int64_t Tmax, const int32_t* pos_ptr, float* ws_val, int32_t* ws_idx, void* stream);

/* Features per argmax tile of pm_dec_linear mode 2 for a given K (64, or 32 when K > 512). */
int pm_dec_argmax_tile(int64_t K);

/* One query per (sequence, head), head_dim 64: out = softmax(q K^T / 8) V (transformer.py:52 with L_q = 1 and NO
 * causal flag - SURVEY.md F3).  q / out f32 (B, H*64); K/V bf16 addressed base + b*stride_b + h*stride_h + key*stride_k;
 * number of keys = (lk_ptr ? *lk_ptr : 0) + lk_add  (self: pos + 1; cross: the constant 1500), <= lk_max <= 4096. */
int pm_dec_attention(const float* q, const void* kc, const void* vc, int64_t stride_b, int64_t stride_h, int64_t stride_k,
                     const int32_t* lk_ptr, int64_t lk_add, int64_t lk_max, float* out, int64_t B, int64_t H, void* stream);

/* pm_dec_linear, plain mode without LayerNorm, with K split over k_split (2..8) workgroups per 16-feature tile: each
 * part reads 1 / k_split of x (at M = 32, K = 2048 every workgroup of the unsplit kernel pulls all 256 KB of x through
 * one CU's L2 path) and of its weight rows; the last part to finish - an agent-scope ticket, no spinning - adds the
 * parts in part order (deterministic) and applies bias / activation / residual.  split_ws: ceil(N / 16) * k_split *
 * mt * 256 floats with mt = ceil(M / 16) rounded up to 1, 2 or 4, 16-byte aligned; split_cnt: ceil(N / 16) * 4 int32 (one ticket per feature tile and
 * row tile), zero before the first launch, zero again after every launch (graph replay).  K / 32 >= k_split. */
int pm_dec_linear_ksplit(const float* x, int64_t ldx, const void* w, int64_t ldw, const float* bias, const float* resid,
                         int64_t ldr, float* out, int64_t ldo, int64_t M, int64_t N, int64_t K, int act, int64_t k_split,
                         float* split_ws, int32_t* split_cnt, void* stream);

/* Fused attention block of the decode step, one launch: LayerNorm(x[b]) -> per-head projection -> attention.
 * self_attn != 0: w = packed [q|k|v] bf16 (3*H*64, d), bias f32 (3*H*64) or NULL; k_h, v_h are rounded to bf16,
 *   appended to the caches kc / vc at position t = *pos_ptr and attended together with the t older keys;
 * self_attn == 0: w = q projection bf16 (H*64, d); kc / vc = the projected cross K / V, lk_const keys.
 * K/V addressing and q/out layout as pm_dec_attention.  d % 64 == 0, d <= 1280.  Replaces pm_dec_linear (mode 0/1)
 * + pm_dec_attention for the same result (same fp32 arithmetic, k/v rounding point unchanged). */
int pm_dec_attention_fused(const float* x, int64_t d, const float* gamma, const float* beta, float eps, const void* w,
                           const float* bias, void* kc, void* vc, int64_t stride_b, int64_t stride_h, int64_t stride_k,
                           const int32_t* pos_ptr, int64_t lk_const, int64_t lk_max, float* out, int64_t B, int64_t H,
                           int self_attn, void* stream);

/* pm_dec_attention_fused with FLOAT K / V caches (kc, vc point to f32, strides in elements): nothing is rounded when a key
 * or value is cached - the reference-accuracy decode of Whisper.generate(exact=True), captured in the same step graph as the
 * bf16-cache path (text/generator.py:23-35 semantics, fp32 throughout).  Same arguments otherwise. */
int pm_dec_attention_fused_kv32(const float* x, int64_t d, const float* gamma, const float* beta, float eps, const void* w,
                                const float* bias, void* kc, void* vc, int64_t stride_b, int64_t stride_h, int64_t stride_k,
                                const int32_t* pos_ptr, int64_t lk_const, int64_t lk_max, float* out, int64_t B, int64_t H,
                                int self_attn, void* stream);

/* pm_dec_attention_fused (kv_f32 = 0) / _kv32 (kv_f32 = 1) as a link of the step's chain of DEFERRED SUMS: two launches per
 * layer and the K-split's ticket leave the chain (transformer.py:50-53 and the residual adds of transformer.py:96-100, same sums).
 *   IN  - the block's input row is  x[b] + x_bias + sum_{p < n_parts} x_parts[p * part_stride + b * part_row_stride + :]  (added in
 *         part order; every (b, h) workgroup forms it itself), and workgroup (b, 0) writes it to x_out (must not alias x) as the
 *         residual stream for what follows.  n_parts = 0: plain x, x_out untouched.  Producers: pm_dec_linear_kparts (fc2 of the
 *         previous layer: part_stride = its part_stride, part_row_stride = ld_parts) and this function's own OUT side
 *         (part_stride = d, part_row_stride = H * d, x_bias = the output projection's bias).
 *   OUT - w_out != NULL (bf16 (d, H*64) row-major): instead of att the block writes the output projection's per-head partial sums
 *         head_parts[(b * H + h) * d + n] = sum_j att[b, h*64 + j] * w_out[n, h*64 + j]  (fp32 FMA chain); `out` is not written.
 *         w_out == NULL: att to `out` as pm_dec_attention_fused.
 * Other arguments as pm_dec_attention_fused. */
int pm_dec_attention_chain(const float* x, int64_t d, const float* gamma, const float* beta, float eps, const void* w,
                           const float* bias, void* kc, void* vc, int64_t stride_b, int64_t stride_h, int64_t stride_k,
                           const int32_t* pos_ptr, int64_t lk_const, int64_t lk_max, int64_t B, int64_t H, int self_attn,
                           int kv_f32, const float* x_parts, int64_t n_parts, int64_t part_stride, int64_t part_row_stride,
                           const float* x_bias, float* x_out, const void* w_out, float* head_parts, float* out, void* stream);

/* pm_dec_linear_ksplit's products WITHOUT the combining pass: part p (K steps [p, p + 1) * K / k_split) of x w^T goes to
 * parts + p * part_stride, row stride ld_parts, as it stands - no bias, no residual, no ticket; the consumer
 * (pm_dec_attention_chain) adds the parts in order.  part_stride >= M * ld_parts, both % 4 == 0, parts 16-byte aligned. */
int pm_dec_linear_kparts(const float* x, int64_t ldx, const void* w, int64_t ldw, float* parts, int64_t ld_parts,
                         int64_t part_stride, int64_t M, int64_t N, int64_t K, int64_t k_split, void* stream);

/* Finish the argmax over the n_tiles tile winners of pm_dec_linear mode 2 (lowest index on ties, like torch.argmax),
 * teacher-force the prompt (next = prompt[b, t+1] while t + 1 < P), write tok_cur[b] and tokens_out[b, t+1]. */
int pm_dec_argmax_reduce(const float* ws_val, const int32_t* ws_idx, int64_t n_tiles, const int32_t* pos_ptr,
                         const int64_t* prompt, int64_t P, int64_t* tok_cur, int64_t* tokens_out, int64_t Ttot,
                         float* margin_out, int64_t B, void* stream);

/* ++*pos_ptr, as its own launch (every workgroup of the step has read t by then). */
int pm_dec_advance(int32_t* pos_ptr, void* stream);

/* Whisper's decoding-time logit filters on the device, between the vocabulary projection (pm_dec_linear mode 0 into logits) and
 * the token choice (pm_dec_sample_topk; k = 1 = arg-max).  The reference has no Whisper decoding (README.md:86-87: TODO); the rules
 * are those of OpenAI's whisper decoding.py (SuppressTokens, SuppressBlank, ApplyTimestampRules) as published - that package is
 * not available here, so the oracle for this entry point (oracle/ref_whisper_rules.py) is "parity unpinned".
 * logits: f32 (B, V), modified in place (-inf); tokens: int64 (B, Ttot) = prompt + generated; n = *pos_ptr + 1 is the index being
 * chosen (nothing happens while n < P); eot < timestamp_begin < V; no_timestamps < 0 = none; max_initial_timestamp < 0 = no cap
 * (else the first timestamp is at most timestamp_begin + max_initial_timestamp); suppress / blank: int32 id lists (blank applies
 * to the first generated token only).  Rules: listed ids; a transcript starts with a timestamp; timestamps come in pairs and
 * never decrease; if the timestamps' total probability exceeds the best text token's, text is masked. */
int pm_dec_whisper_rules(float* logits, int64_t ldl, int64_t V, const int64_t* tokens, int64_t Ttot, const int32_t* pos_ptr,
                         int64_t P, int64_t eot, int64_t no_timestamps, int64_t timestamp_begin, int64_t max_initial_timestamp,
                         const int32_t* suppress, int64_t n_suppress, const int32_t* blank, int64_t n_blank, int64_t B,
                         void* stream);

/* Top-k sampling in place of the arg-max (text/generator.py:30-32: topk, softmax over the k logits, one multinomial draw):
 * logits (B, V) f32, row stride ldl, of the last position (pm_dec_linear mode 0 with the final LayerNorm); per sequence the
 * k (1..64) largest (ties: lowest index first), softmax over them, one draw from a counter-based generator keyed by
 * (seed, position, sequence) - the same seed gives the same ids; then the tail of pm_dec_next_token (prompt forcing,
 * x[b] = emb[token] + pos[t + 1], ticketed advance of *pos_ptr).  k = 1 is the arg-max. */
int pm_dec_sample_topk(const float* logits, int64_t ldl, int64_t V, int64_t k, uint64_t seed, int32_t* pos_ptr,
                       const int64_t* prompt, int64_t P, int64_t* tok_cur, int64_t* tokens_out, int64_t Ttot, const void* emb,
                       const float* pos, float* x, int64_t d, int32_t* ticket, int64_t B, void* stream);

/* pm_dec_argmax_reduce + the NEXT step's pm_dec_embed + pm_dec_advance in one launch (two launches fewer per decode
 * step): sequence b's workgroup picks its token as pm_dec_argmax_reduce does, writes x[b] = emb[token] + pos[t + 1], and
 * the last workgroup to finish - an agent-scope ticket in *ticket (int32, zero before the first launch, zero again after
 * every launch) - stores t + 1 to *pos_ptr.  Before a run's first step: *pos_ptr = 0, tok_cur set, pm_dec_embed once. */
int pm_dec_next_token(const float* ws_val, const int32_t* ws_idx, int64_t n_tiles, int32_t* pos_ptr, const int64_t* prompt,
                      int64_t P, int64_t* tok_cur, int64_t* tokens_out, int64_t Ttot, float* margin_out, const void* emb,
                      const float* pos, float* x, int64_t d, int64_t V, int32_t* ticket, int64_t B, void* stream);

#ifdef __cplusplus
}
#endif
#endif
