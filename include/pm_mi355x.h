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
 * Requires K % 8 == 0, N % 4 == 0, ldy % 4 == 0, ldr % 4 == 0, 16-byte aligned x/w rows. */
int pm_linear_bf16(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias,
                   const void* resid, int64_t ldr, int resid_dtype, void* y, int64_t ldy, int y_dtype,
                   int64_t M, int64_t N, int64_t K, int act, void* stream);

/* nn.LayerNorm over the last dim (transformer.py:87,90,93; vit.py:69; whisper.py:27,45):
 * y[r,:] = (x[r,:] - mean) * rsqrt(var + eps) * gamma + beta, fp32 statistics, biased variance.
 * x: x_dtype (M, d) with row stride ldx; gamma/beta: f32 (d); y: y_dtype.  d % 8 == 0. */
int pm_layernorm(const void* x, int64_t ldx, int x_dtype, const float* gamma, const float* beta, float eps,
                 void* y, int64_t ldy, int y_dtype, int64_t M, int64_t d, void* stream);

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

/* ViT token assembly (vit.py:78-81): Conv2d(3, d, P, stride P) on fp32 NCHW images, flatten,
 * transpose, + pe, prepend cls - im2col-free.  imgs: f32 (N,3,Himg,Wimg); w: bf16 (d, 3*P*P)
 * (the Conv2d weight viewed 2-D); bias: f32 (d); pe: f32 (L, d); cls: f32 (d) or NULL;
 * out: bf16 (N, L + (cls != NULL), d) contiguous.  The cls row is broadcast over the batch
 * (SURVEY.md F1).  Requires P % 2 == 0... see pm_strerror for the exact supported set. */
int pm_vit_tokens(const float* imgs, const void* w, const float* bias, const float* pe, const float* cls,
                  void* out, int64_t N, int64_t Himg, int64_t Wimg, int64_t P, int64_t d, void* stream);

#ifdef __cplusplus
}
#endif
#endif
