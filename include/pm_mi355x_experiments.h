/* pm_mi355x_experiments.h - entry points of the EXPERIMENT kernels (pytorch-models_amd/csrc/experiments/): built, verified,
 * measured slower than what the product uses, kept for A/B runs.  They are NOT part of libpm_mi355x.so: `make experiments`
 * builds build/libpm_mi355x_exp.so = the product library + these (PM_MI355X_LIB=<that file> loads it).  Measurements:
 * DESIGN.md section 8.  Reference sites as in pm_mi355x.h (transformer.py:96-100 for the decode step). */
#ifndef PM_MI355X_EXPERIMENTS_H
#define PM_MI355X_EXPERIMENTS_H
#include "pm_mi355x.h"
#ifdef __cplusplus
extern "C" {
#endif

/* pm_dec_attention_fused with the whole K stream of a (sequence, head) requested up front by seven of the workgroup's
 * eight waves while the eighth loads and normalises the row (csrc/decode_persist.hip): same arguments, lk_max <= 2048. */
int pm_dec_attention_fused_v2(const float* x, int64_t d, const float* gamma, const float* beta, float eps, const void* w,
                              const float* bias, void* kc, void* vc, int64_t stride_b, int64_t stride_h, int64_t stride_k,
                              const int32_t* pos_ptr, int64_t lk_const, int64_t lk_max, float* out, int64_t B, int64_t H,
                              int self_attn, void* stream);

/* One layer of the persistent decode step (pm_dec_layers): device pointers into the decoder's weights (bf16 matrices,
 * row-major (out, in); f32 vectors) and its per-run buffers.  Layer algebra: transformer.py:96-100 (pre-norm).
 * w_q == NULL: no cross-attention block (decoder-only language models). */
typedef struct pm_dec_layer {
  const float* sa_g; const float* sa_b;      /* self-attention LayerNorm weight / bias (d) */
  const void* w_qkv; const float* b_qkv;     /* packed [Wq; Wk; Wv] (3 d, d), bias (3 d) or NULL */
  void* kc; void* vc;                        /* self K / V caches (B, H, Tmax, 64) bf16 */
  const void* w_so; const float* b_so;       /* self out_proj (d, d), bias or NULL */
  const float* ca_g; const float* ca_b;      /* cross-attention LayerNorm */
  const void* w_q; const float* b_q;         /* cross q_proj (d, d), bias or NULL */
  const void* cross_kv;                      /* packed projection of the memory (B, S, [k | v]) bf16 */
  const void* w_co; const float* b_co;       /* cross out_proj */
  const float* mlp_g; const float* mlp_b;    /* MLP LayerNorm */
  const void* w1; const float* b1;           /* linear1 (hid, d) */
  const void* w2; const float* b2;           /* linear2 (d, hid) */
  float sa_eps, ca_eps, mlp_eps, reserved_;
} pm_dec_layer_t;

/* ALL layers of one decode step in one persistent launch (one 512-thread workgroup per CU; pm_dec_layers_grid() of them):
 * the stage chain of pm_dec_attention_fused / pm_dec_linear / pm_dec_linear_ksplit with the activations handed from stage
 * to stage through agent-scope arrival counters instead of kernel boundaries (csrc/decode_persist.hip).  Reads x (B, d) f32
 * = the embedded current token, leaves x = the last layer's output (the final LayerNorm + vocabulary projection are the
 * caller's next launches); appends k, v of position *pos_ptr to the caches.  layers: n_layers structs in DEVICE memory.
 * att (B, d), h (B, ldh >= hid) f32 scratch; counters: n_layers * 24 int32, zero when *pos_ptr == 0 (they count up by
 * epochs of *pos_ptr + 1 and are never reset by the kernel); split_ws: (d / 16) * ceil(B / 16) * k_split * 256 floats,
 * split_cnt: (d / 16) * 4 int32 zero before the first launch; err: one int32, zero before a run - a hand-off that gives
 * up sets it (the launch still drains; later launches return at once) and the caller must treat the run as failed.
 * B <= 64, d % 64 == 0, d <= 1280, H * 64 == d, S, Tmax <= 2048, hid % 32 == 0, act in {NONE, GELU, GELU_TANH},
 * ceil(ceil(hid / 32 / k_split) / 4) <= (4 | 8 | 10 for d <= 512 | 1024 | 1280). */
int pm_dec_layers(const pm_dec_layer_t* layers, int64_t n_layers, int64_t B, int64_t d, int64_t H, int64_t S, int64_t Tmax,
                  int64_t hid, int act, int64_t k_split, const int32_t* pos_ptr, float* x, float* att, float* h, int64_t ldh,
                  int32_t* counters, float* split_ws, int32_t* split_cnt, int32_t* err, void* stream);
int pm_dec_layers_grid(void);

/* The phase-interleaved 256 x 256 x 64 K loop (csrc/experiments/linear_bf16_8ph.hip): y = x w^T, bf16, M % 256 == N % 256 ==
 * K % 64 == 0.  A micro-benchmark entry (tools/gemm8ph_bench.py), not a dispatcher target. */
int pm_gemm8ph_bench(const void* x, int64_t ldx, const void* w, int64_t ldw, void* y, int64_t ldy, int64_t M, int64_t N,
                     int64_t K, void* stream);

#ifdef __cplusplus
}
#endif
#endif
