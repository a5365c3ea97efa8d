"""GPU parity of each C-ABI kernel against the oracle's primitive on the same seeded inputs.

bf16 tolerance (stated): operands are bf16-rounded BEFORE both paths see them, the oracle computes in
fp32, the kernel accumulates in fp32 and rounds its output to bf16 once (2^-9 relative).  We therefore
require |got - want| <= 1e-2 * |want| + 1e-2 * rms(want) for bf16 outputs, and 1e-4-level agreement for
fp32 outputs (accumulation-order differences only).
"""
import math
import os

import pytest
import torch

from oracle import ref_transformer as RT
from oracle import ref_vit as RV
from synthweights import synth_input

pytestmark = pytest.mark.gpu


def _hip_has_experiments() -> bool:
    from pytorch_models import _hip
    return os.path.exists(_hip.LIB_PATH) and _hip.has_experiments()
torch.set_grad_enabled(False)


def bf(x):
    return x.to(torch.bfloat16)


def close_bf16(got, want, rel=1e-2):
    got, want = got.float().cpu(), want.float()
    rms = want.square().mean().sqrt().item()
    torch.testing.assert_close(got, want, rtol=rel, atol=rel * max(rms, 1e-6))


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from pytorch_models._hip import ops as o

    return o


# ---------------------------------------------------------------- linear
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (197, 768, 768), (1, 64, 64), (333, 2304, 768), (50, 192, 3072), (257, 132, 128),
                                   (40000, 388, 64), (33000, 1024, 192)])  # the last two take the 256x128 ring kernel
@pytest.mark.parametrize("act", ["none", "gelu"])
def test_linear_shapes(ops, M, N, K, act):
    x = bf(synth_input("lin_x", (M, K), 1))
    w = bf(synth_input("lin_w", (N, K), 2, scale=1 / math.sqrt(K)))
    b = synth_input("lin_b", (N,), 3, scale=0.1)
    want = x.float() @ w.float().T + b
    want = want if act == "none" else RT.activation(want, act)
    got = ops.linear(x.cuda(), w.cuda(), b.cuda(), act=act)
    assert got.dtype == torch.bfloat16 and got.shape == (M, N)
    close_bf16(got, want)


@pytest.mark.parametrize("act", ["approximate_gelu", "relu", "silu"])
def test_linear_activations(ops, act):
    x = bf(synth_input("lin_x", (70, 128), 1))
    w = bf(synth_input("lin_w", (96, 128), 2, scale=0.1))
    want = RT.activation(x.float() @ w.float().T, act)
    close_bf16(ops.linear(x.cuda(), w.cuda(), None, act=act), want)


@pytest.mark.parametrize("rdt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("odt", [torch.bfloat16, torch.float32])
def test_linear_residual_and_dtypes(ops, rdt, odt):
    M, N, K = 200, 256, 192
    x = bf(synth_input("lin_x", (M, K), 4))
    w = bf(synth_input("lin_w", (N, K), 5, scale=0.07))
    b = synth_input("lin_b", (N,), 6, scale=0.1)
    r = synth_input("lin_r", (M, N), 7).to(rdt)
    want = x.float() @ w.float().T + b + r.float()
    got = ops.linear(x.cuda(), w.cuda(), b.cuda(), resid=r.cuda(), out_dtype=odt)
    assert got.dtype == odt
    if odt == torch.float32:
        torch.testing.assert_close(got.cpu(), want, rtol=1e-4, atol=1e-4)
    else:
        close_bf16(got, want)


def test_linear_strided_views_and_inplace_residual(ops):
    """x as a column slice of a wider buffer (ldx > K); residual aliased with the output."""
    M, N, K = 130, 128, 64
    wide = bf(synth_input("lin_wide", (M, 3 * K), 8)).cuda()
    x = wide[:, K : 2 * K]
    w = bf(synth_input("lin_w", (N, K), 9, scale=0.1)).cuda()
    acc = bf(synth_input("lin_acc", (M, N), 10)).cuda()
    want = x.float().cpu() @ w.float().cpu().T + acc.float().cpu()
    ops.linear(x, w, None, resid=acc, out=acc)
    close_bf16(acc, want)


@pytest.mark.parametrize("act,with_resid", [("none", False), ("gelu", False), ("none", True)])
def test_linear_wide_256x256_kernel(ops, act, with_resid):
    """>= 1024 tiles of 256 x 256 take the wide persistent kernel: ragged M and N, K = 4 steps of 32, every epilogue."""
    M, N, K = 70000, 1032, 128
    x = bf(synth_input("lw_x", (M, K), 60))
    w = bf(synth_input("lw_w", (N, K), 61, scale=0.1))
    b = synth_input("lw_b", (N,), 62, scale=0.1)
    r = bf(synth_input("lw_r", (M, N), 63)) if with_resid else None
    got = ops.linear(x.cuda(), w.cuda(), b.cuda(), act=act, resid=r.cuda() if with_resid else None)
    idx = torch.cat([torch.arange(0, 300), torch.arange(M - 300, M), torch.randint(0, M, (400,), generator=torch.Generator().manual_seed(1))])
    want = x[idx].float() @ w.float().T + b
    want = want if act == "none" else RT.activation(want, act)
    if with_resid:
        want = want + r[idx].float()
    close_bf16(got[idx.cuda()], want)
    assert torch.isfinite(got.float()).all()


@pytest.mark.parametrize("M,N,K", [(12, 64, 32), (70, 96, 72), (300, 132, 200), (5, 8, 8)])
def test_linear_k_tail(ops, M, N, K):
    """K not a multiple of 64 (e.g. an out_proj over n_heads * head_dim = 32): the tail of the last K tile is zero-fed."""
    x = bf(synth_input("lk_x", (M, K), 64))
    w = bf(synth_input("lk_w", (N, K), 65, scale=K ** -0.5))
    b = synth_input("lk_b", (N,), 66, scale=0.1)
    close_bf16(ops.linear(x.cuda(), w.cuda(), b.cuda()), x.float() @ w.float().T + b)


def test_linear_rejects_unsupported(ops):
    x = torch.zeros(4, 68, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(8, 68, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError, match="pm_mi355x error 2"):
        ops.linear(x, w)
    with pytest.raises(RuntimeError, match="HIP devices only"):
        ops.linear(x.cpu(), w.cpu())


def test_linear_linearity_at_full_size(ops):
    """Size-independent property at the BASELINE C2 shape (M = 256*197): f(x + x) == 2 f(x) bit-exactly (no bias)."""
    M, N, K = 50432, 768, 768
    x1 = bf(synth_input("big_x1", (M, K), 11)).cuda()
    w = bf(synth_input("big_w", (N, K), 13, scale=1 / math.sqrt(K))).cuda()
    # x2 = x1, so the sum 2 * x1 is exactly representable in bf16
    y1 = ops.linear(x1, w, None, out_dtype=torch.float32)
    y2 = ops.linear((x1.float() * 2).to(torch.bfloat16), w, None, out_dtype=torch.float32)
    torch.testing.assert_close(y2, 2 * y1, rtol=0, atol=0)  # power-of-two scaling is bit-exact in fp32
    # and a spot check of 64 random rows against the oracle
    idx = torch.randint(0, M, (64,), generator=torch.Generator().manual_seed(0))
    want = x1[idx.cuda()].float().cpu() @ w.float().cpu().T
    torch.testing.assert_close(y1[idx.cuda()].cpu(), want, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("M,N,K,act", [(33000, 1024, 192, "none"), (33000, 1024, 192, "gelu"), (70000, 1032, 128, "gelu")])
def test_linear_layernorm_fold_consumer(ops, M, N, K, act):
    """pm_linear_bf16_ln as the consumer: LN(x) @ w.T + b computed as rstd * (x w'^T - mean * s) + c on the raw rows
    (256x128 ring kernel and 256x256 wide kernel); the oracle normalises first, in fp32."""
    assert ops.linear_ln_supported(M, N, K, act, False)
    x = bf(synth_input("lf_x", (M, K), 70) * 2 + 0.5)
    w = bf(synth_input("lf_w", (N, K), 71, scale=K ** -0.5))
    b = synth_input("lf_b", (N,), 72, scale=0.1)
    g = synth_input("lf_g", (K,), 73, scale=0.2) + 1
    beta = synth_input("lf_beta", (K,), 74, scale=0.2)
    eps = 1e-6
    wl = bf(w.float() * g[None, :])
    s, c = wl.float().sum(1), w.float() @ beta + b
    xf = x.float()
    mean, var = xf.mean(1), xf.var(1, unbiased=False)
    stats = torch.stack([mean, (var + eps).rsqrt()], 1).contiguous()
    got = ops.linear(x.cuda(), wl.cuda(), c.cuda(), act=act, ln_stats=stats.cuda(), ln_s=s.cuda())
    idx = torch.cat([torch.arange(0, 300), torch.arange(M - 300, M), torch.randint(0, M, (400,), generator=torch.Generator().manual_seed(2))])
    want = RT.layernorm({"weight": g, "bias": beta}, "", xf[idx], eps) @ w.float().T + b
    want = want if act == "none" else RT.activation(want, act)
    close_bf16(got[idx.cuda()], want, rel=1.5e-2)  # one extra bf16 rounding (gamma (.) w) relative to the unfused path
    assert torch.isfinite(got.float()).all()


def test_linear_layernorm_fold_producer_row_statistics(ops):
    """The residual GEMM's epilogue emits per-row (sum, sum of squares) of its bf16 outputs per 64-feature block;
    pm_ln_stats_finalize reduces them to (mean, rstd).  The output itself must equal the plain kernel's bit for bit."""
    M, N, K = 33000, 1024, 192
    assert ops.linear_ln_supported(M, N, K, "none", True)
    x = bf(synth_input("lp_x", (M, K), 75)).cuda()
    w = bf(synth_input("lp_w", (N, K), 76, scale=K ** -0.5)).cuda()
    b = synth_input("lp_b", (N,), 77, scale=0.1).cuda()
    r = bf(synth_input("lp_r", (M, N), 78) + 0.3).cuda()
    y, rows = ops.linear(x, w, b, resid=r, want_row_stats=True)
    torch.testing.assert_close(y, ops.linear(x, w, b, resid=r), rtol=0, atol=0)
    yb = y.float().view(M, N // 64, 64)
    torch.testing.assert_close(rows[..., 0], yb.sum(-1), rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(rows[..., 1], yb.square().sum(-1), rtol=1e-5, atol=1e-4)
    stats = ops.ln_stats_finalize(rows, N, 1e-5)
    yf = y.float()
    torch.testing.assert_close(stats[:, 0], yf.mean(1), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(stats[:, 1], (yf.var(1, unbiased=False) + 1e-5).rsqrt(), rtol=1e-4, atol=1e-5)
    assert not ops.linear_ln_supported(100, N, K, "none", True)  # small M: callers keep the LayerNorm kernel


# ---------------------------------------------------------------- layernorm
@pytest.mark.parametrize("M,d", [(5, 64), (197, 192), (1000, 768), (33, 1280), (7, 4096), (64, 8)])
@pytest.mark.parametrize("eps", [1e-5, 1e-6])
def test_layernorm(ops, M, d, eps):
    x = bf(synth_input("ln_x", (M, d), 20)) * 3 + 1
    g = synth_input("ln_g", (d,), 21, scale=0.1) + 1
    b = synth_input("ln_b", (d,), 22, scale=0.1)
    want = RT.layernorm({"weight": g, "bias": b}, "", x.float(), eps)
    close_bf16(ops.layernorm(x.cuda(), g.cuda(), b.cuda(), eps), want)
    got32 = ops.layernorm(x.float().cuda(), g.cuda(), b.cuda(), eps, out_dtype=torch.float32)
    torch.testing.assert_close(got32.cpu(), want, rtol=2e-5, atol=2e-5)


def test_layernorm_strided_rows(ops):
    """Row stride > d: normalising only the cls row of (N, L, d) (ViT pooled-row shortcut)."""
    x = bf(synth_input("ln_x3", (6, 5, 64), 23)).cuda()
    g = torch.ones(64, device="cuda")
    b = torch.zeros(64, device="cuda")
    want = RT.layernorm({"weight": g.cpu(), "bias": b.cpu()}, "", x[:, 0].float().cpu(), 1e-6)
    close_bf16(ops.layernorm(x[:, 0], g, b, 1e-6), want)


# ---------------------------------------------------------------- attention
def ref_attn(q, k, v, H, causal):
    qh, kh, vh = (RT.split_heads(t.float(), H) for t in (q, k, v))
    return RT.merge_heads(RT.sdpa(qh, kh, vh, None, causal))


@pytest.mark.parametrize("B,H,Lq,Lk", [(2, 3, 197, 197), (1, 1, 1, 1), (2, 2, 64, 64), (1, 2, 130, 577), (1, 8, 1500, 1500),
                                       (3, 1, 1, 300), (2, 4, 33, 5), (1, 1, 128, 128), (1, 2, 129, 63)])
def test_attention_noncausal(ops, B, H, Lq, Lk):
    q = bf(synth_input("at_q", (B, Lq, H * 64), 30))
    k = bf(synth_input("at_k", (B, Lk, H * 64), 31))
    v = bf(synth_input("at_v", (B, Lk, H * 64), 32))
    got = ops.attention(q.cuda(), k.cuda(), v.cuda(), H)
    close_bf16(got, ref_attn(q, k, v, H, False), rel=1.5e-2)


@pytest.mark.parametrize("Lk", [1, 5, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 96, 127, 128, 160, 197, 208, 209, 224, 225, 255, 256, 257])
def test_attention_short_key_sequences(ops, Lk):
    """Every 32-key block count of the whole-K/V-in-LDS kernel (Lk <= 256: one-pass softmax), block and 16-key step edges,
    and the first length past it (tiled kernel); queries spanning two workgroups with a ragged last wave."""
    B, H, Lq = 2, 2, 165
    q = bf(synth_input("as_q", (B, Lq, H * 64), 130 + Lk))
    k = bf(synth_input("as_k", (B, Lk, H * 64), 131 + Lk))
    v = bf(synth_input("as_v", (B, Lk, H * 64), 132 + Lk))
    got = ops.attention(q.cuda(), k.cuda(), v.cuda(), H)
    close_bf16(got, ref_attn(q, k, v, H, False), rel=1.5e-2)
    again = ops.attention(q.cuda(), k.cuda(), v.cuda(), H)
    assert torch.equal(got, again)


@pytest.mark.parametrize("B,H,Lq,Lk", [(1, 1, 1, 197), (2, 3, 32, 197), (2, 3, 33, 40), (1, 2, 255, 256), (3, 2, 256, 197),
                                       (70, 12, 70, 70), (300, 3, 5, 33)])
def test_attention_short_heads_walk(ops, B, H, Lq, Lk):
    """The persistent per-head kernel: query-count edges, and more (batch, head) pairs than workgroups so that every
    workgroup walks several heads through both LDS buffers."""
    q = bf(synth_input("ah_q", (B, Lq, H * 64), 140))
    k = bf(synth_input("ah_k", (B, Lk, H * 64), 141))
    v = bf(synth_input("ah_v", (B, Lk, H * 64), 142))
    got = ops.attention(q.cuda(), k.cuda(), v.cuda(), H)
    close_bf16(got, ref_attn(q, k, v, H, False), rel=1.5e-2)
    assert torch.equal(got, ops.attention(q.cuda(), k.cuda(), v.cuda(), H))


@pytest.mark.parametrize("B,H,Lq,Lk", [(2, 2, 4, 4), (1, 3, 200, 200), (2, 1, 448, 448), (1, 2, 6, 9), (1, 1, 130, 40), (1, 1, 1, 5)])
def test_attention_causal_top_left(ops, B, H, Lq, Lk):
    """causal is top-left aligned, including rectangular Lq != Lk."""
    q = bf(synth_input("at_q", (B, Lq, H * 64), 33))
    k = bf(synth_input("at_k", (B, Lk, H * 64), 34))
    v = bf(synth_input("at_v", (B, Lk, H * 64), 35))
    got = ops.attention(q.cuda(), k.cuda(), v.cuda(), H, causal=True)
    close_bf16(got, ref_attn(q, k, v, H, True), rel=1.5e-2)


def test_attention_packed_qkv_and_broadcast_query(ops):
    B, L, H = 2, 77, 2
    qkv = bf(synth_input("at_qkv", (B, L, 3 * H * 64), 36)).cuda()
    q, k, v = qkv[..., : H * 64], qkv[..., H * 64 : 2 * H * 64], qkv[..., 2 * H * 64 :]
    close_bf16(ops.attention(q, k, v, H), ref_attn(q.cpu(), k.cpu(), v.cpu(), H, False), rel=1.5e-2)
    probe = bf(synth_input("at_probe", (1, 1, H * 64), 37)).cuda().expand(B, 1, H * 64)  # stride-0 batch (MAP head)
    close_bf16(ops.attention(probe, k, v, H), ref_attn(probe.cpu(), k.cpu(), v.cpu(), H, False), rel=1.5e-2)


def test_attention_online_softmax_rescale_branch(ops):
    """Force the running max to jump at a late key tile (a spike), so the rescale path is exercised."""
    B, H, L = 1, 1, 300
    q = bf(synth_input("at_q", (B, L, 64), 38))
    k = bf(synth_input("at_k", (B, L, 64), 39))
    v = bf(synth_input("at_v", (B, L, 64), 40))
    k[0, 257] = q[0, 10] * 4  # key 257 (tile 4) dominates query 10
    got = ops.attention(q.cuda(), k.cuda(), v.cuda(), H)
    close_bf16(got, ref_attn(q, k, v, H, False), rel=1.5e-2)


def test_attention_rows_sum_property_at_full_size(ops):
    """Size-independent property at the C2 shape: with v == 1 every output element is exactly 1
    (softmax rows sum to one), for all 256*12 heads and 197 tokens."""
    B, H, L = 256, 12, 197
    q = bf(synth_input("at_qf", (B, L, H * 64), 41)).cuda()
    k = bf(synth_input("at_kf", (B, L, H * 64), 42)).cuda()
    v = torch.ones(B, L, H * 64, dtype=torch.bfloat16, device="cuda")
    out = ops.attention(q, k, v, H).float()
    assert (out - 1).abs().max().item() <= 2 ** -7  # bf16(p) rounding in P.V, renormalised by the fp32 row sum


# ---------------------------------------------------------------- ViT token assembly
@pytest.mark.parametrize("N,img,d,cls", [(1, 224, 192, True), (3, 64, 128, False), (2, 384, 1024, False), (5, 32, 64, True)])
def test_vit_tokens(ops, N, img, d, cls):
    P = 16
    L = (img // P) ** 2
    imgs = synth_input("vt_img", (N, 3, img, img), 50)
    sd = {
        "patch_embed.weight": bf(synth_input("vt_w", (d, 3, P, P), 51, scale=0.04)).float(),
        "patch_embed.bias": synth_input("vt_b", (d,), 52, scale=0.1),
        "pe": synth_input("vt_pe", (1, L, d), 53, scale=0.1),
    }
    if cls:
        sd["cls_token"] = synth_input("vt_cls", (1, 1, d), 54, scale=0.1)
    # the kernel rounds pixels to bf16 before the MFMA: mirror that rounding point in the oracle
    want = RV.tokens(sd, bf(imgs).float())
    got = ops.vit_tokens(imgs.cuda(), bf(sd["patch_embed.weight"]).view(d, -1).cuda(), sd["patch_embed.bias"].cuda(),
                         sd["pe"].view(L, d).cuda(), sd["cls_token"].view(-1).cuda() if cls else None, P)
    assert got.shape == (N, L + int(cls), d)
    close_bf16(got, want)


@pytest.mark.parametrize("kernel", ["6", "7"])
def test_tile_gemm_kernels_forced(kernel):
    """csrc/linear_bf16_tile.hip forced by PM_GEMM_KERNEL: 6 = 256 x 256 tiles, 7 = 320 x 256 tiles, in a fresh process
    (tools/tile_check.py): ViT-B/16's four GEMMs at batch 256 (bias + residual + LayerNorm-fold row partials from the matrix
    pipe; LayerNorm-fold consumer with and without GELU), ragged edges (M = 4104 = 16 tiles + 8 rows, N = 520 / 1000: rows and
    features beyond the operand are loaded clamped and never stored), one tile per workgroup (M = 25216 at 320 rows), the
    smallest shape the dispatcher sends here; values against an fp32 reference on every 97th row and the last 40, row partials
    against sums over the rounded outputs, bit-identical reruns."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "tile_check.py")], env=dict(os.environ, PM_GEMM_KERNEL=kernel),
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.skipif(not _hip_has_experiments(), reason="experiments build only (make experiments; PM_MI355X_LIB)")
@pytest.mark.parametrize("kernel", ["4", "5"])
def test_k_split_gemm_matches_whole_tiles(kernel):
    """csrc/linear_bf16_sk.hip, forced by PM_GEMM_KERNEL: 4 = stream-K (opt-in: K steps dealt out as one stream), 5 = the
    hybrid form (default where it wins: whole tiles + the last round's tiles in two K halves).  Tiles shared by two
    workgroups are combined in fp32 through the workspace.  Same operands -> results within bf16 rounding of the whole-tile
    kernel's (a split tile sums its two K ranges separately), deterministic, tickets left at zero; with the LayerNorm fold's
    row partials and a residual as out_proj / linear2 use it.  Shapes: 258 tiles (one tail tile on two of the XCDs), ViT-B/16's
    out_proj at batch 256 (591 tiles: 9-10 tail tiles per XCD), 628 tiles with GELU (14-15 per XCD), K with an odd number of
    64-steps."""
    import subprocess
    import sys

    code = r"""
import os, sys, torch
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "pytorch-models_amd")]
from pytorch_models._hip import ops
torch.manual_seed(1)
for (M, N, K, act, resid, rows) in [(33000, 512, 1024, "none", True, False), (50432, 768, 768, "none", True, True),
                                    (40000, 1024, 320, "gelu", False, False), (50432, 768, 3072, "none", True, True)]:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16) if resid else None
    got = ops.linear(x, w, b, act=act, resid=r, want_row_stats=rows)
    got, st = got if rows else (got, None)
    rows_chk = slice(0, M, 7)  # every seventh row: all tiles, a seventh of the reference's memory
    ref = x[rows_chk].float() @ w.float().T + b
    if act == "gelu":
        ref = torch.nn.functional.gelu(ref)
    if resid:
        ref = ref + r[rows_chk].float()
    err = (got[rows_chk].float() - ref).abs().max().item()
    assert err <= 2 ** -7 * ref.abs().max().item() + 1e-2, err
    again = ops.linear(x, w, b, act=act, resid=r, want_row_stats=rows)
    again = again[0] if rows else again
    assert torch.equal(got, again)
    if rows:
        blk = got.float().view(M, N // 64, 64)
        torch.testing.assert_close(st[..., 0], blk.sum(-1), rtol=1e-4, atol=1e-3)
        torch.testing.assert_close(st[..., 1], (blk * blk).sum(-1), rtol=1e-4, atol=1e-3)
    ws = list(ops._GEMM_WS.values())
    assert ws and int(ws[0][0][:4096].view(torch.int32).abs().sum()) == 0  # every ticket back at zero
print("ok")
"""
    env = dict(os.environ, PM_GEMM_KERNEL=kernel)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
