"""Randomised shape sweep of pm_linear_bf16 (all three kernels behind one dispatcher: 128x128, 256x128 persistent,
256x256 persistent) and pm_attention_bf16 against fp32 torch on the same bf16 operands.  Seeded: the same 70 + 24
cases every run.  Large cases are checked on a sample of rows."""
import math
import random

import pytest
import torch

from oracle import ref_transformer as RT

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _cases(n, seed):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        kind = rng.random()
        if kind < 0.35:  # small / ragged: the 128x128 kernel
            M, N, K = rng.randint(1, 700), rng.choice([8, 24, 100, 132, 200, 333, 512, 1000]), 8 * rng.randint(1, 40)
        elif kind < 0.7:  # the 256x128 persistent kernel (>= 512 tiles, M >= 4096)
            N = rng.choice([128, 384, 512, 768, 1024])
            M = rng.randint(512 * 256 * 128 // N // 1 + 1, 512 * 256 * 128 // N + 9000) if N < 1024 else rng.randint(20000, 40000)
            K = 64 * rng.randint(1, 12)
        else:  # the 256x256 persistent kernel (>= 1024 tiles)
            N = rng.choice([1536, 2048, 2304, 3072, 1032])
            M = 1024 * 256 * 256 // N + rng.randint(1, 9000)
            K = 64 * rng.randint(1, 8)
        act = rng.choice(["none", "none", "gelu", "relu", "silu", "approximate_gelu"])
        resid = rng.choice([None, None, "bf16", "f32"])
        odt = rng.choice([torch.bfloat16, torch.bfloat16, torch.float32])
        out.append((M, N, K, act, resid, odt, rng.random() < 0.7))
    return out


def _mid_cases(n, seed):
    """M = 4 k .. 24 k, where the dispatcher's cost model moves between the three kernels (partly filled persistent rounds,
    persistent workgroups without a tile)."""
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        M = rng.randint(4096, 24000)
        N = rng.choice([48, 256, 512, 768, 1024, 1536, 2048, 2304, 3072, 4096, 1000])
        K = 64 * rng.randint(1, 16)
        act = rng.choice(["none", "none", "gelu", "relu"])
        resid = rng.choice([None, "bf16", "f32"])
        odt = rng.choice([torch.bfloat16, torch.bfloat16, torch.float32])
        out.append((M, N, K, act, resid, odt, rng.random() < 0.7))
    return out


@pytest.mark.parametrize("case", _cases(70, 1234) + _mid_cases(30, 99), ids=lambda c: f"{c[0]}x{c[1]}x{c[2]}-{c[3]}-{c[4]}-{'f32' if c[5] == torch.float32 else 'bf16'}")
def test_linear_random_shapes(case):
    from pytorch_models._hip import ops

    M, N, K, act, resid, odt, with_bias = case
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N * 3 + K)
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g) * 0.1 if with_bias else None
    r = None
    if resid:
        r = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16 if resid == "bf16" else torch.float32)
    got = ops.linear(x, w, b, act=act, resid=r, out_dtype=odt)
    assert got.shape == (M, N) and got.dtype == odt and torch.isfinite(got.float()).all()
    idx = torch.arange(M, device="cuda") if M <= 2048 else torch.cat([
        torch.arange(0, 300, device="cuda"), torch.arange(M - 300, M, device="cuda"),
        torch.randint(0, M, (600,), device="cuda", generator=g)])
    want = x[idx].float() @ w.float().T
    if b is not None:
        want = want + b
    if act != "none":
        want = RT.activation(want.cpu(), act).cuda()
    if r is not None:
        want = want + r[idx].float()
    tol = 1e-2 if odt == torch.bfloat16 else 2e-3
    rms = want.square().mean().sqrt().item()
    torch.testing.assert_close(got[idx].float(), want, rtol=tol, atol=tol * max(rms, 1e-3))


def _attn_cases(n, seed):
    rng = random.Random(seed)
    return [(rng.randint(1, 6), rng.choice([1, 2, 3, 8]), rng.randint(1, 400), rng.randint(1, 400), rng.random() < 0.4) for _ in range(n)]


@pytest.mark.parametrize("B,H,Lq,Lk,causal", _attn_cases(24, 99))
def test_attention_random_shapes(B, H, Lq, Lk, causal):
    from pytorch_models._hip import ops

    g = torch.Generator(device="cuda").manual_seed(B * 1000 + Lq * 7 + Lk)
    q = torch.randn(B, Lq, H * 64, device="cuda", generator=g).to(torch.bfloat16)
    k = torch.randn(B, Lk, H * 64, device="cuda", generator=g).to(torch.bfloat16)
    v = torch.randn(B, Lk, H * 64, device="cuda", generator=g).to(torch.bfloat16)
    got = ops.attention(q, k, v, H, causal, None)
    qh, kh, vh = (RT.split_heads(t.float().cpu(), H) for t in (q, k, v))
    want = RT.merge_heads(RT.sdpa(qh, kh, vh, None, causal))
    live = torch.ones(Lq, dtype=torch.bool)
    torch.testing.assert_close(got.float().cpu()[:, live], want[:, live], rtol=2e-2, atol=2e-2)


def _dec_cases(n, seed):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        M = rng.choice([1, 2, 7, 16, 17, 31, 32, 33, 48, 64])
        K = 32 * rng.randint(1, 64)
        N = rng.choice([16, 48, 100, 512, 640, 1536, 2048])
        ln = K <= 1280 and (M <= 32 or K <= 512) and rng.random() < 0.5
        ks = rng.choice([0, 0, 2, 3, 4, 8]) if not ln and K // 32 >= 8 else 0
        out.append((M, N, K, ln, ks, rng.choice(["none", "gelu", "approximate_gelu"]), rng.random() < 0.5))
    return out


@pytest.mark.parametrize("M,N,K,ln,ks,act,with_resid", _dec_cases(40, 7))
def test_dec_linear_random_shapes(M, N, K, ln, ks, act, with_resid):
    """pm_dec_linear / pm_dec_linear_ksplit (row tiles and K parts on separate workgroups) stay fp32-exact."""
    from pytorch_models._hip import ops

    g = torch.Generator().manual_seed(M + 31 * N + 977 * K)
    x = torch.randn(M, K, generator=g) * 2
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16)
    b = torch.randn(N, generator=g) * 0.1
    r = torch.randn(M, N, generator=g) if with_resid else None
    xin = x.double()
    lnp = None
    if ln:
        gam, bet = torch.randn(K, generator=g) * 0.1 + 1, torch.randn(K, generator=g) * 0.1
        xin = RT.layernorm({"weight": gam.double(), "bias": bet.double()}, "", xin, 1e-5)
        lnp = (gam.cuda(), bet.cuda(), 1e-5)
    want = xin @ w.double().T + b.double()
    if act != "none":
        want = RT.activation(want, act)
    if r is not None:
        want = want + r.double()
    if ks:
        got = ops.dec_linear_ksplit(x.cuda(), w.cuda(), b.cuda(), k_split=ks, act=act, resid=r.cuda() if r is not None else None)
    else:
        got = ops.dec_linear(x.cuda(), w.cuda(), b.cuda(), ln=lnp, act=act, resid=r.cuda() if r is not None else None)
    torch.testing.assert_close(got.cpu().double(), want, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("d,L,B,seed", [(128, 700, 190, 1), (256, 333, 200, 2), (384, 197, 230, 3), (512, 1500, 24, 4), (768, 197, 256, 5)])
def test_encoder_layernorm_fold_random_geometries(d, L, B, seed):
    """Encoder stacks at sizes where the LayerNorm fold engages (or, for small tile counts, does not): the chained
    forward agrees with running the layers one by one, and a permuted batch gives the permuted output bit for bit."""
    from pytorch_models.transformer import Encoder
    from synthweights import bf16_round_, fill_module

    m = Encoder(2, d, n_heads=d // 64, norm_eps=1e-6)
    fill_module(m, 40 + seed)
    bf16_round_(m)
    m = m.to(torch.bfloat16).cuda().eval()
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(B, L, d, device="cuda", generator=g).to(torch.bfloat16)
    got = m(x)
    step = x
    for layer in m:
        step = layer(step)
    rel = ((got.float() - step.float()).norm() / step.float().norm()).item()
    assert rel < 1e-2 and torch.isfinite(got.float()).all(), rel
    perm = torch.randperm(B, device="cuda", generator=g)
    torch.testing.assert_close(m(x[perm]), got[perm], rtol=0, atol=0)


def _gc_cases(n, seed):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        cg = rng.choice([4, 8, 12, 16, 28, 32, 44, 48, 60, 64])
        G = rng.choice([1, 2, 3, 16])
        k = rng.choice([1, 2, 3, 19, 31, 64, 128])
        stride = rng.choice([1, 1, 2, 3])
        T = rng.randint(max(1, k // 2), 700)
        pl = rng.randint(0, k)
        pr = max(0, k - pl - T) + rng.randint(0, k)
        out.append((rng.randint(1, 3), T, G, cg, k, stride, pl, pr, rng.choice(["none", "gelu"]), rng.random() < 0.5, rng.random() < 0.7))
    return out


@pytest.mark.parametrize("case", _gc_cases(40, 4321), ids=lambda c: "B{}-T{}-G{}-cg{}-k{}-s{}-p{}_{}-{}".format(*c[:9]))
def test_grouped_conv_random_shapes(case):
    """pm_group_windows + pm_grouped_conv_bf16 over random (clips, steps, groups, channels per group, taps, stride, padding):
    both tile heights (64 and 256 steps), ragged last tiles, the zero-weighted K padding, every channel-chunk count."""
    from pytorch_models._hip import ops

    B, T, G, cg, k, stride, pl, pr, act, with_resid, with_bias = case
    d = G * cg
    g = torch.Generator().manual_seed(T * 13 + cg * 5 + k)
    h = torch.randn(B, T, d, generator=g).to(torch.bfloat16)
    w = (torch.randn(d, cg, k, generator=g) / math.sqrt(cg * k)).to(torch.bfloat16)
    b = torch.randn(d, generator=g) * 0.1 if with_bias else None
    Tp = T + pl + pr
    To = (Tp - k) // stride + 1
    r = torch.randn(B * To, d, generator=g).to(torch.bfloat16) if with_resid else None
    cgp = (cg + 7) // 8 * 8
    Kp = (k * cgp + 63) // 64 * 64
    wk = torch.zeros(G, cg, k, cgp, dtype=torch.bfloat16)
    wk[..., :cg] = w.view(G, cg, cg, k).permute(0, 1, 3, 2)
    wp = torch.zeros(G, cg, Kp, dtype=torch.bfloat16)
    wp[..., : k * cgp] = wk.view(G, cg, k * cgp)
    xg = ops.group_windows(h.cuda(), G, cgp, pl, pr)
    got = ops.grouped_conv(xg, wp.cuda(), None if b is None else b.cuda(), k, stride, cg, act, None if r is None else r.cuda())
    hp = torch.nn.functional.pad(h.float().transpose(1, 2), (pl, pr))
    want = torch.nn.functional.conv1d(hp, w.float(), b, stride=stride, groups=G).transpose(1, 2).reshape(B * To, d)
    if act == "gelu":
        want = RT.activation(want, "gelu")
    if r is not None:
        want = want + r.float()
    assert got.shape == want.shape and torch.isfinite(got.float()).all()
    err = (got.float().cpu() - want).abs()
    assert (err <= 1e-2 * want.abs() + 1e-2 * want.pow(2).mean().sqrt()).all(), err.max().item()


def _row_cases(n, seed):
    rng = random.Random(seed)
    return [(rng.randint(1, 3000), 8 * rng.randint(1, 512), rng.choice([torch.bfloat16, torch.float32]), rng.choice([torch.bfloat16, torch.float32]),
             rng.choice(["ln", "ln_noaffine", "rms"]), rng.choice(["none", "gelu"]), rng.choice([None, torch.bfloat16, torch.float32]))
            for _ in range(n)]


@pytest.mark.parametrize("case", _row_cases(30, 777), ids=lambda c: f"{c[0]}x{c[1]}-{c[4]}-{c[5]}")
def test_row_norms_random_shapes(case):
    """pm_layernorm / pm_layernorm_ex / pm_rmsnorm over random (rows, width <= 4096, dtypes, affine, activation, residual)."""
    from pytorch_models._hip import ops

    M, d, xdt, ydt, kind, act, rdt = case
    g = torch.Generator().manual_seed(M * 31 + d)
    x = (torch.randn(M, d, generator=g) * 1.5 + 0.3).to(xdt)
    gamma, beta = torch.randn(d, generator=g) * 0.2 + 1.0, torch.randn(d, generator=g) * 0.1
    xf = x.float()
    if kind == "rms":
        want = xf * torch.rsqrt((xf * xf).mean(-1, keepdim=True) + 1e-5) * gamma
        got = ops.rmsnorm(x.cuda(), gamma.cuda(), 1e-5, ydt)
    else:
        xc = xf - xf.mean(-1, keepdim=True)
        want = xc * torch.rsqrt((xc * xc).mean(-1, keepdim=True) + 1e-5)
        if kind == "ln":
            want = want * gamma + beta
        if act == "gelu":
            want = RT.activation(want, "gelu")
        r = None if rdt is None else torch.randn(M, d, generator=g).to(rdt)
        if r is not None:
            want = want + r.float()
        got = ops.layernorm(x.cuda(), gamma.cuda() if kind == "ln" else None, beta.cuda() if kind == "ln" else None, 1e-5, ydt, act=act,
                            resid=None if r is None else r.cuda())
    assert got.dtype == ydt and got.shape == (M, d)
    tol = dict(rtol=3e-5, atol=3e-5) if ydt == torch.float32 else dict(rtol=4e-3, atol=2e-3)
    torch.testing.assert_close(got.float().cpu(), want, **tol)


@pytest.mark.parametrize("case", [(1, 10, 8, "none"), (2, 400, 64, "instance"), (3, 16000, 512, "layer"), (1, 48000, 512, "instance"),
                                  (5, 3333, 256, "layer"), (2, 1285, 24, "instance"), (4, 2561, 128, "none")],
                         ids=lambda c: f"B{c[0]}-L{c[1]}-C{c[2]}-{c[3]}")
def test_w2v_stem0_shapes(case):
    """pm_w2v_stem0 across clip lengths (chunk boundaries of both passes), channel counts and norms, against conv1d in fp32."""
    from pytorch_models._hip import ops

    B, L, C0, norm = case
    g = torch.Generator().manual_seed(L + C0)
    x = torch.randn(B, L, generator=g) * 0.5 + 0.05
    w = torch.randn(C0, 10, generator=g) / 3.0
    b = torch.randn(C0, generator=g) * 0.1
    gamma, beta = torch.randn(C0, generator=g) * 0.2 + 1.0, torch.randn(C0, generator=g) * 0.1
    h = torch.nn.functional.conv1d(x[:, None], w[:, None], b, stride=5).transpose(1, 2)  # (B, T0, C0)
    if norm == "layer":
        h = torch.nn.functional.layer_norm(h, (C0,), gamma, beta, 1e-5)
    elif norm == "instance":
        hc = h - h.mean(1, keepdim=True)
        h = hc * torch.rsqrt((hc * hc).mean(1, keepdim=True) + 1e-5) * gamma + beta
    want = RT.activation(h, "gelu")
    got = ops.w2v_stem0(x.cuda(), w.cuda(), b.cuda(), norm, gamma.cuda() if norm != "none" else None, beta.cuda() if norm != "none" else None,
                        1e-5, 5)
    torch.testing.assert_close(got.float().cpu(), want, rtol=4e-3, atol=2e-3)
