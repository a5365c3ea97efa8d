"""MobileViT on the GPU (through the C ABI): the small-convolution kernel, the row mean, head dims 20 / 30 of the generic
attention kernel, and the three `from_apple` variants in bf16 against the fp32 oracle on the same (bf16-rounded) weights."""
import pytest
import torch

from oracle import ref_mobile_vit as RM
from oracle import ref_transformer as RT
from synthweights import bf16_round_, fill_module, synth_input

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from pytorch_models._hip import ops as o

    return o


def bf(t):
    return t.to(torch.bfloat16)


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,stride,groups,act,resid", [
    (2, 9, 11, 3, 16, 3, 2, 1, "silu", False),      # the stem: 3 input channels (scalar loads), odd sides
    (2, 8, 8, 32, 32, 3, 1, 32, "silu", False),     # depthwise
    (3, 10, 6, 48, 48, 3, 2, 48, "silu", False),    # depthwise, stride 2
    (1, 6, 6, 96, 48, 3, 1, 1, "silu", False),      # dense (out_fusion)
    (2, 5, 7, 16, 24, 3, 1, 1, "none", True),       # residual after a linear output
    (2, 4, 4, 8, 8, 1, 1, 2, "relu", False),        # grouped 1 x 1
])
def test_conv2d_nhwc(ops, N, H, W, Cin, Cout, k, stride, groups, act, resid):
    x = bf(synth_input("cv_x", (N, Cin, H, W), 70)).float()
    w = bf(synth_input("cv_w", (Cout, Cin // groups, k, k), 71) / (k * (Cin // groups) ** 0.5)).float()
    b = synth_input("cv_b", (Cout,), 72)
    want = RM.conv2d(x, w, stride, groups) + b.view(1, -1, 1, 1)
    want = RT.activation(want, act) if act != "none" else want
    r = bf(synth_input("cv_r", tuple(want.shape), 73)).float() if resid else None
    if resid:
        want = want + r
    got = ops.conv2d_nhwc(bf(x.permute(0, 2, 3, 1)).contiguous().cuda(), bf(w.permute(0, 2, 3, 1)).contiguous().cuda(), b.cuda(), stride,
                          (k - 1) // 2, groups, act, bf(r.permute(0, 2, 3, 1)).contiguous().cuda() if resid else None)
    assert got.shape == (N, want.shape[2], want.shape[3], Cout)
    torch.testing.assert_close(got.float().cpu().permute(0, 3, 1, 2), want, rtol=1e-2, atol=1e-2)


def test_mean_rows(ops):
    x = bf(synth_input("mr_x", (3, 37, 40), 74))
    torch.testing.assert_close(ops.mean_rows(x.cuda()).float().cpu(), x.float().mean(1), rtol=1e-2, atol=1e-3)


@pytest.mark.parametrize("hd,H,Lq,Lk,causal", [(20, 4, 33, 64, False), (30, 4, 16, 16, False), (30, 2, 40, 57, True), (2, 3, 5, 9, False)])
def test_attention_head_dims_multiple_of_two(ops, hd, H, Lq, Lk, causal):
    B, D = 2, H * hd
    q, k, v = (synth_input(f"a2_{n}", (B, L, D), 75) for n, L in (("q", Lq), ("k", Lk), ("v", Lk)))
    want = RT.merge_heads(RT.sdpa(RT.split_heads(q, H), RT.split_heads(k, H), RT.split_heads(v, H), None, causal))
    torch.testing.assert_close(ops.attention_f32(q.cuda(), k.cuda(), v.cuda(), H, causal).cpu(), want, rtol=2e-5, atol=2e-5)
    qb, kb, vb = bf(q), bf(k), bf(v)
    wantb = RT.merge_heads(RT.sdpa(RT.split_heads(qb.float(), H), RT.split_heads(kb.float(), H), RT.split_heads(vb.float(), H), None, causal))
    assert rel(ops.attention(qb.cuda(), kb.cuda(), vb.cuda(), H, causal), wantb) < 1e-2


@pytest.mark.parametrize("variant", ["xxs", "xs", "s"])
def test_mobile_vit_matches_the_oracle(variant):
    """bf16 model against the fp32 oracle on the same bf16-representable weights, 64 x 64 images (the smallest the geometry
    allows) and 128 x 96 (rectangular, more patches per sequence); batch-permutation invariance; training-mode norms refused."""
    from pytorch_models.image.mobile_vit import MobileViT

    m = MobileViT.from_apple(variant).eval()
    fill_module(m, 62)
    bf16_round_(m)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    mg = m.to(torch.bfloat16).cuda()
    for shape in ((2, 3, 64, 64), (3, 3, 128, 192)):
        x = bf(synth_input("mv_x", shape, 61)).float()
        with torch.no_grad():
            want = RM.forward(sd, x)
            got = mg(x.cuda())
        assert got.shape == want.shape and got.dtype == torch.bfloat16
        r = rel(got, want)
        print(f"MobileViT-{variant} {shape}: rel-L2 vs oracle {r:.3e}, max-abs {float((got.float().cpu() - want).abs().max()):.3e}")
        assert r < 1e-2, r  # measured 3.7e-3 .. 4.5e-3 over the three variants and both image sizes
    with torch.no_grad():
        perm = torch.tensor([2, 0, 1])
        assert torch.equal(mg(x[perm].cuda()), got[perm.cuda()])
        stage = mg[1][0]  # a strided MBConv on its own, NCHW in / out like the reference module
        y = synth_input("mv_s", (2, stage.pw1[0].in_channels, 16, 16), 5)
        want1 = RM.mbconv({k[len("1.0."):]: v for k, v in sd.items() if k.startswith("1.0.")}, "", bf(y).float(), 2)
        assert rel(stage(y.cuda()), want1) < 2e-2
    mg.train()
    with pytest.raises(NotImplementedError, match="training mode"):
        mg(x.cuda())
