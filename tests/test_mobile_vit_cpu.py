"""MobileViT on the CPU: the oracle (oracle/ref_mobile_vit.py) against outputs captured from the reference
(tests/golden/mobile_vit.npz, make_golden.py g_mobile_vit), this package's classes against the reference's state_dict layout and
its cvnets loader (digests in tests/golden/mobile_vit_converter.json), and the host-side pieces (patch permutations, BatchNorm
folding, error behaviour).  No kernel runs here."""
import json
import os

import pytest
import torch

import ckpt_synth as C
from oracle import ref_mobile_vit as RM
from synthweights import fill_module, synth_input

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("variant", ["xxs", "xs", "s"])
def test_oracle_matches_the_reference_outputs(golden, variant):
    from pytorch_models.image.mobile_vit import MobileViT

    g = golden("mobile_vit")
    m = MobileViT.from_apple(variant).eval()
    fill_module(m, 62)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = synth_input("mv_x", (2, 3, 64, 64), 61)
    with torch.no_grad():
        outs = RM.stages(sd, x)
        for i, o in enumerate(outs):
            torch.testing.assert_close(o, g[f"{variant}_stage{i}"], rtol=2e-5, atol=2e-5)
        torch.testing.assert_close(RM.forward(sd, x), g[f"{variant}_out"], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("variant", ["xxs", "xs", "s"])
def test_state_dict_layout_and_apple_loader_match_the_reference(variant):
    """Same parameter / buffer names as the reference module, and the cvnets key map (fused qkv split in thirds, BGR flip of
    the first convolution, classifier dropped, every key consumed) lands the same values in the same places."""
    from pytorch_models.image.mobile_vit import MobileViT

    want = json.load(open(os.path.join(ROOT, "tests", "golden", "mobile_vit_converter.json")))[variant]
    channels, d_models, out_dim, expansion = RM.VARIANTS[variant]
    m = MobileViT.from_apple(variant)
    assert sorted(m.state_dict()) == sorted(want)
    ck = C.apple_mobilevit(channels, d_models, out_dim, expansion, seed=63)
    m.load_apple_state_dict(ck)
    got = C.state_digest(m.state_dict())
    for k in want:
        assert got[k] == pytest.approx(want[k], rel=1e-12, abs=1e-12), k
    ck["stray.weight"] = torch.zeros(1)
    with pytest.raises(KeyError, match="unused"):
        MobileViT.from_apple(variant).load_apple_state_dict(ck)


def test_patch_permutations_agree_with_the_nchw_form():
    from pytorch_models.image import mobile_vit as MV

    x = synth_input("mv_perm", (2, 6, 8, 12), 3)  # NCHW
    seq_ref, n_ref = RM.unfold(x, 2)
    seq, n = MV.unfold(x.permute(0, 2, 3, 1).contiguous(), 2)
    assert n == n_ref and torch.equal(seq, seq_ref)
    back = MV.fold(seq, 2, n)
    assert torch.equal(back, x.permute(0, 2, 3, 1)) and torch.equal(RM.fold(seq_ref, 2, n_ref), x)


def test_batchnorm_folding_is_the_eval_affine_map():
    from pytorch_models.image import mobile_vit as MV

    blk = MV.conv_norm_act(6, 10, 3, 2).eval()
    fill_module(blk, 5)
    w, b = MV._folded(blk[0], blk[1])
    assert w.shape == (10, 3, 3, 6) and w.dtype == torch.bfloat16 and b.dtype == torch.float32
    x = synth_input("mv_fold", (2, 6, 9, 9), 4)
    with torch.no_grad():
        want = blk[1](blk[0](x))
        got = torch.nn.functional.conv2d(x, w.float().permute(0, 3, 1, 2), b, 2, 1)
    assert float((got - want).norm() / want.norm()) < 5e-3  # the weight is rounded to bf16, nothing else differs
    v0 = blk[1].running_var.clone()
    blk[1].running_var.mul_(2.0)  # a changed statistic rebuilds the folded tensors
    w2, _ = MV._folded(blk[0], blk[1])
    assert not torch.equal(w, w2) and not torch.equal(v0, blk[1].running_var)


def test_constructor_table_and_error_behaviour():
    from pytorch_models.image.mobile_vit import MBConv, MobileViT, MobileViTBlock

    with pytest.raises(KeyError):
        MobileViT.from_apple("xl")
    with pytest.raises(NotImplementedError, match="no network"):
        MobileViT.from_apple("xxs", pretrained=True)
    m = MobileViT.from_apple("xs")
    assert isinstance(m[2][1], MobileViTBlock) and len(m[3][1].transformer) == 4 and m[3][1].transformer[0].sa.head_dim == 30
    assert isinstance(m[1][0], MBConv) and not m[1][0].residual and m[1][1].residual
    with pytest.raises(RuntimeError, match="HIP devices only"):
        m.eval()(torch.zeros(1, 3, 64, 64))
