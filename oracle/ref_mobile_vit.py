"""Oracle for MobileViT (test infrastructure, see oracle/__init__.py).

Restates /root/reference pytorch_models/image/mobile_vit.py:10-112 in plain fp32 tensor algebra on NCHW tensors: a convolution
is an explicit gather of its windows (``F.unfold``: pure data movement) contracted with the weight, BatchNorm is its eval-mode
affine map, SiLU is x * sigmoid(x); the transformer is oracle.ref_transformer.encoder.  ``sd`` is the reference's state_dict.
Pinned to outputs captured from the reference (tests/golden/mobile_vit.npz, tests/test_mobile_vit_cpu.py).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import Tensor

from . import ref_transformer as T

# (channels, d_models, out_dim, expansion) - mobile_vit.py:105-109
VARIANTS = dict(
    xxs=([16, 24, 48, 64, 80], [64, 80, 96], 320, 2),
    xs=([32, 48, 64, 80, 96], [96, 120, 144], 384, 4),
    s=([32, 64, 96, 128, 160], [144, 192, 240], 640, 4),
)
BN_EPS = 1e-5  # nn.BatchNorm2d default


def conv2d(x: Tensor, w: Tensor, stride: int, groups: int) -> Tensor:
    """nn.Conv2d(kernel k, stride, padding (k - 1) // 2, groups, bias=False) - mobile_vit.py:12."""
    N, C, H, W = x.shape
    Cout, cg, k, _ = w.shape
    pad = (k - 1) // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    win = F.unfold(x, k, padding=pad, stride=stride).view(N, groups, cg, k * k, Ho * Wo)  # (N, g, ci, window, pixel)
    wg = w.view(groups, Cout // groups, cg, k * k)  # (g, co, ci, window)
    return torch.einsum("ngcwp,gocw->ngop", win, wg).reshape(N, Cout, Ho, Wo)


def batchnorm(sd: dict, p: str, x: Tensor) -> Tensor:
    """nn.BatchNorm2d in eval mode - mobile_vit.py:13."""
    scale = sd[p + "weight"] / torch.sqrt(sd[p + "running_var"] + BN_EPS)
    shift = sd[p + "bias"] - sd[p + "running_mean"] * scale
    return x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)


def conv_norm_act(sd: dict, p: str, x: Tensor, stride: int = 1, groups: int = 1, act: bool = True) -> Tensor:
    """conv_norm_act (mobile_vit.py:10-15): p + "0." conv, p + "1." norm, SiLU."""
    y = batchnorm(sd, p + "1.", conv2d(x, sd[p + "0.weight"], stride, groups))
    return T.activation(y, "silu") if act else y


def mbconv(sd: dict, p: str, x: Tensor, stride: int) -> Tensor:
    """MBConv - mobile_vit.py:19-29; the residual exists iff the shapes allow it (in_dim == out_dim, stride 1)."""
    hidden = sd[p + "pw1.0.weight"].shape[0]
    h = conv_norm_act(sd, p + "pw1.", x)
    h = conv_norm_act(sd, p + "dw.", h, stride, groups=hidden)
    h = conv_norm_act(sd, p + "pw2.", h, act=False)
    return x + h if (h.shape == x.shape and stride == 1) else h


def unfold(x: Tensor, ps: int):
    """mobile_vit.py:32-40."""
    N, C, H, W = x.shape
    nH, nW = H // ps, W // ps
    return x.view(N, C, nH, ps, nW, ps).permute(0, 3, 5, 2, 4, 1).reshape(N, ps * ps, nH * nW, C), (nH, nW)


def fold(x: Tensor, ps: int, n_patches) -> Tensor:
    """mobile_vit.py:43-52."""
    nH, nW = n_patches
    N, C = x.shape[0], x.shape[-1]
    return x.view(N, ps, ps, nH, nW, C).permute(0, 5, 3, 1, 4, 2).reshape(N, C, nH * ps, nW * ps)


def mobilevit_block(sd: dict, p: str, x: Tensor, rp=None) -> Tensor:
    """MobileViTBlock.forward - mobile_vit.py:66-69 (patch_size 2, 4 heads, mlp_ratio 2, SiLU: mobile_vit.py:56,62)."""
    h = conv_norm_act(sd, p + "in_conv.0.", x)
    h = conv2d(h, sd[p + "in_conv.1.weight"], 1, 1)
    seq, n_patches = unfold(h, 2)
    seq = T.encoder(sd, p + "transformer.", 4, seq, act="silu", rp=rp)
    seq = T.layernorm(sd, p + "norm.", seq, 1e-5)
    out = conv_norm_act(sd, p + "out_proj.", fold(seq, 2, n_patches))
    return conv_norm_act(sd, p + "out_fusion.", torch.cat([x, out], 1))


def stages(sd: dict, x: Tensor, rp=None) -> list[Tensor]:
    """MobileViT as nn.Sequential - mobile_vit.py:74-101: the outputs of its five convolutional stages."""
    outs = []
    x = mbconv(sd, "0.1.", conv_norm_act(sd, "0.0.", x, 2), 1)
    outs.append(x)
    x = mbconv(sd, "1.0.", x, 2)
    x = mbconv(sd, "1.1.", x, 1)
    x = mbconv(sd, "1.2.", x, 1)
    outs.append(x)
    for s in (2, 3, 4):
        x = mobilevit_block(sd, f"{s}.1.", mbconv(sd, f"{s}.0.", x, 2), rp)
        if s == 4:
            x = conv_norm_act(sd, "4.2.", x)
        outs.append(x)
    return outs


def forward(sd: dict, imgs: Tensor, rp=None) -> Tensor:
    """(N, 3, H, W) -> (N, out_dim): the stages, then AdaptiveAvgPool2d(1) + Flatten (mobile_vit.py:100)."""
    return stages(sd, imgs, rp)[-1].mean((2, 3))
