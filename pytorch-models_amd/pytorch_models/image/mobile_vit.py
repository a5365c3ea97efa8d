"""MobileViT (https://arxiv.org/abs/2110.02178) on the MI355X kernels: drop-in for /root/reference
pytorch_models/image/mobile_vit.py (same classes, constructor arguments, module and parameter names, `from_apple` table and
`load_apple_state_dict` key map), so a reference state_dict loads unchanged.

Execution (eval mode, bf16 parameters, HIP tensors; there is no CPU path):

* activations travel as NHWC bf16 between layers (the reference's NCHW is converted once on the way in): a 1 x 1 convolution is then
  a GEMM over the N*H*W rows (`pm_linear_bf16`, SiLU / residual in its epilogue), `unfold` / `fold` are row permutations;
* `nn.Conv2d(bias=False) + nn.BatchNorm2d` in eval mode is one affine map: the norm's scale goes into the weight, its shift
  becomes the bias (derived tensors, rebuilt when a parameter or running statistic changes) - 3 x 3 dense, strided and
  depthwise convolutions run on `pm_conv2d_nhwc_bf16` with SiLU (and MBConv's residual) in the epilogue;
* the transformer is this package's `Encoder` (n_heads = 4: head dims 16 .. 60 on the generic attention kernel, SiLU MLP);
* `nn.AdaptiveAvgPool2d(1) + Flatten` is `pm_mean_rows_bf16`.
"""
from __future__ import annotations

import torch
from torch import Tensor, nn

from .._hip import ops
from ..transformer import Encoder, _f32, derived, require_bf16_params


def conv_norm_act(in_dim: int, out_dim: int, kernel_size: int, stride: int = 1, groups: int = 1):
    return nn.Sequential(
        nn.Conv2d(in_dim, out_dim, kernel_size, stride, (kernel_size - 1) // 2, groups=groups, bias=False),
        nn.BatchNorm2d(out_dim),
        nn.SiLU(),
    )


def _folded(conv: nn.Conv2d, norm: nn.BatchNorm2d | None):
    """(weight (Cout, kh, kw, Cin / groups) bf16 with the norm's scale folded in, bias f32 or None) of conv [+ eval BatchNorm]."""
    params = [conv.weight] + ([conv.bias] if conv.bias is not None else [])
    if norm is not None:
        params += [norm.weight, norm.bias, norm.running_mean, norm.running_var]

    def build():
        w = conv.weight.detach().float()
        b = conv.bias.detach().float() if conv.bias is not None else None
        if norm is not None:
            scale = norm.weight.detach().float() / torch.sqrt(norm.running_var.detach().float() + norm.eps)
            shift = norm.bias.detach().float() - norm.running_mean.detach().float() * scale
            w = w * scale.view(-1, 1, 1, 1)
            b = shift if b is None else b * scale + shift
        return w.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16), (b.contiguous() if b is not None else None)

    return derived(conv, "folded_nhwc", params, build)


def _conv(x: Tensor, conv: nn.Conv2d, norm: nn.BatchNorm2d | None, act: str, resid: Tensor | None = None) -> Tensor:
    """x (N, H, W, Cin) bf16 -> conv [+ norm] [+ act] [+ resid], NHWC bf16."""
    if norm is not None and norm.training:
        raise NotImplementedError("MobileViT: BatchNorm in training mode is not covered by this build (call .eval())")
    w, b = _folded(conv, norm)
    k, s, g = conv.kernel_size[0], conv.stride[0], conv.groups
    if k == 1 and s == 1 and g == 1:  # a GEMM over the pixels
        N, H, W, C = x.shape
        r2 = resid.reshape(N * H * W, -1) if resid is not None else None
        y = ops.linear(x.reshape(N * H * W, C), w.view(w.shape[0], C), b, act=act, resid=r2)
        return y.view(N, H, W, -1)
    return ops.conv2d_nhwc(x, w, b, s, conv.padding[0], g, act, resid)


def _cna(x: Tensor, block: nn.Sequential, resid: Tensor | None = None) -> Tensor:
    """conv_norm_act / conv_norm blocks: [Conv2d, BatchNorm2d(, SiLU)]."""
    act = "silu" if len(block) > 2 else "none"
    return _conv(x, block[0], block[1], act, resid)


# from MobileNetv2
class MBConv(nn.Sequential):
    def __init__(self, in_dim: int, expansion: int, out_dim: int, stride: int = 1) -> None:
        hidden_dim = in_dim * expansion
        self.residual = (in_dim == out_dim) and (stride == 1)
        super().__init__()
        self.pw1 = conv_norm_act(in_dim, hidden_dim, 1)
        self.dw = conv_norm_act(hidden_dim, hidden_dim, 3, stride, groups=hidden_dim)
        self.pw2 = nn.Sequential(nn.Conv2d(hidden_dim, out_dim, 1, bias=False), nn.BatchNorm2d(out_dim))

    def forward_nhwc(self, x: Tensor) -> Tensor:
        h = _cna(_cna(x, self.pw1), self.dw)
        return _cna(h, self.pw2, x if self.residual else None)

    def forward(self, x: Tensor) -> Tensor:  # NCHW in, NCHW out (the reference's layout)
        return _from_nhwc(self.forward_nhwc(_to_nhwc(x, self)))


def unfold(x: Tensor, patch_size: int) -> tuple[Tensor, tuple[int, int]]:
    """NHWC (N, H, W, C) -> (N, p * p, nH * nW, C): the pixels at the same offset inside every patch form a sequence
    (the reference's unfold works on NCHW: mobile_vit.py:31-39; same result)."""
    N, H, W, C = x.shape
    nH, nW = H // patch_size, W // patch_size
    return (
        x.view(N, nH, patch_size, nW, patch_size, C).permute(0, 2, 4, 1, 3, 5).reshape(N, patch_size * patch_size, nH * nW, C)
    ), (nH, nW)


def fold(x: Tensor, patch_size: int, n_patches: tuple[int, int]) -> Tensor:
    nH, nW = n_patches
    N, C = x.shape[0], x.shape[-1]
    return x.view(N, patch_size, patch_size, nH, nW, C).permute(0, 3, 1, 4, 2, 5).reshape(N, nH * patch_size, nW * patch_size, C)


class MobileViTBlock(nn.Module):
    patch_size = 2

    def __init__(self, in_dim: int, d_model: int, n_layers: int) -> None:
        super().__init__()
        self.in_conv = nn.Sequential(conv_norm_act(in_dim, in_dim, 3), nn.Conv2d(in_dim, d_model, 1, bias=False))
        self.transformer = Encoder(n_layers, d_model, n_heads=4, mlp_ratio=2.0, act="silu")
        self.norm = nn.LayerNorm(d_model)
        self.out_proj = conv_norm_act(d_model, in_dim, 1)
        self.out_fusion = conv_norm_act(in_dim * 2, in_dim, 3)

    def forward_nhwc(self, x: Tensor) -> Tensor:
        h = _conv(_cna(x, self.in_conv[0]), self.in_conv[1], None, "none")
        seq, n_patches = unfold(h, self.patch_size)
        seq = self.transformer(seq)
        d = seq.shape[-1]
        seq = ops.layernorm(seq.reshape(-1, d), _f32(self.norm, "g", self.norm.weight), _f32(self.norm, "b", self.norm.bias),
                            self.norm.eps, out_dtype=torch.bfloat16).view(seq.shape)
        out = _cna(fold(seq, self.patch_size, n_patches), self.out_proj)
        return _cna(torch.cat([x, out], -1), self.out_fusion)

    def forward(self, x: Tensor) -> Tensor:
        return _from_nhwc(self.forward_nhwc(_to_nhwc(x, self)))


def _to_nhwc(x: Tensor, module: nn.Module) -> Tensor:
    require_bf16_params(module, type(module).__name__)
    if x.dim() != 4:
        raise ValueError(f"{type(module).__name__}: expected (N, C, H, W), got {tuple(x.shape)}")
    return x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)


def _from_nhwc(x: Tensor) -> Tensor:
    return x.permute(0, 3, 1, 2).contiguous()


class MobileViT(nn.Sequential):
    def __init__(self, channels: list[int], d_models: list[int], out_dim: int, expansion: int) -> None:
        super().__init__(
            nn.Sequential(
                conv_norm_act(3, 16, 3, 2),
                MBConv(16, expansion, channels[0]),
            ),
            nn.Sequential(
                MBConv(channels[0], expansion, channels[1], 2),
                MBConv(channels[1], expansion, channels[1]),
                MBConv(channels[1], expansion, channels[1]),
            ),
            nn.Sequential(
                MBConv(channels[1], expansion, channels[2], 2),
                MobileViTBlock(channels[2], d_models[0], 2),
            ),
            nn.Sequential(
                MBConv(channels[2], expansion, channels[3], 2),
                MobileViTBlock(channels[3], d_models[1], 4),
            ),
            nn.Sequential(
                MBConv(channels[3], expansion, channels[4], 2),
                MobileViTBlock(channels[4], d_models[2], 3),
                conv_norm_act(channels[4], out_dim, 1),
            ),
            nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Flatten(1)),
        )

    def forward(self, imgs: Tensor) -> Tensor:
        """(N, 3, H, W) with H, W multiples of 64 (five stride-2 stages, then 2 x 2 patches) -> (N, out_dim) bf16."""
        x = _to_nhwc(imgs, self)
        if x.shape[1] % 64 or x.shape[2] % 64:
            raise ValueError(f"MobileViT: image sides must be multiples of 64, got {tuple(imgs.shape[2:])}")
        for stage in list(self)[:-1]:
            for m in stage:
                x = m.forward_nhwc(x) if isinstance(m, (MBConv, MobileViTBlock)) else _cna(x, m)
        N, H, W, C = x.shape
        return ops.mean_rows(x.view(N, H * W, C))

    @staticmethod
    def from_apple(variant: str, *, pretrained: bool = False) -> "MobileViT":
        channels, d_models, out_dim, expansion = dict(
            xxs=([16, 24, 48, 64, 80], [64, 80, 96], 320, 2),
            xs=([32, 48, 64, 80, 96], [96, 120, 144], 384, 4),
            s=([32, 64, 96, 128, 160], [144, 192, 240], 640, 4),
        )[variant]
        m = MobileViT(channels, d_models, out_dim, expansion)
        if pretrained:
            raise NotImplementedError(
                "MobileViT.from_apple(pretrained=True): no network in this build - torch.load the cvnets checkpoint "
                "(mobilevit_{xxs,xs,s}.pt, weights_only=True) yourself and pass it to load_apple_state_dict")
        return m

    @torch.no_grad()
    def load_apple_state_dict(self, state_dict: dict[str, Tensor]) -> None:
        """cvnets (ml-cvnets v0.1) checkpoint -> this module, the key map of the reference's loader (mobile_vit.py:125-196):
        every key must be consumed (the classifier's two are dropped)."""
        sd = dict(state_dict)

        def put(t: Tensor, key: str) -> None:
            t.copy_(sd.pop(key))

        def weight(layer: nn.Module, prefix: str) -> None:
            put(layer.weight, f"{prefix}.weight")
            if getattr(layer, "bias", None) is not None:
                put(layer.bias, f"{prefix}.bias")
            if isinstance(layer, nn.BatchNorm2d):
                for name in ("running_mean", "running_var", "num_batches_tracked"):
                    put(getattr(layer, name), f"{prefix}.{name}")

        def conv_norm(block: nn.Sequential, prefix: str) -> None:
            weight(block[0], f"{prefix}.block.conv")
            weight(block[1], f"{prefix}.block.norm")

        def mbconv(layer: MBConv, prefix: str) -> None:
            for sub, name in ((layer.pw1, "exp_1x1"), (layer.dw, "conv_3x3"), (layer.pw2, "red_1x1")):
                conv_norm(sub, f"{prefix}.{name}")

        def vit_block(layer: MobileViTBlock, prefix: str) -> None:
            conv_norm(layer.in_conv[0], f"{prefix}.local_rep.conv_3x3")
            weight(layer.in_conv[1], f"{prefix}.local_rep.conv_1x1.block.conv")
            for i, enc in enumerate(layer.transformer):
                p = f"{prefix}.global_rep.{i}"
                weight(enc.sa_norm, f"{p}.pre_norm_mha.0")
                for proj, part in zip((enc.sa.q_proj, enc.sa.k_proj, enc.sa.v_proj), range(3)):  # fused qkv, split in thirds
                    proj.weight.copy_(sd[f"{p}.pre_norm_mha.1.qkv_proj.weight"].chunk(3)[part])
                    proj.bias.copy_(sd[f"{p}.pre_norm_mha.1.qkv_proj.bias"].chunk(3)[part])
                del sd[f"{p}.pre_norm_mha.1.qkv_proj.weight"], sd[f"{p}.pre_norm_mha.1.qkv_proj.bias"]
                weight(enc.sa.out_proj, f"{p}.pre_norm_mha.1.out_proj")
                weight(enc.mlp_norm, f"{p}.pre_norm_ffn.0")
                weight(enc.mlp.linear1, f"{p}.pre_norm_ffn.1")
                weight(enc.mlp.linear2, f"{p}.pre_norm_ffn.4")
            weight(layer.norm, f"{prefix}.global_rep.{len(layer.transformer)}")
            conv_norm(layer.out_proj, f"{prefix}.conv_proj")
            conv_norm(layer.out_fusion, f"{prefix}.fusion")

        conv_norm(self[0][0], "conv_1")
        self[0][0][0].weight.copy_(self[0][0][0].weight.flip(1))  # cvnets v0.1 reads images with OpenCV (BGR): mobile_vit.py:178-180
        mbconv(self[0][1], "layer_1.0.block")
        for i in range(3):
            mbconv(self[1][i], f"layer_2.{i}.block")
        for stage, name in ((2, "layer_3"), (3, "layer_4"), (4, "layer_5")):
            mbconv(self[stage][0], f"{name}.0.block")
            vit_block(self[stage][1], f"{name}.1")
        conv_norm(self[4][2], "conv_1x1_exp")
        sd.pop("classifier.fc.weight")
        sd.pop("classifier.fc.bias")
        if sd:
            raise KeyError(f"load_apple_state_dict: unused checkpoint keys {sorted(sd)[:5]}...")
