"""ViT on MI355X, drop-in for /root/reference pytorch_models/image/vit.py (ViT, from_google,
from_facebook, resize_pe; same parameter names: patch_embed, cls_token, pe, layers, norm, pooler.*).

forward = pm_vit_tokens (im2col-free patch projection + pe + cls in one kernel) -> Encoder
(transformer.py) -> final LayerNorm -> pooler.  Unlike the reference, batch > 1 works with a cls
token: the cls row is broadcast over the batch, i.e. the result is the stack of the reference's
batch-1 results (the reference's torch.cat raises there - vit.py:80-81).
"""
from __future__ import annotations

from functools import partial

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from .. import _cpu
from .._hip import ops
from ..transformer import MHA, MLP, Encoder, LayerNorm, _f32, _fused_mlp, _wb, derived


class ClassTokenPooling(nn.Module):
    def forward(self, x: Tensor) -> Tensor:
        return x[:, 0]


class GlobalAveragePooling(nn.Module):
    def forward(self, x: Tensor) -> Tensor:
        return x.mean(1)


class MHAPooling(nn.Module):
    """MAP head (siglip): a learned probe attends over all tokens, then x + mlp(norm(x)) - vit.py:30-43."""

    def __init__(self, d_model: int, n_heads: int, bias: bool = True, mlp_ratio: float = 4.0, norm_eps: float = 1e-6) -> None:
        super().__init__()
        self.probe = nn.Parameter(torch.zeros(1, 1, d_model))
        self.attn = MHA(d_model, n_heads=n_heads, bias=bias)
        self.norm = LayerNorm(d_model, norm_eps)
        self.mlp = MLP(d_model, int(d_model * mlp_ratio))

    def forward(self, x: Tensor) -> Tensor:
        x = self.attn(self.probe, x).squeeze(1)
        return _fused_mlp(self.mlp, self.norm(x), x)


_SIZES = dict(Ti=(12, 192, 3), S=(12, 384, 6), M=(12, 512, 8), B=(12, 768, 12), L=(24, 1024, 16), H=(32, 1280, 16))


class ViT(nn.Module):
    norm_eps = 1e-6

    def __init__(
        self,
        n_layers: int,
        d_model: int,
        n_heads: int,
        patch_size: int,
        img_size: int = 224,
        cls_token: bool = True,
        pool_type: str = "cls_token",
        dropout: float = 0.0,
    ) -> None:
        assert img_size % patch_size == 0
        super().__init__()
        self.patch_embed = nn.Conv2d(3, d_model, patch_size, patch_size)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, d_model)) if cls_token else None
        self.pe = nn.Parameter(torch.zeros(1, (img_size // patch_size) ** 2, d_model))
        self.layers = Encoder(n_layers, d_model, n_heads=n_heads, dropout=dropout, norm_eps=self.norm_eps)
        self.layers.pm_streams = 2  # large batches run as two halves on two HIP streams (transformer.py, Encoder.forward)
        self.norm = LayerNorm(d_model, self.norm_eps)
        poolers = dict(
            cls_token=ClassTokenPooling,
            gap=GlobalAveragePooling,
            mha=partial(MHAPooling, d_model, n_heads, norm_eps=self.norm_eps),
        )
        self.pooler = poolers[pool_type]()

    def tokens(self, imgs: Tensor) -> Tensor:
        """(N, 3, H, W) f32 -> (N, L [+1], d) bf16: patch projection + pe (+ cls) in one kernel."""
        if _cpu.on_cpu(imgs, self.patch_embed.weight):
            return _cpu.vit_tokens(self, imgs)
        if self.patch_embed.weight.dtype == torch.float32:
            return self._tokens_f32(imgs)
        pw = self.patch_embed.weight
        if pw.shape[2] == 16:
            w2d = pw.view(pw.shape[0], -1)
        else:  # other patch sizes: K = 3*P*P zero-padded to a multiple of 64 (derived copy)
            def pad():
                k = 3 * pw.shape[2] * pw.shape[3]
                w = torch.zeros(pw.shape[0], (k + 63) // 64 * 64, dtype=pw.dtype, device=pw.device)
                w[:, :k] = pw.detach().reshape(pw.shape[0], -1)
                return w

            w2d = derived(self, "w2d_pad", (pw,), pad)
        cls = None if self.cls_token is None else _f32(self, "cls", self.cls_token).view(-1)
        return ops.vit_tokens(imgs.float().contiguous(), w2d, _f32(self, "pb", self.patch_embed.bias),
                              _f32(self, "pe", self.pe).view(-1, pw.shape[0]), cls, pw.shape[2])

    def _tokens_f32(self, imgs: Tensor) -> Tensor:
        """fp32 parameters: the patch projection as an fp32 GEMM over patch windows (K order (channel, row, column) = the
        Conv2d weight flattened), + pe in its epilogue; the cls row is prepended (broadcast over the batch: SURVEY F1)."""
        w = self.patch_embed.weight
        d, _, P, _ = w.shape
        N, _, H, W = imgs.shape
        gh, gw = H // P, W // P
        cols = imgs.float().unfold(2, P, P).unfold(3, P, P).permute(0, 2, 3, 1, 4, 5).reshape(N * gh * gw, 3 * P * P)
        pe = self.pe.float().view(-1, d)
        if pe.shape[0] != gh * gw:
            raise ValueError(f"ViT: pe holds {pe.shape[0]} positions, the image has {gh * gw} patches; call resize_pe first")
        y = ops.linear_f32(cols, w.view(d, -1), self.patch_embed.bias, resid=pe.contiguous(), resid_period=gh * gw).view(N, gh * gw, d)
        if self.cls_token is not None:
            y = torch.cat([self.cls_token.float().expand(N, 1, d), y], 1)
        return y

    def forward(self, imgs: Tensor) -> Tensor:
        n_tok = self.pe.shape[1] + (self.cls_token is not None)
        parts = None
        if imgs.dim() == 4 and imgs.is_cuda and self.patch_embed.weight.dtype == torch.bfloat16:
            parts = self.layers.split_sizes(imgs.shape[0], n_tok, torch.bfloat16, imgs.device)
        if parts is None:
            out = self.layers(self.tokens(imgs))
        else:  # a large batch: each part's patch projection runs on the stream its encoder layers run on
            out = self.layers(producers=[lambda lo=lo, hi=hi: self.tokens(imgs[lo:hi]) for lo, hi in parts], device=imgs.device)
        io = self.patch_embed.weight.dtype  # bf16 model -> bf16 features, fp32 model -> fp32 features
        if isinstance(self.pooler, ClassTokenPooling):
            # LayerNorm is row-wise, so normalising only the pooled row equals norm-then-pool (vit.py:83-84)
            return self.norm(out[:, 0], io)
        if out.device.type == "cpu":
            return self.pooler(self.norm(out))
        return self.pooler(self.norm(out)).to(io)

    @torch.no_grad()
    def resize_pe(self, size: int, interpolation_mode: str = "bicubic") -> None:
        """Interpolate the (g, g) grid of position vectors to the grid of a ``size`` x ``size`` image."""
        g_old = int(self.pe.shape[1] ** 0.5)
        g_new = size // self.patch_embed.weight.shape[2]
        grid = self.pe.float().unflatten(1, (g_old, g_old)).permute(0, 3, 1, 2)
        grid = F.interpolate(grid, (g_new, g_new), mode=interpolation_mode)
        self.pe = nn.Parameter(grid.permute(0, 2, 3, 1).flatten(1, 2).to(self.pe.dtype))

    def load_flax_ckpt(self, ckpt, *, big_vision: bool = False, prefix: str = "") -> None:
        """vision_transformer / big_vision ``.npz`` (local path or mapping; nothing is downloaded)."""
        from ..converters import load_flax_vit

        left = load_flax_vit(self, ckpt, big_vision=big_vision, prefix=prefix)
        if left:
            print(left)

    def load_facebook_state_dict(self, state_dict) -> None:
        """DeiT-3 / DINO / DINOv2 state_dict (fused qkv, layer scale folded into out_proj / linear2)."""
        from ..converters import load_facebook_vit

        print(load_facebook_vit(self, state_dict))

    @staticmethod
    def _parse(model_tag: str, default_weights: str):
        tag, weights = model_tag.split("_") if "_" in model_tag else (model_tag, default_weights)
        size, patch = tag.split("/")
        return tag, weights, _SIZES[size], int(patch)

    @staticmethod
    def from_google(model_tag: str, *, pretrained: bool = False, **kwargs) -> "ViT":
        """"Ti/16", "B/16_augreg", "B/16_siglip", ... (default weights: augreg); siglip => no cls token, MAP pooling."""
        _, weights, (n_layers, d_model, n_heads), patch = ViT._parse(model_tag, "augreg")
        extra = dict(cls_token=False, pool_type="mha") if weights == "siglip" else {}
        m = ViT(n_layers, d_model, n_heads, patch, **extra, **kwargs)
        if pretrained:
            if weights not in ("augreg", "siglip"):
                raise ValueError(f"Unsupported weights={weights}")
            _no_download("ViT.from_google")
        return m

    @staticmethod
    def from_facebook(model_tag: str, *, pretrained: bool = False, **kwargs) -> "ViT":
        """"S/16_deit3" (default), "S/16_dino", "S/14_dinov2" (default img_size 518)."""
        _, weights, (n_layers, d_model, n_heads), patch = ViT._parse(model_tag, "deit3")
        if weights in ("deit3", "dino"):
            kwargs.setdefault("img_size", 224)
        elif weights == "dinov2":
            kwargs.setdefault("img_size", 518)
        else:
            raise ValueError(f"Unsupported {weights}")
        m = ViT(n_layers, d_model, n_heads, patch, **kwargs)
        if pretrained:
            _no_download("ViT.from_facebook")
        return m


def _no_download(who: str):
    raise NotImplementedError(
        f"{who}(pretrained=True) needs a network download, which this build does not do; construct with "
        "pretrained=False and load a local checkpoint into the (reference-named) parameters.")
