"""Oracle for ViT.forward (test infrastructure, see oracle/__init__.py).

Restates /root/reference pytorch_models/image/vit.py:20-85 and the geometry tables of
``from_google`` / ``from_facebook`` (vit.py:96-119, 202-239).
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import Tensor

from . import ref_transformer as T

# (n_layers, d_model, n_heads) - vit.py:106-113 and :231-238 (identical tables)
SIZES = dict(Ti=(12, 192, 3), S=(12, 384, 6), M=(12, 512, 8), B=(12, 768, 12), L=(24, 1024, 16), H=(32, 1280, 16))
NORM_EPS = 1e-6  # ViT.norm_eps - vit.py:49


@dataclass
class ViTGeometry:
    n_layers: int
    d_model: int
    n_heads: int
    patch_size: int
    img_size: int = 224
    cls_token: bool = True
    pool_type: str = "cls_token"


def geometry_from_google(tag: str, **kw) -> ViTGeometry:
    """Tag parsing of ViT.from_google - vit.py:98-119: "B/16", "B/16_siglip", default weights augreg;
    siglip => no cls token, MAP-head pooling."""
    tag, weights = tag.split("_") if "_" in tag else (tag, "augreg")
    size, patch = tag.split("/")
    n_layers, d, h = SIZES[size]
    extra = dict(cls_token=False, pool_type="mha") if weights == "siglip" else {}
    return ViTGeometry(n_layers, d, h, int(patch), **extra, **kw)


def geometry_from_facebook(tag: str, **kw) -> ViTGeometry:
    """Tag parsing of ViT.from_facebook - vit.py:204-239: deit3 / dino default to 224, dinov2 to 518."""
    tag, weights = tag.split("_") if "_" in tag else (tag, "deit3")
    size, patch = tag.split("/")
    if weights in ("deit3", "dino"):
        kw.setdefault("img_size", 224)
    elif weights == "dinov2":
        kw.setdefault("img_size", 518)
    else:
        raise ValueError(f"Unsupported {weights}")
    n_layers, d, h = SIZES[size]
    return ViTGeometry(n_layers, d, h, int(patch), **kw)


def patch_embed(sd: dict, imgs: Tensor) -> Tensor:
    """Conv2d(3, d, P, stride P) then flatten(-2).transpose(-1,-2) - vit.py:64,78 - as the GEMM it is:
    (N*L, 3*P*P) x (3*P*P, d), patch order row-major over (H/P, W/P), K order (c, ph, pw)."""
    w = sd["patch_embed.weight"]  # (d, 3, P, P)
    d, c, P, _ = w.shape
    N, _, H, W = imgs.shape
    gh, gw = H // P, W // P
    x = imgs.reshape(N, c, gh, P, gw, P).permute(0, 2, 4, 1, 3, 5).reshape(N, gh * gw, c * P * P)
    return x @ w.reshape(d, -1).T + sd["patch_embed.bias"]


def tokens(sd: dict, imgs: Tensor) -> Tensor:
    """patch-embed + pe (+ cls) - vit.py:78-81.  The cls token is broadcast over the batch; for
    N == 1 this is exactly the reference's torch.cat, for N > 1 it is the stack of the reference's
    batch-1 results (the reference itself raises there: SURVEY.md finding F1)."""
    out = patch_embed(sd, imgs) + sd["pe"]
    if "cls_token" in sd:
        out = torch.cat([sd["cls_token"].expand(out.shape[0], -1, -1), out], dim=-2)
    return out


def mha_pooling(sd: dict, p: str, n_heads: int, x: Tensor, rp=None) -> Tensor:
    """MHAPooling.forward - vit.py:40-43: probe attends over all tokens, then x + mlp(norm(x))."""
    probe = sd[p + "probe"].expand(x.shape[0], -1, -1)
    y = T.mha(sd, p + "attn.", n_heads, probe, x, rp=rp).squeeze(1)
    return y + T.mlp(sd, p + "mlp.", T.layernorm(sd, p + "norm.", y, NORM_EPS), rp=rp)


def forward(sd: dict, geo: ViTGeometry, imgs: Tensor, rp=None) -> Tensor:
    """ViT.forward - vit.py:77-85."""
    x = tokens(sd, imgs)
    if rp is not None:
        x = rp("resid", x)
    x = T.encoder(sd, "layers.", geo.n_heads, x, eps=NORM_EPS, rp=rp)
    x = T.layernorm(sd, "norm.", x, NORM_EPS)
    if rp is not None:
        x = rp("ln", x)
    if geo.pool_type == "cls_token":
        return x[:, 0]  # ClassTokenPooling - vit.py:20-22
    if geo.pool_type == "gap":
        return x.mean(1)  # GlobalAveragePooling - vit.py:25-27
    if geo.pool_type == "mha":
        return mha_pooling(sd, "pooler.", geo.n_heads, x, rp=rp)
    raise KeyError(geo.pool_type)


def resize_pe(pe: Tensor, patch_size: int, size: int, mode: str = "bicubic") -> Tensor:
    """ViT.resize_pe - vit.py:87-94 (bicubic interpolation of the (g, g) grid of position vectors)."""
    old = int(pe.shape[1] ** 0.5)
    new = size // patch_size
    grid = pe.unflatten(1, (old, old)).permute(0, 3, 1, 2)
    grid = torch.nn.functional.interpolate(grid, (new, new), mode=mode)
    return grid.permute(0, 2, 3, 1).flatten(1, 2)
