"""T5 v1.1 / Flan-T5 / mT5 on MI355X: drop-in for /root/reference pytorch_models/text/t5.py (LayerNorm, GEGLU,
RelativePositionBias, T5Block, T5Encoder, T5Decoder, T5Model; same parameter names, so a t5x checkpoint converted the
reference's way loads unchanged).

    LayerNorm (no centring, no bias)  -> pm_rmsnorm
    self- / cross-attention           -> transformer.MHA (packed projections, pm_attention_bias_bf16 with the relative-
                                         position bias; the decoder's causal mask is the kernel's own, not a -1e10 table),
                                         residual add in the out_proj epilogue
    GEGLU + output projection         -> ONE pm_linear_bf16 over the packed [w; v] weight, pm_geglu, pm_linear_bf16 (+residual)
    token embedding, classifier       -> pm_embed_tokens (no positional term), pm_linear_bf16 (fp32 logits)

The bucket table of RelativePositionBias is index arithmetic on the host (as in the reference, t5.py:49-70), cached per
length; the (heads, L, L) bias is gathered from the parameter on the device.  Inference only; no CPU path."""
from __future__ import annotations

import math

import torch
from torch import Tensor, nn

from .._hip import ops
from ..transformer import MHA, Linear, _f32, _wb, derived, require_bf16_params


class LayerNorm(nn.Module):
    """x * rsqrt(mean(x^2) + eps) * weight (t5.py:15-25); statistics in fp32."""

    def __init__(self, dim: int, eps: float = 1e-5) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(dim))
        self.dim = dim
        self.eps = eps

    def forward(self, x: Tensor, out_dtype: torch.dtype | None = None) -> Tensor:
        if not x.is_cuda:
            raise RuntimeError("T5 LayerNorm: HIP devices only (no CPU path)")
        y = ops.rmsnorm(x.reshape(-1, x.shape[-1]), _f32(self, "g", self.weight), self.eps, out_dtype)
        return y.view(*x.shape)


class GEGLU(nn.Module):
    """gelu_tanh(w x) * (v x) (t5.py:29-38): one GEMM over the stacked weight, then the gate kernel."""

    def __init__(self, dim: int, mlp_dim: int) -> None:
        super().__init__()
        self.w = Linear(dim, mlp_dim, False)
        self.v = Linear(dim, mlp_dim, False)
        self.act = nn.GELU(approximate="tanh")

    def _packed(self) -> Tensor:
        return derived(self, "wv", (self.w.weight, self.v.weight),
                       lambda: torch.cat([self.w.weight.detach(), self.v.weight.detach()], 0).to(torch.bfloat16).contiguous())

    def forward(self, x: Tensor) -> Tensor:
        if not x.is_cuda:
            raise RuntimeError("GEGLU: HIP devices only (no CPU path)")
        xb = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
        h = ops.linear(xb.reshape(-1, x.shape[-1]), self._packed(), None)
        return ops.geglu(h).view(*x.shape[:-1], -1)


class RelativePositionBias(nn.Module):
    def __init__(self, n_heads: int, n_buckets: int = 32, max_distance: int = 128) -> None:
        super().__init__()
        self.n_buckets = n_buckets
        self.max_distance = max_distance
        self.bias = nn.Parameter(torch.zeros(n_heads, n_buckets))

    def buckets(self, length: int, bidirection: bool) -> Tensor:
        """(L, L) int64 bucket ids, key position minus query position binned: exact below n/2, logarithmic up to
        max_distance, clipped beyond; separate halves for the two directions when bidirectional (t5.py:49-70)."""
        idx = torch.arange(length)
        rel = idx[None, :] - idx[:, None]
        if bidirection:
            n = self.n_buckets // 2
            offset = (rel > 0).long() * n
            rel = rel.abs()
        else:
            n = self.n_buckets
            offset = torch.zeros_like(rel)
            rel = (-rel).clamp(min=0)
        exact = n // 2
        scale = (n - exact) / math.log(self.max_distance / exact)
        far = exact + (torch.log(rel / exact + torch.finfo(torch.float32).eps) * scale).long()
        far = far.clamp(max=n - 1)
        return torch.where(rel < exact, rel, far) + offset

    def forward(self, length: int, bidirection: bool) -> Tensor:
        cache = self.__dict__.setdefault("_pm_buckets", {})
        key = (length, bidirection, self.bias.device)
        if key not in cache:
            cache[key] = self.buckets(length, bidirection).to(self.bias.device)
        return self.bias[:, cache[key]]  # (heads, L, L)


class T5Block(nn.Module):
    def __init__(self, dim: int, n_heads: int, mlp_dim: int, dropout: float = 0.0, cross_attn: bool = False) -> None:
        super().__init__()
        self.sa_norm = LayerNorm(dim)
        self.sa = MHA(dim, n_heads=n_heads, head_dim=64, bias=False, dropout=dropout)
        self.ca_norm = LayerNorm(dim) if cross_attn else None
        self.ca = MHA(dim, n_heads=n_heads, head_dim=64, bias=False, dropout=dropout) if cross_attn else None
        self.mlp_norm = LayerNorm(dim)
        self.mlp = nn.Sequential(GEGLU(dim, mlp_dim), nn.Dropout(dropout), Linear(mlp_dim, dim, False), nn.Dropout(dropout))

    def forward(self, x: Tensor, memory: Tensor | None = None, attn_bias: Tensor | None = None, causal: bool = False) -> Tensor:
        x = self.sa.attend(self.sa_norm(x), attn_bias=attn_bias, causal=causal, residual=x)
        if self.ca is not None:
            x = self.ca.attend(self.ca_norm(x), memory, residual=x)
        g = self.mlp[0](self.mlp_norm(x))
        wo = self.mlp[2]
        y = ops.linear(g.reshape(-1, g.shape[-1]), _wb(wo, "w", wo.weight), None, resid=x.reshape(-1, x.shape[-1]), out_dtype=x.dtype)
        return y.view(*x.shape)


class T5Encoder(nn.Module):
    def __init__(self, dim: int, n_heads: int, n_layers: int, mlp_dim: int, dropout: float = 0.0) -> None:
        super().__init__()
        self.in_drop = nn.Dropout(dropout)
        self.attn_bias = RelativePositionBias(n_heads)
        self.layers = nn.Sequential(*[T5Block(dim, n_heads, mlp_dim, dropout, False) for _ in range(n_layers)])
        self.norm = LayerNorm(dim)
        self.out_drop = nn.Dropout(dropout)

    def forward(self, x: Tensor) -> Tensor:
        require_bf16_params(self, "T5Encoder")
        bias = self.attn_bias(x.shape[-2], bidirection=True)
        for layer in self.layers:
            x = layer(x, attn_bias=bias)
        return self.norm(x)


class T5Decoder(nn.Module):
    def __init__(self, dim: int, n_heads: int, n_layers: int, mlp_dim: int, dropout: float = 0.0) -> None:
        super().__init__()
        self.in_drop = nn.Dropout(dropout)
        self.attn_bias = RelativePositionBias(n_heads)
        self.layers = nn.Sequential(*[T5Block(dim, n_heads, mlp_dim, dropout, True) for _ in range(n_layers)])
        self.norm = LayerNorm(dim)
        self.out_drop = nn.Dropout(dropout)

    def forward(self, x: Tensor, memory: Tensor) -> Tensor:
        require_bf16_params(self, "T5Decoder")
        bias = self.attn_bias(x.shape[-2], bidirection=False)  # + the causal mask, applied inside the attention kernel
        for layer in self.layers:
            x = layer(x, memory, attn_bias=bias, causal=True)
        return self.norm(x)


_SIZES = dict(small=(512, 6, 8, 1024), base=(768, 12, 12, 2048), large=(1024, 16, 24, 2816), xl=(2048, 32, 24, 5120),
              xxl=(4096, 64, 24, 10240))


class T5Model(nn.Module):
    def __init__(self, vocab_size: int, dim: int, n_heads: int, n_layers: int, mlp_dim: int, dropout: float = 0.0) -> None:
        super().__init__()
        self.token_embs = nn.Embedding(vocab_size, dim)
        self.encoder = T5Encoder(dim, n_heads, n_layers, mlp_dim, dropout)
        self.decoder = T5Decoder(dim, n_heads, n_layers, mlp_dim, dropout)
        self.classifier = Linear(dim, vocab_size, False)

    def _embed(self, x: Tensor) -> Tensor:
        E = _wb(self.token_embs, "E", self.token_embs.weight)
        return ops.embed_tokens(x.reshape(-1, x.shape[-1]), E, None).view(*x.shape, E.shape[1])

    def encode(self, x: Tensor) -> Tensor:
        return self.encoder(self._embed(x))

    def decode(self, x: Tensor, memory: Tensor) -> Tensor:
        h = self.decoder(self._embed(x), memory)
        W = _wb(self.classifier, "w", self.classifier.weight)
        return ops.linear(h.reshape(-1, h.shape[-1]), W, None, out_dtype=torch.float32).view(*x.shape, W.shape[0])

    def forward(self, x: Tensor, targets: Tensor) -> Tensor:
        """token ids (..., S), (..., L) int64 -> logits (..., L, vocab) fp32 (t5.py:144-151)."""
        return self.decode(targets, self.encode(x))

    @torch.no_grad()
    def generate_ids(self, input_ids: Tensor, max_tokens: int = 100, pad_id: int = 0, eos_id: int = 1) -> Tensor:
        """The loop of the reference's T5Generator.generate (t5.py:213-227) on token ids, one sequence: start from the pad
        id, append the arg-max of the last position until eos or max_tokens; the encoder runs once."""
        memory = self.encode(input_ids.view(1, -1))
        out = [pad_id]
        while len(out) < max_tokens:
            logits = self.decode(torch.tensor([out], device=input_ids.device), memory)
            out.append(int(logits[0, -1].argmax()))
            if out[-1] == eos_id:
                break
        return torch.tensor(out, device=input_ids.device)

    @staticmethod
    def from_t5x(model_tag: str, *, pretrained: bool = False, **kwargs) -> "T5Model":
        variant, size = model_tag.split("-")
        dim, n_heads, n_layers, mlp_dim = _SIZES[size]
        vocab_size = 250112 if variant.startswith("mt5") else 32128
        m = T5Model(vocab_size, dim, n_heads, n_layers, mlp_dim, **kwargs)
        if pretrained:
            raise NotImplementedError(
                "T5Model.from_t5x(pretrained=True) streams a t5x checkpoint over HTTP in the reference; this build has no "
                "network: construct with pretrained=False and call load_t5x_checkpoint(dict of the flattened t5x tensors).")
        return m

    @torch.no_grad()
    def load_t5x_checkpoint(self, ckpt: dict) -> None:
        """Flattened t5x ``target`` tensors (name -> array, e.g. what the reference caches under checkpoints/) into this
        model: kernels are stored (in, out) and transposed here; the query / key kernels are scaled by 64^(1/4) each
        because T5 attention has no 1/sqrt(head_dim) while the attention kernel applies one (t5.py:169-176)."""
        sd = {}
        for k, v in ckpt.items():
            v = torch.as_tensor(v)
            if k.endswith("kernel"):
                v = v.T
            if k.endswith(("query.kernel", "key.kernel")):
                v = v * 64**0.25
            sd[_rename_key(k)] = v
        self.load_state_dict(sd)


_RENAMES = (
    ("token_embedder.embedding", "token_embs.weight"), ("decoder.logits_dense.kernel", "classifier.weight"),
    (".encoder_norm.scale", ".norm.weight"), (".decoder_norm.scale", ".norm.weight"),
    (".relpos_bias.rel_embedding", ".attn_bias.bias"), (".layers_", ".layers."),
    (".pre_attention_layer_norm.scale", ".sa_norm.weight"), (".pre_self_attention_layer_norm.scale", ".sa_norm.weight"),
    (".pre_cross_attention_layer_norm.scale", ".ca_norm.weight"), (".pre_mlp_layer_norm.scale", ".mlp_norm.weight"),
    (".attention.", ".sa."), (".self_attention.", ".sa."), (".encoder_decoder_attention.", ".ca."),
    (".query.kernel", ".q_proj.weight"), (".key.kernel", ".k_proj.weight"), (".value.kernel", ".v_proj.weight"),
    (".out.kernel", ".out_proj.weight"), (".wi_0.kernel", ".0.w.weight"), (".wi_1.kernel", ".0.v.weight"),
    (".wo.kernel", ".2.weight"),
)


def _rename_key(key: str) -> str:
    """t5x parameter path -> this module tree (the table of t5.py:230-252, applied in the same order)."""
    for old, new in _RENAMES:
        key = key.replace(old, new)
    return key
