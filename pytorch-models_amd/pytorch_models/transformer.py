"""Shared transformer blocks on MI355X: MHA, MLP, LayerNorm, Encoder/Decoder layers and stacks.

Drop-in for /root/reference pytorch_models/transformer.py (same class names, constructor and
forward signatures, child-module and parameter names - the weight converters index them), but the
forward is a short sequence of hand-written gfx950 kernels reached through the C ABI of
``libpm_mi355x.so``:

    LayerNorm            -> pm_layernorm            (wave-per-row, fp32 statistics)
    q/k/v projections    -> ONE pm_linear_bf16 over the concatenated weight (packed lazily)
    SDPA                 -> pm_attention_bf16, reading the packed projection in place
    out_proj + residual  -> pm_linear_bf16 with the residual add in its epilogue
    linear1 + GELU       -> pm_linear_bf16 with the activation in its epilogue
    linear2 + residual   -> pm_linear_bf16 with the residual add in its epilogue

There is no CPU or eager fallback: tensors must live on a HIP device; anything the kernels do not cover raises.
A module computes in the dtype of its parameters:
  * bf16 parameters (``model.to(torch.bfloat16)``, the throughput form): bf16 activations end to end, bf16 MFMA with fp32
    accumulation, fp32 LayerNorm / softmax statistics;
  * fp32 parameters (the reference's default): fp32 activations end to end on the exact-fp32 MFMA (pm_linear_f32,
    pm_attention_generic_f32, fp32 LayerNorm) - the reference's own tolerances hold (2e-5 / 5e-5:
    tests/test_hip_fp32_models.py) at roughly a tenth of the bf16 path's speed.  Nothing is silently rounded to bf16.
"""
from __future__ import annotations

import os

import torch
from torch import Tensor, nn

from . import _cpu
from ._hip import ops

_ACTS = {
    "gelu": lambda: nn.GELU(),
    "approximate_gelu": lambda: nn.GELU(approximate="tanh"),
    "relu": lambda: nn.ReLU(inplace=True),
    "silu": lambda: nn.SiLU(),
}


def _sig(ts) -> tuple:
    # (tensors created under torch.inference_mode() - e.g. the fp32 twin a generator builds - carry no version counter)
    return tuple((t.data_ptr(), 0 if t.is_inference() else t._version, t.dtype, t.device) for t in ts if t is not None)


def derived(module: nn.Module, key: str, params, build):
    """Value derived from parameters (packed / re-typed weights), rebuilt when any of them changes
    (in-place ``copy_`` bumps ``_version``; ``.to()`` moves storage)."""
    store = module.__dict__.setdefault("_pm_derived", {})
    sig = _sig(params)
    hit = store.get(key)
    if hit is None or hit[0] != sig:
        with torch.no_grad():
            hit = (sig, build())
        store[key] = hit
        # A derived tensor is built by kernels on whatever stream is current and then used from ANY stream (Encoder.forward
        # runs the halves of a large batch on two): wait for the build once, here, instead of ordering every later use.
        # (Not during a graph capture: captures are preceded by a warm-up that has built everything.)
        if torch.cuda.is_available() and any(isinstance(p, Tensor) and p.is_cuda for p in params) and \
                not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream().synchronize()
    return hit[1]


def _f32(module: nn.Module, key: str, t: Tensor | None) -> Tensor | None:
    if t is None:
        return None
    return derived(module, key, (t,), lambda: t.detach().float().contiguous())


_FLOATS = (torch.bfloat16, torch.float32)
_SIDE_STREAMS: dict = {}
_MIN_SPLIT_ROWS = int(os.environ.get("PM_ENCODER_MIN_ROWS", "12288"))  # smallest (batch x tokens) split over two streams: ViT-B/16 gains 5 % at batch 64, 11 % at 128, 6 % at 160, 3-5 % at 256; at batch 32 the doubled launch count makes the host the bottleneck (-39 %)
ENCODER_STREAMS = 0  # 0 = the instance / PM_ENCODER_STREAMS decide; 1 = one stream (bench.py's per-kernel timing pass sets it); 2 = two


def _side_stream(device: torch.device, k: int = 1) -> "torch.cuda.Stream":
    """Extra HIP stream k of a device (Encoder.forward runs the later parts of a large batch on them)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), k)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


def _is32(p: Tensor) -> bool:
    return p.dtype == torch.float32


def _require_bf16(x: Tensor, w: Tensor, who: str) -> None:
    """Device and dtype gate of every block: HIP tensors, bf16 or fp32 (each computed in its own dtype, see above)."""
    if not x.is_cuda or not w.is_cuda:
        raise RuntimeError(
            f"{who}: input on {x.device}, weights on {w.device}: the HIP kernels take operands on ONE HIP device (a module and "
            "its input both on the CPU take the plain-torch CPU forms instead).")
    if w.dtype not in _FLOATS or x.dtype not in _FLOATS:
        raise NotImplementedError(f"{who}: bf16 or fp32 only (weights {w.dtype}, input {x.dtype})")


def require_bf16_params(module: nn.Module, who: str) -> None:
    """Model families whose own kernels exist in bf16 only (the wav2vec2 stems / positional convolutions, T5's RMS norm,
    GEGLU and bias gather): an fp32 instance is refused instead of being computed through bf16 copies."""
    p = next(module.parameters(), None)
    if p is not None and not p.is_cuda:  # the device error first: a CPU model is refused for being on the CPU, whatever its dtype
        raise RuntimeError(f"{who}: HIP devices only - this build has no CPU path (move the model and its inputs to the GPU)")
    if p is not None and p.dtype != torch.bfloat16:
        raise NotImplementedError(
            f"{who}: bf16 parameters only on this build (got {p.dtype}); call model.to(torch.bfloat16).  The fp32-accurate path "
            "covers the shared transformer blocks, ViT, Whisper, GPT / GPT-2 and BERT.")


def _wb(module: nn.Module, key: str, p: Tensor) -> Tensor:
    """The parameter itself if it is bf16, else a cached bf16 copy (rebuilt when the parameter changes)."""
    if p.dtype == torch.bfloat16:
        return p
    return derived(module, key + ":bf16", (p,), lambda: p.detach().to(torch.bfloat16).contiguous())


def _xb(x: Tensor) -> Tensor:
    return x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)


def _like(y: Tensor, ref_dtype: torch.dtype) -> Tensor:
    return y if y.dtype == ref_dtype else y.to(ref_dtype)


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm whose forward is pm_layernorm (parameter names unchanged)."""

    def forward(self, x: Tensor, out_dtype: torch.dtype | None = None) -> Tensor:
        if _cpu.on_cpu(x, self.weight):  # CPU tensors, CPU parameters: plain torch (pytorch_models/_cpu.py)
            y = torch.nn.functional.layer_norm(x.to(self.weight.dtype), self.normalized_shape, self.weight, self.bias, self.eps)
            return y if out_dtype is None else y.to(out_dtype)
        g = _f32(self, "g", self.weight)
        b = _f32(self, "b", self.bias)
        y = ops.layernorm(x.reshape(-1, x.shape[-1]), g, b, self.eps, out_dtype)
        return y.view(*x.shape)


class Linear(nn.Linear):
    """nn.Linear whose forward is pm_linear_bf16 (parameter names unchanged)."""

    def forward(self, x: Tensor) -> Tensor:
        if _cpu.on_cpu(x, self.weight):
            return _cpu.linear(x, self.weight, self.bias)
        _require_bf16(x, self.weight, "Linear")
        if _is32(self.weight):  # fp32 parameters: fp32 arithmetic
            y = ops.linear_f32(x.float().reshape(-1, x.shape[-1]), self.weight, self.bias)
            return y.view(*x.shape[:-1], self.out_features)
        y = ops.linear(_xb(x).reshape(-1, x.shape[-1]), _wb(self, "w", self.weight), _f32(self, "b", self.bias), out_dtype=x.dtype)
        return y.view(*x.shape[:-1], self.out_features)


class MHA(nn.Module):
    def __init__(
        self,
        d_model: int,
        n_heads: int | None = None,
        head_dim: int | None = None,
        bias: bool = True,
        dropout: float = 0.0,
    ) -> None:
        # same resolution order as the reference (transformer.py:18-26): default head_dim 64
        if n_heads is None and head_dim is None:
            head_dim = 64
        if head_dim is None:
            head_dim = d_model // n_heads
        if n_heads is None:
            n_heads = d_model // head_dim
        super().__init__()
        inner = n_heads * head_dim
        self.q_proj = Linear(d_model, inner, bias)
        self.k_proj = Linear(d_model, inner, bias)
        self.v_proj = Linear(d_model, inner, bias)
        self.out_proj = Linear(inner, d_model, bias)
        self.n_heads = n_heads
        self.head_dim = head_dim
        self.dropout = dropout

    # ---- packed projection weights (derived lazily from the four nn.Linear)
    def _pack(self, names: str):
        mods = [getattr(self, f"{n}_proj") for n in names]
        params = [m.weight for m in mods] + [m.bias for m in mods]

        def build():
            w = torch.cat([m.weight.detach() for m in mods], 0).to(torch.bfloat16).contiguous()
            b = None if mods[0].bias is None else torch.cat([m.bias.detach().float() for m in mods], 0).contiguous()
            return w, b

        return derived(self, "pack_" + names, params, build)

    def _pack32(self, names: str):
        """fp32 modules: the concatenated fp32 weight / bias of the named projections."""
        mods = [getattr(self, f"{n}_proj") for n in names]
        params = [m.weight for m in mods] + [m.bias for m in mods]

        def build():
            w = torch.cat([m.weight.detach().float() for m in mods], 0).contiguous()
            b = None if mods[0].bias is None else torch.cat([m.bias.detach().float() for m in mods], 0).contiguous()
            return w, b

        return derived(self, "pack32_" + names, params, build)

    def _pack_ln(self, norm: "LayerNorm"):
        """q/k/v projection with ``norm`` folded in: (w' = bf16(gamma (.) w), s[n] = sum_k w'[n][k],
        c[n] = b[n] + sum_k beta[k] w[n][k]) so that proj(norm(x)) = rstd * (x w'^T - mean * s) + c."""
        w, b = self._pack("qkv")
        return derived(self, "pack_qkv_ln", (w, b, norm.weight, norm.bias), lambda: _fold_ln(w, b, norm))

    def forward(
        self,
        q: Tensor,
        k: Tensor | None = None,
        v: Tensor | None = None,
        attn_bias: Tensor | None = None,
        causal: bool = False,
    ) -> Tensor:
        return self.attend(q, k, v, attn_bias, causal)

    def attend(self, q, k=None, v=None, attn_bias=None, causal=False, residual: Tensor | None = None) -> Tensor:
        """forward() plus an optional residual that is added inside the out_proj kernel's epilogue."""
        if _cpu.on_cpu(q, self.q_proj.weight):
            return _cpu.mha(self, q, k, v, attn_bias, causal, residual)
        _require_bf16(q, self.q_proj.weight, "MHA")
        if self.head_dim % 2 or self.head_dim > 128 or (self.head_dim % 4 and self.head_dim > 64):
            raise NotImplementedError(f"MHA: head_dim {self.head_dim} is not covered by the gfx950 attention kernels (% 4 up to 128, % 2 up to 64)")
        if self.training and self.dropout > 0.0:
            raise NotImplementedError("MHA: inference only (attention dropout is not implemented)")
        H, inner = self.n_heads, self.n_heads * self.head_dim
        Lq = q.shape[-2]
        if _is32(self.q_proj.weight):
            return self._attend_f32(q, k, v, attn_bias, causal, residual)
        io_dtype = q.dtype
        q = _xb(q)
        k = None if k is None else _xb(k)
        v = None if v is None else _xb(v)
        if k is None and v is None:  # self-attention: one projection over the concatenated q/k/v weight
            lead = q.shape[:-2]
            w, b = self._pack("qkv")
            qkv = ops.linear(q.reshape(-1, q.shape[-1]), w, b).view(-1, Lq, 3 * inner)
            qh, kh, vh = qkv[..., :inner], qkv[..., inner : 2 * inner], qkv[..., 2 * inner :]
        else:
            k = q if k is None else k
            Lk = k.shape[-2]
            lead = torch.broadcast_shapes(q.shape[:-2], k.shape[:-2])  # e.g. a (1, 1, d) probe over (N, L, d)
            qh = self.q_proj(q)
            if v is None or v is k:
                w, b = self._pack("kv")
                kv = ops.linear(k.reshape(-1, k.shape[-1]), w, b).view(*k.shape[:-1], 2 * inner)
                kh, vh = kv[..., :inner], kv[..., inner:]
            else:
                kh, vh = self.k_proj(k), self.v_proj(v)
            # broadcast AFTER projecting: an expanded operand is a stride-0 view, never a copy of the projection
            qh = qh.expand(*lead, Lq, inner).reshape(-1, Lq, inner)
            kh = kh.expand(*lead, Lk, inner).reshape(-1, Lk, inner)
            vh = vh.expand(*lead, Lk, inner).reshape(-1, Lk, inner)
        bias4 = None if attn_bias is None else _bias4(attn_bias, lead, Lq, kh.shape[1])
        o = ops.attention(qh, kh, vh, H, causal, bias4)
        res2 = None if residual is None else residual.reshape(-1, residual.shape[-1])
        y = ops.linear(o.view(-1, inner), _wb(self.out_proj, "w", self.out_proj.weight), _f32(self.out_proj, "b", self.out_proj.bias),
                       resid=res2, out_dtype=io_dtype)
        return y.view(*lead, Lq, y.shape[-1])


def _bias4(attn_bias: Tensor, lead, Lq: int, Lk: int) -> Tensor:
    """additive float bias or boolean keep-mask, broadcastable to (*lead, H, Lq, Lk) -> f32 (b, h, Lq, Lk) with unit key
    stride and a non-zero query stride (batch / head strides may be 0)."""
    ab = attn_bias
    if ab.dtype == torch.bool:
        ab = torch.zeros_like(ab, dtype=torch.float32).masked_fill_(~ab, float("-inf"))
    ab = ab.float()
    while ab.dim() < 4:
        ab = ab.unsqueeze(0)
    if ab.dim() > 4:  # several leading batch dims: flatten them like q
        ab = ab.expand(*lead, *ab.shape[-3:]).reshape(-1, *ab.shape[-3:])
    bias4 = ab.expand(ab.shape[0], ab.shape[1], Lq, Lk)
    if bias4.stride(3) != 1 or (Lq > 1 and bias4.stride(2) == 0):
        bias4 = bias4.contiguous()
    return bias4


def _attend_f32(self: MHA, q, k, v, attn_bias, causal, residual):
    """MHA.attend for fp32 parameters: the same call forms, fp32 activations and arithmetic throughout."""
    H, inner = self.n_heads, self.n_heads * self.head_dim
    Lq = q.shape[-2]
    q = q.float()
    k = None if k is None else k.float()
    v = None if v is None else v.float()
    if k is None and v is None:
        lead = q.shape[:-2]
        w, b = self._pack32("qkv")
        qkv = ops.linear_f32(q.reshape(-1, q.shape[-1]), w, b).view(-1, Lq, 3 * inner)
        qh, kh, vh = qkv[..., :inner], qkv[..., inner : 2 * inner], qkv[..., 2 * inner :]
    else:
        k = q if k is None else k
        Lk = k.shape[-2]
        lead = torch.broadcast_shapes(q.shape[:-2], k.shape[:-2])
        qh = self.q_proj(q)
        if v is None or v is k:
            w, b = self._pack32("kv")
            kv = ops.linear_f32(k.reshape(-1, k.shape[-1]), w, b).view(*k.shape[:-1], 2 * inner)
            kh, vh = kv[..., :inner], kv[..., inner:]
        else:
            kh, vh = self.k_proj(k), self.v_proj(v)
        qh = qh.expand(*lead, Lq, inner).reshape(-1, Lq, inner)
        kh = kh.expand(*lead, Lk, inner).reshape(-1, Lk, inner)
        vh = vh.expand(*lead, Lk, inner).reshape(-1, Lk, inner)
    bias4 = None if attn_bias is None else _bias4(attn_bias, lead, Lq, kh.shape[1])
    o = ops.attention_f32(qh, kh, vh, H, causal, bias4)
    res2 = None if residual is None else residual.float().reshape(-1, residual.shape[-1])
    y = ops.linear_f32(o.view(-1, inner), self.out_proj.weight, self.out_proj.bias, resid=res2)
    return y.view(*lead, Lq, y.shape[-1])


MHA._attend_f32 = _attend_f32


def _fold_ln(w: Tensor, b: Tensor | None, norm: "LayerNorm"):
    wf = w.detach().float()
    wl = (wf * norm.weight.detach().float()[None, :]).to(torch.bfloat16).contiguous()
    s = wl.float().sum(1).contiguous()
    c = wf @ norm.bias.detach().float()
    if b is not None:
        c = c + b.float()
    return wl, s, c.contiguous()


class MLP(nn.Module):
    """linear1 -> act -> linear2 -> dropout (child names as in the reference's nn.Sequential, transformer.py:56-67)."""

    def __init__(self, in_dim: int, hidden_dim: float, dropout: float = 0.0, act: str = "gelu") -> None:
        super().__init__()
        self.linear1 = Linear(in_dim, hidden_dim)
        self.act = _ACTS[act]()
        self.linear2 = Linear(hidden_dim, in_dim)
        self.dropout = nn.Dropout(dropout)
        self.act_name = act

    def forward(self, x: Tensor) -> Tensor:
        return self.run(x)

    def run(self, x: Tensor, residual: Tensor | None = None) -> Tensor:
        if _cpu.on_cpu(x, self.linear1.weight):
            return _cpu.mlp(self, x, residual)
        _require_bf16(x, self.linear1.weight, "MLP")
        if self.training and self.dropout.p > 0.0:
            raise NotImplementedError("MLP: inference only (dropout is not implemented)")
        if _is32(self.linear1.weight):
            h = ops.linear_f32(x.float().reshape(-1, x.shape[-1]), self.linear1.weight, self.linear1.bias, act=self.act_name)
            res2 = None if residual is None else residual.float().reshape(-1, residual.shape[-1])
            return ops.linear_f32(h, self.linear2.weight, self.linear2.bias, resid=res2).view(*x.shape)
        x2 = _xb(x).reshape(-1, x.shape[-1])
        h = ops.linear(x2, _wb(self.linear1, "w", self.linear1.weight), _f32(self.linear1, "b", self.linear1.bias), act=self.act_name)
        res2 = None if residual is None else residual.reshape(-1, residual.shape[-1])
        y = ops.linear(h, _wb(self.linear2, "w", self.linear2.weight), _f32(self.linear2, "b", self.linear2.bias), resid=res2,
                       out_dtype=x.dtype)
        return y.view(*x.shape)


def _fused_attend(mha: MHA, x: Tensor, kv: Tensor | None, causal: bool, residual: Tensor) -> Tensor:
    if type(mha).forward is MHA.forward:  # plain MHA: residual add rides in the out_proj epilogue
        return mha.attend(x, kv, None, None, causal, residual=residual)
    return residual + (mha(x, kv, causal=causal) if kv is not None else mha(x, causal=causal))  # subclassed MHA


def _fused_mlp(mlp: MLP, x: Tensor, residual: Tensor) -> Tensor:
    if type(mlp).forward is MLP.forward:
        return mlp.run(x, residual=residual)
    return residual + mlp(x)


class DecoderLayer(nn.Module):
    def __init__(
        self,
        d_model: int,
        n_heads: int | None = None,
        head_dim: int | None = None,
        cross_attn: bool = False,
        bias: bool = True,
        mlp_ratio: float = 4.0,
        dropout: float = 0.0,
        act: str = "gelu",
        pre_norm: bool = True,
        norm_eps: float = 1e-5,
    ) -> None:
        super().__init__()
        self.pre_norm = pre_norm
        self.sa_norm = LayerNorm(d_model, norm_eps)
        self.sa = MHA(d_model, n_heads, head_dim, bias, dropout)
        self.ca_norm = LayerNorm(d_model, norm_eps) if cross_attn else None
        self.ca = MHA(d_model, n_heads, head_dim, bias, dropout) if cross_attn else None
        self.mlp_norm = LayerNorm(d_model, norm_eps)
        self.mlp = MLP(d_model, int(d_model * mlp_ratio), dropout, act)

    _self_causal = True

    def forward(self, x: Tensor, memory: Tensor | None = None) -> Tensor:
        c = self._self_causal
        if self.pre_norm:
            x = _fused_attend(self.sa, self.sa_norm(x), None, c, x)
            if self.ca is not None:
                x = _fused_attend(self.ca, self.ca_norm(x), memory, False, x)
            x = _fused_mlp(self.mlp, self.mlp_norm(x), x)
        else:
            x = self.sa_norm(_fused_attend(self.sa, x, None, c, x))
            if self.ca is not None:
                x = self.ca_norm(_fused_attend(self.ca, x, memory, False, x))
            x = self.mlp_norm(_fused_mlp(self.mlp, x, x))
        return x


class EncoderLayer(DecoderLayer):
    _self_causal = False

    def __init__(
        self,
        d_model: int,
        n_heads: int | None = None,
        head_dim: int | None = None,
        bias: bool = True,
        mlp_ratio: float = 4.0,
        dropout: float = 0.0,
        act: str = "gelu",
        pre_norm: bool = True,
        norm_eps: float = 1e-5,
    ) -> None:
        super().__init__(d_model, n_heads, head_dim, False, bias, mlp_ratio, dropout, act, pre_norm, norm_eps)

    def forward(self, x: Tensor) -> Tensor:
        return DecoderLayer.forward(self, x, None)

    # ---- LayerNorm folded into the GEMMs on either side of it (Encoder.forward drives this)
    def chain_ok(self, x: Tensor) -> bool:
        """True when this layer can run with both LayerNorms folded: plain pre-norm bf16 layer, shapes the
        persistent GEMMs serve (pm_linear_ln_supported)."""
        sa, mlp = self.sa, self.mlp
        if not (self.pre_norm and type(sa).forward is MHA.forward and type(mlp).forward is MLP.forward and x.is_cuda
                and x.dtype == torch.bfloat16 and sa.q_proj.weight.dtype == torch.bfloat16 and not self.training):
            return False
        M, d = x.numel() // x.shape[-1], x.shape[-1]
        inner, hid = sa.n_heads * sa.head_dim, mlp.linear1.out_features
        if sa.head_dim % 8 or sa.head_dim > 128 or mlp.act_name not in ("gelu",):
            return False
        return (ops.linear_ln_supported(M, d, inner, "none", True) and ops.linear_ln_supported(M, d, hid, "none", True)
                and ops.linear_ln_supported(M, 3 * inner, d, "none", False)
                and ops.linear_ln_supported(M, hid, d, mlp.act_name, False))

    def forward_chain(self, x: Tensor, stats: Tensor | None, next_eps: float | None):
        """x + sa(sa_norm(x)), then x + mlp(mlp_norm(x)) with the LayerNorms folded.  ``stats`` = (mean, rstd) of
        the rows of ``x`` under sa_norm (None: run sa_norm as its own kernel); returns (x, stats of the output
        rows under the next layer's sa_norm with ``next_eps``, or None)."""
        sa, mlp = self.sa, self.mlp
        lead, d = x.shape[:-1], x.shape[-1]
        H, inner = sa.n_heads, sa.n_heads * sa.head_dim
        x2 = x.reshape(-1, d)
        if stats is None:
            w, b = sa._pack("qkv")
            qkv = ops.linear(self.sa_norm(x2), w, b)
        else:
            wl, s, c = sa._pack_ln(self.sa_norm)
            qkv = ops.linear(x2, wl, c, ln_stats=stats, ln_s=s)
        qkv = qkv.view(-1, x.shape[-2], 3 * inner)
        o = ops.attention(qkv[..., :inner], qkv[..., inner : 2 * inner], qkv[..., 2 * inner :], H, False, None)
        x2, rows = ops.linear(o.view(-1, inner), sa.out_proj.weight, _f32(sa.out_proj, "b", sa.out_proj.bias), resid=x2,
                              want_row_stats=True)
        st = ops.ln_stats_finalize(rows, d, self.mlp_norm.eps)
        l1, l2 = mlp.linear1, mlp.linear2
        wl, s, c = derived(mlp, "l1_ln", (l1.weight, l1.bias, self.mlp_norm.weight, self.mlp_norm.bias),
                           lambda: _fold_ln(l1.weight, l1.bias, self.mlp_norm))
        h = ops.linear(x2, wl, c, act=mlp.act_name, ln_stats=st, ln_s=s)
        if next_eps is None:
            return ops.linear(h, l2.weight, _f32(l2, "b", l2.bias), resid=x2).view(*lead, d), None
        x2, rows = ops.linear(h, l2.weight, _f32(l2, "b", l2.bias), resid=x2, want_row_stats=True)
        return x2.view(*lead, d), ops.ln_stats_finalize(rows, d, next_eps)


class Encoder(nn.Sequential):
    def __init__(
        self,
        n_layers: int,
        d_model: int,
        n_heads: int | None = None,
        head_dim: int | None = None,
        bias: bool = True,
        mlp_ratio: float = 4.0,
        dropout: float = 0.0,
        act: str = "gelu",
        pre_norm: bool = True,
        norm_eps: float = 1e-5,
    ) -> None:
        super().__init__(*[
            EncoderLayer(d_model, n_heads, head_dim, bias, mlp_ratio, dropout, act, pre_norm, norm_eps)
            for _ in range(n_layers)
        ])

    def forward(self, x: Tensor | None = None, *, producers=None, device: torch.device | None = None) -> Tensor:
        """The layers in order.  Runs of plain pre-norm layers at GEMM-sized M are chained: each residual GEMM
        (out_proj, linear2) also emits the row statistics of its output, and the next GEMM (q/k/v, linear1)
        applies the LayerNorm in its epilogue - no LayerNorm kernel, no LN(x) round trip through HBM.
        ``producers`` (instead of x): one callable per batch part (split_sizes), each called on its part's stream."""
        layers = list(self)
        fold = os.environ.get("PM_LN_FOLD", "1") != "0"

        def plan(t: Tensor):
            return [fold and isinstance(l, EncoderLayer) and type(l).forward is EncoderLayer.forward and l.chain_ok(t) for l in layers]

        def step(i: int, ok, t: Tensor, stats):
            if not ok[i]:
                return layers[i](t), None
            nxt = layers[i + 1].sa_norm.eps if i + 1 < len(layers) and ok[i + 1] else None
            return layers[i].forward_chain(t, stats, nxt)

        halves = self._two_streams(x) if producers is None else None
        if halves is not None and any(plan(h) != plan(halves[0]) for h in halves[1:]):
            halves = None  # the parts would take different paths (LayerNorm folded / not): keep the batch in one piece
        if halves is None and producers is None:
            ok, stats = plan(x), None
            for i in range(len(layers)):
                x, stats = step(i, ok, x, stats)
            return x
        # Two halves of the batch on two HIP streams, layer by layer.  Samples are independent and every kernel is
        # batch-position invariant, so the result is bit-identical; what changes is the schedule: the large-M kernels are
        # persistent (one workgroup per CU) and their last round of tiles leaves most CUs idle - 2.31 rounds run as 3 at
        # ViT-B/16's N = 768 -, and the other half's workgroups start on exactly those CUs.  The vendor library evens such rounds by
        # cutting tiles along K; this evens them across two kernels with whole tiles.
        n_parts = len(halves) if producers is None else len(producers)
        device = halves[0].device if producers is None else device
        cur = torch.cuda.current_stream(device)
        streams = [cur] + [_side_stream(device, k) for k in range(1, n_parts)]
        # The result is allocated on the caller's stream BEFORE the fork: a block the caller's stream freed earlier can
        # then only be handed out while everything the side streams do is still ordered behind it by the wait below; every
        # tensor a side stream allocates also dies on it, so the caching allocator needs no cross-stream bookkeeping.
        out = None
        if producers is None:
            out = torch.empty_like(x)
        elif getattr(self, "_pm_out_shape", None) is not None:
            shape, dtype = self._pm_out_shape
            out = torch.empty(shape, dtype=dtype, device=device)
        for st in streams[1:]:
            st.wait_stream(cur)
        if producers is None:
            state = [(h, None) for h in halves]
        else:  # each part's input is made on the part's stream (ViT: the patch projection of its images)
            state = []
            for k in range(n_parts):
                with torch.cuda.stream(streams[k]):
                    state.append((producers[k](), None))
        oks = [plan(t) for t, _ in state]
        first = [0]
        for t, _ in state:
            first.append(first[-1] + t.shape[0])
        want = ((first[-1],) + tuple(state[0][0].shape[1:]), state[0][0].dtype)
        if out is None or (tuple(out.shape), out.dtype) != want:  # (producers: the shape is known once the first part exists)
            out = torch.empty(want[0], dtype=want[1], device=device)
            for st in streams[1:]:  # allocated behind the fork this once: order the side streams behind the allocation
                st.wait_stream(cur)
        if producers is not None:
            self._pm_out_shape = want  # the next call with this geometry allocates in front of the fork
        if any(o != oks[0] for o in oks[1:]):  # (producers only; rare) different paths per part: one piece on the caller's stream
            for k in range(n_parts):
                with torch.cuda.stream(streams[k]):
                    out[first[k] : first[k + 1]].copy_(state[k][0])
            for st in streams[1:]:
                cur.wait_stream(st)
            return self.forward(out)
        for i in range(len(layers)):
            for k in reversed(range(n_parts)):
                with torch.cuda.stream(streams[k]):
                    state[k] = step(i, oks[k], state[k][0], state[k][1])
                    if i == len(layers) - 1:
                        out[first[k] : first[k + 1]].copy_(state[k][0])
                        state[k] = None
        for st in streams[1:]:
            cur.wait_stream(st)
        return out

    def split_sizes(self, batch: int, tokens: int, dtype: torch.dtype, device: torch.device):
        """[(lo, hi), ...] = the batch ranges forward() would run on separate streams for a (batch, tokens, d) input, or None."""
        want = ENCODER_STREAMS or int(os.environ.get("PM_ENCODER_STREAMS", "0")) or self.pm_streams
        if want < 2 or device.type != "cuda" or batch < 2 or dtype != torch.bfloat16 or batch * tokens < _MIN_SPLIT_ROWS:
            return None
        want = min(want, 4, batch)
        cuts = [batch * k // want for k in range(want + 1)]
        return [(cuts[k], cuts[k + 1]) for k in range(want)]

    pm_streams = 1  # 2 on instances whose owner opts in (ViT): see _two_streams

    def _two_streams(self, x: Tensor):
        """The two halves of a batch that is worth splitting - a 3-D (batch, tokens, d) bf16 input on a HIP device with at least
        _MIN_SPLIT_ROWS rows, where the persistent GEMM / attention kernels run - or None.  Opt-in per instance (`pm_streams = 2`: ViT sets
        it; PM_ENCODER_STREAMS=2 sets it everywhere, =1 or the module switch ENCODER_STREAMS = 1 nowhere).  Not the default for
        every encoder: with HIP-graph replays queued behind it on the caller's stream - Whisper's decode steps, when the host runs
        ahead of the GPU - the fork's cross-stream wait makes every one of those replays slower (measured: +9 ms per 227-replay
        step against 0.15 ms gained in the encoder; with a host synchronisation per step, or no graphs, it costs nothing)."""
        if x.dim() != 3:
            return None
        cuts = self.split_sizes(x.shape[0], x.shape[1], x.dtype, x.device)
        return None if cuts is None else [x[lo:hi] for lo, hi in cuts]


class Decoder(nn.ModuleList):
    def __init__(
        self,
        n_layers: int,
        d_model: int,
        n_heads: int | None = None,
        head_dim: int | None = None,
        cross_attn: bool = False,
        bias: bool = True,
        mlp_ratio: float = 4.0,
        dropout: float = 0.0,
        act: str = "gelu",
        pre_norm: bool = True,
        norm_eps: float = 1e-5,
    ) -> None:
        super().__init__([
            DecoderLayer(d_model, n_heads, head_dim, cross_attn, bias, mlp_ratio, dropout, act, pre_norm, norm_eps)
            for _ in range(n_layers)
        ])

    def forward(self, x: Tensor, memory: Tensor | None = None) -> Tensor:
        for layer in self:
            x = layer(x, memory)
        return x
