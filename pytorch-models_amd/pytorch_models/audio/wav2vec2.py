"""Wav2Vec2 / HuBERT on MI355X, drop-in for /root/reference pytorch_models/audio/wav2vec2.py (FeatureEncoder,
LayerNorm1d, Wav2Vec2, load_hf_state_dict; same child-module and parameter names: feature_encoder.{i}.0 / .2, proj.0 /
proj.1, pe_conv.1, layers, norm).

Data layout (see csrc/wav2vec2.hip): every activation is (clip, time, channel) bf16, so the reference's transposes
never happen and each piece lands on a kernel that already exists for the transformer path:

    feature_encoder[0]      -> pm_w2v_stem0          conv(1 -> C0, k 10, stride 5) + norm + GELU in one pass
    feature_encoder[i >= 1] -> pm_linear_bf16_ex     a strided window of k*C contiguous values per output step,
                                                     GELU in the epilogue (legacy stem: no norm after layer 0) or
                                                     followed by pm_layernorm_ex (channel LayerNorm + GELU)
    proj                    -> pm_layernorm + pm_linear_bf16
    pe_conv (16 groups)     -> pm_group_windows + pm_grouped_conv_bf16 (a tile's distinct input rows resident in LDS, the
                               group's weight streamed; bias, GELU and the "+ x" residual in its epilogue)
    layers, norm            -> transformer.Encoder / LayerNorm

No CPU path, inference only (dropout is the identity).
"""
from __future__ import annotations

import json

import torch
from torch import Tensor, nn

from .._hip import ops
from ..transformer import Encoder, LayerNorm, Linear, _f32, derived, require_bf16_params


class LayerNorm1d(nn.LayerNorm):
    """LayerNorm over the channel dim of (B, C, T) - wav2vec2.py:14-16.  Used time-major by the fused paths; the
    stand-alone forward keeps the reference's interface."""

    def forward(self, x: Tensor) -> Tensor:
        xt = x.transpose(1, 2).contiguous()
        y = ops.layernorm(xt.view(-1, xt.shape[-1]), _f32(self, "g", self.weight), _f32(self, "b", self.bias), self.eps)
        return y.view(xt.shape).transpose(1, 2)


def _norm_params(norm: nn.Module):
    if isinstance(norm, nn.Identity) or getattr(norm, "weight", None) is None:
        return None, None
    return _f32(norm, "g", norm.weight), _f32(norm, "b", norm.bias)


class FeatureEncoder(nn.Sequential):
    def __init__(self, dims, kernels, strides, bias: bool = True, dropout: float = 0.0, legacy: bool = False) -> None:
        super().__init__()
        in_dim = 1
        for i, (out_dim, kernel, stride) in enumerate(zip(dims, kernels, strides)):
            conv = nn.Conv1d(in_dim, out_dim, kernel, stride, bias=bias)
            if legacy:
                norm = nn.InstanceNorm1d(out_dim, affine=True) if i == 0 else nn.Identity()
            else:
                norm = LayerNorm1d(out_dim)
            self.append(nn.Sequential(conv, nn.Dropout(dropout), norm, nn.GELU()))
            in_dim = out_dim

    def _conv_weight(self, conv: nn.Conv1d):
        def build():  # (C', C, k) -> (C', k*C) bf16, K ordered (tap, channel) = the time-major window
            w = conv.weight.detach().permute(0, 2, 1).reshape(conv.out_channels, -1).to(torch.bfloat16).contiguous()
            return w, None if conv.bias is None else conv.bias.detach().float().contiguous()

        return derived(conv, "gemm", (conv.weight, conv.bias), build)

    def time_major(self, x: Tensor) -> Tensor:
        """waveform (B, L) -> bf16 (B, T, C_last)."""
        if not x.is_cuda:
            raise RuntimeError("FeatureEncoder: HIP devices only (no CPU path)")
        if x.dim() == 3:
            x = x.squeeze(1)
        conv0, norm0 = self[0][0], self[0][2]
        if conv0.in_channels != 1:
            raise NotImplementedError("FeatureEncoder: the first conv must read a mono waveform")
        w0 = _f32(conv0, "w0", conv0.weight).view(conv0.out_channels, -1)
        b0 = _f32(conv0, "b0", conv0.bias)
        g0, be0 = _norm_params(norm0)
        kind = "instance" if isinstance(norm0, nn.InstanceNorm1d) else ("layer" if g0 is not None else "none")
        h = ops.w2v_stem0(x.float().contiguous(), w0, b0, kind, g0, be0, getattr(norm0, "eps", 0.0), conv0.stride[0])
        for blk in list(self)[1:]:
            conv, norm = blk[0], blk[2]
            w, b = self._conv_weight(conv)
            B, T, C = h.shape
            k, s = conv.kernel_size[0], conv.stride[0]
            if T < k:
                raise ValueError(f"FeatureEncoder: {T} steps left for a kernel of {k} (waveform too short)")
            To = (T - k) // s + 1
            g, be = _norm_params(norm)
            y = ops.linear_strided(h, M=B * To, K=k * C, row_stride=s * C, rows_per_batch=To, batch_stride=T * C, w=w, bias=b,
                                   act="gelu" if g is None else "none")
            if g is not None:
                y = ops.layernorm(y, g, be, norm.eps, act="gelu")
            h = y.view(B, To, -1)
        return h

    def forward(self, x: Tensor) -> Tensor:
        """(B, 1, L) -> (B, C, T) like the reference's nn.Sequential (a transposed view of the time-major result)."""
        return self.time_major(x).transpose(1, 2)


class Wav2Vec2(nn.Module):
    STEM_DIMS = (512,) * 7
    STEM_KERNELS = (10,) + (3,) * 4 + (2,) * 2
    STEM_STRIDES = (5,) + (2,) * 6

    PE_KERNEL = 128
    PE_GROUPS = 16

    def __init__(self, n_layers: int, d_model: int, stem_bias: bool = True, stem_legacy: bool = False, dropout: float = 0.0,
                 pre_norm: bool = True) -> None:
        super().__init__()
        self._front(d_model, stem_bias, stem_legacy, dropout)
        self.pe_conv = nn.Sequential(
            nn.ConstantPad1d((self.PE_KERNEL // 2, self.PE_KERNEL // 2 - 1), 0),  # "same" padding for an even kernel
            nn.Conv1d(d_model, d_model, self.PE_KERNEL, groups=self.PE_GROUPS),
            nn.GELU(),
        )
        self.layers = Encoder(n_layers, d_model, dropout=dropout, pre_norm=pre_norm)
        self.norm = LayerNorm(d_model)
        self.pre_norm = pre_norm

    def _front(self, d_model: int, stem_bias: bool, stem_legacy: bool, dropout: float) -> None:
        """feature_encoder + proj, shared by the three model classes: conv stem over the waveform, LayerNorm over its channels
        and - only when the widths differ - a Linear onto d_model (wav2vec2.py:62-68)."""
        self.feature_encoder = FeatureEncoder(self.STEM_DIMS, self.STEM_KERNELS, self.STEM_STRIDES, stem_bias, dropout, stem_legacy)
        stem_out = self.STEM_DIMS[-1]
        self.proj = nn.Sequential(LayerNorm(stem_out), *([Linear(stem_out, d_model)] if stem_out != d_model else []))

    # ---- grouped positional conv: pm_group_windows + pm_grouped_conv_bf16 (input span resident in LDS)
    @staticmethod
    def _group_weights(conv: nn.Conv1d):
        def build():
            d, cg, k = conv.weight.shape
            G = conv.groups
            cgp = (cg + 7) // 8 * 8
            Kp = (k * cgp + 63) // 64 * 64
            w = torch.zeros(G, d // G, Kp, dtype=torch.bfloat16, device=conv.weight.device)  # K order (tap, channel), zero tail
            wk = torch.zeros(G, d // G, k, cgp, dtype=torch.bfloat16, device=conv.weight.device)
            wk[..., :cg] = conv.weight.detach().view(G, d // G, cg, k).permute(0, 1, 3, 2).to(torch.bfloat16)
            w[..., : k * cgp] = wk.view(G, d // G, k * cgp)
            return w, conv.bias.detach().float().contiguous(), cgp

        return derived(conv, "groups", (conv.weight, conv.bias), build)

    @staticmethod
    def grouped_conv(conv: nn.Conv1d, h: Tensor, pad: tuple[int, int], act: str, resid: Tensor | None) -> Tensor:
        """act(Conv1d(groups=G)(pad(h))) [+ resid] on time-major bf16 h (B, T, d) -> (B, T_out, d_out)."""
        w, b, cgp = Wav2Vec2._group_weights(conv)
        B, T, _ = h.shape
        G, co, Kp = w.shape
        k, s = conv.kernel_size[0], conv.stride[0]
        Tp = T + pad[0] + pad[1]
        To = (Tp - k) // s + 1
        cg = conv.in_channels // G
        xg = ops.group_windows(h, G, cgp, pad[0], pad[1])  # (B, G, Tp, cgp)
        r2 = None if resid is None else resid.reshape(B * To, G * co)
        if co == cg and ops.grouped_conv_supported(cg, cgp):
            return ops.grouped_conv(xg, w, b, k, s, cg, act, r2).view(B, To, G * co)
        # channel counts the resident-span kernel does not cover: one strided-window GEMM per group
        out = torch.empty((B * To, G * co), dtype=torch.bfloat16, device=h.device)
        xf = xg.view(-1)
        for g in range(G):
            cols = slice(g * co, (g + 1) * co)
            ops.linear_strided(xf[g * Tp * cgp:], M=B * To, K=k * cgp, row_stride=s * cgp, rows_per_batch=To, batch_stride=G * Tp * cgp,
                               w=w[g][:, : k * cgp], bias=b[cols], act=act, resid=None if r2 is None else r2[:, cols], out=out[:, cols])
        return out.view(B, To, G * co)

    def _features(self, x: Tensor) -> Tensor:
        require_bf16_params(self, type(self).__name__)
        return self.proj(self.feature_encoder.time_major(x))  # (B, T, d) bf16

    def forward(self, x: Tensor) -> Tensor:
        """waveform (B, L) -> (B, T, d)."""
        h = self._features(x)
        pad = self.pe_conv[0].padding
        h = self.grouped_conv(self.pe_conv[1], h, (pad[0], pad[1]), "gelu", h)  # x + GELU(conv(x))
        out_dtype = self.norm.weight.dtype
        if self.pre_norm:
            return self.norm(self.layers(h), out_dtype)
        return self.layers(self.norm(h)).to(out_dtype)

    @classmethod
    def from_hf(cls, model_tag: str, *, pretrained: bool = False, config: dict | str | None = None, **kwargs):
        """The reference fetches config.json (and the checkpoint) over HTTP (wav2vec2.py:89-111); this build has no
        network: pass the model's ``config`` (dict or path to a local config.json) and load a local checkpoint with
        ``load_hf_state_dict``."""
        if config is None or pretrained:
            raise NotImplementedError(
                f"{cls.__name__}.from_hf({model_tag!r}): no network in this build - pass config=<dict | path to config.json> "
                "with pretrained=False, then load_hf_state_dict(torch.load(path, weights_only=True)).")
        if isinstance(config, str):
            with open(config) as f:
                config = json.load(f)
        assert config["hidden_size"] == config["num_attention_heads"] * 64
        _kwargs = dict(n_layers=config["num_hidden_layers"], d_model=config["hidden_size"], stem_bias=config["conv_bias"])
        if "feat_extract_norm" in config:
            _kwargs["stem_legacy"] = config["feat_extract_norm"] == "group"
        if "do_stable_layer_norm" in config:
            _kwargs["pre_norm"] = config["do_stable_layer_norm"]
        return cls(**_kwargs, **kwargs)

    _HF_FLAVOUR = "wav2vec2"

    def load_hf_state_dict(self, state_dict: dict[str, Tensor]) -> None:
        """Hugging Face Wav2Vec2Model / HubertModel (Data2VecAudioModel, SEWModel in the subclasses) state_dict, keys without the
        model prefix (wav2vec2.py:113-152; converters.load_hf_wav2vec2).  Prints the keys it did not use, like the reference."""
        from ..converters import load_hf_wav2vec2

        left = load_hf_wav2vec2(self, state_dict, flavour=self._HF_FLAVOUR)
        print(dict.fromkeys(left).keys())
