"""The CPU-device forms of the shared blocks, ViT and Whisper: plain torch on CPU tensors (BASELINE.json configs[0]: "ViT-Ti/16
... on the repo's CPU path (plumbing, no GPU)"; the reference's modules run on whatever device holds them).

Selected ONLY by ``device.type == "cpu"`` of BOTH the input and the parameters (`on_cpu`); a HIP tensor never comes here, and a
HIP call whose library is missing still raises (pytorch_models._hip.lib) - this is not a fallback of the HIP path.  Nothing
is imported from ``oracle/`` (that is test infrastructure).  The arithmetic follows the reference's module code line for line
in meaning (file:line cited per function) and is pinned to the reference's own vectors at the reference's own tolerances by
tests/test_cpu_device.py (tests/golden/vit.npz, blocks.npz, mha.npz, whisper.npz, audio.npz)."""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import Tensor


def on_cpu(x: Tensor, *params) -> bool:
    """True when the input AND every given parameter live on the CPU; mixed placements are an error, as everywhere."""
    ps = [p for p in params if p is not None]
    if x.device.type == "cpu":
        for p in ps:
            if p.device.type != "cpu":
                raise RuntimeError(f"pytorch_models: input on {x.device} but parameters on {p.device}; move both to one device")
        return True
    return False


_ACT = {
    "gelu": F.gelu,
    "approximate_gelu": lambda x: F.gelu(x, approximate="tanh"),
    "relu": F.relu,
    "silu": F.silu,
    "none": lambda x: x,
}


def linear(x: Tensor, w: Tensor, b: Tensor | None) -> Tensor:
    return F.linear(x.to(w.dtype), w, b)


def mha(m, q: Tensor, k, v, attn_bias, causal: bool, residual: Tensor | None) -> Tensor:
    """reference transformer.py:36-53: q/k/v projections, heads split by unflatten + transpose, SDPA, merge, out_proj."""
    k = q if k is None else k
    v = k if v is None else v
    dt = m.q_proj.weight.dtype
    qh = F.linear(q.to(dt), m.q_proj.weight, m.q_proj.bias).unflatten(-1, (m.n_heads, m.head_dim)).transpose(-2, -3)
    kh = F.linear(k.to(dt), m.k_proj.weight, m.k_proj.bias).unflatten(-1, (m.n_heads, m.head_dim)).transpose(-2, -3)
    vh = F.linear(v.to(dt), m.v_proj.weight, m.v_proj.bias).unflatten(-1, (m.n_heads, m.head_dim)).transpose(-2, -3)
    if attn_bias is not None and attn_bias.dtype != torch.bool:
        attn_bias = attn_bias.to(dt)
    o = F.scaled_dot_product_attention(qh, kh, vh, attn_bias, 0.0, causal)
    y = F.linear(o.transpose(-2, -3).flatten(-2), m.out_proj.weight, m.out_proj.bias)
    return y if residual is None else residual.to(dt) + y


def mlp(m, x: Tensor, residual: Tensor | None) -> Tensor:
    """reference transformer.py:56-67."""
    dt = m.linear1.weight.dtype
    h = _ACT[m.act_name](F.linear(x.to(dt), m.linear1.weight, m.linear1.bias))
    y = F.linear(h, m.linear2.weight, m.linear2.bias)
    return y if residual is None else residual.to(dt) + y


def vit_tokens(vit, imgs: Tensor) -> Tensor:
    """reference image/vit.py:78-81: patch_embed -> flatten -> + pe -> cls token prepended (batch-broadcast: SURVEY F1)."""
    dt = vit.patch_embed.weight.dtype
    out = F.conv2d(imgs.to(dt), vit.patch_embed.weight, vit.patch_embed.bias, stride=vit.patch_embed.stride).flatten(2).transpose(1, 2)
    if out.shape[1] != vit.pe.shape[1]:
        raise ValueError(f"ViT: pe holds {vit.pe.shape[1]} positions, the image has {out.shape[1]} patches; call resize_pe first")
    out = out + vit.pe
    if vit.cls_token is not None:
        out = torch.cat([vit.cls_token.expand(out.shape[0], 1, -1), out], 1)
    return out


def whisper_stem(enc, x: Tensor) -> Tensor:
    """reference audio2text/whisper.py:29-33: conv stem, transpose to (B, T, d), + pos_embs."""
    dt = enc.stem[0].weight.dtype
    y = enc.stem(x.to(dt)).transpose(1, 2)
    return y + enc.pos_embs[: y.shape[1]].to(dt)


def embed_tokens(tok: Tensor, E: Tensor, pos: Tensor | None) -> Tensor:
    """reference whisper.py:48 / text models: token embedding (+ learned positions)."""
    y = F.embedding(tok, E)
    return y if pos is None else y + pos[: tok.shape[-1]].to(E.dtype)


def spectrogram(x: Tensor, window: Tensor, n_fft: int, hop: int) -> Tensor:
    """reference audio/spectrogram.py:15-16: centred STFT (reflect padding) -> power spectrogram."""
    return torch.stft(x, n_fft, hop, window=window, return_complex=True).abs().square()


def mel_spectrogram(x: Tensor, window: Tensor, filters: Tensor, n_fft: int, hop: int) -> Tensor:
    """reference audio/spectrogram.py:44-45."""
    return filters @ spectrogram(x, window, n_fft, hop)


def whisper_log_mel(x: Tensor, window: Tensor, filters: Tensor) -> Tensor:
    """reference audio2text/whisper.py:143-148: last frame dropped, log10, floored at the PER-SAMPLE max - 8, (x + 4) / 4
    (the -inf of an all-zero clip propagates exactly as the reference's does)."""
    y = mel_spectrogram(x, window, filters, 400, 160)[..., :-1]
    y = y.clamp(0).log10()
    y = y.maximum(y.flatten(-2).max(-1, keepdim=True)[0].unsqueeze(-1) - 8)
    return (y + 4) / 4
