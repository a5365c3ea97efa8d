"""Whisper on MI355X, drop-in for /root/reference pytorch_models/audio2text/whisper.py
(WhisperEncoder, WhisperDecoder, Whisper, WhisperPreprocessor, Whisper.from_openai; same parameter
names: stem.0 / stem.2, pos_embs, layers, norm, token_embs).

New capability (absent in the reference, README.md:86): ``Whisper.generate`` - batched greedy decoding
with a KV cache, see generate.py.
"""
from __future__ import annotations

import torch
from torch import Tensor, nn

from .. import _cpu
from .._hip import ops
from ..audio.spectrogram import MelSpectrogram
from ..transformer import Decoder, Encoder, LayerNorm, _f32, _wb, derived

_SIZES = {  # tag -> (n_layers, d_model); "base" is 8 layers exactly as the reference builds it
    "tiny": (4, 384), "tiny.en": (4, 384), "base": (8, 512), "base.en": (8, 512),
    "small": (12, 768), "small.en": (12, 768), "medium": (24, 1024), "medium.en": (24, 1024),
    "large-v1": (32, 1280), "large-v2": (32, 1280), "large-v3": (32, 1280),
}


class WhisperEncoder(nn.Module):
    max_seq_len = 3000

    def __init__(self, n_layers: int, d_model: int, n_mels: int = 80, dropout: float = 0.0) -> None:
        super().__init__()
        self.stem = nn.Sequential(
            nn.Conv1d(n_mels, d_model, 3, 1, 1),
            nn.GELU(),
            nn.Conv1d(d_model, d_model, 3, 2, 1),
            nn.GELU(),
        )
        self.register_buffer("pos_embs", torch.zeros(self.max_seq_len // 2, d_model))
        self.pos_embs: Tensor
        self.layers = Encoder(n_layers, d_model, dropout=dropout)
        self.norm = LayerNorm(d_model)

    def _stem_weights(self):
        c1, c2 = self.stem[0], self.stem[2]

        def build():
            d, n_mels, _ = c1.weight.shape
            cpad = (n_mels + 63) // 64 * 64
            w1 = torch.zeros(d, 3, cpad, dtype=torch.bfloat16, device=c1.weight.device)  # (out, tap, channel) K-major
            w1[:, :, :n_mels] = c1.weight.detach().permute(0, 2, 1).to(torch.bfloat16)
            w2 = c2.weight.detach().permute(0, 2, 1).contiguous().view(d, 3 * d).to(torch.bfloat16)
            return w1.view(d, 3 * cpad), c1.bias.detach().float().contiguous(), w2, c2.bias.detach().float().contiguous()

        return derived(self, "stem", (c1.weight, c1.bias, c2.weight, c2.bias), build)

    def forward(self, x: Tensor) -> Tensor:
        """(B, n_mels, T) -> (B, T // 2, d): conv stem (both convs as MFMA GEMMs with fused GELU, the second
        also adding pos_embs in its epilogue), Encoder, LayerNorm."""
        if _cpu.on_cpu(x, self.stem[0].weight):  # CPU tensors and parameters: plain torch (pytorch_models/_cpu.py)
            return self.norm(self.layers(_cpu.whisper_stem(self, x)))
        if self.stem[0].weight.dtype == torch.float32:
            return self._forward_f32(x)
        w1, b1, w2, b2 = self._stem_weights()
        B, _, T = x.shape
        d = w2.shape[0]
        y1 = ops.whisper_stem1(x.float().contiguous(), w1, b1)  # (B, T + 2, d) bf16, rows 0 and T + 1 zero
        L = (T - 1) // 2 + 1
        pos = _f32(self, "pos", self.pos_embs)[:L]
        # Conv1d(d, d, 3, stride 2, pad 1): output row t' reads buffer rows 2t', 2t'+1, 2t'+2 = 3*d contiguous values
        y2 = ops.linear_strided(y1, M=B * L, K=3 * d, row_stride=2 * d, rows_per_batch=L, batch_stride=(T + 2) * d,
                                w=w2, bias=b2, act="gelu", resid=pos, resid_period=L)
        return self.norm(self.layers(y2.view(B, L, d)), self.stem[0].weight.dtype)  # fp32 model -> fp32 memory


def _encoder_forward_f32(self: WhisperEncoder, x: Tensor) -> Tensor:
    """fp32 parameters: both convolutions as fp32 GEMMs over windows (a window view + one contiguous copy each is layout,
    the arithmetic is pm_linear_f32 with GELU and, for the second, + pos_embs in its epilogue), fp32 Encoder, fp32 LayerNorm."""
    c1, c2 = self.stem[0], self.stem[2]
    B, C, T = x.shape
    d = c1.weight.shape[0]
    xp = torch.nn.functional.pad(x.float(), (1, 1))
    cols = xp.unfold(2, 3, 1).permute(0, 2, 1, 3).reshape(B * T, C * 3)  # (B T, C * 3): K order (channel, tap) = weight.view(d, -1)
    y1 = ops.linear_f32(cols, c1.weight.view(d, C * 3), c1.bias, act="gelu").view(B, T, d)
    L = (T - 1) // 2 + 1
    yp = torch.nn.functional.pad(y1, (0, 0, 1, 1))
    cols = yp.unfold(1, 3, 2).reshape(B * L, d * 3)  # window dim last: (B, L, d, 3) -> K order (channel, tap)
    pos = self.pos_embs.float()[:L].contiguous()
    y2 = ops.linear_f32(cols, c2.weight.view(d, d * 3), c2.bias, act="gelu", resid=pos, resid_period=L)
    return self.norm(self.layers(y2.view(B, L, d)))


WhisperEncoder._forward_f32 = _encoder_forward_f32


class WhisperDecoder(nn.Module):
    max_seq_len = 448

    def __init__(self, vocab_size: int, n_layers: int, d_model: int, dropout: float = 0.0) -> None:
        super().__init__()
        self.token_embs = nn.Embedding(vocab_size, d_model)
        self.pos_embs = nn.Parameter(torch.zeros(self.max_seq_len, d_model))
        self.layers = Decoder(n_layers, d_model, cross_attn=True, dropout=dropout)
        self.norm = LayerNorm(d_model)

    def forward(self, x: Tensor, memory: Tensor) -> Tensor:
        """tokens (B, L) int64, memory (B, S, d) -> logits (B, L, V) (tied embeddings), teacher-forced."""
        if _cpu.on_cpu(x, self.token_embs.weight):
            E = self.token_embs.weight
            h = self.norm(self.layers(_cpu.embed_tokens(x, E, self.pos_embs), memory.to(E.dtype)))
            return h @ E.T
        if self.token_embs.weight.dtype == torch.float32:  # fp32 parameters: fp32 throughout
            E = self.token_embs.weight
            h = self.norm(self.layers(ops.embed_tokens(x, E, self.pos_embs), memory.float()))
            return ops.linear_f32(h.view(-1, h.shape[-1]), E).view(*x.shape, E.shape[0])
        E = _wb(self.token_embs, "E", self.token_embs.weight)
        h = ops.embed_tokens(x, E, _f32(self, "pos", self.pos_embs))  # (B, L, d) bf16
        h = self.norm(self.layers(h, memory))
        return ops.linear(h.view(-1, h.shape[-1]), E, None, out_dtype=torch.float32).view(*x.shape, E.shape[0])

    @torch.no_grad()
    def generate(self, memory: Tensor, prompt: Tensor, max_new_tokens: int, *, graph: bool = True, rules=None,
                 path: str = "auto") -> Tensor:
        """Batched greedy decoding with a KV cache: (B, P) int64 prompt -> (B, P + max_new_tokens) ids.  ``rules``: a
        generate.WhisperRules (token suppression, timestamp pairing / monotonicity) applied to the logits on the device;
        ``path``: "persistent" / "launches" / "auto" (generate.GreedyDecoder)."""
        from .generate import greedy_decode, greedy_exact

        if self.token_embs.weight.dtype == torch.float32:  # fp32 parameters: the fp32 end-to-end loop (no bf16 storage points)
            if rules is not None:
                raise NotImplementedError("WhisperDecoder.generate: the decoding rules run on the bf16 model's step kernels")
            return greedy_exact(self, memory, prompt, max_new_tokens)
        return greedy_decode(self, memory, prompt, max_new_tokens, graph=graph, rules=rules, path=path)


class Whisper(nn.Module):
    def __init__(self, vocab_size: int, n_layers: int, d_model: int, n_mels: int = 80, dropout: float = 0.0, *,
                 n_decoder_layers: int | None = None) -> None:
        """``n_decoder_layers`` (superset of the reference signature): a shallower decoder than encoder, the geometry of
        the distilled checkpoints the reference lists as TODO (README.md:87; e.g. distil-large-v2 = 32 encoder / 2 decoder
        layers).  Parameter names are unchanged, so such a checkpoint loads through load_openai_state_dict."""
        super().__init__()
        self.encoder = WhisperEncoder(n_layers, d_model, n_mels, dropout=dropout)
        self.decoder = WhisperDecoder(vocab_size, n_layers if n_decoder_layers is None else n_decoder_layers, d_model, dropout=dropout)

    def forward(self, x: Tensor, targets: Tensor) -> Tensor:
        return self.decoder(targets, self.encoder(x))

    @torch.no_grad()
    def generate(self, x: Tensor, prompt: Tensor, max_new_tokens: int, *, graph: bool = True, rules=None, path: str = "auto",
                 exact: bool = False) -> Tensor:
        """log-mel (B, n_mels, T) + prompt ids (B, P) -> greedy ids (B, P + max_new_tokens).
        ``exact=True`` (bf16 model): the whole pipeline - encoder, cross K/V, decoder, caches - in fp32 on an fp32 copy of
        this model's (bf16-valued) weights: the ids are those of the reference's fp32 forward on the same weights, bit for
        bit (tests/test_hip_exact.py); costs ~10x the bf16 encoder and an eager fp32 step loop.  A model whose
        parameters are fp32 always decodes this way."""
        if exact and self.decoder.token_embs.weight.dtype != torch.float32:
            # fp32 encoder on the fp32 twin of the (bf16-valued) weights; the decode step is the throughput path's own launch
            # list with fp32 K / V caches - its projections are fp32-exact as they stand (bf16 weights x activations in three
            # bf16 terms) - replayed as one HIP graph per token.  Geometries that list does not cover (more than 256
            # (sequence, head) pairs) fall back to the eager fp32 loop of the twin.
            from .generate import greedy_decode

            twin = self.exact_copy()
            H = self.decoder.layers[0].sa.n_heads
            if rules is None and prompt.shape[0] * H <= 256 and prompt.shape[0] <= 64:
                return greedy_decode(self.decoder, twin.encoder(x), prompt, max_new_tokens, graph=graph, kv32=True)
            return twin.generate(x, prompt, max_new_tokens)
        return self.decoder.generate(self.encoder(x), prompt, max_new_tokens, graph=graph, rules=rules, path=path)

    def exact_copy(self) -> "Whisper":
        """fp32 twin of this model (same values: bf16 -> fp32 is exact), rebuilt when a parameter changes."""
        import copy

        params = list(self.parameters()) + list(self.buffers())
        def build():
            twin = copy.deepcopy(self).float()
            for m in twin.modules():  # the twin derives its own packed / re-typed weights
                m.__dict__.pop("_pm_derived", None)
            return twin

        return derived(self, "exact32", params, build)

    def load_openai_state_dict(self, state_dict) -> None:
        """OpenAI ``model_state_dict`` (e.g. ``torch.load(path, weights_only=True)["model_state_dict"]``)."""
        from ..converters import load_openai_whisper

        left = load_openai_whisper(self, state_dict)
        if left:
            print(left)

    @staticmethod
    def from_openai(model_tag: str, *, pretrained: bool = False, **kwargs) -> "Whisper":
        n_layers, d_model = _SIZES[model_tag]
        if model_tag == "large-v3":
            n_mels, vocab_size = 128, 51866
        else:
            n_mels, vocab_size = 80, (51864 if model_tag.endswith(".en") else 51865)
        m = Whisper(vocab_size, n_layers, d_model, n_mels, **kwargs)
        if pretrained:
            raise NotImplementedError(
                "Whisper.from_openai(pretrained=True) needs a network download, which this build does not do; "
                "construct with pretrained=False and load a local checkpoint into the (reference-named) parameters.")
        return m


class WhisperPreprocessor(MelSpectrogram):
    def __init__(self, variant: str = "tiny") -> None:
        n_mels = 128 if variant == "large-v3" else 80
        super().__init__(400, 160, n_mels, 16_000)

    def forward(self, x: Tensor) -> Tensor:
        """(..., T) waveform -> (..., n_mels, T // 160) log-mel: last frame dropped, log10, floored at the
        PER-SAMPLE max - 8, (x + 4) / 4 - all inside the two logmel kernels."""
        if _cpu.on_cpu(x, self.window, self.filters):
            return _cpu.whisper_log_mel(x, self.window, self.filters)
        return ops.stft_mel(x, self._tables(), 400, 160, x.shape[-1] // 160, 2, self._csr(), self.filters.shape[0])
