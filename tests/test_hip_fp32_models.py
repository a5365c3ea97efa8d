"""Models left in the reference's default fp32 run on the same kernels: weights through cached bf16 copies, fp32 at the
interface (inputs rounded to bf16 on the way in, every module returns the dtype it was given, fp32 residual stream).
Checked against the fp32 oracle on the same (bf16-representable) weights and against the native bf16 model."""
import copy

import pytest
import torch

from oracle import ref_text as RX
from oracle import ref_transformer as RT
from oracle import ref_vit as RV
from oracle import ref_whisper as RW
from synthweights import bf16_round_, fill_module, synth_input, synth_tokens

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def rel(got, want):
    got, want = got.float().cpu(), want.float().cpu()
    return ((got - want).norm() / want.norm()).item()


def pair(m, seed):
    """(fp32 model on the GPU, its bf16 twin, state_dict) with bf16-representable weights."""
    fill_module(m, seed)
    bf16_round_(m)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    return m.cuda().eval(), copy.deepcopy(m).to(torch.bfloat16).cuda().eval(), sd


def test_blocks_keep_the_dtype_they_are_given():
    from pytorch_models.transformer import DecoderLayer, Encoder

    m32, m16, sd = pair(Encoder(2, 128, n_heads=2), 91)
    x = synth_input("f32_x", (3, 40, 128), 91)
    y32 = m32(x.cuda())
    assert y32.dtype == torch.float32
    want = RT.encoder(sd, "", 2, x.to(torch.bfloat16).float())
    assert rel(y32, want) < 2e-2 and rel(y32, m16(x.to(torch.bfloat16).cuda())) < 2e-2
    # bf16 activations into an fp32 module stay bf16 (only the weights are copied)
    assert m32(x.to(torch.bfloat16).cuda()).dtype == torch.bfloat16
    d32, _, sd = pair(DecoderLayer(64, cross_attn=True), 92)
    xq, mem = synth_input("f32_q", (2, 9, 64), 92), synth_input("f32_m", (2, 5, 64), 92)
    y = d32(xq.cuda(), mem.cuda())
    assert y.dtype == torch.float32 and rel(y, RT.decoder_layer(sd, "", 1, xq.to(torch.bfloat16).float(), mem.to(torch.bfloat16).float())) < 2e-2
    # a weight update invalidates the cached bf16 copies
    with torch.no_grad():
        for p in d32.parameters():
            p.mul_(0.5)
    assert rel(d32(xq.cuda(), mem.cuda()), y) > 1e-3


def test_vit_whisper_gpt2_bert_in_fp32():
    from pytorch_models.audio2text import Whisper
    from pytorch_models.image import ViT
    from pytorch_models.text import BERT, GPT2

    v32, v16, sd = pair(ViT.from_google("Ti/16"), 93)
    imgs = synth_input("f32_img", (2, 3, 224, 224), 93)
    f32 = v32(imgs.cuda())
    assert f32.dtype == torch.float32 and f32.shape == (2, 192)
    assert rel(f32, v16(imgs.cuda())) < 2e-2
    assert rel(f32, RV.forward(sd, RV.geometry_from_google("Ti/16"), imgs)) < 2e-2
    w32, w16, sd = pair(Whisper(100, 2, 64), 94)
    mel, toks = synth_input("f32_mel", (2, 80, 16), 94), synth_tokens("f32_tok", (2, 12), 100, 94)
    lg = w32(mel.cuda(), toks.cuda())
    assert lg.dtype == torch.float32 and rel(lg, RW.forward(sd, mel, toks)) < 3e-2 and rel(lg, w16(mel.cuda(), toks.cuda())) < 3e-2
    assert w32.encoder(mel.cuda()).dtype == torch.float32
    with pytest.raises(NotImplementedError, match="bf16"):
        w32.generate(mel.cuda(), toks[:, :2].cuda(), 2)  # the KV-cached decode wants the native bf16 model
    tok = synth_tokens("text_tok", (2, 16), 2000, 71)
    g32, g16, sd = pair(GPT2(2, 128), 95)
    assert rel(g32(tok.cuda()), RX.gpt2(sd, tok)) < 2e-2
    b32, _, sd = pair(BERT(2000, 2, 128), 96)
    h = b32(tok.cuda())
    assert h.dtype == torch.float32 and rel(h, RX.bert(sd, tok)) < 2e-2
