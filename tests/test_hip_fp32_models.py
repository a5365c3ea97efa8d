"""Models left in the reference's default fp32 compute in fp32 (pm_linear_f32 / pm_attention_generic_f32 / fp32 LayerNorm):
checked against the fp32 oracle on the SAME fp32 weights at fp32 tolerances - nothing is rounded to bf16 on the way
(the first round ran such models through bf16 copies at 2e-2).  tests/test_hip_exact.py holds the comparisons
with the reference's own vectors."""
import copy

import pytest
import torch

from oracle import ref_text as RX
from oracle import ref_transformer as RT
from oracle import ref_vit as RV
from oracle import ref_whisper as RW
from synthweights import fill_module, synth_input, synth_tokens

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
F32 = dict(rtol=5e-5, atol=5e-5)


def pair(m, seed):
    """(fp32 model on the GPU, its bf16 twin, state_dict): plain fp32 weights, NOT bf16-representable."""
    fill_module(m, seed)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    return m.cuda().eval(), copy.deepcopy(m).to(torch.bfloat16).cuda().eval(), sd


def test_blocks_compute_in_the_dtype_of_their_parameters():
    from pytorch_models.transformer import DecoderLayer, Encoder

    m32, m16, sd = pair(Encoder(2, 128, n_heads=2), 91)
    x = synth_input("f32_x", (3, 40, 128), 91)
    y32 = m32(x.cuda())
    assert y32.dtype == torch.float32
    torch.testing.assert_close(y32.cpu(), RT.encoder(sd, "", 2, x), **F32)
    assert m16(x.to(torch.bfloat16).cuda()).dtype == torch.bfloat16
    d32, _, sd = pair(DecoderLayer(64, cross_attn=True), 92)
    xq, mem = synth_input("f32_q", (2, 9, 64), 92), synth_input("f32_m", (2, 5, 64), 92)
    y = d32(xq.cuda(), mem.cuda())
    torch.testing.assert_close(y.cpu(), RT.decoder_layer(sd, "", 1, xq, mem), **F32)
    with torch.no_grad():  # a weight update invalidates the derived (packed) copies
        for p in d32.parameters():
            p.mul_(0.5)
    assert ((d32(xq.cuda(), mem.cuda()) - y).norm() / y.norm()).item() > 1e-3


def test_vit_whisper_gpt2_bert_in_fp32():
    from pytorch_models.audio2text import Whisper
    from pytorch_models.image import ViT
    from pytorch_models.text import BERT, GPT2

    v32, _, sd = pair(ViT.from_google("Ti/16"), 93)
    imgs = synth_input("f32_img", (2, 3, 224, 224), 93)
    f32 = v32(imgs.cuda())
    assert f32.dtype == torch.float32 and f32.shape == (2, 192)
    torch.testing.assert_close(f32.cpu(), RV.forward(sd, RV.geometry_from_google("Ti/16"), imgs), **F32)
    w32, _, sd = pair(Whisper(100, 2, 64), 94)
    mel, toks = synth_input("f32_mel", (2, 80, 16), 94), synth_tokens("f32_tok", (2, 12), 100, 94)
    lg = w32(mel.cuda(), toks.cuda())
    assert lg.dtype == torch.float32
    torch.testing.assert_close(lg.cpu(), RW.forward(sd, mel, toks), **F32)
    ids = w32.generate(mel.cuda(), toks[:, :2].cuda(), 6)  # an fp32 model decodes through the fp32 end-to-end loop
    want, _ = RW.greedy_cached(sd, "decoder.", toks[:, :2], RW.encoder(sd, "encoder.", mel), 6)
    assert torch.equal(ids.cpu(), want)
    tok = synth_tokens("text_tok", (2, 16), 2000, 71)
    g32, _, sd = pair(GPT2(2, 128), 95)
    torch.testing.assert_close(g32(tok.cuda()).cpu(), RX.gpt2(sd, tok), rtol=2e-4, atol=2e-4)  # 50257 logits of magnitude ~1
    b32, _, sd = pair(BERT(2000, 2, 128), 96)
    h = b32(tok.cuda())
    assert h.dtype == torch.float32
    torch.testing.assert_close(h.cpu(), RX.bert(sd, tok), **F32)
