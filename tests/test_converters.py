"""CPU: the weight converters load synthetic upstream-format checkpoints into exactly the
parameters the reference's converters produce (digests captured in tests/golden/converters.json by
tests/golden/make_golden.py from the reference import), and invalidate the packed-weight caches of the HIP path."""
import json
import os

import numpy as np
import pytest
import torch

import ckpt_synth as C
from pytorch_models.audio2text import Whisper
from pytorch_models.image import ViT

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "converters.json")))


def check(model, name):
    got = C.state_digest(model.state_dict())
    want = GOLD[name]
    assert sorted(got) == sorted(want)
    for k in want:
        np.testing.assert_allclose(got[k], want[k], rtol=1e-9, atol=1e-9, err_msg=f"{name}: {k}")


def test_flax_augreg(tmp_path):
    ck = C.flax_vit(2, 128, 2, 16, 16, big_vision=False, cls=True, map_head=False, seed=61)
    m = ViT(2, 128, 2, 16, img_size=64)
    m.load_flax_ckpt(ck)
    check(m, "flax_augreg")
    path = tmp_path / "ck.npz"  # the same through a local .npz file
    np.savez(path, **ck)
    m2 = ViT(2, 128, 2, 16, img_size=64)
    m2.load_flax_ckpt(str(path))
    check(m2, "flax_augreg")
    with pytest.raises(FileNotFoundError):
        m2.load_flax_ckpt("augreg/not_downloaded.npz")


def test_flax_siglip_big_vision():
    m = ViT(2, 128, 2, 16, img_size=64, cls_token=False, pool_type="mha")
    m.load_flax_ckpt(C.flax_vit(2, 128, 2, 16, 16, big_vision=True, cls=False, map_head=True, seed=62, prefix="params/img/"),
                     big_vision=True, prefix="params/img/")
    check(m, "flax_siglip")


@pytest.mark.parametrize("name,cls_slot,ls,seed", [("fb_deit3", False, "gamma", 63), ("fb_dinov2", True, "ls", 64), ("fb_dino", True, None, 65)])
def test_facebook(name, cls_slot, ls, seed, capsys):
    m = ViT(2, 128, 2, 16, img_size=64)
    m.load_facebook_state_dict(C.facebook_vit(2, 128, 16, 16, pe_has_cls=cls_slot, layer_scale=ls, seed=seed))
    check(m, name)
    assert capsys.readouterr().out.strip() == "[]"  # every upstream key consumed


def test_openai_whisper_and_cache_invalidation():
    w = Whisper(100, 2, 64)
    qkv_before, _ = w.decoder.layers[0].sa._pack("qkv")
    w.load_openai_state_dict(C.openai_whisper(2, 64, 80, 100, seed=66))
    check(w, "openai_whisper")
    assert w.decoder.layers[0].sa.k_proj.bias.abs().sum() == 0  # OpenAI's key projection has no bias
    qkv_after, _ = w.decoder.layers[0].sa._pack("qkv")
    assert qkv_after is not qkv_before and torch.equal(
        qkv_after[:64], w.decoder.layers[0].sa.q_proj.weight.to(torch.bfloat16)
    )  # packed copies are bf16 whatever the module dtype
