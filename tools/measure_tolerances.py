"""Measured error of the bf16 path per BASELINE config (the test bounds are set from it):
rel-L2 and max-abs of the LayerNorm-ed outputs against (a) the fp32 oracle on the same bf16-rounded weights, (b) the
reference's fp32 goldens (fp32 weights).  Also the cost of the fp32 / exact modes.  Writes JSON to stdout."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import numpy as np
import torch
from oracle import ref_spectrogram as RS, ref_vit as RV, ref_whisper as RW
from synthweights import bf16_round_, fill_module, synth_input, synth_tokens
from pytorch_models.audio2text import Whisper, WhisperPreprocessor
from pytorch_models.image import ViT
torch.set_grad_enabled(False)
torch.set_num_threads(16)


def G(name):
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False)
    return {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}


def err(got, want):
    got, want = got.float().cpu(), want.float()
    return {"rel_l2": round(((got - want).norm() / want.norm()).item(), 5), "max_abs": round((got - want).abs().max().item(), 5)}


res = {}
gv, gw = G("vit"), G("whisper")
for key, mk, seed, xname, shape, geo, gold in [
    ("C1 ViT-Ti/16 b1", lambda: ViT.from_google("Ti/16"), 31, "vit_ti", (1, 3, 224, 224), "Ti/16", "ti16_b1"),
    ("C2 ViT-B/16 (first 4 of the batch)", lambda: ViT.from_google("B/16"), 32, "vit_b", (4, 3, 224, 224), "B/16", "b16_first4"),
    ("C5 ViT-L/16 siglip @384 b2", lambda: ViT.from_google("L/16_siglip", img_size=384), 34, "vit_ls", (2, 3, 384, 384), None, "l16_siglip384_b2"),
]:
    m = mk().eval(); fill_module(m, seed); bf16_round_(m)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = synth_input(xname, shape, seed)
    got = m.to(torch.bfloat16).cuda()(x.cuda())
    e = {"vs_reference_golden": err(got, gv[gold])}
    if geo:
        e["vs_oracle_same_weights"] = err(got, RV.forward(sd, RV.geometry_from_google(geo), x))
    res[key] = e
for tag, seed, key in (("base", 56, "C3 Whisper-base encoder memory (2 clips)"), ("tiny", 55, "Whisper-tiny encoder memory (2 clips)")):
    w = Whisper.from_openai(tag).eval(); fill_module(w, seed); bf16_round_(w)
    sd = {k: v.clone() for k, v in w.state_dict().items()}
    wave = synth_input(f"w_wave_{tag}", (2, 480000), seed, scale=0.1)
    mem = w.to(torch.bfloat16).cuda().encoder(WhisperPreprocessor(tag).cuda()(wave.cuda()))
    want = RW.encoder(sd, "encoder.", RS.whisper_log_mel(wave, 80, "rfft"))
    res[key] = {"vs_oracle_same_weights": err(mem, want),
                "vs_reference_golden_slice": err(mem[:, ::100, ::32], gw[f"greedy_{tag}_memory_slice"])}
# costs of the accurate modes (Whisper-base, 32 clips; ViT-B/16, 64 images)
w = Whisper.from_openai("base").eval(); fill_module(w, 56); bf16_round_(w); w = w.to(torch.bfloat16).cuda()
mel = WhisperPreprocessor("base").cuda()(synth_input("tol_wave", (32, 480000), 1, scale=0.1).cuda())
prompt = synth_tokens("tol_p", (32, 4), 51865, 1).cuda()
def timed(fn, n=2):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
res["cost_ms Whisper-base b32 224 tokens"] = {"default bf16": round(timed(lambda: w.generate(mel, prompt, 224)), 1),
                                              "exact=True (fp32 twin)": round(timed(lambda: w.generate(mel, prompt, 224, exact=True), 1), 1),
                                              "exact encoder alone": round(timed(lambda: w.exact_copy().encoder(mel)), 1),
                                              "bf16 encoder alone": round(timed(lambda: w.encoder(mel)), 1)}
v = ViT.from_google("B/16").eval(); fill_module(v, 32)
x = synth_input("tol_img", (64, 3, 224, 224), 2).cuda()
v32, v16 = v.cuda(), None
t32 = timed(lambda: v32(x))
import copy
v16 = copy.deepcopy(v).to(torch.bfloat16).cuda()
res["cost_ms ViT-B/16 b64"] = {"fp32 model": round(t32, 2), "bf16 model": round(timed(lambda: v16(x)), 2)}
print(json.dumps(res, indent=1))
