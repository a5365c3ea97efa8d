"""pytorch_models - MI355X (gfx950) build of the gau-nernst/pytorch-models hot path.

Same import paths as the reference for the path BASELINE.json names:
``pytorch_models.transformer``, ``pytorch_models.image.ViT``,
``pytorch_models.audio.spectrogram`` and ``pytorch_models.audio2text``.
Everything on a HIP device computes through hand-written HIP kernels (``libpm_mi355x.so``; no fallback: a missing library raises).
A module and its input both on the CPU take the plain-torch CPU forms of ``_cpu.py`` (BASELINE configs[0]); a HIP tensor never does.
"""
