"""pytorch_models - MI355X (gfx950) build of the gau-nernst/pytorch-models hot path.

Same import paths as the reference for the path BASELINE.json names:
``pytorch_models.transformer``, ``pytorch_models.image.ViT``,
``pytorch_models.audio.spectrogram`` and ``pytorch_models.audio2text``.
Everything computes through hand-written HIP kernels (``libpm_mi355x.so``); there is no CPU path.
"""
