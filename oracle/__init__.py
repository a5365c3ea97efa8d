"""CPU oracle: a plain fp32 PyTorch restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pytorch-models_amd/`` imports this package; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do, and
there only as the checker / reported baseline - never as the thing that is shipped or measured
as the product.  The product path (``pytorch_models`` in this repo) raises if the HIP library
is missing; it never falls back to this code.

Parity pin: PINNED.  Every function here is checked (``tests/test_oracle_golden.py``) against
golden vectors captured by importing the reference (gau-nernst/pytorch-models @ /root/reference)
on CPU in the build container with the deterministic weights of ``synthweights.py``; the
generating script is ``tests/golden/make_golden.py``.  The reference's own upstream comparators
(timm, openai-whisper, librosa: /root/reference tests/image/test_vit.py:2,
tests/audio2text/test_whisper.py:5, tests/audio/test_spectrogram.py:3) and pretrained
checkpoints are not available offline, so "equal to the upstream pretrained models at 2e-5 / 5e-5"
is parity unpinned; "equal to the reference code on identical weights" is what is pinned.

The oracle is written functionally over a ``state_dict`` (name -> tensor) plus a key prefix, so it
consumes the product modules' and the reference modules' weights alike: the parameter names are
the shared contract.
"""
