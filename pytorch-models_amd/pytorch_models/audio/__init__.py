"""Audio encoders: Wav2Vec2 / HuBERT, data2vec-audio, SEW.  Same import path and class names as
/root/reference pytorch_models/audio/__init__.py (EnCodec is not built)."""
from .data2vec_audio import Data2VecAudio
from .sew import SEW
from .wav2vec2 import Wav2Vec2

__all__ = ["Data2VecAudio", "SEW", "Wav2Vec2"]
