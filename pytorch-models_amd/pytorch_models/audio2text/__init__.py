from .whisper import Whisper, WhisperDecoder, WhisperEncoder, WhisperPreprocessor
