"""Text models on the shared transformer blocks: BERT / RoBERTa, GPT, GPT-2 and a KV-cached
greedy generator.  Same import path and class names as /root/reference pytorch_models/text/__init__.py (T5 is not built)."""
from .bert import BERT
from .generator import DecoderGenerator
from .gpt import GPT
from .gpt2 import GPT2

__all__ = ["BERT", "DecoderGenerator", "GPT", "GPT2"]
