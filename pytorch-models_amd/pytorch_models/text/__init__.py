"""Text models on the shared transformer blocks: BERT / RoBERTa, GPT, GPT-2 and a KV-cached
greedy generator.  Same import path and class names as /root/reference pytorch_models/text/__init__.py (T5Generator's tokenizer download is not built: T5Model.generate_ids
runs its loop on token ids)."""
from .bert import BERT
from .generator import DecoderGenerator
from .gpt import GPT
from .gpt2 import GPT2
from .t5 import T5Decoder, T5Encoder, T5Model

__all__ = ["BERT", "DecoderGenerator", "GPT", "GPT2", "T5Decoder", "T5Encoder", "T5Model"]
