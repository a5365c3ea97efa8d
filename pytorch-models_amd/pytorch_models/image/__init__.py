from .vit import ViT
