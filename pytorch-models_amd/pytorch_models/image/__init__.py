from .mobile_vit import MobileViT
from .vit import ViT
