from .vit_sr_supernet import *  # noqa: F401,F403
from .vision_transformer_supernet import FlexibleDistillVisionTransformer  # noqa: F401,E402
