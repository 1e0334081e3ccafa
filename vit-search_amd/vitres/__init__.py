"""vitres: MI355X-native ViT-Res / ViT-ResNAS (super)network hot path.

Python host side mirroring the reference's timm-style surface (create_model / network_def /
supernet_config / engine), executing through libvitres_hip.so (include/vitres_hip.h).
"""
from . import supernet_config  # noqa: F401
from .nets import vit_sr_supernet  # noqa: F401  (registers the model factories)
from .registry import create_model, list_models, register_model  # noqa: F401

__all__ = ["create_model", "register_model", "list_models", "supernet_config"]
