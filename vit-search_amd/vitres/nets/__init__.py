from .vit_sr_supernet import *  # noqa: F401,F403
