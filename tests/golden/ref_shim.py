"""Import shim for the *reference* (yilunliao/vit-search) in the dev container.

TEST INFRASTRUCTURE ONLY.  Used by tests/golden/make_golden.py to generate the
committed golden vectors; never imported by the product, by bench.py or by any
test that runs on the GPU box (/root/reference does not exist there).

What it does:
  1. installs a stub `timm` exposing only the symbols the hot path imports
     (nets/vit_sr_supernet.py:9-11, nets/patch_conv.py:5, engine.py:17-18, utils.py:21);
  2. registers synthetic packages `nets`, `supernet_config`, `network_utils` whose
     __path__ points into /root/reference so nets/__init__.py (which pulls
     torchvision) is skipped;
  3. turns the hard-coded `.cuda()` calls (nets/vit_sr_supernet.py:99,
     nets/channel_drop.py:87,151) into no-ops so the reference runs on CPU.
"""
import importlib
import math
import sys
import types

import torch
import torch.nn as nn

REF = "/root/reference"


def _trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    # timm 0.3.2 timm/models/layers/weight_init.py semantics == torch.nn.init.trunc_normal_
    return torch.nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


def _to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


class _PatchEmbed(nn.Module):
    """timm 0.3.2 PatchEmbed: Conv2d(k=s=patch) -> flatten(2).transpose(1, 2)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        img_size = _to_2tuple(img_size)
        patch_size = _to_2tuple(patch_size)
        self.img_size = img_size
        self.patch_size = patch_size
        self.num_patches = (img_size[1] // patch_size[1]) * (img_size[0] // patch_size[0])
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


def _cfg(url="", **kwargs):
    return {"url": url, "num_classes": 1000, "input_size": (3, 224, 224)}


_REGISTRY = {}


def _register_model(fn):
    _REGISTRY[fn.__name__] = fn
    return fn


def _create_model(model_name, pretrained=False, **kwargs):
    kwargs = {k: v for k, v in kwargs.items() if v is not None}
    return _REGISTRY[model_name](pretrained=pretrained, **kwargs)


def _accuracy(output, target, topk=(1,)):
    maxk = max(topk)
    batch_size = target.size(0)
    _, pred = output.topk(maxk, 1, True, True)
    pred = pred.t()
    correct = pred.eq(target.reshape(1, -1).expand_as(pred))
    return [correct[:k].reshape(-1).float().sum(0) * 100.0 / batch_size for k in topk]


def install():
    if "timm" in sys.modules and getattr(sys.modules["timm"], "_is_shim", False):
        return
    timm = types.ModuleType("timm")
    timm._is_shim = True
    models = types.ModuleType("timm.models")
    vt = types.ModuleType("timm.models.vision_transformer")
    layers = types.ModuleType("timm.models.layers")
    registry = types.ModuleType("timm.models.registry")
    data = types.ModuleType("timm.data")
    utils = types.ModuleType("timm.utils")
    vt._cfg = _cfg
    vt.PatchEmbed = _PatchEmbed
    layers.to_2tuple = _to_2tuple
    layers.trunc_normal_ = _trunc_normal_
    registry.register_model = _register_model
    models.create_model = _create_model
    models.vision_transformer = vt
    models.layers = layers
    models.registry = registry
    data.Mixup = type("Mixup", (), {})
    utils.accuracy = _accuracy
    utils.ModelEma = type("ModelEma", (), {})
    timm.models, timm.data, timm.utils = models, data, utils
    for name, mod in [("timm", timm), ("timm.models", models), ("timm.models.vision_transformer", vt),
                      ("timm.models.layers", layers), ("timm.models.registry", registry),
                      ("timm.data", data), ("timm.utils", utils)]:
        sys.modules[name] = mod

    for pkg in ("nets", "supernet_config", "network_utils"):
        m = types.ModuleType(pkg)
        m.__path__ = [f"{REF}/{pkg}"]
        sys.modules[pkg] = m

    # hard-coded .cuda() -> no-op on this CPU-only host
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.synchronize = lambda *a, **k: None
    if REF not in sys.path:
        sys.path.append(REF)


def ref_modules():
    """Return the reference hot-path modules (imported lazily)."""
    install()
    out = types.SimpleNamespace()
    out.vit_sr = importlib.import_module("nets.vit_sr_supernet")
    out.blocks = importlib.import_module("nets.supernet_blocks")
    out.mln = importlib.import_module("nets.masked_layer_norm")
    out.channel_drop = importlib.import_module("nets.channel_drop")
    out.net_utils = importlib.import_module("nets.net_utils")
    out.patch_conv = importlib.import_module("nets.patch_conv")
    out.flop = importlib.import_module("network_utils.compute_flop_mac")
    out.cfg = {n: importlib.import_module(f"supernet_config.{n}")
               for n in ("sr_tiny", "sr_small", "sr_tiny_mh", "sr_small_mh", "sr_tiny_666")}
    out.create_model = _create_model
    return out
