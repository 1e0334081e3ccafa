"""timm-free model registry with timm 0.3.2's `create_model` calling convention
(reference main.py:329-348 builds kwargs and calls timm.models.create_model(**kwargs))."""
_MODELS = {}


def register_model(fn):
    _MODELS[fn.__name__] = fn
    try:                                    # also expose through timm when it is installed (drop-in for main.py)
        from timm.models.registry import register_model as _timm_register
        _timm_register(fn)
    except Exception:
        pass
    return fn


def create_model(model_name, pretrained=False, **kwargs):
    """None-valued kwargs are dropped, as timm 0.3.2 does (drop_block_rate=None in main.py:331)."""
    if model_name not in _MODELS:
        raise RuntimeError('Unknown model (%s)' % model_name)
    kwargs = {k: v for k, v in kwargs.items() if v is not None}
    return _MODELS[model_name](pretrained=pretrained, **kwargs)


def list_models():
    return sorted(_MODELS)
