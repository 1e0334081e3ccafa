"""Search spaces of the shipped ViT-ResNAS supernets (reference supernet_config/*.py), restated.

`<space>.num_channels_to_keep` is a list aligned 1:1 with `network_def`: np.ndarray of width choices for
embed / spatial-reduction entries, {'attn','mlp','layer'} dict for transformer blocks ('layer' may be
None; a 0 choice = drop the block), None for the head.  `<space>.network_def` is the largest network.
"""
import copy
import types

import numpy as np


def _space(name, embed_entry, stages, head=1000):
    """stages: list of (dim, heads, head_dim, hidden, n_blocks, embed_choices, attn, mlp, layer_choices, skip_fn)"""
    keep, ndef = [], [embed_entry]
    for si, (dim, heads, hd, hidden, nblk, emb, attn, mlp, layer, skip) in enumerate(stages):
        if si > 0:
            ndef.append((3, stages[si - 1][0], dim))
        keep.append(np.array(emb))
        blk = {'attn': np.array(attn), 'mlp': np.array(mlp), 'layer': None}
        blk_skip = copy.deepcopy(blk)
        blk_skip['layer'] = np.array(layer) if layer is not None else None
        for b in range(nblk):
            keep.append(blk_skip if skip(b) else blk)
            ndef.append((1, (dim, heads, hd), (dim, hidden), 1))
    keep.append(None)
    ndef.append((2, stages[-1][0], head))
    m = types.ModuleType(__name__ + '.' + name)
    m.num_channels_to_keep = keep
    m.network_def = tuple(ndef)
    return m


_odd = lambda b: b % 2 == 1          # blocks 2,4,6 (1-based) can be skipped
_odd_not_last = lambda b: b % 2 == 1
_never = lambda b: False

_E1, _E2, _E3 = [256, 224, 192, 176, 160], [512, 448, 384, 352, 320], [1024, 896, 768, 704, 640]
_S1, _S2, _S3 = [320, 280, 240, 220, 200], [640, 560, 480, 440, 400], [1280, 1120, 960, 880, 800]

sr_tiny = _space('sr_tiny', (0, 256), [
    (256, 4, 64, 768, 7, _E1, [256, 192, 128], [768, 640, 512, 384], [256, 256, 256, 0], _odd),
    (512, 8, 64, 1536, 7, _E2, [512, 384, 256], [1536, 1280, 1024, 768], [512, 512, 512, 0], _odd),
    (1024, 12, 64, 3072, 4, _E3, [768, 640, 512], [3072, 2560, 2048, 1536], None, _never)])

sr_tiny_666 = _space('sr_tiny_666', (0, 256), [
    (256, 4, 64, 768, 6, _E1, [256, 192, 128], [768, 640, 512, 384], [256, 256, 256, 0], _odd),
    (512, 8, 64, 1536, 6, _E2, [512, 384, 256], [1536, 1280, 1024, 768], [512, 512, 512, 0], _odd),
    (1024, 12, 64, 3072, 6, _E3, [768, 640, 512], [3072, 2560, 2048, 1536], [1024, 1024, 1024, 0], _odd)])

sr_tiny_mh = _space('sr_tiny_mh', (4, 256), [
    (256, 6, 32, 768, 6, _E1, [192, 160, 128, 96], [768, 704, 640, 576, 512, 448, 384], [256, 256, 0, 0], _odd),
    (512, 12, 48, 1536, 6, _E2, [576, 480, 384, 288], [1536, 1408, 1280, 1152, 1024, 896, 768], [512, 512, 0, 0], _odd),
    (1024, 12, 64, 3072, 6, _E3, [768, 640, 512, 384], [3072, 2816, 2560, 2304, 2048, 1792, 1536],
     [1024, 1024, 0, 0], _odd)])

sr_small = _space('sr_small', (5, 320, 32), [
    (320, 8, 32, 960, 7, _S1, [256, 224, 192, 160], [960, 880, 800, 720, 640, 560, 480], [320, 320, 0, 0], _odd),
    (640, 12, 48, 1920, 7, _S2, [576, 480, 384, 288], [1920, 1760, 1600, 1440, 1280, 1120, 960], [640, 640, 0, 0], _odd),
    (1280, 12, 64, 3840, 7, _S3, [768, 640, 512, 384], [3840, 3520, 3200, 2880, 2560, 2240, 1920],
     [1280, 1280, 0, 0], _odd)])

sr_small_mh = _space('sr_small_mh', (4, 320), [
    (320, 8, 32, 960, 7, _S1, [256, 224, 192, 160], [960, 880, 800, 720, 640, 560, 480], [320, 320, 0, 0], _odd),
    (640, 16, 48, 1920, 7, _S2, [768, 672, 576, 480], [1920, 1760, 1600, 1440, 1280, 1120, 960], [640, 640, 0, 0], _odd),
    (1280, 16, 64, 3840, 7, _S3, [1024, 896, 768, 640], [3840, 3520, 3200, 2880, 2560, 2240, 1920],
     [1280, 1280, 0, 0], _odd)])


# ---- single-stage patch-16 spaces of the sibling ViT supernet (reference supernet_config/tiny.py, tiny_deep.py, small_deep.py) ----
# The reference ships only the choice tables; `network_def` here is the largest member of each table (heads of 64 channels).
def _flat_space(name, dim, embed, attn, mlp, pattern, layer_a, layer_b):
    """pattern: per block 'b' (no layer choice), 'a' / 'c' (removable with layer choices layer_a / layer_b)."""
    blk = {'attn': np.array(attn), 'mlp': np.array(mlp), 'layer': None}
    tab = {'b': blk}
    for key, layer in (('a', layer_a), ('c', layer_b)):
        t = copy.deepcopy(blk)
        t['layer'] = np.array(layer)
        tab[key] = t
    keep = [np.array(embed)] + [tab[ch] for ch in pattern] + [None]
    m = types.ModuleType(__name__ + '.' + name)
    m.num_channels_to_keep = keep
    m.network_def = ((0, dim),) + ((1, (dim, attn[0] // 64, 64), (dim, mlp[0]), 1),) * len(pattern) + ((2, dim, 1000),)
    return m


tiny = _flat_space('tiny', 240, [240, 224, 208, 192], [512, 384, 256, 128], [1024, 768, 512, 256],
                   'b' + 'bbac' * 3 + 'b', [240, 240, 0], [240, 0])
tiny_deep = _flat_space('tiny_deep', 240, [240, 224, 208, 192], [384, 320, 256, 192], [960, 800, 640, 480],
                        'bb' + 'baba' * 3 + 'bb', [240, 240, 240, 0], [240, 240, 0, 0])
small_deep = _flat_space('small_deep', 384, [384, 352, 320, 288], [512, 448, 384, 320], [1536, 1280, 1024, 768],
                         'bb' + 'baba' * 3 + 'bb', [384, 384, 384, 0], [384, 384, 0, 0])
