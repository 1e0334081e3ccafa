"""Deterministic weight / input recipe shared by the golden generator and the tests.

Uses numpy's legacy RandomState (bit-stable across numpy versions) so that the same tensors can be
rebuilt on the GPU box without shipping multi-MB weight files.  Data only -- no reference code.
"""
import zlib

import numpy as np
import torch

# micro nets: img 56 -> 4x4 -> 2x2 -> 1x1 patches; dims satisfy the HIP kernels' alignment rules
MICRO_BLOCKS = ((1, (32, 2, 32), (32, 64), 1), (1, (32, 2, 32), (32, 64), 1), (3, 32, 64),
                (1, (64, 2, 48), (64, 128), 1), (1, (64, 2, 48), (64, 128), 1), (3, 64, 96),
                (1, (96, 2, 64), (96, 192), 1), (2, 96, 10))
MICRO_DEFS = {
    0: ((0, 32),) + MICRO_BLOCKS,
    4: ((4, 32),) + MICRO_BLOCKS,
    5: ((5, 32, 16),) + MICRO_BLOCKS,
}
MICRO_IMG = 56
MICRO_CLASSES = 10


def micro_keep_config():
    """num_channels_to_keep for the micro supernet (same grammar as supernet_config/*.py)."""
    b1 = {'attn': np.array([64, 32]), 'mlp': np.array([64, 48, 32]), 'layer': None}
    b1s = dict(b1, layer=np.array([32, 32, 0]))
    b2 = {'attn': np.array([96, 48]), 'mlp': np.array([128, 96, 64]), 'layer': None}
    b2s = dict(b2, layer=np.array([64, 0]))
    b3 = {'attn': np.array([128, 64]), 'mlp': np.array([192, 128]), 'layer': None}
    return [np.array([32, 24, 16]), b1, b1s, np.array([64, 48, 40]), b2, b2s,
            np.array([96, 80]), b3, None]


# four evo-search style candidates carved out of the micro supernet (incl. a removed block)
MICRO_CANDIDATES = [
    ((0, 24), (1, (24, 1, 32), (24, 48), 1), (1, (24, 2, 32), (24, 32), 0), (3, 24, 48),
     (1, (48, 2, 48), (48, 96), 1), (1, (48, 1, 48), (48, 64), 1), (3, 48, 80),
     (1, (80, 1, 64), (80, 128), 1), (2, 80, 10)),
    ((0, 32), (1, (32, 2, 32), (32, 64), 1), (1, (32, 2, 32), (32, 64), 1), (3, 32, 64),
     (1, (64, 2, 48), (64, 128), 1), (1, (64, 2, 48), (64, 128), 1), (3, 64, 96),
     (1, (96, 2, 64), (96, 192), 1), (2, 96, 10)),
    ((0, 16), (1, (16, 1, 32), (16, 32), 1), (1, (16, 1, 32), (16, 48), 1), (3, 16, 40),
     (1, (40, 1, 48), (40, 64), 0), (1, (40, 2, 48), (40, 128), 1), (3, 40, 96),
     (1, (96, 2, 64), (96, 128), 1), (2, 96, 10)),
    ((0, 24), (1, (24, 2, 32), (24, 64), 1), (1, (24, 1, 32), (24, 48), 1), (3, 24, 64),
     (1, (64, 1, 48), (64, 96), 1), (1, (64, 2, 48), (64, 64), 1), (3, 64, 80),
     (1, (80, 2, 64), (80, 192), 1), (2, 80, 10)),
]

REF_TINY_DEF = ((4, 192),) + ((1, (192, 3, 64), (192, 768), 1),) * 4 + ((3, 192, 384),) + \
    ((1, (384, 6, 64), (384, 1536), 1),) * 4 + ((3, 384, 768),) + \
    ((1, (768, 12, 64), (768, 3072), 1),) * 4 + ((2, 768, 1000),)
SR_TINY_DEF = ((0, 256),) + ((1, (256, 4, 64), (256, 768), 1),) * 7 + ((3, 256, 512),) + \
    ((1, (512, 8, 64), (512, 1536), 1),) * 7 + ((3, 512, 1024),) + \
    ((1, (1024, 12, 64), (1024, 3072), 1),) * 4 + ((2, 1024, 1000),)
SR_SMALL_DEF = ((5, 320, 32),) + ((1, (320, 8, 32), (320, 960), 1),) * 7 + ((3, 320, 640),) + \
    ((1, (640, 12, 48), (640, 1920), 1),) * 7 + ((3, 640, 1280),) + \
    ((1, (1280, 12, 64), (1280, 3840), 1),) * 7 + ((2, 1280, 1000),)
SR_TINY_MH_DEF = ((4, 256),) + ((1, (256, 6, 32), (256, 768), 1),) * 6 + ((3, 256, 512),) + \
    ((1, (512, 12, 48), (512, 1536), 1),) * 6 + ((3, 512, 1024),) + \
    ((1, (1024, 12, 64), (1024, 3072), 1),) * 6 + ((2, 1024, 1000),)
SR_SMALL_MH_DEF = ((4, 320),) + ((1, (320, 8, 32), (320, 960), 1),) * 7 + ((3, 320, 640),) + \
    ((1, (640, 16, 48), (640, 1920), 1),) * 7 + ((3, 640, 1280),) + \
    ((1, (1280, 16, 64), (1280, 3840), 1),) * 7 + ((2, 1280, 1000),)
README_SEARCHED_DEF = None  # filled by make_golden from the reference's own __main__ block if needed


def fill_state_dict(shapes, seed):
    """shapes: ordered list of (key, shape).  Returns {key: float32 tensor} (int64 for counters)."""
    rs = np.random.RandomState(seed)
    out = {}
    for key, shape in shapes:
        shape = tuple(shape)
        if key.endswith('num_batches_tracked'):
            out[key] = torch.zeros((), dtype=torch.int64)
            continue
        n = rs.standard_normal(shape).astype(np.float32) if len(shape) else np.float32(rs.standard_normal())
        if key.endswith('running_var'):
            v = np.abs(n) + 0.5
        elif key.endswith('running_mean'):
            v = 0.1 * n
        elif len(shape) >= 2 and not key.endswith('pos_embed') and key != 'tokens':
            fan_in = int(np.prod(shape[1:]))
            v = n * (0.8 / np.sqrt(fan_in))
        elif key.endswith('pos_embed') or key == 'tokens':
            v = 0.2 * n
        elif key.endswith('.weight'):          # LN / BN scale
            v = 1.0 + 0.1 * n
        else:                                  # biases
            v = 0.05 * n
        out[key] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return out


def inputs(seed, batch, img, classes, n_patches):
    rs = np.random.RandomState(seed)
    x = torch.from_numpy(rs.standard_normal((batch, 3, img, img)).astype(np.float32))
    t = rs.standard_normal((batch, classes)).astype(np.float32)
    t = np.exp(t) / np.exp(t).sum(-1, keepdims=True)
    pt = rs.standard_normal((batch, n_patches, classes)).astype(np.float32)
    pt = np.exp(pt) / np.exp(pt).sum(-1, keepdims=True)
    labels = rs.randint(0, classes, size=(batch,)).astype(np.int64)
    return x, torch.from_numpy(t), torch.from_numpy(pt), torch.from_numpy(labels)


def checksum(sd):
    c = 0
    for k in sd:
        c = zlib.crc32(sd[k].detach().cpu().contiguous().numpy().tobytes(), c)
    return c


def toy_candidate_score(network_def):
    """Deterministic stand-in for a candidate's accuracy (F13: drives the evolver without a trained supernet)."""
    s = 0.0
    for i, e in enumerate(network_def):
        if e[0] == 1 and e[3]:
            s += (1.0 + 0.013 * i) * (e[1][1] * e[1][2] * 1.7 + e[2][1] * 0.41) * (e[1][0] ** 0.5)
    return round(s / 1e4, 6)


def toy_teacher(classes):
    """Fixed stand-in for the KD teacher (the reference uses a pretrained RegNet from timm, main.py:370-382): global average of the
    image -> Linear(3 -> classes) with recipe weights.  Any module mapping images to logits works; this one travels with the repo."""
    import torch
    lin = torch.nn.Linear(3, classes)
    g = torch.Generator().manual_seed(77)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(classes, 3, generator=g))
        lin.bias.copy_(torch.randn(classes, generator=g) * 0.1)

    class Teacher(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = lin

        def forward(self, x):
            return self.lin(x.float().mean(dim=(2, 3)))
    return Teacher()


# single-stage patch-16 sibling (nets/vision_transformer_supernet.py), micro size: 64 px -> 4 x 4 patches
VIT16_IMG = 64
VIT16_DEF = ((0, 48),) + ((1, (48, 2, 32), (48, 96), 1),) * 4 + ((2, 48, MICRO_CLASSES),)


def vit16_keep_config():
    import numpy as np
    blk = {'attn': np.array([64, 32]), 'mlp': np.array([96, 64, 48]), 'layer': None}
    skip = {'attn': np.array([64, 32]), 'mlp': np.array([96, 64, 48]), 'layer': np.array([48, 0])}
    return [np.array([48, 40, 32]), blk, skip, blk, skip, None]


# ---- F18: DropPath draws and gradient sampling -------------------------------------------------------------------
GRAD_SAMPLES = 384


def drop_path_noise(seed, rates, batch):
    """Uniform [0,1) draws of the DropPath calls of one forward, call order (two per block with rate > 0: attention branch, MLP
    branch), caller sample order.  In every row one sample is forced below the rate (dropped) and one above it (kept), so
    that zeros and 1/keep_prob scales both occur whatever the seed."""
    rs = np.random.RandomState(seed)
    rows = []
    for j, r in enumerate(rates):
        for h in range(2):
            u = rs.uniform(0.0, 1.0, size=batch).astype(np.float32)
            u[(2 * j + h) % batch] = np.float32(0.5 * r)
            u[(2 * j + h + 3) % batch] = np.float32(0.5 * (1.0 + r))
            rows.append(u)
    return rows


def grad_sample_index(name, numel):
    """Deterministic element positions at which a parameter gradient is sampled into the full-size fixtures."""
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7fffffff)
    n = min(GRAD_SAMPLES, numel)
    return np.sort(rs.choice(numel, size=n, replace=False)).astype(np.int64) if numel > n else np.arange(numel, dtype=np.int64)
