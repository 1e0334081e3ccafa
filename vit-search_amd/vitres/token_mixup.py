"""SwitchTokenMix for the HIP path (reference token_mixup.py:39-162; constructed in main.py:316-322 as `patch_mixup_fn`).

Every call mixes the first half of the batch at patch level (a random box of the patch_len x patch_len grid is taken from a
permuted partner; the per-patch soft targets follow the box) and the second half at image level (mixup with lam ~ Beta(.8,.8)).
The random draws use the SAME generators in the SAME order as the reference -- torch's CPU generator for the two
permutations, numpy's global generator for the Beta samples and the box -- so a run seeded like the reference sees the same
mixes; the tensor work is one `vr_token_mix` call (two streaming kernels).  Returns the reference's 4-tuple
`(samples, targets, patch_targets, 'seq')`; unlike the reference the input batch is not modified in place (`out=` lets the
caller mix straight into e.g. the static input buffer of a captured hipGraph).
"""
import numpy as np
import torch

from . import _lib
from .kernels import _p, _stream


class SwitchTokenMix:
    def __init__(self, patch_len, switch_prob=0.5, num_classes=1000, smoothing=0.1):
        self.patch_len, self.switch_prob, self.num_classes, self.smoothing = patch_len, switch_prob, num_classes, smoothing

    def __repr__(self):
        return '(patch_len={}, switch_prob={})'.format(self.patch_len, self.switch_prob)

    def draw(self, batch):
        """Host-side randomness of one call (reference order: token_mixup.py:149-150 -> 104-106 -> 74-95, then 126-127)."""
        pl, h = self.patch_len, batch // 2
        perm_a = torch.randperm(h)

        def randint(lo, hi, size=None):
            return np.random.randint(lo, lo + 1 if lo == hi else hi, size=size)
        lam = np.random.beta(1., 1.)
        area = int(pl * pl * lam)
        ch = randint(1, max(1, min(pl, area) - 1))
        cw = area // ch
        if cw > pl:
            cw, ch = pl, area // pl
        y0 = int(randint(0, max(0, pl - ch), size=2)[1])
        x0 = int(randint(0, max(0, pl - cw), size=2)[1])
        lam_patch = 1 - (ch * cw + 0.0) / (pl * pl)
        perm_b = torch.randperm(batch - h)
        lam_img = np.random.beta(0.8, 0.8)
        partner = torch.cat([perm_a, perm_b + h])
        return dict(half=h, partner=partner, box=(y0, y0 + int(ch), x0, x0 + int(cw)), lam_patch=float(lam_patch),
                    lam_img=float(lam_img))

    def __call__(self, samples, targets, out=None, draw=None):
        if not samples.is_cuda:
            raise RuntimeError('vitres.token_mixup runs on the GPU through libvitres_hip.so (CPU restatement: oracle/)')
        B, C, H, W = samples.shape
        d = draw or self.draw(B)
        K, pl = self.num_classes, self.patch_len
        x = samples.contiguous().float()
        out = torch.empty_like(x) if out is None else out
        new_targets = torch.empty((B, K), dtype=torch.float32, device=x.device)
        patch_targets = torch.empty((B, pl * pl, K), dtype=torch.float32, device=x.device)
        partner = d["partner"].to(x.device, non_blocking=True)
        labels = targets.to(x.device).long().contiguous()
        off = self.smoothing / K
        on = 1. - self.smoothing + off
        y0, y1, x0, x1 = d["box"]
        f32 = lambda v: float(np.float32(v))                              # noqa: E731  torch rounds python scalars to fp32
        _lib.check(_lib.lib().vr_token_mix(_p(x), _p(out), _p(labels), _p(partner), _p(new_targets), _p(patch_targets), B, C, H,
                                           W, K, pl, d["half"], y0, y1, x0, x1, f32(d["lam_patch"]),
                                           f32(1. - d["lam_patch"]), f32(d["lam_img"]), f32(1. - d["lam_img"]), f32(on),
                                           f32(off), _stream()), "vr_token_mix")
        return out, new_targets, patch_targets, 'seq'
