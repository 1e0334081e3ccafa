"""DropPath (stochastic depth per sample), mirror of reference nets/drop.py:11-39.

In the HIP path the per-sample factor floor(keep_prob + U) / keep_prob is a float[B] vector that the
GEMM epilogue of the branch's last Linear multiplies in (include/vitres_hip.h: vr_gemm `scale`).
"""
import torch
import torch.nn as nn


def drop_path_scale(batch, drop_prob, device, generator=None):
    keep = 1.0 - drop_prob
    u = torch.rand(batch, dtype=torch.float32, device=device, generator=generator)
    return torch.floor(keep + u) / keep


class DropPath(nn.Module):
    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def scale(self, batch, device):
        if not self.drop_prob or not self.training:
            return None
        return drop_path_scale(batch, self.drop_prob, device)

    def extra_repr(self):
        return 'drop_prob={}'.format(self.drop_prob)
