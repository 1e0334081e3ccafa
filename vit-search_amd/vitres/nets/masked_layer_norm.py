"""MaskedLayerNorm: LayerNorm over each sample's active channel prefix (reference
nets/masked_layer_norm.py:91-130), executed by the vr_ln_fwd / vr_ln_bwd HIP kernels."""
import torch
import torch.nn as nn

from .. import kernels as K


class _MaskedLayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, keep, out_dtype):
        xs = x.contiguous().float()
        rows = xs.shape[-2] if xs.dim() >= 3 else 1
        y, mean, rstd = K.ln_fwd(xs, weight, bias, keep, rows, eps, out_dtype)
        ctx.save_for_backward(xs, weight, mean, rstd)
        ctx.keep, ctx.rows = keep, rows
        return y

    @staticmethod
    def backward(ctx, gy):
        xs, weight, mean, rstd = ctx.saved_tensors
        dw = torch.zeros_like(weight)
        db = torch.zeros_like(weight)
        gy = gy.contiguous()
        if gy.dtype not in (torch.float32, torch.bfloat16):
            gy = gy.float()
        dx = K.ln_bwd(gy, xs, weight, mean, rstd, ctx.keep, ctx.rows, None, dw, db)
        return dx, dw, db, None, None, None


class MaskedLayerNorm(nn.Module):
    """x: (B, N, C) fp32; mask: None, an int32 keep-count vector (B,) on the device, or the reference's
    (B,1,C) bool prefix mask.  Always affine, eps 1e-6."""

    def __init__(self, num_channels, eps=1e-6):
        super().__init__()
        self.register_parameter('weight', nn.Parameter(torch.ones(num_channels)))
        self.register_parameter('bias', nn.Parameter(torch.zeros(num_channels)))
        self.eps = eps
        self.num_channels = num_channels
        self.normalized_shape = (num_channels,)

    def forward(self, x, mask=None, out_dtype=torch.float32):
        keep = mask
        if mask is not None and mask.dtype == torch.bool:
            keep = mask.reshape(mask.shape[0], -1).sum(dim=1).to(torch.int32)
        return _MaskedLayerNormFn.apply(x, self.weight, self.bias, self.eps, keep, out_dtype)

    def extra_repr(self):
        return 'num_channels={}, eps={}'.format(self.num_channels, self.eps)
