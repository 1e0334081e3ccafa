"""Loss functions of the training loop (reference main.py:390-398 picks timm's SoftTargetCrossEntropy when
mixup / patch-mixup is on; engine.py:153-157 sums the class-token and patch-token terms)."""
import torch
import torch.nn as nn

from . import kernels as K


class _SoftCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        rows = logits.numel() // logits.shape[-1]
        loss_rows, dlogits = K.softce(logits.contiguous(), target.contiguous(), 1.0 / rows,
                                      want_grad=ctx.needs_input_grad[0])
        ctx.save_for_backward(dlogits)
        return loss_rows.mean()

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return d * g, None


class SoftTargetCrossEntropy(nn.Module):
    """mean over rows of sum_k -target * log_softmax(x)  (timm 0.3.2 semantics; works for (B,K) and (B,P,K))."""

    def forward(self, x, target):
        return _SoftCE.apply(x.float(), target.float())
