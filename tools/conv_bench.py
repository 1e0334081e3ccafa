#!/usr/bin/env python
"""The stem's direct 3x3 convolution alone at the workloads' shapes (dev tool): us and TB/s of input + output bytes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-search_amd"))
import torch
from vitres import kernels as K


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


print("VITRES_CONV_WGS=%s" % os.environ.get("VITRES_CONV_WGS", "3"))
for B, m in ((256, 32), (128, 32), (128, 24), (64, 32), (128, 16)):
    H = W = 112
    a = torch.randn(B * H * W, m, device="cuda").bfloat16()
    w = (torch.randn(m, 9 * m, device="cuda") * 0.1).bfloat16()
    bias = torch.randn(m, device="cuda")
    for name, fn in (("conv3x3 bf16", lambda: K.conv3x3(a, w, B, H, W, m, m, torch.bfloat16)),
                     ("conv3x3 f32 ", lambda: K.conv3x3(a, w, B, H, W, m, m, torch.float32)),
                     ("bias_relu   ", lambda: K.conv3x3_bias_relu(a, w, bias, a, B, H, W, m, m, torch.bfloat16))):
        t = timeit(fn)
        by = B * H * W * m * (2 + (4 if "f32" in name else 2) + (2 if "bias" in name else 0))
        print("B%-3d m%-2d %s %7.1f us  %5.2f TB/s  %6.1f TF/s" % (B, m, name, t * 1e6, by / t / 1e12, 2.0 * B * H * W * m * 9 * m / t / 1e12))
