#!/usr/bin/env python
"""LayerNorm forward / backward at the step's shapes: us and algorithmic TB/s (dev tool; run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-search_amd"))
import torch
from vitres import kernels as K
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for (B, N, C) in [(128, 257, 256), (128, 65, 512), (128, 17, 1024), (64, 257, 320)]:
    M = B * N
    x = torch.randn(B, N, C, device="cuda"); w = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    keep = torch.full((B,), C, dtype=torch.int32, device="cuda")
    y, mean, rstd = K.ln_fwd(x, w, b, keep, N, 1e-6, torch.bfloat16)
    dy = torch.randn(B, N, C, device="cuda").to(torch.bfloat16); g = torch.randn(B, N, C, device="cuda")
    dw = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    sc = torch.ones(B, device="cuda")
    # pre-allocate outputs? the wrappers allocate: include a plain run for reference
    t1 = timeit(lambda: K.ln_fwd(x, w, b, keep, N, 1e-6, torch.bfloat16))
    t2 = timeit(lambda: K.ln_bwd(dy, x, w, mean, rstd, keep, N, g, dw, db, next_cast=(sc, keep)))
    part = torch.zeros(2, 64, C, device="cuda")
    t3 = timeit(lambda: K.ln_bwd(dy, x, w, mean, rstd, keep, N, g, part[0], part[1], next_cast=(sc, keep), copies=64))
    t4 = timeit(lambda: K.ln_grad_reduce([(part[0], part[1], dw, db)] * 27, 64))
    print("%-16s fwd %6.1f us %5.2f TB/s   bwd %6.1f us %5.2f TB/s   bwd (64 partial rows) %6.1f us %5.2f TB/s   reduce x27 %5.1f us"
          % ("%d,%d,%d" % (B, N, C), t1 * 1e6, M * C * 6 / t1 / 1e12, t2 * 1e6, M * C * 16 / t2 / 1e12, t3 * 1e6,
             M * C * 16 / t3 / 1e12, t4 * 1e6))
