#!/usr/bin/env python
"""The first-stage GEMMs of a candidate-scoring step (C5: 256 images, sr_small widths) alone, by tile choice (dev tool):
VITRES_NT_TILE=0/1/2/3 python tools/c5_gemm_bench.py"""
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


dev, bf = "cuda", torch.bfloat16
M = 256 * 257
print("VITRES_NT_TILE=%s" % os.environ.get("VITRES_NT_TILE", "0"))
for (Kd, N, kind) in [(320, 960, "fc1"), (320, 960, "relu"), (320, 960, "qkv"), (320, 768, "fc1"), (320, 768, "qkv"), (960, 320, "fc2"),
                      (256, 320, "proj"), (320, 1280, "fc1"), (320, 1280, "qkv")]:
    x = torch.randn(M, Kd, device=dev).to(bf); w = (torch.randn(1280, 1280, device=dev) * Kd ** -0.5).to(bf)[:N, :Kd]
    ldw = 1280
    bias = torch.randn(N, device=dev)
    if kind in ("qkv", "fc1", "relu"):
        y = torch.empty(M, N, device=dev, dtype=bf)
        fn = lambda: K.gemm(x, w, y, M=M, N=N, K=Kd, lda=Kd, ldb=ldw, ldc=N, bias=bias, act=(1 if kind == "fc1" else (3 if kind == "relu" else 0)), rows_in=257)
        by = (M * Kd + N * Kd + M * N) * 2
    else:
        y = torch.empty(M, N, device=dev); r = torch.randn(M, N, device=dev)
        fn = lambda: K.gemm(x, w, y, M=M, N=N, K=Kd, lda=Kd, ldb=ldw, ldc=N, bias=bias, resid=r, rows_in=257)
        by = (M * Kd + N * Kd) * 2 + 2 * M * N * 4
    t = timeit(fn)
    print("%-22s %8.1f us %8.1f TF/s %6.2f TB/s" % ("%s K%d N%d" % (kind, Kd, N), t * 1e6, 2.0 * M * N * Kd / t / 1e12, by / t / 1e12))
