#!/usr/bin/env python
"""Forward / dgrad (K-contiguous weights) GEMM shapes of the step with the epilogues they run with (dev tool)."""
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

dev = "cuda"; bf = torch.bfloat16
print("%-34s %8s %8s %8s" % ("case", "us", "TF/s", "TB/s"))
for (M, Kd, N, kind) in [(32896, 256, 768, "qkv"), (32896, 256, 768, "fc1"), (32896, 768, 256, "fc2"), (32896, 256, 256, "proj"),
                         (32896, 768, 256, "dgelu"), (8320, 512, 1536, "fc1"), (8320, 1536, 512, "fc2"), (2176, 1024, 3072, "fc1"), (2176, 3072, 1024, "fc2")]:
    x = torch.randn(M, Kd, device=dev).to(bf); w = (torch.randn(N, Kd, device=dev) * Kd ** -0.5).to(bf)
    bias = torch.randn(N, device=dev)
    if kind in ("qkv",):
        y = torch.empty(M, N, device=dev, dtype=bf)
        fn = lambda: K.gemm(x, w, y, M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, bias=bias, rows_in=257)
        by = (M * Kd + N * Kd + M * N) * 2
    elif kind == "fc1":
        y = torch.empty(M, N, device=dev, dtype=bf); h = torch.empty(M, N, device=dev, dtype=bf)
        fn = lambda: K.gemm(x, w, y, out2=h, M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, bias=bias, act=1, rows_in=257)
        by = (M * Kd + N * Kd + 2 * M * N) * 2
    elif kind in ("fc2", "proj"):
        y = torch.empty(M, N, device=dev); r = torch.randn(M, N, device=dev)
        fn = lambda: K.gemm(x, w, y, M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, bias=bias, resid=r, rows_in=257)
        by = (M * Kd + N * Kd) * 2 + 2 * M * N * 4
    else:
        # du = (gt @ W2) * gelu'(u): M x K(=C) @ [N(=F) x K]^T
        Kd2, N2 = N, Kd
        g = torch.randn(M, Kd2, device=dev).to(bf); wt = (torch.randn(N2, Kd2, device=dev) * 0.05).to(bf)
        u = torch.randn(M, N2, device=dev).to(bf); du = torch.empty(M, N2, device=dev, dtype=bf)
        fn = lambda: K.gemm(g, wt, du, M=M, N=N2, K=Kd2, lda=Kd2, ldb=Kd2, ldc=N2, dact_u=u, ldu=N2, rows_in=257)
        by = (M * Kd2 + N2 * Kd2 + 2 * M * N2) * 2
    t = timeit(fn)
    print("%-34s %8.1f %8.1f %8.2f" % ("%s %d,%d,%d" % (kind, M, Kd, N), t * 1e6, 2.0 * M * N * Kd / t / 1e12, by / t / 1e12))
