#!/usr/bin/env python
"""Split-K form of the 4-wave GEMM (sched bit 64) against the default schedule on the long-K shapes of stages 2 / 3 (dev tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-search_amd"))
import torch
from vitres import kernels as K

dev = "cuda"


def timeit(fn, n=40):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


print("%-30s %10s %10s" % ("form M,N,K", "default", "split"))
for M, N, Kd, rows in [(8320, 512, 1536, 65), (2176, 1024, 3072, 17), (2176, 1024, 2304, 17), (8320, 512, 512, 65), (2176, 1024, 768, 17),
                       (2176, 768, 1024, 17)]:
    a = torch.randn(M, Kd, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, Kd, device=dev) * Kd ** -0.5).to(torch.bfloat16)
    wt = (torch.randn(Kd, N, device=dev) * Kd ** -0.5).to(torch.bfloat16)
    bias, res = torch.randn(N, device=dev), torch.randn(M, N, device=dev)
    scale = torch.ones(M // rows, device=dev)
    o32, o16 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * N * Kd
    for name, fn in (("res", lambda s: K.gemm(a, w, o32, M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, bias=bias, resid=res, scale=scale,
                                              rows_in=rows, sched=s)),
                     ("dgrad", lambda s: K.gemm(a, wt, o16, M=M, N=N, K=Kd, lda=Kd, ldb=N, ldc=N, b_trans=True, rows_in=rows, sched=s))):
        t0, t1 = timeit(lambda: fn(0)), timeit(lambda: fn(64))
        print("%-30s %7.1f us %6.0f TF   %7.1f us %6.0f TF" % ("%s %d,%d,%d" % (name, M, N, Kd), t0 * 1e6, fl / t0 / 1e12, t1 * 1e6,
                                                             fl / t1 / 1e12))
