#!/usr/bin/env python
"""Micro-benchmark of vr_gemm on the shapes of the hot path (dev tool; run on the GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-search_amd"))
import torch  # noqa: E402
from vitres import kernels as K  # noqa: E402

dev = "cuda"
dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
SCHED = int(os.environ.get("GEMM_SCHED", "0"))


def timeit(fn, n=20):
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


shapes = [  # (tokens M, in K, out N) of Linear layers at B=128
    (32896, 320, 960), (32896, 320, 1280), (32896, 1280, 320), (32896, 256, 768), (32896, 256, 256), (32896, 768, 256), (8320, 512, 1536), (8320, 1536, 512),
    (2176, 1024, 3072), (2176, 3072, 1024), (2176, 1024, 1024), (8320, 512, 512), (8320, 1536, 512), (2176, 1024, 2304), (32896, 192, 576), (32896, 192, 768), (32896, 768, 192),
]
print("%-26s %10s %10s %10s" % ("M,K,N", "fwd TF", "dgrad TF", "wgrad TF"))
for M, Kd, N in shapes:
    x = torch.randn(M, Kd, device=dev).to(dt)
    w = (torch.randn(N, Kd, device=dev) * Kd ** -0.5).to(dt)
    y = torch.empty(M, N, device=dev, dtype=dt)
    dy = torch.randn(M, N, device=dev).to(dt)
    dx = torch.empty(M, Kd, device=dev, dtype=dt)
    dw = torch.zeros(N, Kd, device=dev)
    bias = torch.randn(N, device=dev)
    fl = 2.0 * M * N * Kd
    t1 = timeit(lambda: K.gemm(x, w, y, M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, bias=bias, rows_in=257, sched=SCHED))
    t2 = timeit(lambda: K.gemm(dy, w, dx, M=M, N=Kd, K=N, lda=N, ldb=Kd, ldc=Kd, b_trans=True, rows_in=257, sched=SCHED))
    sk = max(1, min(64, M // 512))
    t3 = timeit(lambda: K.gemm(dy, x, dw, M=N, N=Kd, K=M, lda=N, ldb=Kd, ldc=Kd, a_trans=True, b_trans=True, atomic=True,
                               split_k=0, sched=SCHED))
    print("%-26s %10.1f %10.1f %10.1f   (%.0f / %.0f / %.0f us)" % ("%d,%d,%d" % (M, Kd, N), fl / t1 / 1e12, fl / t2 / 1e12,
                                                                 fl / t3 / 1e12, t1 * 1e6, t2 * 1e6, t3 * 1e6))
