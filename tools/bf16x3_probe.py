#!/usr/bin/env python
"""What a split-bf16 ("bf16x3") operand mode would cost and buy on the first stage's Linears (dev probe).

x = x_hi + x_lo, W = W_hi + W_lo (bf16 each); x W^T ~= x_hi W_hi^T + x_lo W_hi^T + x_hi W_lo^T with fp32 accumulation: exactly ONE
bf16 GEMM on K-concatenated operands [x_hi | x_lo | x_hi] . [W_hi | W_hi | W_lo]^T -- three times the slices and operand bytes of the
bf16 GEMM in the same kernels (no new code path): timed here against the bf16 and the exact-fp32 kernels, with the error of each
against an fp64 product."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-search_amd"))
import torch
from vitres import kernels as K
dev, bf = "cuda", torch.bfloat16


def timeit(fn, n=20):
    g = torch.cuda.CUDAGraph()
    fn(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * n) * 1e-3


print("%-26s %9s %9s %9s   %10s %10s %10s" % ("Linear (M, K -> N)", "bf16 us", "bf16x3 us", "fp32 us", "err bf16", "err bf16x3", "err fp32"))
tot = [0.0, 0.0, 0.0]
for (M, Kd, N, name) in [(32896, 256, 768, "qkv"), (32896, 256, 768, "fc1"), (32896, 256, 256, "proj"), (32896, 768, 256, "fc2"),
                         (8320, 512, 1536, "s2 qkv/fc1"), (8320, 1536, 512, "s2 fc2"), (2176, 1024, 3072, "s3 fc1"), (2176, 3072, 1024, "s3 fc2")]:
    x = torch.randn(M, Kd, device=dev); w = torch.randn(N, Kd, device=dev) * Kd ** -0.5
    xh, wh = x.to(bf), w.to(bf)
    xl, wl = (x - xh.float()).to(bf), (w - wh.float()).to(bf)
    x3 = torch.cat([xh, xl, xh], 1).contiguous(); w3 = torch.cat([wh, wh, wl], 1).contiguous()
    ref = (x[:4096].double() @ w.double().t())
    y1 = torch.empty(M, N, device=dev, dtype=bf); y3 = torch.empty(M, N, device=dev); yf = torch.empty(M, N, device=dev); y1f = torch.empty(M, N, device=dev)
    f1 = lambda: K.gemm(xh, wh, y1, M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, rows_in=257)
    f1f = lambda: K.gemm(xh, wh, y1f, M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, rows_in=257)
    f3 = lambda: K.gemm(x3, w3, y3, M=M, N=N, K=3 * Kd, lda=3 * Kd, ldb=3 * Kd, ldc=N, rows_in=257)
    ff = lambda: K.gemm(x, w, yf, M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, rows_in=257)
    t1, t3, tf = timeit(f1), timeit(f3), timeit(ff)
    f1f(); torch.cuda.synchronize()
    err = lambda y: float((y[:4096].double() - ref).abs().max() / ref.abs().max())
    print("%-26s %9.1f %9.1f %9.1f   %10.2e %10.2e %10.2e" % ("%s %d, %d -> %d" % (name, M, Kd, N), t1 * 1e6, t3 * 1e6, tf * 1e6, err(y1f), err(y3), err(yf)))
    for i, t in enumerate((t1, t3, tf)):
        tot[i] += t
print("%-26s %9.1f %9.1f %9.1f" % ("sum", tot[0] * 1e6, tot[1] * 1e6, tot[2] * 1e6))
