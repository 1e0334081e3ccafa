#!/usr/bin/env python
"""vr_gemm_ln (Linear + LayerNorm fused) at the step's shapes: us and algorithmic TB/s per mode (dev tool; run on the GPU box).
   usage: tools/ntln_bench.py [masked]      masked: per-sample kept widths like a supernet step (3/4 of the rows at 3/4 width)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-search_amd"))
import torch
from vitres import kernels as K

masked = len(sys.argv) > 1 and sys.argv[1] == "masked"
NROT = int(os.environ.get("NROT", "3"))    # rotate over operand sets: more than the 256 MB of last-level cache at stage 1


def timeit(fn, n=30):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for (B, T, C, Kd) in [(128, 257, 256, 768), (128, 257, 256, 256), (128, 65, 512, 1536), (128, 65, 512, 512), (64, 257, 320, 960)]:
    M = B * T
    dev = "cuda"
    sets = []
    for r in range(NROT):
        g = torch.Generator(device=dev).manual_seed(r)
        a = (torch.randn(M, Kd, device=dev, generator=g) * 0.5).to(torch.bfloat16)
        w = (torch.randn(C, Kd, device=dev, generator=g) * 0.05).to(torch.bfloat16)
        res = torch.randn(M, C, device=dev, generator=g)
        sets.append((a, w, res, torch.empty(M, C, device=dev)))
    lw, lb, bias = torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    keep = kk = None
    if masked:
        keep = torch.full((B,), C, dtype=torch.int32, device=dev)
        keep[: 3 * B // 4] = C * 3 // 4 // 8 * 8
        kk = torch.full((B,), Kd, dtype=torch.int32, device=dev)
        kk[: 3 * B // 4] = Kd * 3 // 4 // 64 * 64
    y, mean, rstd = K.gemm_ln_fwd(sets[0][0], sets[0][1], sets[0][3], lw, lb, keep, 1e-6, M=M, N=C, K=Kd, lda=Kd, ldb=Kd, ldc=C, bias=bias,
                                  resid=sets[0][2], rows_in=T, keep_k=kk, keep_n=keep)
    dw, db = torch.zeros(64, C, device=dev), torch.zeros(64, C, device=dev)
    sc = torch.ones(B, device=dev)
    kp = keep if keep is not None else torch.full((B,), C, dtype=torch.int32, device=dev)

    def fwd(i):
        a, w, res, out = sets[i % NROT]
        K.gemm_ln_fwd(a, w, out, lw, lb, keep, 1e-6, M=M, N=C, K=Kd, lda=Kd, ldb=Kd, ldc=C, bias=bias, resid=res, rows_in=T, keep_k=kk,
                      keep_n=keep)

    def bwd(i):
        a, w, res, out = sets[i % NROT]
        K.gemm_ln_bwd(a, w, out.view(B, T, C), lw, mean, rstd, keep, res, dw, db, next_cast=(sc, kp), M=M, N=C, K=Kd, lda=Kd, ldb=Kd,
                      rows_in=T, keep_k=kk, copies=64)

    if not K.gemm_ln_supported(sets[0][0], C, C):
        continue
    t0, t1 = timeit(fwd), timeit(bwd)
    b0 = M * Kd * 2 + M * C * 10
    b1 = M * Kd * 2 + M * C * 14
    print("M%-6d N%-4d K%-5d fwd %6.1f us %5.2f TB/s %6.1f TF   bwd %6.1f us %5.2f TB/s %6.1f TF" %
          (M, C, Kd, t0 * 1e6, b0 / t0 / 1e12, 2.0 * M * C * Kd / t0 / 1e12, t1 * 1e6, b1 / t1 / 1e12, 2.0 * M * C * Kd / t1 / 1e12))
