#!/usr/bin/env python
"""Run a fixed list of vr_gemm launches (3x each) for rocprofv3 --kernel-trace; tools/rocpd_list.py prints durations in order."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-search_amd"))
import torch
from vitres import kernels as K
dev = "cuda"; dt = torch.bfloat16
M = 32896
cases = []
for Kd in (64, 256, 512, 1024, 2048):
    cases.append(("fwd_bf16out K=%d N=768" % Kd, M, Kd, 768, "fwd"))
for N in (128, 256, 768, 1536):
    cases.append(("fwd_bf16out K=256 N=%d" % N, M, 256, N, "fwd"))
cases.append(("gelu K=256 N=768", M, 256, 768, "gelu"))
cases.append(("nostore K=64", M, 64, 768, "act100"))
cases.append(("nobias K=64", M, 64, 768, "act101"))
cases.append(("nostore K=256", M, 256, 768, "act100"))
cases.append(("resid_f32 K=768 N=256", M, 768, 256, "resid"))
cases.append(("dgrad K=768 N=256", M, 768, 256, "dgrad"))
cases.append(("wgrad 768x256", M, 256, 768, "wgrad"))
cases.append(("fwd small M=2176 K=1024 N=3072", 2176, 1024, 3072, "fwd"))
for name, m, k, n, kind in cases:
    x = torch.randn(m, k, device=dev).to(dt); w = (torch.randn(n, k, device=dev) * k ** -0.5).to(dt)
    bias = torch.randn(n, device=dev)
    if kind == "fwd":
        y = torch.empty(m, n, device=dev, dtype=dt)
        f = lambda: K.gemm(x, w, y, M=m, N=n, K=k, lda=k, ldb=k, ldc=n, bias=bias, rows_in=257)
    elif kind.startswith("act"):
        y = torch.empty(m, n, device=dev, dtype=dt); a_ = int(kind[3:])
        f = lambda: K.gemm(x, w, y, M=m, N=n, K=k, lda=k, ldb=k, ldc=n, bias=bias, rows_in=257, act=a_)
    elif kind == "gelu":
        y = torch.empty(m, n, device=dev, dtype=dt); y2 = torch.empty_like(y)
        f = lambda: K.gemm(x, w, y, out2=y2, M=m, N=n, K=k, lda=k, ldb=k, ldc=n, bias=bias, act=1, rows_in=257)
    elif kind == "resid":
        y = torch.empty(m, n, device=dev); r = torch.randn(m, n, device=dev)
        f = lambda: K.gemm(x, w, y, M=m, N=n, K=k, lda=k, ldb=k, ldc=n, bias=bias, resid=r, rows_in=257)
    elif kind == "dgrad":
        dy = torch.randn(m, n, device=dev).to(dt); dx = torch.empty(m, k, device=dev, dtype=dt)
        f = lambda: K.gemm(dy, w, dx, M=m, N=k, K=n, lda=n, ldb=k, ldc=k, b_trans=True, rows_in=257)
    else:
        dy = torch.randn(m, n, device=dev).to(dt); dw = torch.zeros(n, k, device=dev)
        f = lambda: K.gemm(dy, x, dw, M=n, N=k, K=m, lda=n, ldb=k, ldc=k, a_trans=True, b_trans=True, atomic=True, split_k=64)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    print(name)
