#!/usr/bin/env python
"""One launch shape of each bf16 GEMM kernel, a handful of launches, single stream: the workload for rocprofv3 --pmc
passes (HBM traffic per launch).  dev tool; run on the GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-search_amd"))
import torch  # noqa: E402
from vitres import kernels as K  # noqa: E402

M, Kd, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (32896, 320, 960)))
dt = torch.bfloat16
x = torch.randn(M, Kd, device="cuda").to(dt)
w = (torch.randn(N, Kd, device="cuda") * Kd ** -0.5).to(dt)
y = torch.empty(M, N, device="cuda", dtype=dt)
dy = torch.randn(M, N, device="cuda").to(dt)
dw = torch.zeros(N, Kd, device="cuda")
bias = torch.randn(N, device="cuda")
for _ in range(4):
    K.gemm(x, w, y, M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, bias=bias, rows_in=257)
    K.gemm(dy, x, dw, M=N, N=Kd, K=M, lda=N, ldb=Kd, ldc=Kd, a_trans=True, b_trans=True, atomic=True, split_k=0)
torch.cuda.synchronize()
print("algorithmic bytes: nt %.1f MB, tn %.1f MB" % ((M * Kd + N * Kd + M * N) * 2 / 1e6, ((M * N + M * Kd) * 2 + N * Kd * 4) / 1e6))
