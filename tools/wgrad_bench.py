#!/usr/bin/env python
"""The weight-gradient group launches of the step alone (dev tool): one vr_gemm_group per transformer block of each stage, replayed
from a hipGraph (the host needs ~10 us per eager launch).  VITRES_DBG_TN=1 / 2: without the atomics / without the K loop."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-search_amd"))
import torch
from vitres import kernels as K
dev, bf = "cuda", torch.bfloat16
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("VITRES_"))
print("# " + (tag or "default"))
for (T, C, F, HD, name) in [(32896, 256, 768, 256, "stage 1"), (8320, 512, 1536, 512, "stage 2"), (2176, 1024, 3072, 768, "stage 3")]:
    r = lambda *s: torch.randn(*s, device=dev).to(bf)
    xn, dqkv, ao, gt, xn2, du, h = r(T, C), r(T, 3 * HD), r(T, HD), r(T, C), r(T, C), r(T, F), r(T, F)
    dws = [torch.zeros(3 * HD, C, device=dev), torch.zeros(C, HD, device=dev), torch.zeros(F, C, device=dev), torch.zeros(C, F, device=dev)]
    dbs = [torch.zeros(3 * HD, device=dev), torch.zeros(C, device=dev), torch.zeros(F, device=dev), torch.zeros(C, device=dev)]
    def call(dy, x, dw, db):
        No, Ki = dw.shape
        return (dy, x, dw, dict(M=No, N=Ki, K=T, lda=No, ldb=Ki, ldc=Ki, a_trans=True, b_trans=True, atomic=True, split_k=0, bias_grad=db))
    calls = [call(gt, h, dws[3], dbs[3]), call(du, xn2, dws[2], dbs[2]), call(gt, ao, dws[1], dbs[1]), call(dqkv, xn, dws[0], dbs[0])]
    K.ensure_workspaces(torch.device(dev, torch.cuda.current_device()), roles=(0, 1))
    for _ in range(3):
        K.gemm_group(calls)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            K.gemm_group(calls)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 50 * 1e-3
    fl = 2.0 * T * (3 * HD * C + C * HD + 2 * F * C)
    by = 2.0 * T * (3 * HD + C + HD + C + F + C + F + C)
    print("%-8s group: %7.1f us  %6.0f TF/s dense  %5.2f TB/s operands once" % (name, t * 1e6, fl / t / 1e12, by / t / 1e12))
