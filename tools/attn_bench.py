#!/usr/bin/env python
"""Micro-benchmark + parity check of vr_attn_fwd / vr_attn_bwd at the hot path's shapes (dev tool; run on the GPU box).
usage: attn_bench.py [check]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-search_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from vitres import kernels as K  # noqa: E402

dev = "cuda"
CHECK = len(sys.argv) > 1 and sys.argv[1] == "check"


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


shapes = [  # (B, N, H, D, kept heads per arch group)
    (128, 257, 4, 64, (4, 3)), (128, 65, 8, 64, (8, 6)), (128, 17, 12, 64, (12, 10)),
    (64, 257, 8, 32, (8, 6)), (64, 65, 12, 48, (12, 9)), (64, 17, 12, 64, (12, 12)),
    (128, 257, 6, 32, (6, 4)),
]
print("%-22s %9s %9s %9s %9s" % ("B,N,H,D", "fwd us", "fwd TF", "bwd us", "bwd TF"))
for B, N, H, D, kh in shapes:
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B, N, 3 * H * D, generator=g).to(dev).to(torch.bfloat16)
    d_o = torch.randn(B, N, H * D, generator=g).to(dev).to(torch.bfloat16)
    keep = torch.tensor([kh[0] * D] * (B // 2) + [kh[1] * D] * (B - B // 2), dtype=torch.int32, device=dev)
    scale = D ** -0.5
    o, lse = K.attn_fwd(qkv, keep, B, N, H, D, scale)
    heads = (kh[0] * (B // 2) + kh[1] * (B - B // 2))
    fl = 4.0 * N * N * D * heads
    t1 = timeit(lambda: K.attn_fwd(qkv, keep, B, N, H, D, scale))
    t2 = timeit(lambda: K.attn_bwd(qkv, o, d_o, lse, keep, B, N, H, D, scale))
    print("%-22s %9.1f %9.1f %9.1f %9.1f" % ("%d,%d,%d,%d" % (B, N, H, D), t1 * 1e6, fl / t1 / 1e12, t2 * 1e6, 2.5 * fl / t2 / 1e12))
    if CHECK:
        import emu_kernels as E
        nb = min(B, 6)
        sel = torch.cat([torch.arange(nb // 2), torch.arange(B - (nb - nb // 2), B)])
        qc, dc, kc = qkv[sel].cpu(), d_o[sel].cpu(), keep[sel].cpu()
        orf, lser = E.attn_fwd(qc, kc, nb, N, H, D, scale)
        dqr = E.attn_bwd(qc, orf, dc, lser, kc, nb, N, H, D, scale)
        dq = K.attn_bwd(qkv, o, d_o, lse, keep, B, N, H, D, scale)

        def rel(a, b):
            return float((a.float().cpu() - b.float()).abs().max() / b.float().abs().max())
        print("    check: o %.2e  lse %.2e  dqkv %.2e" % (rel(o[sel], orf), rel(lse[sel], lser), rel(dq[sel], dqr)))
