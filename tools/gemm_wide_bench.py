#!/usr/bin/env python
"""A/B of the forward / data-gradient GEMM kernels on the block Linears of the step (dev tool; run on the GPU box).

Every shape is timed with gemm_nt.hip's kernels (sched bit 0x100), the lean-loop kernels with three slices in flight (gemm_ntk.hip,
sched 3 << 9) and the library's own choice; `python tools/gemm_wide_bench.py [stage ...]`.  Operands are standard-normal bf16 (never zeros: DVFS)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-search_amd"))
import torch  # noqa: E402
from vitres import kernels as K  # noqa: E402

dev = "cuda"
bf = torch.bfloat16


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def case(M, N, Kd, kind, rows_in):
    """kind: fwd (bias, bf16) | gelu (bias, act 2, two outputs) | res (bias + residual + DropPath scale, fp32) | dgrad (k-major W)
    | dmul (k-major W, times the saved gelu')"""
    B = M // rows_in
    x = torch.randn(M, Kd, device=dev).to(bf)
    bias = torch.randn(N, device=dev)
    kw = dict(M=M, N=N, K=Kd, lda=Kd, ldc=N, rows_in=rows_in)
    if kind in ("dgrad", "dmul"):
        w = (torch.randn(Kd, N, device=dev) * Kd ** -0.5).to(bf)          # forward weight [out = K][in = N]
        kw.update(ldb=N, b_trans=True)
    else:
        w = (torch.randn(N, Kd, device=dev) * Kd ** -0.5).to(bf)
        kw.update(ldb=Kd)
    if kind == "fwd":
        out = torch.empty(M, N, device=dev, dtype=bf)
        kw.update(bias=bias)
    elif kind == "gelu":
        out = torch.empty(M, N, device=dev, dtype=bf)
        kw.update(bias=bias, act=2, out2=torch.empty(M, N, device=dev, dtype=bf))
    elif kind == "res":
        out = torch.empty(M, N, device=dev)
        kw.update(bias=bias, resid=torch.randn(M, N, device=dev), scale=torch.rand(B, device=dev) + 0.5)
    elif kind == "dmul":
        out = torch.empty(M, N, device=dev, dtype=bf)
        kw.update(dact_u=torch.randn(M, N, device=dev).to(bf), ldu=N, act=2)
    else:
        out = torch.empty(M, N, device=dev, dtype=bf)
    return x, w, out, kw


STAGES = {
    "s2": [(8320, 1536, 512, "fwd"), (8320, 1536, 512, "gelu"), (8320, 512, 1536, "res"), (8320, 512, 512, "res"),
           (8320, 512, 1536, "dgrad"), (8320, 1536, 512, "dmul"), (8320, 512, 512, "dgrad")],
    "s3": [(2176, 2304, 1024, "fwd"), (2176, 3072, 1024, "gelu"), (2176, 1024, 3072, "res"), (2176, 1024, 768, "res"),
           (2176, 1024, 3072, "dgrad"), (2176, 3072, 1024, "dmul"), (2176, 1024, 2304, "dgrad"), (2176, 768, 1024, "dgrad")],
    "s1": [(32896, 768, 256, "fwd"), (32896, 768, 256, "gelu"), (32896, 256, 768, "res"), (32896, 256, 256, "res"),
           (32896, 256, 768, "dgrad"), (32896, 768, 256, "dmul")],
    # 256 tiles of 256 x 256 (1024 of 128 x 128): one / four per CU exactly -- time = fixed cost per tile + slices x slope
    "probe": [(8192, 2048, k, kind) for kind in ("fwd", "res", "gelu", "dgrad") for k in (128, 256, 512, 1024, 2048, 4096)],
    # one 128 x 128 tile per CU (two 64 x 128): time = fixed cost + slices x slope
    "kprobe": [(8192, 512, k, kind) for kind in ("fwd", "res", "dgrad") for k in (128, 256, 512, 1024, 2048, 4096)],
    "skp": [(8320, 512, 1536, "res"), (8320, 512, 1536, "dgrad"), (2176, 1024, 3072, "res"), (2176, 1024, 2304, "dgrad"), (8320, 512, 512, "res")],
    "small": [(4160, 1920, 640, "fwd"), (4160, 640, 1920, "res"), (1088, 3840, 1280, "gelu"), (1088, 1280, 3840, "res"),
              (16448, 960, 320, "gelu"), (16448, 320, 960, "res")],
}
ROWS = {8320: 65, 2176: 17, 32896: 257, 4160: 65, 1088: 17, 16448: 257, 8192: 64}

def main():
    which = sys.argv[1:] or ["s2", "s3", "s1"]
    print("%-30s %9s %9s %9s   %s" % ("M N K kind", "old us", "lean3 us", "auto us", "dense TF/s (old / lean3 / auto)"))
    for st in which:
        for M, N, Kd, kind in STAGES[st]:
            x, w, out, kw = case(M, N, Kd, kind, ROWS[M])
            ts = []
            for sched in (0x100, 3 << 9, 0):
                ts.append(timeit(lambda: K.gemm(x, w, out, sched=sched, **kw)))
            fl = 2.0 * M * N * Kd
            print("%-30s %9.1f %9.1f %9.1f   %6.0f / %6.0f / %6.0f" % ("%d %d %d %s" % (M, N, Kd, kind), ts[0] * 1e6, ts[1] * 1e6,
                                                                     ts[2] * 1e6, fl / ts[0] / 1e12, fl / ts[1] / 1e12, fl / ts[2] / 1e12))


if __name__ == "__main__":
    main()
