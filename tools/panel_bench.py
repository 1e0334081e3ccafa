#!/usr/bin/env python
"""Round 6: the stage-1 forward / data-gradient GEMMs (K = 256 / 320) alone, hipGraph of 20 launches each: tiled lean kernel
(sched 0x100000) | panel-resident kernel, 144-row / 80-row form | its no-MFMA byte probe (sched 0x400000) | the library's rule.
`python tools/panel_bench.py [--masked]`.  GB/s = algorithmic bytes (A once, weight once, outputs / side tensors once)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
from gemm_wide_bench import K  # noqa: E402

dev, bf = "cuda", torch.bfloat16
masked = "--masked" in sys.argv


def graph_time(fn, n=20, reps=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e-3)
    return best
NO, FORCE, P80, PROBE = 0x100000, 0x200000, 0x800000, 0x400000


def case(M, N, Kd, kind, rows):
    B = M // rows
    x = torch.randn(M, Kd, device=dev).to(bf)
    w = (torch.randn(N, Kd, device=dev) * Kd ** -0.5).to(bf)
    kw = dict(M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, rows_in=rows)
    out = torch.empty(M, N, device=dev, dtype=bf)
    by = (M * Kd + N * Kd + M * N) * 2
    if kind == "fwd":
        kw.update(bias=torch.randn(N, device=dev))
    elif kind == "gelu":
        kw.update(bias=torch.randn(N, device=dev), act=2, out2=torch.empty(M, N, device=dev, dtype=bf))
        by += M * N * 2
    elif kind == "dmul":
        kw.update(dact_u=torch.randn(M, N, device=dev).to(bf), ldu=N, act=2)
        by += M * N * 2
    if masked:
        kk = torch.full((B,), Kd, dtype=torch.int32)
        kn = torch.full((B,), N, dtype=torch.int32)
        kk[B // 2:] = 160
        kn[B // 2:] = (N * 5 // 8) // 64 * 64
        x = (x * (torch.arange(Kd, device=dev)[None, :] < kk.cuda().long().repeat_interleave(rows)[:, None])).contiguous()   # the contract of keep_k
        kw.update(keep_k=kk.cuda(), keep_n=kn.cuda(), m_groups=2)
    return x, w, out, kw, by


SHAPES = [(32896, 768, 256, "fwd", 257), (32896, 768, 256, "gelu", 257), (32896, 768, 256, "dmul", 257), (32896, 256, 256, "dgrad", 257),
          (16448, 960, 320, "fwd", 257), (16448, 960, 320, "gelu", 257), (16448, 960, 320, "dmul", 257), (16448, 320, 320, "dgrad", 257),
          (65792, 960, 320, "fwd", 257)]
COLS = [("tiled", NO), ("panel144", FORCE), ("panel80", FORCE | P80), ("probe144", FORCE | PROBE), ("noW144", FORCE | 0x1000000),
        ("noMFMA144", FORCE | 0x2000000), ("probe80", FORCE | P80 | PROBE)]
print("%-28s" % "M N K kind" + "".join("%20s" % c[0] for c in COLS))
for M, N, Kd, kind, rows in SHAPES:
    x, w, out, kw, by = case(M, N, Kd, kind, rows)
    line = "%-28s" % ("%d %d %d %s" % (M, N, Kd, kind))
    ref = None
    for name, sched in COLS:
        if name.endswith("144") and Kd > 256:
            line += "%20s" % "-"
            continue
        t = graph_time(lambda: K.gemm(x, w, out, sched=sched, **kw))
        tag = ""
        if name == "tiled":
            ref = out.float().clone()
        elif name.startswith("panel"):
            err = float((out.float() - ref).abs().max() / ref.abs().max())
            tag = "" if err < 2e-2 else " BAD%.0e" % err
        line += "%20s" % ("%.1f us %.2f TB/s%s" % (t * 1e6, by / t / 1e12, tag))
    print(line, flush=True)
