#!/usr/bin/env python
"""Round 6: the stage 2 / 3 forward and data-gradient GEMMs alone (hipGraph of 20 launches each: no host floor) under the lean
kernel's tile x ring x K-share choices -- `python tools/ksplit_sweep.py [s2 s3] [--masked]`.  Columns: the library's rule, then
explicit (tile, ring, shares) combinations; every split result is checked against the unsplit one."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
from gemm_wide_bench import case, STAGES, ROWS, K  # noqa: E402

masked = "--masked" in sys.argv
stages = [a for a in sys.argv[1:] if not a.startswith("--")] or ["s2", "s3"]
COMBOS = [("auto", 0, 0, 0), ("nosplit", 0, 0, 1),
          ("t1 r2 s2", 1, 2, 2), ("t1 r2 s3", 1, 2, 3), ("t1 r2 s4", 1, 2, 4), ("t1 r3 s2", 1, 3, 2), ("t1 r1 s4", 1, 1, 4),
          ("t2 r3 s2", 2, 3, 2), ("t2 r2 s2", 2, 2, 2), ("t2 r3 s3", 2, 3, 3), ("t2 r2 s4", 2, 2, 4), ("t2 r3 s4", 2, 3, 4),
          ("t3 r3 s2", 3, 3, 2),
          ("t1 r4 s1", 1, 4, 1), ("t2 r4 s1", 2, 4, 1), ("t2 r6 s1", 2, 6, 1), ("t3 r6 s1", 3, 6, 1), ("t2 r3 s1", 2, 3, 1), ("t1 r3 s1", 1, 3, 1)]


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


K.ensure_workspaces("cuda:0", roles=(0,))
print("%-26s" % "M N K kind" + "".join("%10s" % c[0] for c in COMBOS))
tot = [0.0] * len(COMBOS)
for st in stages:
    for M, N, Kd, kind in STAGES[st]:
        rows = ROWS[M]
        x, w, out, kw = case(M, N, Kd, kind, rows)
        B = M // rows
        mg = None
        if masked:          # two architecture groups: the second keeps 5/8 of K and 3/4 of N (64-aligned)
            kk = torch.full((B,), Kd, dtype=torch.int32)
            kn = torch.full((B,), N, dtype=torch.int32)
            kk[B // 2:] = (Kd * 5 // 8) // 64 * 64
            kn[B // 2:] = (N * 3 // 4) // 64 * 64
            kw.update(keep_k=kk.cuda(), keep_n=kn.cuda(), m_groups=2)
        ref = None
        line = "%-26s" % ("%d %d %d %s" % (M, N, Kd, kind))
        for ci, (name, tile, ring, shares) in enumerate(COMBOS):
            sched = tile << 11
            try:
                t = graph_time(lambda: K.gemm(x, w, out, sched=sched, ring=ring, k_shares=shares, **kw))
            except Exception as e:      # noqa: BLE001
                line += "%10s" % "err"
                continue
            if name == "nosplit":
                ref = out.float().clone()
            elif ref is not None:
                err = float((out.float() - ref).abs().max() / ref.abs().max())
                if err > 2e-2 or err != err:
                    line += "  BAD%.0e" % err
                    continue
            tot[ci] += t
            line += "%10.1f" % (t * 1e6)
        print(line, flush=True)
print("%-26s" % "sum" + "".join("%10.1f" % (t * 1e6) for t in tot))
