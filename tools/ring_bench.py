#!/usr/bin/env python
"""The stage 2 / 3 forward and data-gradient GEMMs alone, library's own choice (dev tool): one column per process environment --
`for r in 0 1 2 3 4; do VITRES_NT_RING=$r python tools/ring_bench.py; done`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_wide_bench import case, STAGES, ROWS, timeit, K  # noqa: E402

tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("VITRES_"))
print("# " + (tag or "default"))
tot = 0.0
for st in (sys.argv[1:] or ["s2", "s3"]):
    for M, N, Kd, kind in STAGES[st]:
        x, w, out, kw = case(M, N, Kd, kind, ROWS[M])
        t = timeit(lambda: K.gemm(x, w, out, **kw), n=50)
        tot += t
        print("%-28s %8.1f us %7.0f TF/s" % ("%d %d %d %s" % (M, N, Kd, kind), t * 1e6, 2.0 * M * N * Kd / t / 1e12))
print("%-28s %8.1f us" % ("sum", tot * 1e6))
