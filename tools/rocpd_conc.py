#!/usr/bin/env python
"""Concurrency profile of the last <ms> milliseconds of a rocpd kernel trace: kernels per queue and time with exactly k kernels
running (dev tool; used on tools/two_chain_probe.py)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
win = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
dis = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
rows = list(cur.execute("select start, end, queue_id from %s order by start" % dis))
tend = max(r[1] for r in rows)
rows = [r for r in rows if r[0] >= tend - win * 1e6]
qs = {}
for r in rows:
    qs.setdefault(r[2], []).append(r)
for q, rs in sorted(qs.items()):
    print("queue %s: %d kernels, busy %.3f ms" % (q, len(rs), sum(r[1] - r[0] for r in rs) / 1e6))
pts = sorted([(r[0], 1) for r in rows] + [(r[1], -1) for r in rows])
lvl, last, hist = 0, pts[0][0], {}
for t, d in pts:
    hist[lvl] = hist.get(lvl, 0) + t - last
    lvl += d
    last = t
print("time by number of concurrently running kernels (ms):", {k: round(v / 1e6, 3) for k, v in sorted(hist.items())})
