#!/usr/bin/env python
"""Timeline summary of the last N ms of a rocpd kernel trace: busy union, idle gaps, per-queue busy time (dev tool)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
span_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
dis = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % dis)]
scols = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
namecol = "display_name" if "display_name" in scols else "kernel_name"
qcol = "queue_id" if "queue_id" in cols else None
rows = list(cur.execute("select s.%s, d.start, d.end%s from %s d join %s s on d.kernel_id = s.id order by d.start" % (
    namecol, (", d." + qcol) if qcol else "", dis, sym)))
t_end = max(r[2] for r in rows)
rows = [r for r in rows if r[1] >= t_end - span_ms * 1e6]
t0 = rows[0][1]
busy, cur_s, cur_e = 0, None, None
gaps = []
for r in sorted(rows, key=lambda r: r[1]):
    if cur_s is None:
        cur_s, cur_e = r[1], r[2]
    elif r[1] <= cur_e:
        cur_e = max(cur_e, r[2])
    else:
        busy += cur_e - cur_s
        gaps.append((r[1] - cur_e, r[0][:60]))
        cur_s, cur_e = r[1], r[2]
busy += cur_e - cur_s
tot = t_end - t0
print("window %.3f ms  busy(union) %.3f ms  idle %.3f ms  sum-of-kernels %.3f ms  kernels %d" % (
    tot / 1e6, busy / 1e6, (tot - busy) / 1e6, sum(r[2] - r[1] for r in rows) / 1e6, len(rows)))
if qcol:
    q = {}
    for r in rows:
        q[r[3]] = q.get(r[3], 0) + r[2] - r[1]
    print("per queue busy ms:", {k: round(v / 1e6, 3) for k, v in q.items()})
gaps.sort(reverse=True)
print("gap histogram (us): >20: %d  5-20: %d  2-5: %d  <2: %d ; total idle in gaps<5us: %.3f ms" % (
    sum(g[0] > 20e3 for g in gaps), sum(5e3 < g[0] <= 20e3 for g in gaps), sum(2e3 < g[0] <= 5e3 for g in gaps),
    sum(g[0] <= 2e3 for g in gaps), sum(g[0] for g in gaps if g[0] <= 5e3) / 1e6))
print("largest gaps (us, next kernel):")
for g in gaps[:12]:
    print("  %8.1f  %s" % (g[0] / 1e3, g[1]))
