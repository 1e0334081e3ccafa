#!/usr/bin/env python
"""One training step out of a rocpd kernel trace (between two consecutive adamw launches ~one step apart): per-kernel-name totals,
the forward/backward split, and the in-order list (dev tool).  usage: rocpd_step.py db [k-th step from the end] [list]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
kth = int(sys.argv[2]) if len(sys.argv) > 2 else 6
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
dis = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
scols = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
namecol = "display_name" if "display_name" in scols else "kernel_name"
rows = list(cur.execute("select s.%s, d.start, d.end, d.queue_id from %s d join %s s on d.kernel_id = s.id order by d.start" % (namecol, dis, sym)))


def step_marks(rows):
    """Indices of the adamw dispatch that CLOSES a step.  With the optimizer inside the graph (round 4) a step has an early, capped
    update of the arena's tail beside the backward as well: that one is followed at once by backward kernels, the closing one by
    the next step's gather / copies."""
    ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
    out = []
    for i in ad:
        nxt = [r[0] for r in rows[i + 1:i + 4]]
        if any(("nt_kernel" in n or "tn_group_kernel" in n or "ntln_kernel" in n or "vr_attn_mfma" in n or "ln_bwd" in n) for n in nxt):
            continue
        out.append(i)
    return out if len(out) >= 3 else ad

ad = step_marks(rows)
a, b = ad[-kth - 1], ad[-kth]
step = rows[a + 1:b + 1]
t0 = step[0][1]
print("step window %.3f ms, %d kernels, sum %.3f ms" % ((step[-1][2] - t0) / 1e6, len(step), sum(r[2] - r[1] for r in step) / 1e6))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return n[:70]
tot = {}
for r in step:
    k = short(r[0]); t = tot.setdefault(k, [0, 0.0]); t[0] += 1; t[1] += (r[2] - r[1]) / 1e3
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%5d %9.1f us  %s" % (v[0], v[1], k))
if len(sys.argv) > 3:
    for r in step:
        print("%9.1f %8.1f q%-3s %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], short(r[0])))
