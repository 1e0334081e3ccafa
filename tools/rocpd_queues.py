#!/usr/bin/env python
"""Per-queue timeline of the last full step in a rocpd kernel trace of a hipGraph-replayed bench run (dev tool).
Finds the last two adamw_kernel dispatches (step boundaries), then reports per queue: busy time, number of kernels, idle gaps
between consecutive kernels of that queue (histogram + which kernels follow the largest gaps)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
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

marks = [rows[i][1] for i in step_marks(rows)]
t0, t1 = marks[-2], marks[-1]
step = [r for r in rows if t0 <= r[1] < t1]
print("step span %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(step)))
qs = {}
for r in step:
    qs.setdefault(r[3], []).append(r)
for q, rs in sorted(qs.items(), key=lambda kv: -len(kv[1])):
    busy = sum(r[2] - r[1] for r in rs)
    gaps = [(rs[i + 1][1] - rs[i][2], rs[i][0], rs[i + 1][0]) for i in range(len(rs) - 1)]
    pos = [g for g in gaps if g[0] > 0]
    print("queue %s: %d kernels, busy %.3f ms, sum of gaps %.3f ms; gaps <2us %d, 2-5 %d, 5-10 %d, 10-30 %d, >30 %d" % (
        q, len(rs), busy / 1e6, sum(g[0] for g in pos) / 1e6, sum(g[0] <= 2e3 for g in pos), sum(2e3 < g[0] <= 5e3 for g in pos),
        sum(5e3 < g[0] <= 10e3 for g in pos), sum(10e3 < g[0] <= 30e3 for g in pos), sum(g[0] > 30e3 for g in pos)))
    for g in sorted(pos, reverse=True)[:8]:
        print("    %7.1f us  after %-40s before %s" % (g[0] / 1e3, g[1].split("(")[0][-40:], g[2].split("(")[0][-40:]))

# union over all queues: when is NOTHING running?
ev = sorted((r[1], r[2], r[0]) for r in step)
cur_e, idle, gaps = ev[0][1], 0, []
for i in range(1, len(ev)):
    s, e, n = ev[i]
    if s > cur_e:
        idle += s - cur_e
        gaps.append((s - cur_e, n))
    cur_e = max(cur_e, e)
print("union: idle %.3f ms of %.3f ms; gaps: <2us %d, 2-5 %d, 5-10 %d, >10 %d" % (
    idle / 1e6, (t1 - t0) / 1e6, sum(g[0] <= 2e3 for g in gaps), sum(2e3 < g[0] <= 5e3 for g in gaps),
    sum(5e3 < g[0] <= 10e3 for g in gaps), sum(g[0] > 10e3 for g in gaps)))
for g in sorted(gaps, reverse=True)[:10]:
    print("    %7.1f us before %s" % (g[0] / 1e3, g[1].split("(")[0][-60:]))
# concurrency profile: time with exactly k kernels running
pts = sorted([(r[1], 1) for r in step] + [(r[2], -1) for r in step])
lvl, last, hist = 0, pts[0][0], {}
for t, d in pts:
    hist[lvl] = hist.get(lvl, 0) + t - last
    lvl += d
    last = t
print("time by number of concurrently running kernels (ms):", {k: round(v / 1e6, 3) for k, v in sorted(hist.items())})
