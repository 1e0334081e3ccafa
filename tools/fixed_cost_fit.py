#!/usr/bin/env python
"""Fixed-cost fit of the step's kernels (dev tool): two (or three) in-order step listings from tools/rocpd_step.py ... list, measured at
different batch sizes, aligned launch by launch per hardware queue -> t = a + b * B per launch; printed per kernel family sorted by
sum(a).  usage: fixed_cost_fit.py B1:step1.txt B2:step2.txt [B3:step3.txt]"""
import re, sys
import numpy as np


def read(path):
    rows = []
    for ln in open(path):
        m = re.match(r"\s*([0-9.]+)\s+([0-9.]+) q(\S+)\s+(.*)", ln)
        if m:
            rows.append((float(m.group(1)), float(m.group(2)), m.group(3), m.group(4).strip()))
    return rows


def fam(n):
    n = re.sub(r"<.*", "", n)
    return n.split("(")[0][:48]


runs = []
for arg in sys.argv[1:]:
    b, p = arg.split(":")
    runs.append((int(b), read(p)))
qs = sorted({r[2] for _, rows in runs for r in rows})
tot = {}
for q in qs:
    per = [(b, [r for r in rows if r[2] == q]) for b, rows in runs]
    n = min(len(p) for _, p in per)
    if any(len(p) != n for _, p in per):
        print("# queue %s: launch counts differ %s -- aligned on the first %d" % (q, [len(p) for _, p in per], n))
    B = np.array([b for b, _ in per], float)
    for i in range(n):
        t = np.array([p[i][1] for _, p in per])
        bb, aa = np.polyfit(B, t, 1)
        names = {fam(p[i][3]) for _, p in per}
        key = (q, "/".join(sorted(names)))
        e = tot.setdefault(key, [0, 0.0, 0.0, np.zeros(len(per))])
        e[0] += 1; e[1] += aa; e[2] += bb; e[3] += t
print("%-3s %-50s %4s %9s %9s  %s" % ("q", "family", "n", "sum a us", "b us/img", "  ".join("t(B=%d)" % b for b, _ in runs)))
for (q, k), e in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-3s %-50s %4d %9.1f %9.3f  %s" % (q, k, e[0], e[1], e[2], "  ".join("%8.1f" % x for x in e[3])))
for q in qs:
    sa = sum(e[1] for (qq, _), e in tot.items() if qq == q); sb = sum(e[2] for (qq, _), e in tot.items() if qq == q)
    print("queue %s: sum a = %.1f us, sum b = %.2f us/img, launches %d" % (q, sa, sb, sum(e[0] for (qq, _), e in tot.items() if qq == q)))
