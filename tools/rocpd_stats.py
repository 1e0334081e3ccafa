#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database as a per-kernel table (name, calls, total/avg/min/max us, %)."""
import re
import sqlite3
import sys


def main(path, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    dis = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
    namecol = "display_name" if "display_name" in cols else "kernel_name"
    rows = cur.execute("select s.%s, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from %s d "
                       "join %s s on d.kernel_id = s.id group by s.%s order by 3 desc" % (namecol, dis, sym, namecol)).fetchall()
    total = sum(r[2] for r in rows)
    print("%-90s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for name, n, tot, mn, mx in rows[:top]:
        name = re.sub(r"\s+", " ", name)[:90]
        print("%-90s %7d %12.1f %10.2f %10.2f %10.2f %6.2f" % (name, n, tot / 1e3, tot / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    print("TOTAL kernel time: %.3f ms over %d kernels" % (total / 1e6, sum(r[1] for r in rows)))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
