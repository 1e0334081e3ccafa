#!/usr/bin/env python
"""Per-kernel average of the PMC counters in a rocpd database (rocprofv3 --pmc ... --kernel-trace)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
def tab(prefix):
    return [t for t in tabs if t.startswith(prefix)][0]
sym, dis, ev, info = tab("rocpd_info_kernel_symbol"), tab("rocpd_kernel_dispatch"), tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
dcols = [r[1] for r in cur.execute("pragma table_info(%s)" % dis)]
scols = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
namecol = "display_name" if "display_name" in scols else "kernel_name"
evcol = "event_id" if "event_id" in dcols else "id"
q = ("select s.%s, i.name, i.units, count(*), avg(e.value), min(e.value), max(e.value) from %s e join %s d on e.event_id = d.%s "
     "join %s s on d.kernel_id = s.id join %s i on e.pmc_id = i.id group by s.%s, i.name order by avg(e.value) desc" % (
         namecol, ev, dis, evcol, sym, info, namecol))
print("%-80s %-12s %6s %14s %14s %14s  units" % ("kernel", "counter", "n", "avg", "min", "max"))
for name, cname, units, n, avg, mn, mx in cur.execute(q):
    print("%-80s %-12s %6d %14.1f %14.1f %14.1f  %s" % (name.replace("(anonymous namespace)::", "")[:80], cname, n, avg, mn, mx, units))
