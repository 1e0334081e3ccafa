#!/usr/bin/env python
"""Print the gemm kernel dispatches of a rocpd database in launch order (name suffix, duration us, grid)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
dis = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % dis)]
scols = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
namecol = "display_name" if "display_name" in scols else "kernel_name"
gx = "grid_size_x" if "grid_size_x" in cols else None
q = "select s.%s, d.start, d.end%s from %s d join %s s on d.kernel_id = s.id order by d.start" % (
    namecol, (", d.grid_size_x, d.grid_size_y, d.grid_size_z" if gx else ""), dis, sym)
pat = sys.argv[2] if len(sys.argv) > 2 else "gemm_kernel"
for row in cur.execute(q):
    if pat in row[0]:
        print("%-70s %9.2f us  grid %s" % (row[0].replace("(anonymous namespace)::", "")[:70], (row[2] - row[1]) / 1e3, row[3:] if gx else ""))
