#!/usr/bin/env python
"""MFMA-busy and wave-state summary per kernel family from a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES,
SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY, GRBM_GUI_ACTIVE); counter instances are summed per dispatch."""
import re, sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
tab = lambda p: [t for t in tabs if t.startswith(p)][0]
sym, dis, ev, info = tab("rocpd_info_kernel_symbol"), tab("rocpd_kernel_dispatch"), tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
dcols = [r[1] for r in cur.execute("pragma table_info(%s)" % dis)]
scols = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
namecol = "display_name" if "display_name" in scols else "kernel_name"
evcol = "event_id" if "event_id" in dcols else "id"
q = ("select s.%s, d.%s, i.name, sum(e.value) from %s e join %s d on e.event_id = d.%s join %s s on d.kernel_id = s.id "
     "join %s i on e.pmc_id = i.id group by d.%s, i.name" % (namecol, evcol, ev, dis, evcol, sym, info, evcol))
fam = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(set)
for name, did, cname, val in cur.execute(q):
    f = re.sub(r"\(anonymous namespace\)::", "", name); f = re.sub(r"^void ", "", f); f = re.sub(r"[<(].*", "", f)
    fam[f][cname] += val; cnt[f].add(did)
print("# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (GRBM_GUI_ACTIVE per XCD x 1024); wave-state columns = share of SQ_WAVE_CYCLES")
print("%-36s %7s %9s %9s %9s %9s %9s" % ("kernel family", "launch", "us/launch", "mfma_util", "wait_any", "wait_inst", "active"))
rows = []
for f, c in fam.items():
    busy, wave = c.get("SQ_BUSY_CYCLES", 0), c.get("SQ_WAVE_CYCLES", 0)
    if not busy or not wave:
        continue
    gui = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    rows.append((gui, f, len(cnt[f]), gui / len(cnt[f]) / 2400.0, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(gui * 1024.0, 1.0),
                 c.get("SQ_WAIT_ANY", 0) / wave, c.get("SQ_WAIT_INST_ANY", 0) / wave, c.get("SQ_ACTIVE_INST_ANY", 0) / wave))
for _, f, n, us, a, b, c_, d in sorted(rows, reverse=True)[:16]:
    print("%-36s %7d %9.1f %9.3f %9.3f %9.3f %9.3f" % (f[:36], n, us, a, b, c_, d))
