#!/usr/bin/env python
"""Average duration of every kernel FAMILY inside the replayed hipGraph, from a rocprofv3 kernel trace of a bench run (rocpd database):
usage: rocpd_family_json.py trace.db out.json workload:batch:dtype.  bench.py reports the dominant family's figure as
roofline.graph (beside the live eager-launch timing) when workload, per-GPU batch and dtype match."""
import json, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
dis = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
scols = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
namecol = "display_name" if "display_name" in scols else "kernel_name"
fam = {}
for name, n, tot in cur.execute("select s.%s, count(*), sum(d.end - d.start) from %s d join %s s on d.kernel_id = s.id group by s.%s" % (namecol, dis, sym, namecol)):
    k = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("vr_gemm_nt::ntk_kernel", "vr_gemm_nt::nt_kernel").replace("vr_gemm_tn::tn8_group_kernel", "vr_gemm_tn::tn_group_kernel")   # (lean-loop kernels: same family)
    for f in ("vr_gemm_nt::nt_kernel", "vr_gemm_tn::tn_group_kernel", "vr_gemm_tn::tn_kernel", "vr_gemm_ntln::ntln_kernel", "ln_bwd_kernel", "ln_fwd_kernel",
              "vr_attn_mfma::fwd_kernel", "vr_attn_mfma::bwd_dq_kernel", "vr_attn_mfma::bwd_dkv_kernel", "vr_attn_mfma::bwd_short_kernel", "adamw_kernel"):
        if k.startswith(f):
            k = f
            break
    else:
        k = k.split("(")[0][:60]
    a = fam.setdefault(k, [0, 0])
    a[0] += n
    a[1] += tot
wl, b, dt = sys.argv[3].split(":")
json.dump({"key": {"workload": wl, "batch": int(b), "dtype": dt},
           "source": "rocprofv3 --kernel-trace over `bench.py --steps 20 --warmup 5` (hipGraph replay, two streams): all launches of the run",
           "kernels": {k: {"launches": n, "avg_us": t / n / 1e3} for k, (n, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])}},
          open(sys.argv[2], "w"), indent=1)
