#!/usr/bin/env python
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB per dispatch).

gfx950 corrections (MI355X_MICROARCH.md, HBM section; re-checked here on torch's fp32->bf16 copy of a known size):
FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> doubled; WRITE_SIZE is exact.
usage: rocpd_traffic.py fetch.db write.db [out.json]
"""
import json, sqlite3, sys


def per_kernel(path):
    db = sqlite3.connect(path); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    tab = lambda p: [t for t in tabs if t.startswith(p)][0]
    sym, dis, ev = tab("rocpd_info_kernel_symbol"), tab("rocpd_kernel_dispatch"), tab("rocpd_pmc_event")
    dcols = [r[1] for r in cur.execute("pragma table_info(%s)" % dis)]
    scols = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
    namecol = "display_name" if "display_name" in scols else "kernel_name"
    evcol = "event_id" if "event_id" in dcols else "id"
    q = "select s.%s, count(*), sum(e.value) from %s e join %s d on e.event_id = d.%s join %s s on d.kernel_id = s.id group by s.%s" % (
        namecol, ev, dis, evcol, sym, namecol)
    return {n: (c, v) for n, c, v in cur.execute(q)}


def family(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("vr_gemm_nt::ntk_kernel", "vr_gemm_nt::nt_kernel").replace("vr_gemm_tn::tn8_group_kernel", "vr_gemm_tn::tn_group_kernel")   # (lean-loop kernels: same family)
    for fam in ("vr_gemm_nt::nt_kernel", "vr_gemm_tn::tn_group_kernel", "vr_gemm_tn::tn_kernel", "gemm_kernel", "ln_bwd_kernel", "ln_fwd_kernel",
                "vr_attn_mfma::fwd_kernel", "vr_attn_mfma::bwd_dq_kernel", "vr_attn_mfma::bwd_dkv_kernel"):
        if name.startswith(fam):
            return fam
    return name.split("(")[0][:60]


f, w = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
fam = {}
for n in set(f) | set(w):
    a = fam.setdefault(family(n), [0, 0.0, 0.0])
    a[0] += f.get(n, (0, 0))[0]
    a[1] += 2.0 * f.get(n, (0, 0))[1] * 1024
    a[2] += w.get(n, (0, 0))[1] * 1024
rows = sorted(fam.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))
print("%-44s %7s %14s %14s %14s" % ("kernel family", "launch", "read MB/launch", "write MB/launch", "total MB/step"))
out = {}
steps = 2
tot = sum(v[1] + v[2] for v in fam.values()) / steps
print('# all kernels: %.1f MB per step' % (tot / 1e6))
for k, (n, rd, wr) in rows[:28]:
    if n == 0:
        continue
    print("%-44s %7d %14.2f %14.2f %14.1f" % (k, n, rd / n / 1e6, wr / n / 1e6, (rd + wr) / steps / 1e6))
    out[k] = {"launches": n, "read_bytes_per_launch": rd / n, "write_bytes_per_launch": wr / n}
if len(sys.argv) > 3:
    key = None
    if len(sys.argv) > 4:                       # workload:batch:dtype of the run (bench.load_traffic refuses a file measured on another one)
        wl, b, dt = sys.argv[4].split(":")
        key = {"workload": wl, "batch": int(b), "dtype": dt}
    json.dump({"key": key, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --steps 1 --warmup 1 --no-graph` "
                         "single stream; FETCH_SIZE doubled (gfx950 correction), WRITE_SIZE as reported", "kernels": out},
              open(sys.argv[3], "w"), indent=1)
