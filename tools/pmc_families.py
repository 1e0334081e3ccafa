#!/usr/bin/env python
"""Kernel-family summary of tools/rocpd_pmc.py tables (tools/pmc_step.sh): counters summed over the launches of a family.
usage: python tools/pmc_families.py gpurun_out/<tag>_g_pmc.txt gpurun_out/<tag>_h_pmc.txt"""
import re
import sys
from collections import defaultdict

FAM = ("vr_gemm_nt::nt_kernel", "vr_gemm_tn::tn_group_kernel", "vr_gemm_tn::tn_kernel", "vr_gemm_ntln::ntln_kernel", "vr_attn_mfma",
       "ln_bwd_kernel", "ln_fwd_kernel", "adamw_kernel")
tot = defaultdict(lambda: defaultdict(float))
for path in sys.argv[1:]:
    for ln in open(path).read().splitlines()[1:]:
        m = re.match(r"(.{80}) (\S+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", ln)
        if not m:
            continue
        name, ctr, n, avg = m.group(1), m.group(2), int(m.group(3)), float(m.group(4))
        for f in FAM:
            if f in name:
                tot[f][ctr] += avg * n
                tot[f]["_n_" + ctr] += n
                break
print("%-30s %9s %9s %9s %9s %10s %10s %9s" % ("family", "VALU/MFMA", "SALU/MFMA", "LDS/MFMA", "VMEMrd/MFMA", "LDS confl", "L2 hit", "launches"))
for f in FAM:
    t = tot.get(f)
    if not t:
        continue
    mf = t.get("SQ_INSTS_MFMA", 0.0)
    r = lambda k: ("%9.2f" % (t[k] / mf)) if (mf and k in t) else "%9s" % "-"
    confl = ("%9.3f" % (t["SQ_LDS_BANK_CONFLICT"] / t["SQ_LDS_IDX_ACTIVE"])) if t.get("SQ_LDS_IDX_ACTIVE") else "%9s" % "-"
    hit = ("%9.3f" % (t["TCC_HIT_sum"] / (t["TCC_HIT_sum"] + t["TCC_MISS_sum"]))) if t.get("TCC_HIT_sum") else "%9s" % "-"
    print("%-30s %s %s %s   %s %10s %10s %9d" % (f, r("SQ_INSTS_VALU"), r("SQ_INSTS_SALU"), r("SQ_INSTS_LDS"), r("SQ_INSTS_VMEM_RD"), confl, hit,
                                                   int(t.get("_n_SQ_WAVE_CYCLES", t.get("_n_TCC_HIT_sum", 0)))))
