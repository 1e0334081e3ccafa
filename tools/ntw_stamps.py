#!/usr/bin/env python
"""Where a workgroup of the stream-K GEMM (gemm_ntw.hip) spends its time: wall-clock stamps of the first segments of workgroups
0..63 (VITRES_NTW_DBG=8).  slots: 0 kernel entry, 1 segment set up, 2 first slices landed (loop entry), 3 loop done, 4 partial
sums exchanged, 5 epilogue done.  python tools/ntw_stamps.py M N K kind [sched]"""
import os
import sys

os.environ["VITRES_NTW_DBG"] = str(8 | int(os.environ.get("VITRES_NTW_DBG", "0")))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-search_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from vitres import kernels as K  # noqa: E402
import gemm_wide_bench as gb  # noqa: E402,F401  (prints its default table header; harmless)


def main():
    M, N, Kd = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    kind = sys.argv[4] if len(sys.argv) > 4 else "fwd"
    sched = int(sys.argv[5]) if len(sys.argv) > 5 else 8
    x, w, out, kw = gb.case(M, N, Kd, kind, gb.ROWS.get(M, M))
    ws = K._workspace(x.device)
    for _ in range(3):
        K.gemm(x, w, out, sched=sched, **kw)
    torch.cuda.synchronize()
    st = ws[2048 * 4: 4096 * 4].view(torch.int64)
    st.zero_()
    K.gemm(x, w, out, sched=sched, **kw)
    torch.cuda.synchronize()
    v = st.cpu().view(64, 16)
    st.zero_()
    t0 = min(int(r[0]) >> 4 for r in v if int(r[0]))
    print("%s %d %d %d sched %d: us since the first workgroup entered (slot:time)" % (kind, M, N, Kd, sched))
    for wg in list(range(0, 8)) + [16, 32, 63]:
        row = [(int(x_) & 15, ((int(x_) >> 4) - t0) / 100.0) for x_ in v[wg] if int(x_)]
        print("wg %2d: " % wg + "  ".join("%d:%.2f" % (s_, t_) for s_, t_ in row))


if __name__ == "__main__":
    main()
