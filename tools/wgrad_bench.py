#!/usr/bin/env python
"""The weight-gradient group launches of the step alone (dev tool): one vr_gemm_group per transformer block of each stage, replayed
from a hipGraph (the host needs ~10 us per eager launch): the 4-wave kernel (sched 64 on the first problem; VITRES_DBG_TN=1 / 2:
without the atomics / without the K loop) against the 8-wave, double-buffered one (round 6), dense and with two architecture groups
(`--masked`: the second group keeps 5/8 of every width)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-search_amd"))
import torch
from vitres import kernels as K
dev, bf = "cuda", torch.bfloat16
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("VITRES_"))
print("# " + (tag or "default"))
for (T, C, F, HD, name) in [(32896, 256, 768, 256, "stage 1"), (8320, 512, 1536, 512, "stage 2"), (2176, 1024, 3072, 768, "stage 3")]:
    r = lambda *s: torch.randn(*s, device=dev).to(bf)
    xn, dqkv, ao, gt, xn2, du, h = r(T, C), r(T, 3 * HD), r(T, HD), r(T, C), r(T, C), r(T, F), r(T, F)
    dws = [torch.zeros(3 * HD, C, device=dev), torch.zeros(C, HD, device=dev), torch.zeros(F, C, device=dev), torch.zeros(C, F, device=dev)]
    dbs = [torch.zeros(3 * HD, device=dev), torch.zeros(C, device=dev), torch.zeros(F, device=dev), torch.zeros(C, device=dev)]
    rows = {32896: 257, 8320: 65, 2176: 17}[T]
    B = T // rows
    masked = "--masked" in sys.argv

    def keep(w, period=0):
        if not masked:
            return None
        k = torch.full((B,), period or w, dtype=torch.int32)
        k[B // 2:] = ((period or w) * 5 // 8) // 64 * 64
        return k.cuda()

    def zero_masked(x, k, period=0):             # the contract of keep_k / keep_n: masked channels hold exact zeros
        if k is None:
            return x
        W = x.shape[1]
        col = torch.arange(W, device=dev) % (period or W)
        return (x * (col[None, :] < k.long().repeat_interleave(rows)[:, None])).contiguous()

    def call(dy, x, dw, db, sched, kr=None, kc=None, rp=0):
        No, Ki = dw.shape
        kw = dict(M=No, N=Ki, K=T, lda=No, ldb=Ki, ldc=Ki, a_trans=True, b_trans=True, atomic=True, split_k=0, bias_grad=db, sched=sched)
        if masked:
            kw.update(keep_k=kr, keep_n=kc, k_period=rp, rows_in=rows, m_groups=2)
        return (dy, x, dw, kw)
    kC, kF, kH = keep(C), keep(F), keep(HD)
    xn, xn2, gt, ao, du, h, dqkv = zero_masked(xn, kC), zero_masked(xn2, kC), zero_masked(gt, kC), zero_masked(ao, kH), zero_masked(du, kF), zero_masked(h, kF), zero_masked(dqkv, kH, HD)
    res = []
    for sched in (64, 0, 0x10000):
        calls = [call(gt, h, dws[3], dbs[3], sched, kC, kF), call(du, xn2, dws[2], dbs[2], sched, kF, kC), call(gt, ao, dws[1], dbs[1], sched, kC, kH),
                 call(dqkv, xn, dws[0], dbs[0], sched, kH, kC, HD)]
        for d in dws + dbs:
            d.zero_()
        K.gemm_group(calls)
        torch.cuda.synchronize()
        snap = [d.clone() for d in dws + dbs]
        for _ in range(2):
            K.gemm_group(calls)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10):
                K.gemm_group(calls)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        res.append((e0.elapsed_time(e1) / 50 * 1e-3, snap))
    err = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-6)) for r_ in res[1:] for a, b in zip(r_[1], res[0][1]))
    fl = 2.0 * T * (3 * HD * C + C * HD + 2 * F * C)
    by = 2.0 * T * (3 * HD + C + HD + C + F + C + F + C)
    for nm, (t, _) in zip(("4-wave", "4-lean", "8-wave"), res):
        print("%-8s %-7s group: %7.1f us  %6.0f TF/s dense  %5.2f TB/s operands once" % (name, nm, t * 1e6, fl / t / 1e12, by / t / 1e12))
    print("         against 4-wave: max relative difference %.2e" % err)
