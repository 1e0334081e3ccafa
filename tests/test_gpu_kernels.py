"""Unit parity of every HIP kernel (through the C ABI) against the torch statement of its semantics
(tests/emu_kernels.py) on seeded inputs.  fp32 kernels: ~1e-5; bf16 kernels: bf16 rounding of the outputs."""
import numpy as np
import pytest
import torch

import emu_kernels as E
import vitres.kernels as K

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-6))


def tol(dtype):
    return 2e-5 if dtype == torch.float32 else 1.2e-2


def tol32(dtype):
    """fp32 OUTPUT of a GEMM: with bf16 operands the emulation multiplies the same bf16-rounded values and accumulates in fp32
    like the MFMAs do -- only the summation order differs.  (1.2e-2 is for bf16 OUTPUTS: one ulp of the largest value; at that
    width a wrong fragment in one of 80 K slices would pass.)"""
    return 2e-5 if dtype == torch.float32 else 1e-4


def both(fn_name, cpu_args, cpu_kwargs=None, tensor_outs=()):
    """Run emu on CPU args and the real kernel on .cuda() copies; returns (real, ref)."""
    cpu_kwargs = cpu_kwargs or {}
    to = lambda v: v.to(DEV) if isinstance(v, torch.Tensor) else v
    ref = getattr(E, fn_name)(*[a.clone() if isinstance(a, torch.Tensor) else a for a in cpu_args],
                              **{k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in cpu_kwargs.items()})
    real = getattr(K, fn_name)(*[to(a) for a in cpu_args], **{k: to(v) for k, v in cpu_kwargs.items()})
    torch.cuda.synchronize()
    return real, ref


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K_", [(257 * 3, 576, 192), (130, 1000, 96), (64, 10, 32), (2176, 768, 3072), (8, 24, 40)])
def test_gemm_forward_epilogues(dtype, M, N, K_):
    rows_in = 257 if M % 257 == 0 else 0
    Bn = max(M // rows_in, 1) if rows_in else 1
    a = rnd(M, K_, seed=1).to(dtype)
    b = rnd(N, K_, seed=2, scale=K_ ** -0.5).to(dtype)
    bias = rnd(N, seed=3)
    keep = torch.randint(1, N + 1, (Bn,), generator=torch.Generator().manual_seed(4)).int()
    scale = torch.rand(Bn, generator=torch.Generator().manual_seed(5)) + 0.5
    resid = rnd(M, N, seed=6)
    for variant in ("plain", "resid", "gelu"):
        out_dtype = torch.float32 if variant == "resid" else dtype
        out = torch.zeros(M, N, dtype=out_dtype)
        kw = dict(M=M, N=N, K=K_, lda=K_, ldb=K_, ldc=N, bias=bias, rows_in=rows_in, keep_n=keep if rows_in else None)
        if variant == "resid":
            kw.update(scale=scale if rows_in else None, resid=resid)
        out2 = None
        if variant == "gelu":
            out2 = torch.zeros(M, N, dtype=out_dtype)
            kw.update(act=1, out2=out2)
        to = lambda v: v.to(DEV) if isinstance(v, torch.Tensor) else v
        ref = E.gemm(a, b, out.clone(), **{k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()})
        kw_d = {k: to(v) for k, v in kw.items()}
        real = K.gemm(a.to(DEV), b.to(DEV), out.to(DEV), **kw_d)
        torch.cuda.synchronize()
        assert relerr(real, ref) < (tol32(dtype) if variant == "resid" else tol(dtype)), (variant, relerr(real, ref))
        if variant == "gelu":
            ref2 = torch.zeros(M, N, dtype=out_dtype)
            kw2 = dict(kw); kw2["out2"] = ref2
            E.gemm(a, b, out.clone(), **kw2)
            assert relerr(kw_d["out2"], ref2) < tol(dtype)


def _wide_case(M, N, K_, variant, rows_in, seed=0):
    """Operands of one forward / data-gradient GEMM form of the transformer blocks (bf16) with prefix masks on both sides."""
    Bn = M // rows_in
    g = torch.Generator().manual_seed(100 + seed)
    bt = variant in ("dgrad", "dmul")
    a = rnd(M, K_, seed=seed + 1).to(torch.bfloat16)
    keep_k = torch.randint(K_ // 3, K_ + 1, (Bn,), generator=g).int()
    keep_k[: Bn // 2] = K_                                                  # (two architecture groups: one dense, one pruned)
    keep_k[Bn // 2:] = int(keep_k[Bn // 2])
    a = a * (torch.arange(K_)[None, :] < keep_k.long().repeat_interleave(rows_in)[:, None])      # the contract of keep_k
    b = rnd(K_, N, seed=seed + 2, scale=K_ ** -0.5).to(torch.bfloat16) if bt else rnd(N, K_, seed=seed + 2, scale=K_ ** -0.5).to(torch.bfloat16)
    keep_n = torch.randint(N // 4, N + 1, (Bn,), generator=g).int()
    keep_n[: Bn // 2] = N
    kw = dict(M=M, N=N, K=K_, lda=K_, ldb=N if bt else K_, ldc=N, b_trans=bt, rows_in=rows_in, keep_n=keep_n, keep_k=keep_k)
    out_dtype = torch.bfloat16
    if variant == "fwd":
        kw.update(bias=rnd(N, seed=seed + 3))
    elif variant == "gelu":
        kw.update(bias=rnd(N, seed=seed + 3), act=2, out2=torch.zeros(M, N, dtype=torch.bfloat16))
    elif variant == "res":
        out_dtype = torch.float32
        kw.update(bias=rnd(N, seed=seed + 3), resid=rnd(M, N, seed=seed + 4), scale=torch.rand(Bn, generator=g) + 0.5)
    elif variant == "dmul":
        kw.update(dact_u=rnd(M, N, seed=seed + 5).to(torch.bfloat16), ldu=N, act=2)
    return a, b, torch.zeros(M, N, dtype=out_dtype), kw


@pytest.mark.parametrize("M,N,K_,rows_in", [(2176, 1024, 3072, 17), (8320, 512, 1536, 65), (2176, 3072, 1024, 17), (1300, 320, 640, 65),
                                            (4 * 257, 512, 256, 257)])
@pytest.mark.parametrize("variant", ["fwd", "gelu", "res", "dgrad", "dmul"])
@pytest.mark.parametrize("groups", [1, 2])
def test_gemm_k_shares(M, N, K_, rows_in, variant, groups):
    """K-split of the lean-loop kernel (gemm_ntk.hip SPLIT, vr_gemm_args.k_shares; round 6): 2 - 4 workgroups share a tile's live
    slices round-robin, their fp32 partial tiles meet in vr_gemm_args.ws by plain stores and the last arriver sums them in share
    order.  Every block-Linear form x tile x ring x share count, prefix masks on both sides (shares without a live slice), one and
    two architecture groups (the group-pure grid), launch after launch beside a busy second stream (nobody may wait for anybody):
    against the emulation, against the unsplit kernel, bit-identical from run to run (the sum does not depend on arrival order),
    tickets back at zero."""
    a, b, out, kw = _wide_case(M, N, K_, variant, rows_in)
    ref = E.gemm(a, b, out.clone(), **{k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()})
    to = lambda v: v.to(DEV) if isinstance(v, torch.Tensor) else v
    kw_d = {k: to(v) for k, v in kw.items()}
    ad, bd = a.to(DEV), b.to(DEV)
    t_ = 1e-4 if out.dtype == torch.float32 else 8e-3
    side = torch.cuda.Stream()
    busy = torch.randn(4096, 4096, device=DEV)
    with torch.cuda.stream(side):
        for _ in range(3):
            busy = torch.tanh(busy @ busy * 1e-2)
    ws = K._workspace(torch.device(DEV))
    assert ws is not None
    for tile in (1, 2, 3):
        plain = K.gemm(ad, bd, torch.full_like(out, float("nan")).to(DEV), sched=tile << 11, k_shares=1, m_groups=groups, **kw_d).clone()
        assert relerr(plain, ref) < t_
        for ring in (1, 2, 3):
            for shares in (2, 3, 4):
                runs = []
                for rep in range(2):
                    real = K.gemm(ad, bd, torch.full_like(out, float("nan")).to(DEV), sched=tile << 11, ring=ring, k_shares=shares,
                                  m_groups=groups, **kw_d)
                    torch.cuda.synchronize()
                    assert relerr(real, ref) < t_, (variant, tile, ring, shares, rep, relerr(real, ref))
                    assert relerr(real, plain) < (2e-5 if out.dtype == torch.float32 else 8e-3)
                    runs.append(real.clone())
                assert torch.equal(runs[0], runs[1]), (variant, tile, ring, shares)
        if variant == "gelu":
            ref2 = torch.zeros(M, N, dtype=torch.bfloat16)
            kw2 = dict(kw); kw2["out2"] = ref2
            E.gemm(a, b, out.clone(), **kw2)
            assert relerr(kw_d["out2"], ref2) < t_
    assert int(ws[:16384].view(torch.int32).abs().sum()) == 0          # tickets back at zero
    # the library's own rule on the same problem (whatever it chooses) agrees too
    auto = K.gemm(ad, bd, torch.full_like(out, float("nan")).to(DEV), m_groups=groups, **kw_d)
    assert relerr(auto, ref) < t_


@pytest.mark.parametrize("M,N,K_,rows_in", [(12 * 257, 768, 256, 257), (6 * 257, 776, 320, 257), (10 * 65, 256, 192, 65), (4 * 257, 1024, 256, 257)])
@pytest.mark.parametrize("variant", ["fwd", "gelu", "dmulk", "dgradk"])
@pytest.mark.parametrize("groups", [1, 2])
def test_gemm_panel_resident(M, N, K_, rows_in, variant, groups):
    """gemm_panel.hip (round 6): the A panel resident in LDS, the weight strips in registers, both panel heights (144 rows x 8
    waves, 80 rows x 4 waves; sched 0x200000 / 0x800000 force them), the four stage-1 forms (qkv-like bias store with per-head
    periodic column masks, fc1 GELU pair, fc2 data gradient times the saved gelu', plain data gradient -- the last two on a
    K-contiguous weight), prefix masks on K (k-steps beyond a panel's widest sample are never loaded or multiplied), ragged last
    panels and column strips, one and two architecture groups; against the emulation and the tiled lean kernel (sched 0x100000)."""
    Bn = M // rows_in
    g = torch.Generator().manual_seed(7)
    a = rnd(M, K_, seed=1).to(torch.bfloat16)
    keep_k = torch.full((Bn,), K_, dtype=torch.int32)
    keep_k[Bn // 2:] = (K_ * 5 // 8) // 32 * 32
    a = a * (torch.arange(K_)[None, :] < keep_k.long().repeat_interleave(rows_in)[:, None])
    b = rnd(N, K_, seed=2, scale=K_ ** -0.5).to(torch.bfloat16)
    period = 0
    keep_n = torch.full((Bn,), N, dtype=torch.int32)
    if variant == "fwd" and N % 3 == 0 and (N // 3) % 64 == 0:         # qkv layout: per-head prefixes inside three sections
        period = N // 3
        keep_n[:] = period
        keep_n[Bn // 2:] = period // 2
    else:
        keep_n[Bn // 2:] = (N * 5 // 8) // 8 * 8
    keep_n[1] = 0                                                        # a fully masked sample inside a live group
    kw = dict(M=M, N=N, K=K_, lda=K_, ldb=K_, ldc=N, rows_in=rows_in, keep_n=keep_n, keep_k=keep_k, n_period=period)
    if variant == "fwd":
        kw.update(bias=rnd(N, seed=3))
    elif variant == "gelu":
        kw.update(bias=rnd(N, seed=3), act=2, out2=torch.zeros(M, N, dtype=torch.bfloat16))
    elif variant == "dmulk":
        kw.update(dact_u=rnd(M, N, seed=5).to(torch.bfloat16), ldu=N, act=2)
    out = torch.zeros(M, N, dtype=torch.bfloat16)
    ref = E.gemm(a, b, out.clone(), **{k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()})
    to = lambda v: v.to(DEV) if isinstance(v, torch.Tensor) else v
    kw_d = {k: to(v) for k, v in kw.items()}
    ad, bd = a.to(DEV), b.to(DEV)
    tiled = K.gemm(ad, bd, torch.full_like(out, float("nan")).to(DEV), sched=0x100000, m_groups=groups, **kw_d).clone()
    assert relerr(tiled, ref) < 8e-3
    for form in ((0x200000,) if K_ > 256 else (0x200000, 0x200000 | 0x800000)):
        if K_ > 256:
            form |= 0x800000
        for rep in range(2):
            real = K.gemm(ad, bd, torch.full_like(out, float("nan")).to(DEV), sched=form, m_groups=groups, **kw_d)
            torch.cuda.synchronize()
            assert relerr(real, ref) < 8e-3, (variant, hex(form), rep, relerr(real, ref))
            assert relerr(real, tiled) < 8e-3
        if variant == "gelu":
            ref2 = torch.zeros(M, N, dtype=torch.bfloat16)
            kw2 = dict(kw); kw2["out2"] = ref2
            E.gemm(a, b, out.clone(), **kw2)
            assert relerr(kw_d["out2"], ref2) < 8e-3


@pytest.mark.parametrize("form", [0x200000, 0x200000 | 0x800000])
def test_gemm_panel_resident_leaves_masked_strips_unwritten(form):
    """sched 0x40000 on the panel kernel: a 32-column strip is left unwritten only when the whole 64-wide slice it lies in is beyond
    the width of the panel's architecture group (the readers' granule) -- everything below is written, zeros included; a
    DropPath-dropped sample (keep -(k + 2)) gets its zeros below its group's width k."""
    rows_in, Bn, N, K_ = 257, 8, 768, 256
    M = Bn * rows_in
    a = rnd(M, K_, seed=1).to(torch.bfloat16)
    b = rnd(N, K_, seed=2, scale=K_ ** -0.5).to(torch.bfloat16)
    keep_n = torch.tensor([768, 768, -(768 + 2), 768, 416, 416, -(416 + 2), 416], dtype=torch.int32)
    kw = dict(M=M, N=N, K=K_, lda=K_, ldb=K_, ldc=N, rows_in=rows_in, keep_n=keep_n.to(DEV), bias=rnd(N, seed=3).to(DEV), act=2,
              out2=torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV))
    out = K.gemm(a.to(DEV), b.to(DEV), torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV), sched=form | 0x40000,
                 m_groups=2, **kw).cpu().float()
    out2 = kw["out2"].cpu().float()
    half = M // 2
    for o in (out, out2):
        assert torch.isfinite(o[:half]).all()                                  # group 0: full width
        assert torch.isfinite(o[half:, :448]).all() and torch.isnan(o[half:, 448:]).all()   # group 1: 416 -> slices below 448 written
        assert (o[2 * rows_in:3 * rows_in] == 0).all()                         # DropPath-dropped sample of group 0: zeros, written
        assert (o[6 * rows_in:7 * rows_in, :448] == 0).all()
        assert (o[half:, 416:448] == 0).all()
    ref_kn = keep_n.clone()
    ref_kn[ref_kn < 0] = 0
    ref = E.gemm(a, b, torch.zeros(M, N, dtype=torch.bfloat16), M=M, N=N, K=K_, lda=K_, ldb=K_, ldc=N, rows_in=rows_in, keep_n=ref_kn,
                 bias=rnd(N, seed=3), act=2, out2=torch.zeros(M, N, dtype=torch.bfloat16))
    assert relerr(torch.nan_to_num(out), ref) < 8e-3


@pytest.mark.parametrize("M,N,K_,rows_in", [(8320, 512, 1536, 65), (2176, 1024, 768, 17), (2176, 1024, 3072, 17), (20 * 65, 512, 512, 65),
                                            (6 * 257, 320, 640, 257), (4 * 17, 2048, 256, 17)])
@pytest.mark.parametrize("masked,groups,with_scale", [(False, 1, True), (True, 2, True), (True, 1, False)])
def test_gemm_ln_fold_matches_separate_kernels(M, N, K_, rows_in, masked, groups, with_scale, monkeypatch):
    """vr_gemm_ln_fold (round 6, gemm_ntk.hip LNF): the Linear with bias + DropPath scale + residual on the tiled kernel and the
    LayerNorm of its fp32 result in ONE launch -- every tile takes its row block's ticket, the last arriver normalises the block's
    rows.  Against vr_gemm followed by vr_ln_fwd on the same inputs: the residual stream bit for bit (same kernel, write-through
    stores), y / mean / rstd to the last bit or two (the same row routine); launch after launch beside a busy second stream (the
    tickets return to zero, nobody waits for anybody), ragged last row blocks, prefix masks on K, N and the LayerNorm width, one
    and two architecture groups."""
    monkeypatch.setattr(K, "LN_FOLD", True)                  # (opt-in in the product: slower inside the step)
    Bn = M // rows_in
    g = torch.Generator().manual_seed(11)
    a = rnd(M, K_, seed=1).to(torch.bfloat16)
    keep_k = keep_n = ln_keep = None
    if masked:
        keep_k = torch.full((Bn,), K_, dtype=torch.int32)
        keep_k[Bn // 2:] = (K_ * 5 // 8) // 64 * 64
        a = a * (torch.arange(K_)[None, :] < keep_k.long().repeat_interleave(rows_in)[:, None])
        keep_n = torch.full((Bn,), N, dtype=torch.int32)
        keep_n[Bn // 2:] = (N * 3 // 4) // 8 * 8
        ln_keep = keep_n.clone()
    b = rnd(N, K_, seed=2, scale=K_ ** -0.5).to(torch.bfloat16)
    resid = rnd(M, N, seed=4)
    if masked:
        resid = resid * (torch.arange(N)[None, :] < ln_keep.long().repeat_interleave(rows_in)[:, None])      # the stream's masked channels are zeros
    kw = dict(M=M, N=N, K=K_, lda=K_, ldb=K_, ldc=N, rows_in=rows_in, keep_n=keep_n, keep_k=keep_k, bias=rnd(N, seed=3), resid=resid,
              scale=(torch.rand(Bn, generator=g) + 0.5) if with_scale else None)
    lw, lb = rnd(N, seed=6) + 1.0, rnd(N, seed=7)
    to = lambda v: v.to(DEV) if isinstance(v, torch.Tensor) else v
    kw_d = {k: to(v) for k, v in kw.items()}
    ad, bd = a.to(DEV), b.to(DEV)
    K.M_GROUPS[0] = groups
    try:
        x_sep = K.gemm(ad, bd, torch.full((M, N), float("nan"), device=DEV), sched=2 << 11, **kw_d)
        y_sep, mu_sep, rs_sep = K.ln_fwd(x_sep, to(lw), to(lb), to(ln_keep), rows_in, 1e-6, torch.bfloat16)
        side = torch.cuda.Stream()
        busy = torch.randn(4096, 4096, device=DEV)
        with torch.cuda.stream(side):
            for _ in range(3):
                busy = torch.tanh(busy @ busy * 1e-2)
        for rep in range(3):
            x_f = torch.full((M, N), float("nan"), device=DEV)
            got = K.gemm_ln_fold_fwd(ad, bd, x_f, to(lw), to(lb), to(ln_keep), 1e-6, **kw_d)
            assert got is not None, "the form must be covered"
            y_f, mu_f, rs_f = got
            torch.cuda.synchronize()
            assert torch.equal(x_f, x_sep), rep
            assert relerr(mu_f, mu_sep) < 1e-6 and relerr(rs_f, rs_sep) < 1e-6
            assert relerr(y_f, y_sep) < 8e-3, (rep, relerr(y_f, y_sep))
            assert float((y_f.float() != y_sep.float()).float().mean()) < 1e-3
        ws = K._workspace(torch.device(DEV))
        assert int(ws[:16384].view(torch.int32).abs().sum()) == 0          # tickets back at zero
    finally:
        K.M_GROUPS[0] = 1


@pytest.mark.parametrize("tile,ring", [(1, 4), (2, 4), (2, 5), (2, 6), (3, 4), (3, 6)])
@pytest.mark.parametrize("variant", ["fwd", "res", "dgrad", "dmul"])
def test_gemm_lean_loop_deep_rings(tile, ring, variant):
    """Rings of 4 - 6 slice buffers (round 6: the rule's choice where a grid gives a CU a single workgroup with >= 16 slices)."""
    M, N, K_, rows_in = 2176, 1024, 2304, 17
    a, b, out, kw = _wide_case(M, N, K_, variant, rows_in)
    ref = E.gemm(a, b, out.clone(), **{k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()})
    to = lambda v: v.to(DEV) if isinstance(v, torch.Tensor) else v
    kw_d = {k: to(v) for k, v in kw.items()}
    for rep in range(2):
        real = K.gemm(a.to(DEV), b.to(DEV), torch.full_like(out, float("nan")).to(DEV), sched=tile << 11, ring=ring, k_shares=1, **kw_d)
        assert relerr(real, ref) < (1e-4 if out.dtype == torch.float32 else 8e-3), (tile, ring, variant, relerr(real, ref))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_rowmaps_and_pos(dtype):
    B, P, C, Kd = 3, 16, 64, 592
    N = P + 1
    a = rnd(B * P, Kd, seed=1).to(dtype)
    w = rnd(C, Kd, seed=2, scale=Kd ** -0.5).to(dtype)
    bias, pos = rnd(C, seed=3), rnd(P, C, seed=4)
    keep = torch.tensor([64, 40, 17], dtype=torch.int32)
    out = rnd(B, N, C, seed=9)
    kw = dict(M=B * P, N=C, K=Kd, lda=Kd, ldb=Kd, ldc=C, bias=bias, pos=pos, keep_n=keep, rows_in=P, c_map=(P, N, 1))
    ref = E.gemm(a, w, out.clone(), **kw)
    real = K.gemm(a.to(DEV), w.to(DEV), out.to(DEV), **{k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in kw.items()})
    assert relerr(real, ref) < tol32(dtype)
    # a_map on the input rows (token 0 of every sample)
    y = rnd(B, N, C, seed=5).to(dtype)
    w2 = rnd(24, C, seed=6, scale=0.1).to(dtype)
    out = torch.zeros(B, 24)
    kw = dict(M=B, N=24, K=C, lda=C, ldb=C, ldc=24, a_map=(1, N, 0))
    ref = E.gemm(y, w2, out.clone(), **kw)
    real = K.gemm(y.to(DEV), w2.to(DEV), out.to(DEV), **kw)
    assert relerr(real, ref) < tol32(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K_", [(257 * 2, 192, 576), (65 * 4, 1536, 384), (34, 96, 10)])
def test_gemm_dgrad(dtype, M, N, K_):
    """dx[M,N] = dy[M,K_] @ W[K_,N] (b_trans), with gelu' and keep masks."""
    ldk = (K_ + 7) // 8 * 8
    dy = torch.zeros(M, ldk, dtype=dtype)
    dy[:, :K_] = rnd(M, K_, seed=1).to(dtype)
    w = rnd(K_, N, seed=2, scale=K_ ** -0.5).to(dtype)
    u = rnd(M, N, seed=3).to(dtype)
    rows_in = M // 2
    keep = torch.tensor([N, max(N // 2 - 3, 1)], dtype=torch.int32)
    for use_u in (False, True):
        out = torch.zeros(M, N, dtype=dtype)
        kw = dict(M=M, N=N, K=K_, lda=ldk, ldb=N, ldc=N, b_trans=True, keep_n=keep, rows_in=rows_in)
        if use_u:
            kw.update(dact_u=u, ldu=N)
        ref = E.gemm(dy, w, out.clone(), **kw)
        real = K.gemm(dy.to(DEV), w.to(DEV), out.to(DEV), **{k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in kw.items()})
        assert relerr(real, ref) < tol(dtype), (use_u, relerr(real, ref))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("T,No,Ki,split", [(257 * 4, 576, 192, 2), (2176, 768, 3072, 4), (1300, 10, 96, 3), (40, 64, 592, 1)])
def test_gemm_wgrad(dtype, T, No, Ki, split):
    """dW[No,Ki] += dy[T,No]^T x[T,Ki]  (both contraction-major, split-K atomics)."""
    dy = rnd(T, No, seed=1).to(dtype)
    x = rnd(T, Ki, seed=2).to(dtype)
    out = rnd(No, Ki, seed=3)
    kw = dict(M=No, N=Ki, K=T, lda=No, ldb=Ki, ldc=Ki, a_trans=True, b_trans=True, atomic=True, split_k=split)
    bg_ref, bg = rnd(No, seed=4), rnd(No, seed=4).to(DEV)
    ref = E.gemm(dy, x, out.clone(), bias_grad=bg_ref, **kw)
    real = K.gemm(dy.to(DEV), x.to(DEV), out.to(DEV), bias_grad=bg, **kw)
    t = 5e-5 if dtype == torch.float32 else 1e-4
    assert relerr(real, ref) < t, relerr(real, ref)
    assert relerr(bg, bg_ref) < 2e-4, relerr(bg, bg_ref)       # fused bias gradient (exact sums of the stored values)


@pytest.mark.parametrize("groups", [2, 3, 5])
def test_gemm_group_interleaved_tile_order_changes_nothing(groups):
    """vr_gemm_args.m_groups is a pure scheduling hint (row tiles / token splits of the architecture groups interleaved in the
    XCD-contiguous workgroup order): forward, data-gradient, LayerNorm-fused and weight-gradient results are those of the plain
    order -- bit for bit where no atomics are involved -- for tile counts that do and do not divide by the group count."""
    for M, N, K_, rows_in in ((65 * 30, 512, 384, 65), (257 * 12, 256, 768, 257), (17 * 45, 1024, 512, 17)):
        for variant in ("fwd", "res", "dgrad"):
            a, b, out, kw = _wide_case(M, N, K_, variant, rows_in, seed=7)
            to = lambda v: v.to(DEV) if isinstance(v, torch.Tensor) else v
            kw_d = {k: to(v) for k, v in kw.items()}
            res = []
            for g_ in (1, groups):
                K.M_GROUPS[0] = g_
                try:
                    res.append(K.gemm(a.to(DEV), b.to(DEV), torch.full_like(out, float("nan")).to(DEV), sched=16, **kw_d).clone())
                    if g_ > 1:                                # the lean-loop kernel (gemm_ntk.hip: the default path), three slices in flight
                        res.append(K.gemm(a.to(DEV), b.to(DEV), torch.full_like(out, float("nan")).to(DEV), sched=3 << 9, **kw_d).clone())
                finally:
                    K.M_GROUPS[0] = 1
            assert torch.equal(res[0], res[1]), (M, N, K_, variant)
            assert relerr(res[2], res[0]) < (2e-5 if out.dtype == torch.float32 else 8e-3)      # (same products; bf16 outputs may round the other way)
        # weight gradient: token splits interleaved
        T = M
        dy, x = rnd(T, N, seed=3).to(torch.bfloat16).to(DEV), rnd(T, K_, seed=4).to(torch.bfloat16).to(DEV)
        keep = torch.full((T // rows_in,), N, dtype=torch.int32, device=DEV)
        outs = []
        for g_ in (1, groups):
            K.M_GROUPS[0] = g_
            try:
                o = torch.zeros(N, K_, device=DEV)
                K.gemm(dy, x, o, M=N, N=K_, K=T, lda=N, ldb=K_, ldc=K_, a_trans=True, b_trans=True, atomic=True, split_k=0,
                       rows_in=rows_in, keep_k=keep)
                outs.append(o)
            finally:
                K.M_GROUPS[0] = 1
        assert relerr(outs[1], outs[0]) < 1e-5


@pytest.mark.parametrize("T,No,Ki,rps", [(2176, 3072, 1024, 17), (17 * 8, 768, 1024, 17), (65 * 6, 200, 328, 65)])
def test_gemm_wgrad_store_form(T, No, Ki, rps):
    """atomic == 2: dW and the bias gradient are OVERWRITTEN (one workgroup per tile over all tokens, plain stores): the
    destination starts as NaN; tiles with no kept row / column for any sample must come back as exact zeros; alone and as
    members of a vr_gemm_group launch beside an atomic-form problem."""
    B = T // rps
    dy, x = rnd(T, No, seed=1).to(torch.bfloat16), rnd(T, Ki, seed=2).to(torch.bfloat16)
    kr = torch.tensor([max(No // 2 - 8 * (i % 3), 8) for i in range(B)], dtype=torch.int32)       # upper row tiles fully masked
    kc = torch.tensor([max(Ki - 128 - 16 * (i % 2), 8) for i in range(B)], dtype=torch.int32)     # last column tile fully masked
    dy = dy * (torch.arange(No)[None, :] < kr.long().repeat_interleave(rps)[:, None])
    x = x * (torch.arange(Ki)[None, :] < kc.long().repeat_interleave(rps)[:, None])
    kw = dict(M=No, N=Ki, K=T, lda=No, ldb=Ki, ldc=Ki, a_trans=True, b_trans=True, atomic=2, split_k=1, rows_in=rps)
    ref, bg_ref = torch.full((No, Ki), float("nan")), torch.full((No,), float("nan"))
    E.gemm(dy, x, ref, bias_grad=bg_ref, keep_k=kr, keep_n=kc, **kw)
    out, bg = torch.full((No, Ki), float("nan"), device=DEV), torch.full((No,), float("nan"), device=DEV)
    K.gemm(dy.to(DEV), x.to(DEV), out, bias_grad=bg, keep_k=kr.to(DEV), keep_n=kc.to(DEV), **kw)
    assert torch.isfinite(out).all() and torch.isfinite(bg).all()
    assert relerr(out, ref) < 1e-4 and relerr(bg, bg_ref) < 2e-4
    assert float(out[int(kr.max()):].abs().max()) == 0.0 and float(out[:, int(kc.max()):].abs().max()) == 0.0
    # grouped: [store, atomic, store]
    outs = [torch.full((No, Ki), float("nan"), device=DEV), torch.zeros(No, Ki, device=DEV), torch.full((No, Ki), float("nan"), device=DEV)]
    bgs = [torch.full((No,), float("nan"), device=DEV), torch.zeros(No, device=DEV), torch.full((No,), float("nan"), device=DEV)]
    calls = []
    for i in range(3):
        kwi = dict(kw, atomic=(True if i == 1 else 2), split_k=(0 if i == 1 else 1), keep_k=kr.to(DEV), keep_n=kc.to(DEV), bias_grad=bgs[i])
        calls.append((dy.to(DEV), x.to(DEV), outs[i], kwi))
    K.gemm_group(calls)
    torch.cuda.synchronize()
    for o_, b_ in zip(outs, bgs):
        assert relerr(o_, ref) < 1e-4 and relerr(b_, bg_ref) < 2e-4
    # the ranges kernel that clears what the store form does not write
    buf = torch.full((1000003,), 3.0, device=DEV)
    K.zero_ranges(buf, [(0, 5), (7, 7), (13, 100001), (500002, 1000003)])
    want = torch.full((1000003,), 3.0)
    for lo, hi in ((0, 5), (13, 100001), (500002, 1000003)):
        want[lo:hi] = 0
    assert torch.equal(buf.cpu(), want)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_wgrad_rowmaps(dtype):
    B, No_, Ni, Co, C = 6, 5, 17, 64, 32
    gt = rnd(B, No_, Co, seed=1).to(dtype)
    y = rnd(B, Ni, C, seed=2).to(dtype)
    out = torch.zeros(Co, C)
    kw = dict(M=Co, N=C, K=B, lda=Co, ldb=C, ldc=C, a_trans=True, b_trans=True, atomic=True, split_k=1,
              a_map=(1, No_, 0), b_map=(1, Ni, 0))
    ref = E.gemm(gt, y, out.clone(), **kw)
    real = K.gemm(gt.to(DEV), y.to(DEV), out.to(DEV), **kw)
    assert relerr(real, ref) < tol32(dtype)
    col = rnd(B * 4, 9 * C, seed=3).to(dtype)
    out = torch.zeros(Co, 9 * C)
    kw = dict(M=Co, N=9 * C, K=B * 4, lda=Co, ldb=9 * C, ldc=9 * C, a_trans=True, b_trans=True, atomic=True,
              split_k=2, a_map=(4, No_, 1))
    ref = E.gemm(gt, col, out.clone(), **kw)
    real = K.gemm(gt.to(DEV), col.to(DEV), out.to(DEV), **kw)
    assert relerr(real, ref) < tol32(dtype)


@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [32, 192, 320, 1280])
@pytest.mark.parametrize("masked", [False, True])
def test_layernorm_fwd_bwd(out_dtype, C, masked):
    B, N = 5, 17
    keep = torch.tensor([C, C // 2, C - 4, 8, C], dtype=torch.int32) if masked else None
    x = rnd(B, N, C, seed=1) + 0.3
    if masked:
        x = x * (torch.arange(C)[None, None, :] < keep.long()[:, None, None])
    w, b = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    (y, mean, rstd), (yr, meanr, rstdr) = both("ln_fwd", (x, w, b, keep, N, 1e-6, out_dtype))
    assert relerr(y, yr) < tol(out_dtype)
    assert relerr(mean, meanr) < 1e-5 and relerr(rstd, rstdr) < 1e-5
    dy = rnd(B, N, C, seed=4).to(out_dtype)
    gin = rnd(B, N, C, seed=5)
    dwr, dbr = torch.zeros(C), torch.zeros(C)
    dxr = E.ln_bwd(dy, x, w, meanr, rstdr, keep, N, gin, dwr, dbr)
    dw, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dx = K.ln_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), mean, rstd, None if keep is None else keep.to(DEV), N, gin.to(DEV), dw, db)
    assert relerr(dx, dxr) < 3e-5
    assert relerr(dw, dwr) < 3e-5 and relerr(db, dbr) < 3e-5
    # fused second output: the next backward branch's gradient (DropPath scale, its own prefix mask, compute dtype)
    keep2 = torch.tensor([C // 2, C, 4, C, C - 8], dtype=torch.int32)
    scale2 = torch.tensor([1.0, 0.0, 1.25, 1.25, 1.0])
    for sc, kp in ((scale2, keep2), (None, keep2), (scale2, None), (None, None)):
        dw2, db2 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        dx2, gt = K.ln_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), mean, rstd, None if keep is None else keep.to(DEV), N, gin.to(DEV),
                           dw2, db2, next_cast=(None if sc is None else sc.to(DEV), None if kp is None else kp.to(DEV)))
        assert torch.equal(dx2, dx) and gt.dtype == dy.dtype
        ref = E.scale_mask_cast(dx.cpu(), sc, kp, N, dy.dtype)
        assert torch.equal(gt.cpu(), ref)


def test_cast_transpose_batch_wide_and_ragged_tiles():
    """W^T bf16 shadows: interior 64 x 64 tiles of aligned matrices take the vector path, everything else the scalar one."""
    shapes = [(256, 768, 256), (1000, 1024, 1000), (72, 68, 72), (130, 64, 136), (64, 66, 64), (8, 8, 8)]
    offs, entries, src_parts, n_dst = 0, [], [], 0
    for i, (rows, cols, ld) in enumerate(shapes):
        pad = 0 if i != 3 else 2                       # a source that is only 8-byte aligned
        src_parts.append(torch.zeros(pad))
        offs += pad
        entries.append((offs, n_dst, rows, cols, ld))
        src_parts.append(rnd(rows * cols, seed=10 + i))
        offs += rows * cols
        tail = (-offs) % 4
        src_parts.append(torch.zeros(tail))
        offs += tail
        n_dst += cols * ld
    src = torch.cat(src_parts).to(DEV)
    dst = torch.zeros(n_dst, dtype=torch.bfloat16, device=DEV)
    K.cast_transpose_batch(src, dst, K.tr_descs(entries, DEV))
    for (so, do, rows, cols, ld) in entries:
        want = src[so:so + rows * cols].view(rows, cols).t().to(torch.bfloat16)
        got = dst[do:do + cols * ld].view(cols, ld)
        assert torch.equal(got[:, :rows], want), (rows, cols, ld)
        assert float(got[:, rows:].float().abs().sum()) == 0.0


@pytest.mark.parametrize("C,copies", [(256, 64), (320, 7), (512, 64)])
def test_layernorm_grad_partial_rows(C, copies):
    """grad_copies: the workgroups spread their dgamma / dbeta sums over `copies` rows and vr_ln_grad_reduce folds them into
    the gradients (which already hold a value) and leaves the rows zero; vr_gemm_ln mode 1 likewise."""
    B, N = 24, 65
    x = (rnd(B, N, C, seed=1) + 0.3).to(DEV)
    w, b = (1 + 0.1 * rnd(C, seed=2)).to(DEV), (0.1 * rnd(C, seed=3)).to(DEV)
    keep = torch.tensor([C, C // 2, C - 4, 8] * (B // 4), dtype=torch.int32, device=DEV)
    y, mean, rstd = K.ln_fwd(x, w, b, keep, N, 1e-6, torch.bfloat16)
    dy = rnd(B, N, C, seed=4).to(torch.bfloat16).to(DEV)
    gin = rnd(B, N, C, seed=5).to(DEV)
    base_w, base_b = rnd(C, seed=6).to(DEV), rnd(C, seed=7).to(DEV)
    dw0, db0 = base_w.clone(), base_b.clone()
    dx0 = K.ln_bwd(dy, x, w, mean, rstd, keep, N, gin, dw0, db0)
    part = torch.zeros(2, copies, C, device=DEV)
    dw1, db1 = base_w.clone(), base_b.clone()
    dx1 = K.ln_bwd(dy, x, w, mean, rstd, keep, N, gin, part[0], part[1], copies=copies)
    assert torch.equal(dx0, dx1)
    assert torch.equal(dw1, base_w) and int((part[0].abs().sum(1) > 0).sum()) == copies
    K.ln_grad_reduce([(part[0], part[1], dw1, db1)], copies)
    assert relerr(dw1, dw0) < 1e-5 and relerr(db1, db0) < 1e-5
    assert float(part.abs().max()) == 0.0
    if K.gemm_ln_supported(dy, C, C):
        Kd = 192
        du, wt = _bf(rnd(B * N, Kd, seed=8)).to(DEV), _bf(rnd(C, Kd, seed=9, scale=Kd ** -0.5)).to(DEV)
        dwa, dba, dwb, dbb = base_w.clone(), base_b.clone(), base_w.clone(), base_b.clone()
        kw = dict(M=B * N, N=C, K=Kd, lda=Kd, ldb=Kd, rows_in=N)
        dxa = K.gemm_ln_bwd(du, wt, x, w, mean, rstd, keep, gin, dwa, dba, **kw)
        dxb = K.gemm_ln_bwd(du, wt, x, w, mean, rstd, keep, gin, part[0], part[1], copies=copies, **kw)
        K.ln_grad_reduce([(part[0], part[1], dwb, dbb)], copies)
        assert torch.equal(dxa, dxb)
        assert relerr(dwb, dwa) < 1e-5 and relerr(dbb, dba) < 1e-5 and float(part.abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,H,D", [(17, 2, 64), (65, 3, 48), (257, 2, 32), (257, 3, 64), (5, 2, 48), (2, 2, 64), (40, 2, 64),
                                   (50, 3, 32), (96, 2, 48), (197, 3, 64), (130, 2, 32), (32, 2, 64), (64, 2, 32)])
def test_attention_fwd_bwd(dtype, N, H, D):
    B = 3
    qkv = rnd(B, N, 3 * H * D, seed=1).to(dtype)
    keep = torch.tensor([H * D, D, H * D], dtype=torch.int32)
    scale = D ** -0.5
    (o, lse), (orf, lser) = both("attn_fwd", (qkv, keep, B, N, H, D, scale))
    t = 3e-5 if dtype == torch.float32 else 1.5e-2
    assert relerr(o, orf) < t
    assert relerr(lse, lser) < 1e-4
    d_o = rnd(B, N, H * D, seed=2).to(dtype)
    dq = K.attn_bwd(qkv.to(DEV), o, d_o.to(DEV), lse, keep.to(DEV), B, N, H, D, scale)
    dqr = E.attn_bwd(qkv, orf, d_o, lser, keep, B, N, H, D, scale)
    assert relerr(dq, dqr) < (1e-4 if dtype == torch.float32 else 2.5e-2), relerr(dq, dqr)


@pytest.mark.parametrize("N,H,D", [(17, 2, 64), (65, 3, 48), (257, 2, 32), (257, 3, 64)])
def test_attention_propagates_non_finite_inputs(N, H, D):
    """attn_mfma.hip is built with -fno-honor-nans (no v_max canonicalisation): a NaN / Inf in q, k or v of a diverged step must
    still reach o, lse and the gradients -- train_one_epoch's only divergence guard is math.isfinite(loss) (engine.py:170-173)."""
    B = 2
    scale = D ** -0.5
    keep = torch.tensor([H * D, H * D], dtype=torch.int32, device=DEV)
    for bad, where in ((float("nan"), 0), (float("inf"), H * D), (float("nan"), 2 * H * D)):        # q, k, v of head 0
        qkv = rnd(B, N, 3 * H * D, seed=1).to(torch.bfloat16).to(DEV)
        qkv[1, N // 2, where + 3] = bad
        o, lse = K.attn_fwd(qkv, keep, B, N, H, D, scale)
        torch.cuda.synchronize()
        assert torch.isfinite(o[0].float()).all() and torch.isfinite(lse[0]).all()             # the other sample is untouched
        assert not torch.isfinite(o[1, :, :D].float()).all(), (bad, where)
        d_o = rnd(B, N, H * D, seed=2).to(torch.bfloat16).to(DEV)
        dq = K.attn_bwd(qkv, o, d_o, lse, keep, B, N, H, D, scale)
        assert not torch.isfinite(dq[1].float()).all(), (bad, where)
        assert torch.isfinite(dq[0].float()).all()


def test_relayout_and_zero():
    """vr_relayout (the [out, in, taps] <-> [out, (taps, in)] copies around the convolution-shaped weights, row copies into padded
    rows) and zero_ against torch."""
    for sd, dd in ((torch.float32, torch.float32), (torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32)):
        w = rnd(20, 7, 9, seed=1).to(sd)
        out = torch.full((20, 9 * 7), 5.0, dtype=dd, device=DEV)
        K.relayout(w.to(DEV), out, 20, 7, 9)
        assert torch.equal(out.cpu(), w.permute(0, 2, 1).reshape(20, 63).to(dd))
    w = rnd(256, 588, seed=2)
    out = torch.full((256, 592), 3.0, dtype=torch.bfloat16, device=DEV)
    K.relayout(w.to(DEV), out, 256, 1, 588, dst_ld=592)
    assert torch.equal(out[:, :588].cpu(), w.to(torch.bfloat16)) and float((out[:, 588:].float() - 3.0).abs().max()) == 0.0
    wp = rnd(256, 592, seed=4)                                                   # padded rows -> dense rows (src_ld)
    dense = torch.empty(256, 588, device=DEV)
    K.relayout(wp.to(DEV), dense, 256, 1, 588, src_ld=592)
    assert torch.equal(dense.cpu(), wp[:, :588])
    g = rnd(24, 49, 8, seed=3)                                                   # gradient [C, (kh, kw), m] -> [C, m, kh, kw]
    back = torch.empty(24, 8, 7, 7, device=DEV)
    K.relayout(g.to(DEV), back, 24, 49, 8)
    assert torch.equal(back.cpu(), g.view(24, 7, 7, 8).permute(0, 3, 1, 2))
    for t in (torch.full((3, 1001), 2.0, device=DEV), torch.full((1000,), 2.0, dtype=torch.bfloat16, device=DEV), torch.ones(1, device=DEV)):
        assert float(K.zero_(t).float().abs().max()) == 0.0


def test_stem_glue_kernels():
    """vr_bn_finalize (train-mode BatchNorm2d between the statistics and the normalise pass, running statistics included) against
    torch.nn.BatchNorm2d itself; vr_conv_w_flip and vr_conv3x3_res against their torch statements."""
    C, R = 24, 4096
    z = rnd(R, C, seed=1) * 2 + 0.5
    bn = torch.nn.BatchNorm2d(C).to(DEV)
    ref = torch.nn.BatchNorm2d(C).to(DEV)
    with torch.no_grad():
        for m_ in (bn, ref):
            m_.weight.copy_(1 + 0.1 * rnd(C, seed=2).to(DEV)); m_.bias.copy_(0.1 * rnd(C, seed=3).to(DEV))
            m_.running_mean.copy_(rnd(C, seed=4).to(DEV)); m_.running_var.copy_(rnd(C, seed=5).abs().to(DEV) + 0.5)
    zd = z.to(DEV)
    sq = torch.zeros(2, C, device=DEV)
    K.bn_stats(zd, sq[0], sq[1])
    scale, shift, mean, rstd = K.bn_finalize(sq, R, bn, bn.momentum, True)
    ref.train()
    y_ref = ref(zd.t().reshape(1, C, R, 1))                              # [N=1, C, H=R, W=1]: statistics over the R rows
    y = zd * scale + shift
    assert relerr(y, y_ref.reshape(C, R).t()) < 1e-5
    assert relerr(bn.running_mean, ref.running_mean) < 1e-6 and relerr(bn.running_var, ref.running_var) < 1e-5
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked) == 1
    assert relerr(mean, zd.mean(0)) < 1e-5 and relerr(rstd, torch.rsqrt(zd.var(0, unbiased=False) + bn.eps)) < 1e-5
    w = rnd(24, 24, 3, 3, seed=6)
    for dt in (torch.float32, torch.bfloat16):
        assert torch.equal(K.conv_w_flip(w.to(DEV), dt).cpu(), E.conv_w_flip(w, dt))
    B, H, W = 2, 20, 36
    a, wt, res = rnd(B * H * W, 24, seed=7).bfloat16(), (rnd(24, 9 * 24, seed=8) * 0.1).bfloat16(), rnd(B * H * W, 24, seed=9).bfloat16()
    real = K.conv3x3_res(a.to(DEV), wt.to(DEV), res.to(DEV), B, H, W, 24, 24, torch.bfloat16)
    assert relerr(real, E.conv3x3_res(a, wt, res, B, H, W, 24, 24, torch.bfloat16)) < 8e-3
    plain = K.conv3x3(a.to(DEV), wt.to(DEV), B, H, W, 24, 24, torch.float32)
    assert relerr(plain, E.conv3x3(a, wt, B, H, W, 24, 24, torch.float32)) < 1e-4
    # the evaluation stem's last convolution writing the projection's patchify operand: == the NHWC form + vr_patch_unfold, bit for bit
    B, H, W, P = 2, 28, 42, 7
    a, res, bias = rnd(B * H * W, 24, seed=10).bfloat16().to(DEV), rnd(B * H * W, 24, seed=11).bfloat16().to(DEV), rnd(24, seed=12).to(DEV)
    nhwc = K.conv3x3_bias_relu(a, wt.to(DEV), bias, res, B, H, W, 24, 24, torch.bfloat16)
    col = K.conv3x3_bias_relu_patch(a, wt.to(DEV), bias, res, B, H, W, 24, 24, P, torch.bfloat16)
    assert torch.equal(col, K.patch_unfold(nhwc, B, H // P, W // P, P, 24))
    # training-mode patchify without the unfold / fold passes (round 4): BatchNorm + ReLU (+ skip) writing patch order, BatchNorm
    # backward and the skip connection's add reading it -- each == the NHWC form around vr_patch_unfold
    z = (rnd(B * H * W, 24, seed=13) * 2).to(DEV)
    sc, sh = (1 + 0.1 * rnd(24, seed=14)).to(DEV), (0.1 * rnd(24, seed=15)).to(DEV)
    mu, rs = (0.2 * rnd(24, seed=16)).to(DEV), (1 + 0.1 * rnd(24, seed=17).abs()).to(DEV)
    for zz in (z, z.bfloat16()):
        colp = K.bn_relu_patch(zz, sc, sh, res, B, H, W, P, torch.bfloat16)
        assert torch.equal(colp, K.patch_unfold(K.bn_relu(zz, sc, sh, res, torch.bfloat16), B, H // P, W // P, P, 24))
        dcol = rnd(B * (H // P) * (W // P), P * P * 24, seed=18).bfloat16().to(DEV)
        sg = torch.zeros(4, 24, device=DEV)
        dz_p = K.bn_bwd_patch(dcol, zz, sc, sh, mu, rs, sg[0], sg[1], True, B, H, W, P)
        dz_r = K.bn_bwd(K.patch_fold(dcol, B, H // P, W // P, P, 24), zz, sc, sh, mu, rs, sg[2], sg[3], True)
        assert relerr(sg[0], sg[2]) < 1e-5 and relerr(sg[1], sg[3]) < 1e-5 and relerr(dz_p, dz_r) < 8e-3
    got = K.conv3x3_res_patch(a, wt.to(DEV), dcol, B, H, W, 24, 24, P, torch.bfloat16)
    assert torch.equal(got, K.conv3x3_res(a, wt.to(DEV), K.patch_fold(dcol, B, H // P, W // P, P, 24), B, H, W, 24, 24, torch.bfloat16))


def test_softce():
    for R, Kc in ((8, 10), (128 * 16, 1000)):
        x = rnd(R, Kc, seed=1) * 3
        t = torch.softmax(rnd(R, Kc, seed=2), -1)
        (l, d), (lr, dr) = both("softce", (x, t, 1.0 / R))
        assert relerr(l, lr) < 1e-5 and relerr(d, dr) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_small_kernels(dtype):
    B, N, C = 4, 17, 96
    x = rnd(B, N, C, seed=1)
    keep = torch.tensor([96, 40, 8, 77], dtype=torch.int32)
    scale = torch.tensor([1.25, 0.0, 1.25, 1.0])
    r, e = both("scale_mask_cast", (x, scale, keep, N, dtype))
    assert relerr(r, e) < tol(dtype)
    xt = x.to(dtype)
    out = torch.zeros(C)
    ref = E.colsum(xt, out.clone(), B * (N - 1), C, C, (N - 1, N, 1))
    real = K.colsum(xt.to(DEV), out.to(DEV), B * (N - 1), C, C, (N - 1, N, 1))
    assert relerr(real, ref) < 1e-4
    ref = E.batchsum(x, torch.zeros(N, C))
    real = K.batchsum(x.to(DEV), torch.zeros(N, C, device=DEV))
    assert relerr(real, ref) < 1e-5
    img = rnd(2, 3, 56, 56, seed=3)
    ldk = 592 if dtype == torch.bfloat16 else 588
    r, e = both("im2col_patch", (img, 14, ldk, dtype))
    assert relerr(r, e) < tol(dtype) and r.shape == e.shape
    tokens, pos = rnd(1, 1, C, seed=4), rnd(1, N, C, seed=5)
    r, e = both("embed_cls", (tokens, pos, x.clone(), keep))
    assert relerr(r, e) < 1e-6
    r, e = both("mask_rows", (x.clone(), keep, N))
    assert relerr(r, e) < 1e-6
    # spatial-reduction helpers (grid 4 -> 2)
    g = 4
    y = rnd(B, 1 + g * g, C, seed=6).to(dtype)
    r, e = both("sr_im2col", (y, B, g, C))
    assert relerr(r, e) < 1e-6
    dcol = rnd(B * 4, 9 * C, seed=7).to(dtype)
    dyr = torch.zeros(B, 1 + g * g, C, dtype=dtype)
    E.sr_col2im(dcol, dyr, B, g, C)
    dy = torch.zeros(B, 1 + g * g, C, dtype=dtype, device=DEV)
    K.sr_col2im(dcol.to(DEV), dy, B, g, C)
    assert relerr(dy[:, 1:], dyr[:, 1:]) < tol(dtype)
    xs = rnd(B, 1 + g * g, 64, seed=8)
    r, e = both("sr_resid", (xs, B, g, 64, C))
    assert relerr(r, e) < 1e-6
    do = rnd(B, 5, C, seed=9)
    r, e = both("sr_resid_bwd", (do, B, g, 64, C))
    assert relerr(r, e) < 1e-6
    src = rnd(1000, seed=10)
    dst = torch.empty(1000, dtype=torch.bfloat16, device=DEV)
    K.cast_bf16(src.to(DEV), dst)
    assert torch.equal(dst.cpu(), src.to(torch.bfloat16))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_stem_kernels(dtype):
    B, H, W, m = 2, 12, 12, 24
    img = rnd(B, 3, 2 * H, 2 * W, seed=1)
    r, e = both("im2col3x3_image", (img, 2, 32, dtype))
    assert r.shape == e.shape and relerr(r, e) < tol(dtype)
    a = rnd(B * H * W, m, seed=2).to(dtype)
    r, e = both("im2col3x3", (a, B, H, W, m))
    assert relerr(r, e) < 1e-6
    dcol = rnd(B * H * W, 9 * m, seed=3).to(dtype)
    r, e = both("col2im3x3", (dcol, B, H, W, m))
    assert relerr(r, e) < tol(dtype)
    z = rnd(B * H * W, m, seed=4) + 0.2
    s_ref, q_ref = torch.zeros(m), torch.zeros(m)
    E.bn_stats(z, s_ref, q_ref)
    sd, qd = torch.zeros(m, device=DEV), torch.zeros(m, device=DEV)
    K.bn_stats(z.to(DEV), sd, qd)
    assert relerr(sd, s_ref) < 1e-5 and relerr(qd, q_ref) < 1e-5
    scale, shift = 1 + 0.1 * rnd(m, seed=5), 0.1 * rnd(m, seed=6)
    res = rnd(B * H * W, m, seed=7).to(dtype)
    r, e = both("bn_relu", (z, scale, shift, res, dtype))
    assert relerr(r, e) < tol(dtype)
    r, e = both("bn_relu", (z, scale, shift, None, dtype))
    assert relerr(r, e) < tol(dtype)
    mean, rstd = z.mean(0), 1.0 / z.var(0, unbiased=False).add(1e-5).sqrt()
    da = rnd(B * H * W, m, seed=8).to(dtype)
    for training in (True, False):
        sg_r, sgz_r = torch.zeros(m), torch.zeros(m)
        dz_r = E.bn_bwd(da, z, scale, shift, mean, rstd, sg_r, sgz_r, training)
        sg, sgz = torch.zeros(m, device=DEV), torch.zeros(m, device=DEV)
        dz = K.bn_bwd(da.to(DEV), z.to(DEV), scale.to(DEV), shift.to(DEV), mean.to(DEV), rstd.to(DEV), sg, sgz, training)
        assert relerr(sg, sg_r) < 1e-4 and relerr(sgz, sgz_r) < 1e-4
        assert relerr(dz, dz_r) < (1e-4 if dtype == torch.float32 else 1.5e-2)
    if dtype == torch.bfloat16:        # the pre-BatchNorm tensor stored in bf16 (z_dtype): same kernels, fp32 sums
        zb = z.to(torch.bfloat16)
        s_ref, q_ref = torch.zeros(m), torch.zeros(m)
        E.bn_stats(zb, s_ref, q_ref)
        sd, qd = torch.zeros(m, device=DEV), torch.zeros(m, device=DEV)
        K.bn_stats(zb.to(DEV), sd, qd)
        assert relerr(sd, s_ref) < 1e-5 and relerr(qd, q_ref) < 1e-5
        r, e = both("bn_relu", (zb, scale, shift, res, dtype))
        assert relerr(r, e) < tol(dtype)
        sg_r, sgz_r = torch.zeros(m), torch.zeros(m)
        dz_r = E.bn_bwd(da, zb, scale, shift, mean, rstd, sg_r, sgz_r, True)
        sg, sgz = torch.zeros(m, device=DEV), torch.zeros(m, device=DEV)
        dz = K.bn_bwd(da.to(DEV), zb.to(DEV), scale.to(DEV), shift.to(DEV), mean.to(DEV), rstd.to(DEV), sg, sgz, True)
        assert relerr(sg, sg_r) < 1e-4 and relerr(sgz, sgz_r) < 1e-4 and relerr(dz, dz_r) < 1.5e-2
    a3 = rnd(B * 14 * 14, m, seed=9).to(dtype)
    r, e = both("patch_unfold", (a3, B, 2, 2, 7, m))
    assert torch.equal(r.cpu(), e)
    r2, e2 = both("patch_fold", (e, B, 2, 2, 7, m))
    assert torch.equal(r2.cpu(), a3) and torch.equal(e2, a3)


@pytest.mark.parametrize("M,N,Kd,act", [(700, 256, 768, 0), (1300, 512, 1536, 2), (130, 72, 200, 0), (2176, 1024, 3072, 1),
                                       (300, 64, 64, 2), (8320, 512, 512, 0)])
def test_gemm_dgrad_reads_forward_weight(M, N, Kd, act):
    """b_trans: dX = dY W with W [K = out, N = in] as the forward stores it (k-major weight slices, transposing LDS reads) --
    equal to the statement on the transposed copy, with and without the gelu' epilogues, masks, K tails and narrow N."""
    rows_in = 65 if M % 65 == 0 else 0
    dy = _bf(rnd(M, Kd, seed=1))
    w = _bf(rnd(Kd, N, seed=2, scale=Kd ** -0.5))
    kw = dict(M=M, N=N, K=Kd, lda=Kd, ldb=N, ldc=N, b_trans=True)
    if rows_in:
        nb = M // rows_in
        kw.update(rows_in=rows_in, keep_n=torch.tensor([N, N // 2, 8, N - 8] * (nb // 4 + 1), dtype=torch.int32)[:nb],
                  keep_k=torch.tensor([Kd, Kd // 2, Kd, Kd // 4] * (nb // 4 + 1), dtype=torch.int32)[:nb])
    if act:
        u = _bf(rnd(M, N, seed=3))
        kw.update(dact_u=u, ldu=N, act=2 if act == 2 else 0)
    if rows_in:
        dy = _bf(dy.float() * (torch.arange(Kd)[None, :] < kw["keep_k"].long().repeat_interleave(rows_in)[:, None]))
    ref = E.gemm(dy, w, torch.zeros(M, N, dtype=torch.bfloat16), **kw)
    to = lambda v: v.to(DEV) if isinstance(v, torch.Tensor) else v
    real = K.gemm(dy.to(DEV), w.to(DEV), torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV), **{k: to(v) for k, v in kw.items()})
    assert relerr(real, ref) < tol(torch.bfloat16)
    # and equal (same products, same order per k slice) to the K-contiguous form on W^T
    kw2 = dict(kw, ldb=Kd, b_trans=False)
    real2 = K.gemm(dy.to(DEV), w.t().contiguous().to(DEV), torch.zeros((M, N), dtype=torch.bfloat16, device=DEV),
                   **{k: to(v) for k, v in kw2.items()})
    assert relerr(real, real2.cpu()) < 2e-3


@pytest.mark.parametrize("M,N,Kd,form", [(8320, 512, 1536, "res"), (8320, 512, 512, "plain"), (8320, 1536, 512, "gelu2"),
                                          (128, 1024, 1000, "plain"), (128, 512, 1024, "res"), (2176, 1024, 768, "res"),
                                          (2176, 768, 1024, "bias")])
def test_gemm_several_slices_per_round(M, N, Kd, form):
    """The grid-dependent K-loop forms of the LDS-DMA kernel (two / four slices per round on grids of <= 3 / <= 1 workgroups per
    CU, the three-buffer ring) at the step's second- and third-stage shapes, with masks, against the emulation."""
    rows_in = 65 if M % 65 == 0 else (17 if M % 17 == 0 else 1)
    nb = M // rows_in
    a = _bf(rnd(M, Kd, seed=1))
    w = _bf(rnd(N, Kd, seed=2, scale=Kd ** -0.5))
    keep_k = torch.tensor([Kd, Kd // 2, Kd, (Kd // 4) // 8 * 8] * (nb // 4 + 1), dtype=torch.int32)[:nb]
    keep_n = torch.tensor([N, N // 2, 8, N - 8] * (nb // 4 + 1), dtype=torch.int32)[:nb]
    a = _bf(a.float() * (torch.arange(Kd)[None, :] < keep_k.long().repeat_interleave(rows_in)[:, None]))
    kw = dict(M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, rows_in=rows_in, keep_k=keep_k, keep_n=keep_n)
    out_dt = torch.bfloat16
    outs = {}
    if form == "res":
        out_dt = torch.float32
        kw.update(bias=rnd(N, seed=3), resid=rnd(M, N, seed=4), scale=torch.tensor([1.25, 0.0, 1.0, 1.25] * (nb // 4 + 1))[:nb])
    elif form == "bias":
        kw.update(bias=rnd(N, seed=3))
    elif form == "gelu2":
        kw.update(bias=rnd(N, seed=3), act=2)
        outs = dict(out2=torch.zeros(M, N, dtype=out_dt))
    ref2 = dict(outs)
    ref = E.gemm(a, w, torch.zeros(M, N, dtype=out_dt), **kw, **ref2)
    to = lambda v: v.to(DEV) if isinstance(v, torch.Tensor) else v
    dev2 = {k: torch.zeros_like(v).to(DEV) for k, v in outs.items()}
    real = K.gemm(a.to(DEV), w.to(DEV), torch.full((M, N), 7.0, dtype=out_dt, device=DEV), **{k: to(v) for k, v in kw.items()}, **dev2)
    assert relerr(real, ref) < (2e-3 if out_dt == torch.float32 else tol(torch.bfloat16))
    for k in outs:
        assert relerr(dev2[k], ref2[k]) < tol(torch.bfloat16)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_masked_work_skipping(dtype):
    """keep_k / keep_n / periods only skip work that is zero by contract: results equal the dense statement."""
    B, Nt, C, H, D = 6, 65, 256, 4, 64
    HD, M = H * D, 6 * 65
    g = torch.Generator().manual_seed(3)
    ek = torch.tensor([256, 160, 256, 64, 192, 160], dtype=torch.int32)          # embed keep per sample
    ak = torch.tensor([256, 128, 64, 192, 128, 256], dtype=torch.int32)          # head-prefix keep (multiples of D)
    rowmask = lambda keep, width, period=0: ((torch.arange(width) % period if period else torch.arange(width))[None, :]
                                             < keep.long().repeat_interleave(Nt)[:, None])
    to = lambda v: v.to(DEV) if isinstance(v, torch.Tensor) else v
    # forward qkv: A = y masked by ek, columns kept per head prefix
    y = (rnd(M, C, seed=1) * rowmask(ek, C)).to(dtype)
    w = rnd(3 * HD, C, seed=2, scale=C ** -0.5).to(dtype)
    bias = rnd(3 * HD, seed=3)
    kw = dict(M=M, N=3 * HD, K=C, lda=C, ldb=C, ldc=3 * HD, bias=bias, rows_in=Nt, keep_k=ek, keep_n=ak, n_period=HD)
    ref = E.gemm(y, w, torch.zeros(M, 3 * HD, dtype=dtype), **kw)
    real = K.gemm(y.to(DEV), w.to(DEV), torch.full((M, 3 * HD), 7.0, dtype=dtype, device=DEV), **{k: to(v) for k, v in kw.items()})
    assert relerr(real, ref) < tol(dtype)
    # dgrad through qkv: A = dqkv zero in dropped heads (periodic k skipping), output masked by ek
    dqkv = (rnd(M, 3 * HD, seed=4) * rowmask(ak, 3 * HD, HD)).to(dtype)
    kw = dict(M=M, N=C, K=3 * HD, lda=3 * HD, ldb=C, ldc=C, b_trans=True, rows_in=Nt, keep_k=ak, k_period=HD, keep_n=ek)
    ref = E.gemm(dqkv, w, torch.zeros(M, C, dtype=dtype), **kw)
    real = K.gemm(dqkv.to(DEV), w.to(DEV), torch.full((M, C), 7.0, dtype=dtype, device=DEV), **{k: to(v) for k, v in kw.items()})
    assert relerr(real, ref) < tol(dtype)
    # wgrad: dW[3HD, C] += dqkv^T y with row (periodic) and column keeps, bias gradient fused
    out = torch.zeros(3 * HD, C)
    bg_ref, bg = torch.zeros(3 * HD), torch.zeros(3 * HD, device=DEV)
    kw = dict(M=3 * HD, N=C, K=M, lda=3 * HD, ldb=C, ldc=C, a_trans=True, b_trans=True, atomic=True, split_k=6, rows_in=Nt,
              keep_k=ak, k_period=HD, keep_n=ek)
    ref = E.gemm(dqkv, y, out.clone(), bias_grad=bg_ref, **kw)
    real = K.gemm(dqkv.to(DEV), y.to(DEV), out.to(DEV), bias_grad=bg, **{k: to(v) for k, v in kw.items()})
    assert relerr(real, ref) < (5e-5 if dtype == torch.float32 else 1e-4)
    assert relerr(bg, bg_ref) < 2e-4
    # residual epilogue with a fully masked sample (layer drop: keep 0) and prefix keep_k
    ok = torch.tensor([256, 0, 256, 64, 0, 160], dtype=torch.int32)
    o = (rnd(M, HD, seed=5) * rowmask(ak, HD)).to(dtype)
    wp = rnd(C, HD, seed=6, scale=HD ** -0.5).to(dtype)
    resid = rnd(M, C, seed=7) * rowmask(ek, C)
    scale = torch.tensor([1.25, 1.25, 0.0, 1.25, 1.25, 1.0])
    kw = dict(M=M, N=C, K=HD, lda=HD, ldb=HD, ldc=C, bias=rnd(C, seed=8), scale=scale, keep_n=ok, resid=resid, rows_in=Nt, keep_k=ak)
    ref = E.gemm(o, wp, torch.zeros(M, C), **kw)
    real = K.gemm(o.to(DEV), wp.to(DEV), torch.full((M, C), 7.0, device=DEV), **{k: to(v) for k, v in kw.items()})
    assert relerr(real, ref) < tol32(dtype)


@pytest.mark.parametrize("Cin,Cout", [(24, 24), (32, 32), (16, 24), (24, 8)])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_direct_conv3x3(Cin, Cout, out_dtype):
    """vr_conv3x3 (implicit-GEMM 3x3 / stride 1 / pad 1 on MFMA) == conv2d of the same bf16 operands; H, W not multiples of
    the 16 x 16 tile exercise the halo and the edge tiles."""
    B, H, W = 3, 37, 20
    a = rnd(B * H * W, Cin, seed=1).to(torch.bfloat16)
    w = (rnd(Cout, 9 * Cin, seed=2) * (9 * Cin) ** -0.5).to(torch.bfloat16)
    ref = E.conv3x3(a, w, B, H, W, Cin, Cout, out_dtype)
    out = K.conv3x3(a.to(DEV), w.to(DEV), B, H, W, Cin, Cout, out_dtype)
    assert out.dtype == out_dtype and out.shape == ref.shape
    assert relerr(out, ref) < (2e-5 if out_dtype == torch.float32 else 1e-2)
    # the data gradient is the same kernel with flipped / transposed weights: check the adjoint identity <conv(a), y> = <a, conv_t(y)>
    y = rnd(B * H * W, Cout, seed=3).to(torch.bfloat16)
    wt = w.float().view(Cout, 3, 3, Cin).flip(1, 2).permute(3, 1, 2, 0).reshape(Cin, 9 * Cout).to(torch.bfloat16)
    if K.conv3x3_supported(y, Cout, Cin):
        back = K.conv3x3(y.to(DEV), wt.to(DEV), B, H, W, Cout, Cin, torch.float32).cpu()
        fwd = E.conv3x3(a, w, B, H, W, Cin, Cout, torch.float32)
        lhs, rhs = float((fwd * y.float()).sum()), float((a.float() * back).sum())
        assert abs(lhs - rhs) < 2e-3 * max(abs(lhs), 1.0)


@pytest.mark.parametrize("Cin,Cout,with_res", [(24, 24, True), (32, 32, False), (16, 24, True)])
def test_direct_conv3x3_bias_relu(Cin, Cout, with_res):
    """vr_conv3x3_bias_relu (evaluation stem, BatchNorm folded): relu(conv + bias) (+ residual) in the convolution's epilogue."""
    B, H, W = 2, 37, 20
    a = rnd(B * H * W, Cin, seed=1).to(torch.bfloat16)
    w = (rnd(Cout, 9 * Cin, seed=2) * (9 * Cin) ** -0.5).to(torch.bfloat16)
    bias = 0.3 * rnd(Cout, seed=3)
    res = rnd(B * H * W, Cout, seed=4).to(torch.bfloat16) if with_res else None
    ref = E.conv3x3_bias_relu(a, w, bias, res, B, H, W, Cin, Cout, torch.bfloat16)
    out = K.conv3x3_bias_relu(a.to(DEV), w.to(DEV), bias.to(DEV), None if res is None else res.to(DEV), B, H, W, Cin, Cout,
                              torch.bfloat16)
    assert relerr(out, ref) < 1e-2
    assert float(out.float().min()) >= (0.0 if res is None else float(res.float().min()) - 1e-6)


@pytest.mark.parametrize("H,W,Cout,fold", [(56, 56, 24, False), (224, 224, 32, True), (37, 61, 16, True), (30, 20, 8, False)])
def test_direct_conv1_from_image(H, W, Cout, fold):
    """vr_conv1_direct: 3x3 / stride 2 / pad 1 from the fp32 NCHW image == conv2d on the bf16-rounded image and weights;
    odd sizes exercise the padding and the edge tiles; fold = the evaluation epilogue relu(. + bias)."""
    B = 2
    img = rnd(B, 3, H, W, seed=1)
    w = torch.zeros(Cout, 32, dtype=torch.bfloat16)
    w[:, :27] = (rnd(Cout, 27, seed=2) * 27 ** -0.5).to(torch.bfloat16)
    bias = 0.3 * rnd(Cout, seed=3) if fold else None
    for od in (torch.float32, torch.bfloat16):
        ref = E.conv1_direct(img, w, bias, fold, od)
        out = K.conv1_direct(img.to(DEV), w.to(DEV), None if bias is None else bias.to(DEV), fold, od)
        assert out.shape == ref.shape and out.dtype == od
        assert relerr(out, ref) < (2e-5 if od == torch.float32 else 1e-2)
    # the gather + GEMM form it replaces gives the same numbers
    col = K.im2col3x3_image(img.to(DEV), 2, 32, torch.bfloat16)
    z = torch.empty((col.shape[0], Cout), dtype=torch.float32, device=DEV)
    K.gemm(col, w.to(DEV), z, M=col.shape[0], N=Cout, K=32, lda=32, ldb=32, ldc=Cout)
    assert relerr(K.conv1_direct(img.to(DEV), w.to(DEV), None, False, torch.float32), z.cpu()) < 2e-5


@pytest.mark.parametrize("M,N,Kd", [(1000, 24, 32), (300, 256, 192), (130, 72, 200)])
def test_gemm_relu_epilogue(M, N, Kd):
    """act = 3: C = relu(A B^T + bias), single bf16 store (the conv1 GEMM of the BatchNorm-folded evaluation stem)."""
    a, w, bias = _bf(rnd(M, Kd, seed=1)), _bf(rnd(N, Kd, seed=2, scale=Kd ** -0.5)), rnd(N, seed=3)
    kw = dict(M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, bias=bias, act=3)
    ref = E.gemm(a, w, torch.zeros(M, N, dtype=torch.bfloat16), **kw)
    real = K.gemm(a.to(DEV), w.to(DEV), torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV),
                  **{k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in kw.items()})
    assert relerr(real, ref) < tol(torch.bfloat16)
    assert float(real.float().min()) == 0.0


@pytest.mark.parametrize("C", [24, 32, 16])
def test_direct_conv3x3_wgrad(C):
    """vr_conv3x3_wgrad == the weight gradient of conv2d on the same bf16 operands (edge tiles, accumulation into dw)."""
    B, H, W = 3, 37, 20
    a = rnd(B * H * W, C, seed=1).to(torch.bfloat16)
    dz = rnd(B * H * W, C, seed=2).to(torch.bfloat16)
    ref = E.conv3x3_wgrad(a, dz, torch.ones(C, 9 * C), B, H, W, C, C)
    dw = torch.ones(C, 9 * C, device=DEV)
    K.conv3x3_wgrad(a.to(DEV), dz.to(DEV), dw, B, H, W, C, C)
    assert relerr(dw, ref) < 3e-5


@pytest.mark.parametrize("N,H,D", [(401, 2, 64), (785, 2, 32), (300, 3, 48), (577, 1, 64)])
def test_attention_long_sequences(N, H, D):
    """N > 288 (280 / 336 / 392 px fine-tuning: 401 / 577 / 785 tokens): the block-streaming MFMA kernels -- two-pass softmax
    forward, key-blocked dQ, query-blocked dK / dV -- against the fp32 restatement; one head dropped by the prefix mask."""
    B = 2
    dtype = torch.bfloat16
    qkv = rnd(B, N, 3 * H * D, seed=1).to(dtype)
    keep = torch.tensor([H * D, max(D, (H - 1) * D)], dtype=torch.int32)
    scale = D ** -0.5
    (o, lse), (orf, lser) = both("attn_fwd", (qkv, keep, B, N, H, D, scale))
    assert relerr(o, orf) < 1.5e-2
    assert relerr(lse, lser) < 1e-4
    d_o = rnd(B, N, H * D, seed=2).to(dtype)
    dq = K.attn_bwd(qkv.to(DEV), o, d_o.to(DEV), lse, keep.to(DEV), B, N, H, D, scale)
    dqr = E.attn_bwd(qkv, orf, d_o, lser, keep, B, N, H, D, scale)
    assert relerr(dq, dqr) < 2.5e-2, relerr(dq, dqr)


# ---- vr_gemm_ln: Linear with the adjacent LayerNorm in its epilogue (gemm_nt_ln.hip) -------------------------------
def _bf(t):
    return t.to(torch.bfloat16)


# (130 x 257: not a multiple of 16 rows; sched is passed through -- vr_gemm_ln has one kernel)
@pytest.mark.parametrize("B,Nt,C,Kd,masked,sched", [
    (6, 257, 256, 768, True, 16), (4, 65, 512, 1536, True, 0), (3, 50, 192, 256, False, 16), (5, 17, 448, 128, True, 0),
    (2, 33, 8, 72, False, 16), (5, 257, 320, 1280, True, 0), (3, 70, 296, 320, False, 0),
    (21, 257, 248, 200, True, 0), (150, 257, 256, 256, True, 0), (130, 257, 256, 768, True, 0)])
def test_gemm_ln_forward(B, Nt, C, Kd, masked, sched):
    """mode 0 == vr_gemm (residual epilogue) followed by vr_ln_fwd: same residual stream bit for bit (same MFMA order is not
    required: compared with tolerance), LayerNorm output / statistics within bf16 / fp32 rounding."""
    M = B * Nt
    a, w = _bf(rnd(M, Kd, seed=1)), _bf(rnd(C, Kd, seed=2, scale=Kd ** -0.5))
    bias, resid = rnd(C, seed=3), rnd(M, C, seed=4)
    lw, lb = 1 + 0.1 * rnd(C, seed=5), 0.1 * rnd(C, seed=6)
    scale = (torch.rand(B, generator=torch.Generator().manual_seed(7)) > 0.3).float() / 0.7 if masked else None
    keep_n = torch.tensor([C - 8 * (i % 3) for i in range(B)], dtype=torch.int32) if masked else None
    ln_keep = torch.tensor([C - 16 * (i % 2) - 8 * (i % 3) for i in range(B)], dtype=torch.int32) if masked else None
    keep_k = torch.tensor([Kd - 64 * (i % 2) for i in range(B)], dtype=torch.int32) if masked else None
    if keep_k is not None:                                   # the producer zeroes masked K columns (contract of keep_k)
        a = a.view(B, Nt, Kd).clone()
        for i in range(B):
            a[i, :, int(keep_k[i]):] = 0
        a = a.view(M, Kd)
    kw = dict(M=M, N=C, K=Kd, lda=Kd, ldb=Kd, ldc=C, bias=bias, scale=scale, keep_n=keep_n, resid=resid, rows_in=Nt, keep_k=keep_k)
    out_ref = torch.empty(M, C)
    y_ref, mu_ref, rs_ref = E.gemm_ln_fwd(a, w, out_ref, lw, lb, ln_keep, 1e-6, **kw)
    cu = lambda t: None if t is None else t.to(DEV)
    out = torch.empty(M, C, device=DEV)
    kw_d = {k: (cu(v) if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}
    assert K.gemm_ln_supported(cu(a), C, C)
    y, mu, rs = K.gemm_ln_fwd(cu(a), cu(w), out, cu(lw), cu(lb), cu(ln_keep), 1e-6, sched=sched, **kw_d)
    torch.cuda.synchronize()
    assert relerr(out, out_ref) < 2e-5
    assert relerr(mu, mu_ref) < 2e-5 and relerr(rs, rs_ref) < 1e-4
    assert relerr(y, y_ref) < 1.2e-2
    # against the two separate kernels on the device: the LayerNorm of the SAME residual stream
    y2, mu2, rs2 = K.ln_fwd(out, cu(lw), cu(lb), cu(ln_keep), Nt, 1e-6, torch.bfloat16)
    assert relerr(mu, mu2) < 1e-5 and relerr(rs, rs2) < 1e-4
    assert float((y.float() - y2.float()).abs().max()) <= 2 ** -6 * float(y2.float().abs().max())


@pytest.mark.parametrize("B,Nt,C,Kd,masked,nxt,sched", [
    (6, 257, 256, 768, True, True, 16), (4, 65, 512, 1536, True, True, 0), (3, 50, 192, 256, False, False, 16),
    (5, 17, 448, 192, True, False, 0), (2, 33, 8, 72, False, True, 16), (5, 257, 320, 960, True, True, 0),
    (3, 70, 296, 320, False, False, 0),
    (21, 257, 248, 200, True, True, 0), (150, 257, 256, 256, True, False, 0), (130, 257, 256, 768, True, True, 0)])
def test_gemm_ln_backward(B, Nt, C, Kd, masked, nxt, sched):
    """mode 1 == data-gradient GEMM (fp32 result) followed by vr_ln_bwd."""
    M = B * Nt
    du, wt = _bf(rnd(M, Kd, seed=1)), _bf(rnd(C, Kd, seed=2, scale=Kd ** -0.5))
    x, dx_in = rnd(M, C, seed=3), rnd(M, C, seed=4)
    lw = 1 + 0.1 * rnd(C, seed=5)
    ln_keep = torch.tensor([C - 16 * (i % 2) - 8 * (i % 3) for i in range(B)], dtype=torch.int32) if masked else None
    keep_k = torch.tensor([Kd - 64 * (i % 2) for i in range(B)], dtype=torch.int32) if masked else None
    if keep_k is not None:
        du = du.view(B, Nt, Kd).clone()
        for i in range(B):
            du[i, :, int(keep_k[i]):] = 0
        du = du.view(M, Kd)
    _, mean, rstd = E.ln_fwd(x, lw, torch.zeros(C), ln_keep, Nt, 1e-6, torch.float32)
    nc = None
    if nxt:
        nc = ((torch.rand(B, generator=torch.Generator().manual_seed(9)) > 0.3).float() / 0.7,
              torch.tensor([C - 8 * (i % 4) for i in range(B)], dtype=torch.int32))
    dw_ref, db_ref = torch.zeros(C), torch.zeros(C)
    ref = E.gemm_ln_bwd(du, wt, x, lw, mean, rstd, ln_keep, dx_in, dw_ref, db_ref, next_cast=nc, M=M, N=C, K=Kd, lda=Kd, ldb=Kd,
                        rows_in=Nt, keep_k=keep_k)
    cu = lambda t: None if t is None else t.to(DEV)
    dw, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    real = K.gemm_ln_bwd(cu(du), cu(wt), cu(x), cu(lw), cu(mean), cu(rstd), cu(ln_keep), cu(dx_in), dw, db,
                         next_cast=None if nc is None else (cu(nc[0]), cu(nc[1])), M=M, N=C, K=Kd, lda=Kd, ldb=Kd, rows_in=Nt,
                         keep_k=cu(keep_k), sched=sched)
    torch.cuda.synchronize()
    dx, dx_ref = (real[0], ref[0]) if nxt else (real, ref)
    assert relerr(dx, dx_ref) < 5e-5
    assert relerr(dw, dw_ref) < 1e-4 and relerr(db, db_ref) < 1e-4
    if nxt:
        assert relerr(real[1], ref[1]) < 1.2e-2
        if masked:
            for i in range(B):
                tail = real[1].view(B, Nt, C)[i, :, int(nc[1][i]):]
                assert tail.numel() == 0 or float(tail.abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("T", [1, 2])
def test_token_rows_of_embed_and_spatial_reduction_kernels(dtype, T):
    """num_tokens = 1 (class token) and 2 (class + distillation token) in vr_embed_cls and the vr_sr_* gathers."""
    B, g, C, Cin = 3, 6, 72, 40
    N = T + g * g
    keep = torch.tensor([72, 40, 8], dtype=torch.int32)
    tokens, pos, x = rnd(1, T, C, seed=4), rnd(1, N, C, seed=5), rnd(B, N, C, seed=6)
    r, e = both("embed_cls", (tokens, pos, x.clone(), keep, T))
    assert relerr(r, e) < 1e-6 and torch.equal(r[:, T:].cpu(), x[:, T:])          # patch rows untouched
    y = rnd(B, N, C, seed=7).to(dtype)
    r, e = both("sr_im2col", (y, B, g, C, T))
    assert relerr(r, e) < 1e-6
    go = g // 2
    dcol = rnd(B * go * go, 9 * C, seed=8).to(dtype)
    dyr = torch.full((B, N, C), 7.0, dtype=dtype)
    E.sr_col2im(dcol, dyr, B, g, C, T)
    dy = torch.full((B, N, C), 7.0, dtype=dtype, device=DEV)
    K.sr_col2im(dcol.to(DEV), dy, B, g, C, T)
    assert relerr(dy, dyr) < tol(dtype) and float((dy[:, :T].float() - 7.0).abs().max()) == 0.0      # token rows untouched
    xs = rnd(B, N, Cin, seed=9)
    r, e = both("sr_resid", (xs, B, g, Cin, C, T))
    assert r.shape == (B, T + go * go, C) and relerr(r, e) < 1e-6
    do = rnd(B, T + go * go, C, seed=10)
    r, e = both("sr_resid_bwd", (do, B, g, Cin, C, T))
    assert r.shape == (B, N, Cin) and relerr(r, e) < 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("R,rps,K_,mapped", [(8, 1, 1000, True), (8 * 16, 16, 1000, True), (6, 1, 10, False), (12, 4, 37, True)])
def test_softce_train(dtype, R, rps, K_, mapped):
    """vr_softce_train: mean soft-target CE of internally ordered logits against caller-ordered targets, loss accumulated into
    one scalar, gradient in the head GEMMs' layout (row pitch rounded up to 8, pad zeroed)."""
    B = R // rps
    logits, target = rnd(R, K_, seed=1, scale=2.0), torch.softmax(rnd(R, K_, seed=2), -1)
    smap = torch.randperm(B, generator=torch.Generator().manual_seed(3)) if mapped else None
    acc_ref = torch.full((1,), 0.5)
    ref = E.softce_train(logits, target, smap, rps, acc_ref, dtype)
    acc = torch.full((1,), 0.5, device=DEV)
    real = K.softce_train(logits.to(DEV), target.to(DEV), None if smap is None else smap.to(DEV), rps, acc, dtype)
    torch.cuda.synchronize()
    assert real.shape == ref.shape == (R, (K_ + 7) // 8 * 8)
    assert abs(acc.item() - acc_ref.item()) < 1e-5 * abs(acc_ref.item())
    assert relerr(real, ref) < (1e-5 if dtype == torch.float32 else 1e-2)
    assert float(real[:, K_:].float().abs().max() if K_ % 8 else 0.0) == 0.0
    # against the two-step torch statement: gather targets, mean CE, autograd
    x = logits.clone().requires_grad_(True)
    src = (smap.repeat_interleave(rps) * rps + torch.arange(R) % rps) if smap is not None else torch.arange(R)
    loss = torch.sum(-target[src] * torch.log_softmax(x, -1), -1).mean()
    loss.backward()
    assert abs(acc.item() - 0.5 - loss.item()) < 1e-5 * abs(loss.item())
    assert relerr(real[:, :K_], x.grad) < (1e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("T,C,F,HD", [(257 * 8, 64, 192, 64), (65 * 16, 128, 384, 96), (17 * 6, 256, 768, 192), (257 * 16, 256, 768, 256),
                                      (65 * 32, 512, 1536, 512)])
@pytest.mark.parametrize("sched", [0, 64, 0x10000])
def test_gemm_group_equals_the_launches_one_by_one(T, C, F, HD, sched):
    """vr_gemm_group: the four weight gradients of a transformer block as one launch (masked tiles, bias gradients, per-head
    periods included) against the same calls issued through vr_gemm -- and against the torch statement.  sched 0: the 4-wave
    kernel with the lean instruction stream of round 6 (tn8_group_kernel<4>: LDS-DMA through a buffer descriptor, tokens past a
    split's end and masked channel chunks through its range check); 64: tn_body's stream; 0x10000: the 8-wave, double-buffered
    form (one workgroup per CU, one token split for the whole group)."""
    import functools
    dt = torch.bfloat16
    rps = T // 8 if T % 8 == 0 else T // 2
    nb = T // rps
    keep_c = torch.tensor([C, C // 2] * (nb // 2), dtype=torch.int32)
    keep_f = torch.tensor([F, F // 3] * (nb // 2), dtype=torch.int32)
    keep_h = torch.tensor([HD, HD // 2] * (nb // 2), dtype=torch.int32)
    shapes = [("fc2", C, F, keep_c, keep_f, 0), ("fc1", F, C, keep_f, keep_c, 0), ("proj", C, HD, keep_c, keep_h, 0),
              ("qkv", 3 * HD, C, keep_h, keep_c, HD)]
    calls_cpu, calls_gpu, outs_gpu, outs_one = [], [], [], []
    for i, (name, out_f, in_f, kr, kc, period) in enumerate(shapes):
        dy = rnd(T, out_f, seed=10 + i).to(dt)
        x = rnd(T, in_f, seed=20 + i).to(dt)
        for s_ in range(nb):                                         # masked channels are exact zeros by contract
            lim = int(kr[s_])
            if period:
                dy[s_ * rps:(s_ + 1) * rps].view(rps, 3, period)[:, :, lim:] = 0
            else:
                dy[s_ * rps:(s_ + 1) * rps, lim:] = 0
            x[s_ * rps:(s_ + 1) * rps, int(kc[s_]):] = 0
        kw = dict(M=out_f, N=in_f, K=T, lda=out_f, ldb=in_f, ldc=in_f, a_trans=True, b_trans=True, atomic=True, split_k=0,
                  k_period=period, rows_in=rps, sched=sched)
        dw_ref, db_ref = torch.zeros(out_f, in_f), torch.zeros(out_f)
        calls_cpu.append((dy, x, dw_ref, dict(kw, keep_k=kr, keep_n=kc, bias_grad=db_ref)))
        dw_g, db_g = torch.zeros(out_f, in_f, device=DEV), torch.zeros(out_f, device=DEV)
        dw_1, db_1 = torch.zeros(out_f, in_f, device=DEV), torch.zeros(out_f, device=DEV)
        calls_gpu.append((dy.to(DEV), x.to(DEV), dw_g, dict(kw, keep_k=kr.to(DEV), keep_n=kc.to(DEV), bias_grad=db_g)))
        outs_gpu.append((dw_g, db_g))
        outs_one.append((dw_1, db_1, dict(kw, keep_k=kr.to(DEV), keep_n=kc.to(DEV), bias_grad=db_1)))
    K.gemm_group(calls_gpu)
    E.gemm_group(calls_cpu)
    for (a, b, _, _), (dw_1, db_1, kw1) in zip(calls_gpu, outs_one):
        K.gemm(a, b, dw_1, **kw1)
    torch.cuda.synchronize()
    for (dw_g, db_g), (dw_1, db_1, _), (_, _, dw_ref, kwr) in zip(outs_gpu, outs_one, calls_cpu):
        assert relerr(dw_g, dw_1) < 2e-5 and relerr(db_g, db_1) < 2e-5          # same products, another split / summation order
        assert relerr(dw_g, dw_ref) < 1e-4 and relerr(db_g, kwr["bias_grad"]) < 2e-4      # fp32 results of bf16 operands


@pytest.mark.parametrize("M,C,F", [(257 * 4, 256, 768), (65 * 3, 128, 392)])
def test_gemm_saved_gelu_derivative_pair(M, C, F):
    """act = 2: fc1 stores (gelu'(u), gelu(u)); fc2's data gradient multiplies by the saved derivative -- the same du as the
    (u, gelu(u)) / gelu'(u)-in-the-epilogue pair of act = 1, and the torch statement."""
    dt = torch.bfloat16
    B = M // (257 if M % 257 == 0 else 65)
    rps = M // B
    y, w1 = _bf(rnd(M, C, seed=1)), _bf(rnd(F, C, seed=2, scale=C ** -0.5))
    b1 = rnd(F, seed=3)
    keep = torch.tensor([F - 64 * (i % 2) for i in range(B)], dtype=torch.int32)
    kw = dict(M=M, N=F, K=C, lda=C, ldb=C, ldc=F, bias=b1, keep_n=keep, rows_in=rps)
    u_ref, h_ref = torch.empty(M, F, dtype=dt), torch.empty(M, F, dtype=dt)
    d_ref, h2_ref = torch.empty(M, F, dtype=dt), torch.empty(M, F, dtype=dt)
    E.gemm(y, w1, u_ref, out2=h_ref, act=1, **kw)
    E.gemm(y, w1, d_ref, out2=h2_ref, act=2, **kw)
    cu = lambda t_: t_.to(DEV) if isinstance(t_, torch.Tensor) else t_
    kwd = {k: cu(v) for k, v in kw.items()}
    d, h = torch.empty(M, F, dtype=dt, device=DEV), torch.empty(M, F, dtype=dt, device=DEV)
    K.gemm(cu(y), cu(w1), d, out2=h, act=2, **kwd)
    assert relerr(h, h_ref) < 1.2e-2 and relerr(d, d_ref) < 1.2e-2
    for i in range(B):                                                 # masked hidden units: derivative stored as 0
        tail = d.view(B, rps, F)[i, :, int(keep[i]):]
        assert tail.numel() == 0 or float(tail.abs().max()) == 0.0
    # data gradient of fc2: du = (gt @ W2) * gelu'(u)
    gt, w2t = _bf(rnd(M, C, seed=5)), _bf(rnd(F, C, seed=6, scale=C ** -0.5))      # W2^T rows: [F, C] K-contiguous
    kw2 = dict(M=M, N=F, K=C, lda=C, ldb=C, ldc=F, ldu=F, keep_n=keep, rows_in=rps)
    du_ref = torch.empty(M, F, dtype=dt)
    E.gemm(gt, w2t, du_ref, dact_u=u_ref, **kw2)
    du = torch.empty(M, F, dtype=dt, device=DEV)
    K.gemm(cu(gt), cu(w2t), du, dact_u=d, act=2, **{k: cu(v) for k, v in kw2.items()})
    assert relerr(du, du_ref) < 2e-2, relerr(du, du_ref)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K_,rows_in", [(2176, 1024, 3072, 17), (8320, 512, 1536, 65), (1300, 320, 640, 65), (2176, 768, 1024, 17),
                                            (4 * 257, 1536, 512, 257), (33 * 256, 256, 768, 256), (5 * 257, 776, 256, 257)])
@pytest.mark.parametrize("variant", ["fwd", "gelu", "res", "dgrad", "dmul"])
def test_gemm_lean_loop(M, N, K_, rows_in, variant):
    """gemm_ntk.hip (round 5: buffer-addressed LDS-DMA, live-slice bit mask, 1 / 2 / 3 slices in flight) in every tile x depth
    combination (sched bits 0x1800 / 0x600) against the emulation and against gemm_nt.hip's kernel (sched 0x100): prefix masks on
    both sides, ragged row and column tiles, launch after launch beside a busy second stream."""
    a, b, out, kw = _wide_case(M, N, K_, variant, rows_in)
    ref = E.gemm(a, b, out.clone(), **{k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()})
    to = lambda v: v.to(DEV) if isinstance(v, torch.Tensor) else v
    kw_d = {k: to(v) for k, v in kw.items()}
    ad, bd = a.to(DEV), b.to(DEV)
    t_ = 1e-4 if out.dtype == torch.float32 else 8e-3
    old = K.gemm(ad, bd, torch.full_like(out, float("nan")).to(DEV), sched=0x100, **kw_d).clone()
    assert relerr(old, ref) < t_
    side = torch.cuda.Stream()
    busy = torch.randn(4096, 4096, device=DEV)
    with torch.cuda.stream(side):
        for _ in range(3):
            busy = torch.tanh(busy @ busy * 1e-2)
    for tile in (1, 2, 3):
        for nbuf in (1, 2, 3):
            sched = (tile << 11) | (nbuf << 9)
            for rep in range(2):
                real = K.gemm(ad, bd, torch.full_like(out, float("nan")).to(DEV), sched=sched, **kw_d)
                torch.cuda.synchronize()
                assert relerr(real, ref) < t_, (variant, tile, nbuf, rep, relerr(real, ref))
                # same products, same summation order per output as the old kernel of the same tile would use: tight agreement
                assert relerr(real, old) < (2e-5 if out.dtype == torch.float32 else 8e-3)
            if variant == "gelu":
                ref2 = torch.zeros(M, N, dtype=torch.bfloat16)
                kw2 = dict(kw); kw2["out2"] = ref2
                E.gemm(a, b, out.clone(), **kw2)
                assert relerr(kw_d["out2"], ref2) < t_


@pytest.mark.gpu
@pytest.mark.parametrize("nbuf", [1, 2, 3])
def test_gemm_lean_loop_periodic_masks_and_rowmaps(nbuf):
    """Lean-loop kernel with the per-head (periodic) k masks of the attention projection's data gradient -- live slices that are not
    a prefix, slice 0 dead for some tiles -- fully masked samples, and mapped token rows (cls / patch rows of the embedding)."""
    B, Nt, H, D, C = 6, 65, 4, 64, 256
    HD, M = H * D, B * Nt
    sched = (1 << 11) | (nbuf << 9)
    to = lambda v: v.to(DEV) if isinstance(v, torch.Tensor) else v
    ek = torch.tensor([256, 160, 256, 64, 192, 160], dtype=torch.int32)
    for ak in (torch.tensor([256, 128, 64, 192, 128, 256], dtype=torch.int32), torch.tensor([0, 0, 0, 0, 0, 0], dtype=torch.int32),
               torch.tensor([64, 64, 64, 64, 64, 64], dtype=torch.int32)):
        cols = torch.arange(3 * HD) % HD
        dqkv = (rnd(M, 3 * HD, seed=4) * (cols[None, :] < ak.long().repeat_interleave(Nt)[:, None])).to(torch.bfloat16)
        w = rnd(3 * HD, C, seed=2, scale=C ** -0.5).to(torch.bfloat16)
        kw = dict(M=M, N=C, K=3 * HD, lda=3 * HD, ldb=C, ldc=C, b_trans=True, rows_in=Nt, keep_k=ak, k_period=HD, keep_n=ek)
        ref = E.gemm(dqkv, w, torch.zeros(M, C, dtype=torch.bfloat16), **kw)
        real = K.gemm(dqkv.to(DEV), w.to(DEV), torch.full((M, C), 7.0, dtype=torch.bfloat16, device=DEV), sched=sched,
                      **{k: to(v) for k, v in kw.items()})
        assert relerr(real, ref) < tol(torch.bfloat16)
    # periodic masks whose first slice is dead: period 128, keep 0 in the first half is impossible for a prefix -- use a row map
    # instead: rows gathered through a_map / scattered through c_map
    rpi, rps = 64, 65                                                    # patch rows of [B, 65, C]: skip the cls row of every sample
    Mp = B * rpi
    x = rnd(B * rps, C, seed=9).to(torch.bfloat16)
    w2 = rnd(512, C, seed=10, scale=C ** -0.5).to(torch.bfloat16)
    kw = dict(M=Mp, N=512, K=C, lda=C, ldb=C, ldc=512, bias=rnd(512, seed=11), a_map=(rpi, rps, 1), rows_in=rpi)
    ref = E.gemm(x, w2, torch.zeros(Mp, 512, dtype=torch.bfloat16), **kw)
    real = K.gemm(x.to(DEV), w2.to(DEV), torch.zeros(Mp, 512, dtype=torch.bfloat16, device=DEV), sched=sched, **{k: to(v) for k, v in kw.items()})
    assert relerr(real, ref) < tol(torch.bfloat16)


@pytest.mark.gpu
@pytest.mark.parametrize("tile", [1, 2, 3])
def test_gemm_masked_tiles_unwritten_and_readers_refused(tile, monkeypatch):
    """Round 5, vr_gemm_args.sched 0x40000 / 0x80000 and the per-group grid (m_groups divides the rows): four architecture groups of
    widths 512 / 256 / 128 / 0 (a dropped layer), one sample of the widest group masked on its own (keep -(512 + 2)).  Written: every
    tile that holds a kept column of its GROUP -- values of the emulation, zeros beyond a row's width, zeros for the marked sample;
    left as they were (NaN here): the tiles beyond the group's width and the whole dropped group.  Without the bit everything is
    written.  A reader of such an operand (bit 0x80000) that the group-by-group kernels do not cover (K % 64 != 0 -> gemm_nt.hip)
    fails instead of running."""
    G, spg, rows_in, N, K_ = 4, 2, 65, 512, 256
    M = G * spg * rows_in                                           # 130 rows per group: two short tiles at 128, three at 64
    widths = [512, 256, 128, 0]
    keep = torch.tensor([w for w in widths for _ in range(spg)], dtype=torch.int32)
    keep[1] = -(512 + 2)
    a, b = rnd(M, K_, seed=1).to(torch.bfloat16), rnd(N, K_, seed=2, scale=K_ ** -0.5).to(torch.bfloat16)
    bias = rnd(N, seed=3)
    kw = dict(M=M, N=N, K=K_, lda=K_, ldb=K_, ldc=N, bias=bias, keep_n=keep, rows_in=rows_in)
    ref = E.gemm(a, b, torch.zeros(M, N, dtype=torch.bfloat16), **kw)
    to = lambda v: v.to(DEV) if isinstance(v, torch.Tensor) else v
    kw_d = {k: to(v) for k, v in kw.items()}
    BN = 64 if tile == 3 else 128
    K.M_GROUPS[0] = G
    try:
        for skip in (False, True):
            K.WRITE_SKIP[0] = skip
            out = K.gemm(a.to(DEV), b.to(DEV), torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV), sched=tile << 11,
                         **kw_d).cpu()
            for g_, w in enumerate(widths):
                rows = slice(g_ * spg * rows_in, (g_ + 1) * spg * rows_in)
                wr = (w + BN - 1) // BN * BN if skip else N           # columns of the group's written tiles
                assert torch.isfinite(out[rows, :wr]).all(), (skip, g_)
                if wr:
                    assert relerr(out[rows, :wr], ref[rows, :wr]) < 8e-3, (skip, g_)
                assert torch.isnan(out[rows, wr:]).all(), (skip, g_)  # untouched
            assert float(out[rows_in:2 * rows_in].abs().max()) == 0.0  # the sample masked on its own: zeros, stored
        # a reader the group-by-group kernels do not cover
        K.WRITE_SKIP[0] = False
        K2 = 96
        a2, b2 = rnd(M, K2, seed=4).to(torch.bfloat16).to(DEV), rnd(N, K2, seed=5).to(torch.bfloat16).to(DEV)
        kk = torch.tensor([w * K2 // 512 for w in widths for _ in range(spg)], dtype=torch.int32, device=DEV)
        o2 = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
        K.gemm(a2, b2, o2, M=M, N=N, K=K2, lda=K2, ldb=K2, ldc=N, keep_k=kk, rows_in=rows_in)       # fine without the bit
        with pytest.raises(RuntimeError, match="VR_EUNSUPPORTED"):
            K.gemm(a2, b2, o2, M=M, N=N, K=K2, lda=K2, ldb=K2, ldc=N, keep_k=kk, rows_in=rows_in, sched=K.READS_SKIPPED_BIT)
    finally:
        K.M_GROUPS[0] = 1
        K.WRITE_SKIP[0] = False
