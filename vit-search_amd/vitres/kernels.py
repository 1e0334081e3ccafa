"""Tensor-level wrappers over the C ABI (include/vitres_hip.h).

Each function takes torch CUDA tensors, passes raw device pointers + sizes to libvitres_hip.so on
torch's current stream, and returns torch tensors it allocated for the outputs.  PyTorch is used
for device memory and streams only.  Non-CUDA tensors raise: there is no CPU path.
"""
import ctypes

import torch

from . import _lib
from ._lib import GemmArgs, LnEpilogue, RowMap, VR_BF16, VR_F32

IDENT = (0, 0, 0)

# bench.py instrumentation: when a list, every vr_gemm launch is bracketed by HIP events on torch's current stream
# (the stream the kernel is launched on) and (kind, flops, bytes, ev0, ev1) is appended.
PROFILE = None
PROFILE_ATTN = None       # with PROFILE: (kept FLOPs, event, event) per attention launch
PROFILE_DESC = None       # with PROFILE: one description per entry (tools/gemm_launches.py)


def _dt(t):
    if t.dtype == torch.float32:
        return VR_F32
    if t.dtype == torch.bfloat16:
        return VR_BF16
    raise TypeError("unsupported dtype %s" % t.dtype)


def _dtcode(dtype):
    return VR_F32 if dtype == torch.float32 else VR_BF16


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("vitres kernels need CUDA/HIP tensors (got %s); there is no CPU fallback" % t.device)
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _rm(m):
    return RowMap(*(m or IDENT))


# Workspaces of the split-K GEMM form (vr_gemm_args.ws): one per (device, role).  A role is a chain of launches that never overlap
# one another: 0 = the main stream, 1.. = the side streams of vitres.functional (which switches the role around what it runs
# there).  Created zeroed on first use -- outside graph capture (engine.GraphedTrainStep calls ensure_workspaces first): a launch
# that finds none while capturing simply runs without tile sharing.
_WS = {}
_WS_ROLE = [0]
# dev aid (A/B inside the step): bits OR-ed into vr_gemm_args.sched / value of k_shares of every forward / data-gradient launch
_DBG_SCHED_OR = int(__import__("os").environ.get("VITRES_DBG_SCHED_OR", "0"), 0)
_DBG_K_SHARES = int(__import__("os").environ.get("VITRES_DBG_K_SHARES", "0"))


class ws_role:
    def __init__(self, role):
        self.role = role

    def __enter__(self):
        self.prev, _WS_ROLE[0] = _WS_ROLE[0], self.role

    def __exit__(self, *exc):
        _WS_ROLE[0] = self.prev


def _workspace(dev, role=None):
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), _WS_ROLE[0] if role is None else role)
    ws = _WS.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        ws = _WS[key] = torch.zeros(int(_lib.lib().vr_gemm_ws_bytes()), dtype=torch.uint8, device=dev)
    return ws


def ensure_workspaces(dev, roles=(0, 1)):
    for r in roles:
        _workspace(torch.device(dev), r)


# Architecture groups of the batch being processed (vr_gemm_args.m_groups): the model sets it from its plan at the start of a
# forward / backward (G = B / example_per_arch contiguous groups of samples in the arch-grouped execution order, each with its own
# keep rows); every GEMM that carries keep arrays passes it on, so that the kernels deal every group to every XCD.
# VITRES_GROUP_INTERLEAVE=0 keeps the plain tile order (measurement).
M_GROUPS = [1]
# The model vouches (vit_sr_supernet.sample_plan -> plan.skip_writes) that every tile / token split of the bf16 kernels sees ONE
# architecture of the batch being processed and that every masked Linear of the network is one the group-pure kernels cover: the
# masked GEMMs then carry sched bit 0x40000 -- producers leave fully masked output tiles unwritten, and a launch that would have
# to read such an operand through a kernel that tiles across groups fails (VR_EUNSUPPORTED) instead of reading them.
WRITE_SKIP = [False]
SKIP_WRITES_BIT = 0x40000
READS_SKIPPED_BIT = 0x80000


def reads_skipped():
    """sched bit for a GEMM whose operand (the MLP's hidden activation or its gradient) was produced under WRITE_SKIP: the launch
    must go to a kernel that tiles group by group, or fail."""
    return READS_SKIPPED_BIT if WRITE_SKIP[0] else 0


# test aid (tests/test_gpu_model.py): fill the output of every launch that may leave tiles unwritten with NaN first -- a reader of
# an unwritten tile then changes the loss / gradients instead of quietly reading whatever the arena held
DBG_POISON = [__import__("os").environ.get("VITRES_DBG_POISON", "0") != "0"]
_GROUP_INTERLEAVE = __import__("os").environ.get("VITRES_GROUP_INTERLEAVE", "1") != "0"


def _gemm_args(a, b, out, *, M, N, K, lda, ldb, ldc, a_trans=False, b_trans=False, out2=None, bias=None, pos=None,
               scale=None, keep_n=None, resid=None, dact_u=None, ldu=0, act=0, atomic=False, split_k=1, rows_in=0,
               a_map=None, b_map=None, c_map=None, bias_grad=None, keep_k=None, n_period=0, k_period=0, sched=0, ws="auto",
               ring=0, k_shares=0, m_groups=None):
    args = GemmArgs()
    if isinstance(ws, str):         # "auto": the role's workspace wherever the K-split kernels (gemm_ntk.hip) could be chosen
        ws = _workspace(a.device) if (not a_trans and a.dtype == torch.bfloat16 and a.is_cuda and K >= 512 and k_shares != 1) else None
    if ws is not None:
        args.ws, args.ws_bytes = ws.data_ptr(), ws.numel() * ws.element_size()
    args.ring, args.k_shares = ring, (k_shares or (_DBG_K_SHARES if not a_trans else 0))
    if not a_trans:
        sched |= _DBG_SCHED_OR
    args.A, args.B, args.C, args.C2 = _p(a), _p(b), _p(out), _p(out2)
    args.bias, args.pos, args.scale, args.keep_n = _p(bias), _p(pos), _p(scale), _p(keep_n)
    args.resid, args.dact_u, args.bias_grad, args.keep_k = _p(resid), _p(dact_u), _p(bias_grad), _p(keep_k)
    args.n_period, args.k_period, args.sched = n_period, k_period, sched
    mg = M_GROUPS[0] if m_groups is None else m_groups      # (closures launched later pass the value of their own forward / backward)
    args.m_groups = mg if (_GROUP_INTERLEAVE and (keep_k is not None or keep_n is not None) and rows_in > 0) else 0
    if WRITE_SKIP[0] and (keep_k is not None or keep_n is not None) and rows_in > 0 and (mg <= 1 or args.m_groups > 1):
        args.sched |= SKIP_WRITES_BIT
    args.M, args.N, args.K = M, N, K
    args.lda, args.ldb, args.ldc, args.ldu = lda, ldb, ldc, ldu
    args.a_trans, args.b_trans = int(a_trans), int(b_trans)
    args.in_dtype, args.out_dtype = _dt(a), _dt(out)
    assert b.dtype == a.dtype, "A and B must share a dtype"
    args.act, args.atomic, args.split_k, args.rows_in = act, int(atomic), split_k, rows_in
    args.a_map, args.b_map, args.c_map = _rm(a_map), _rm(b_map), _rm(c_map)
    return args


def gemm_ln_supported(a, N, ldc):
    """vr_gemm_ln covers this Linear (bf16 operands, whole rows in one tile)."""
    return a.dtype == torch.bfloat16 and N == ldc and bool(_lib.lib().vr_gemm_ln_supported(N))


def _kept_flops(M, N, K, rows_in, keep_k, keep_n, k_period=0):
    def kept(keep, dim, period):
        if keep is None:
            return None
        k = keep.detach().to("cpu", torch.float64)
        return torch.clamp(k, min=0, max=period) * (dim // period) if period else torch.clamp(k, min=0, max=dim)
    kk, kn = kept(keep_k, K, k_period), kept(keep_n, N, 0)
    if kk is None and kn is None:
        return 2.0 * M * N * K
    nb = len(kk if kk is not None else kn)
    kk = kk if kk is not None else torch.full((nb,), float(K), dtype=torch.float64)
    kn = kn if kn is not None else torch.full((nb,), float(N), dtype=torch.float64)
    return float((2.0 * rows_in * kk * kn).sum())


def _launch_gemm_ln(args, ln, a, M, N, K, rows_in, keep_k, keep_n, k_period, extra_bytes):
    if PROFILE is None:
        _lib.check(_lib.lib().vr_gemm_ln(ctypes.byref(args), ctypes.byref(ln), _stream()), "vr_gemm_ln")
        return
    flops = _kept_flops(M, N, K, rows_in, keep_k, keep_n, k_period)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(_lib.lib().vr_gemm_ln(ctypes.byref(args), ctypes.byref(ln), _stream()), "vr_gemm_ln")
    e1.record()
    # algorithmic bytes from the KEPT widths, like _gemm_work: sample b reads rows x kept K of A; the weights once for the widest
    # sample; the row-wide side tensors (residual stream, LayerNorm output ...) in full
    if keep_k is not None and rows_in > 0:
        kk = torch.clamp(keep_k.detach().to("cpu", torch.float64), min=0, max=k_period) * (K // k_period) if k_period else \
            torch.clamp(keep_k.detach().to("cpu", torch.float64), min=0, max=K)
        a_bytes, k_w = float((rows_in * kk).sum()) * 2, float(kk.max())
    else:
        a_bytes, k_w = float(M) * K * 2, float(K)
    PROFILE.append((("bf16", 0, 0, 2), flops, 2.0 * M * N * K, a_bytes + N * k_w * 2 + extra_bytes, e0, e1))
    if PROFILE_DESC is not None:
        PROFILE_DESC.append("ln%d M%d N%d K%d" % (ln.mode, M, N, K))


def gemm_ln_fwd(a, b, out, ln_w, ln_b, ln_keep, eps, *, M, N, K, lda, ldb, ldc, bias=None, scale=None, keep_n=None,
                resid=None, rows_in=0, keep_k=None, k_period=0, sched=0):
    """out = resid + scale * mask(a @ b^T + bias) (fp32) and (y, mean, rstd) = masked LayerNorm(out) -- vr_gemm_ln mode 0."""
    args = _gemm_args(a, b, out, M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=ldc, bias=bias, scale=scale, keep_n=keep_n, resid=resid,
                      rows_in=rows_in, keep_k=keep_k, k_period=k_period, sched=sched)
    y = torch.empty(out.shape, dtype=torch.bfloat16, device=out.device)
    mean = torch.empty(M, dtype=torch.float32, device=out.device)
    rstd = torch.empty(M, dtype=torch.float32, device=out.device)
    ln = LnEpilogue()
    ln.mode, ln.eps = 0, eps
    ln.w, ln.b, ln.keep, ln.y, ln.mean, ln.rstd = _p(ln_w), _p(ln_b), _p(ln_keep), _p(y), _p(mean), _p(rstd)
    _launch_gemm_ln(args, ln, a, M, N, K, rows_in, keep_k, keep_n, k_period, M * N * (4 + 4 + 2))
    return y, mean, rstd


# vr_gemm_ln_fold (rows of several 128-column tiles; narrower rows go to vr_gemm_ln) is OPT-IN: measured round 6 inside the sr_tiny step
# 7.51 against 6.97 ms (52 - 73 us per folded launch against 22 - 35 + 5 - 8 for the two kernels, profiles/r06_ln_fold.txt): the tiles'
# fp32 rows must be visible to a workgroup on ANOTHER XCD inside the launch, i.e. stored write-through (partial-line writes to
# memory) and waited for, and the last arriver's row loop is a tail nothing runs beside.
LN_FOLD = __import__("os").environ.get("VITRES_LN_FOLD", "0") != "0"


def gemm_ln_fold_fwd(a, b, out, ln_w, ln_b, ln_keep, eps, *, M, N, K, lda, ldb, ldc, bias=None, scale=None, keep_n=None,
                     resid=None, rows_in=0, keep_k=None, k_period=0, sched=0):
    """gemm_ln_fwd for rows that span several tiles (vr_gemm_ln_fold): out = resid + scale * mask(a @ b^T + bias) (fp32) and
    (y, mean, rstd) = masked LayerNorm(out), one launch of the tiled kernel -- or None (nothing launched) when the form is not
    covered: the caller then issues the Linear and the LayerNorm separately."""
    if not (LN_FOLD and a.is_cuda and a.dtype == torch.bfloat16 and out.dtype == torch.float32 and N == ldc and 256 < N <= 2048 and
            bias is not None and resid is not None and K % 64 == 0):
        return None
    args = _gemm_args(a, b, out, M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=ldc, bias=bias, scale=scale, keep_n=keep_n, resid=resid,
                      rows_in=rows_in, keep_k=keep_k, k_period=k_period, sched=sched, k_shares=0,
                      ws=_workspace(a.device))
    if not args.ws:
        return None
    y = torch.empty(out.shape, dtype=torch.bfloat16, device=out.device)
    mean = torch.empty(M, dtype=torch.float32, device=out.device)
    rstd = torch.empty(M, dtype=torch.float32, device=out.device)
    ln = LnEpilogue()
    ln.mode, ln.eps = 0, eps
    ln.w, ln.b, ln.keep, ln.y, ln.mean, ln.rstd = _p(ln_w), _p(ln_b), _p(ln_keep), _p(y), _p(mean), _p(rstd)
    if PROFILE is not None:
        flops, alg_bytes = _gemm_work(a, out, M, N, K, False, rows_in, keep_k, keep_n, k_period, 0, resid=resid, bias=bias)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = _lib.lib().vr_gemm_ln_fold(ctypes.byref(args), ctypes.byref(ln), _stream())
    if rc == -3:                                   # VR_EUNSUPPORTED: nothing was launched
        return None
    _lib.check(rc, "vr_gemm_ln_fold")
    if PROFILE is not None:
        e1.record()
        PROFILE.append((("bf16", 0, 0, 0), flops, 2.0 * M * N * K, alg_bytes + M * N * 2, e0, e1))       # (+ the LayerNorm's output)
        if PROFILE_DESC is not None:
            PROFILE_DESC.append("nt+ln M%d N%d K%d bias res%s f32out" % (M, N, K, " scale" if scale is not None else ""))
    return y, mean, rstd


def gemm_ln_bwd(du, wt, x, ln_w, mean, rstd, ln_keep, dx_in, dw, db, next_cast=None, *, M, N, K, lda, ldb, rows_in=0,
                keep_k=None, k_period=0, copies=1, sched=0):
    """LayerNorm backward of dy = du @ wt^T without writing dy -- vr_gemm_ln mode 1; returns dx or (dx, gt) like ln_bwd
    (copies: dw / db are [copies, N] partial rows, see ln_bwd)."""
    dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    gt = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if next_cast is not None else None
    sc, kp = next_cast if next_cast is not None else (None, None)
    args = _gemm_args(du, wt, dx, M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=N, resid=dx_in, rows_in=rows_in, keep_k=keep_k,
                      k_period=k_period, sched=sched)
    ln = LnEpilogue()
    ln.mode, ln.eps = 1, 0.0
    ln.w, ln.keep, ln.mean, ln.rstd, ln.x = _p(ln_w), _p(ln_keep), _p(mean), _p(rstd), _p(x)
    ln.dw, ln.db, ln.gt_out, ln.gt_scale, ln.gt_keep = _p(dw), _p(db), _p(gt), _p(sc), _p(kp)
    ln.grad_copies = copies
    _launch_gemm_ln(args, ln, du, M, N, K, rows_in, keep_k, ln_keep, k_period, M * N * (4 + 4 + 4 + 2))
    return dx if next_cast is None else (dx, gt)


def gemm(a, b, out, *, M, N, K, lda, ldb, ldc, a_trans=False, b_trans=False, out2=None, bias=None, pos=None,
         scale=None, keep_n=None, resid=None, dact_u=None, ldu=0, act=0, atomic=False, split_k=1, rows_in=0,
         a_map=None, b_map=None, c_map=None, bias_grad=None, keep_k=None, n_period=0, k_period=0, sched=0, ws="auto",
         ring=0, k_shares=0, m_groups=None):
    """out[M,N] = epilogue(a[M,K] @ b[N,K]^T) -- see vr_gemm in include/vitres_hip.h.  ws: K-split workspace (a uint8 tensor),
    None, or "auto" = the current role's (see _workspace)."""
    args = _gemm_args(a, b, out, M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=ldc, a_trans=a_trans, b_trans=b_trans, out2=out2,
                      bias=bias, pos=pos, scale=scale, keep_n=keep_n, resid=resid, dact_u=dact_u, ldu=ldu, act=act,
                      atomic=atomic, split_k=split_k, rows_in=rows_in, a_map=a_map, b_map=b_map, c_map=c_map,
                      bias_grad=bias_grad, keep_k=keep_k, n_period=n_period, k_period=k_period, sched=sched, ws=ws,
                      ring=ring, k_shares=k_shares, m_groups=m_groups)
    if DBG_POISON[0] and (args.sched & SKIP_WRITES_BIT) and keep_n is not None and not a_trans and resid is None and \
            out.dtype == torch.bfloat16:
        out.fill_(float("nan"))
        if out2 is not None:
            out2.fill_(float("nan"))
    if PROFILE is None:
        _lib.check(_lib.lib().vr_gemm(ctypes.byref(args), _stream()), "vr_gemm")
        return out
    flops, alg_bytes = _gemm_work(a, out, M, N, K, a_trans, rows_in, keep_k, keep_n, k_period, n_period, out2=out2, resid=resid,
                                  dact_u=dact_u, pos=pos, bias=bias, atomic=atomic,
                                  skip_writes=bool(args.sched & SKIP_WRITES_BIT))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(_lib.lib().vr_gemm(ctypes.byref(args), _stream()), "vr_gemm")
    e1.record()
    PROFILE.append((("bf16" if args.in_dtype == VR_BF16 else "f32", int(a_trans), int(b_trans),
                     int(a_map is not None or b_map is not None)), flops, 2.0 * M * N * K, alg_bytes, e0, e1))
    if PROFILE_DESC is not None:
        PROFILE_DESC.append("%s M%d N%d K%d%s%s%s%s%s%s%s" % (
            "tn" if a_trans else ("nn" if b_trans else "nt"), M, N, K, " act%d" % act if act else "", " bias" if bias is not None else "",
            " res" if resid is not None else "", " scale" if scale is not None else "", " dact" if dact_u is not None else "",
            " map" if (a_map or b_map or c_map) else "", " f32out" if out.dtype == torch.float32 and not a_trans else ""))
    return out


def gemm_group(calls):
    """calls: [(a, b, out, kwargs)] -- the arguments of gemm() for each problem.  One vr_gemm_group launch (weight gradients of
    a transformer block share the CUs); semantics are exactly those of issuing the calls one after the other."""
    if len(calls) == 1:
        a, b, out, kw = calls[0]
        return gemm(a, b, out, **kw)
    arr = (GemmArgs * len(calls))()
    for i, (a, b, out, kw) in enumerate(calls):
        arr[i] = _gemm_args(a, b, out, **kw)
    if PROFILE is None:
        _lib.check(_lib.lib().vr_gemm_group(arr, len(calls), _stream()), "vr_gemm_group")
        return
    flops = dense = alg = 0.0
    for a, b, out, kw in calls:
        f, by = _gemm_work(a, out, kw["M"], kw["N"], kw["K"], kw.get("a_trans", False), kw.get("rows_in", 0), kw.get("keep_k"),
                           kw.get("keep_n"), kw.get("k_period", 0), kw.get("n_period", 0), atomic=kw.get("atomic", False))
        flops, dense, alg = flops + f, dense + 2.0 * kw["M"] * kw["N"] * kw["K"], alg + by
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(_lib.lib().vr_gemm_group(arr, len(calls), _stream()), "vr_gemm_group")
    e1.record()
    a0, kw0 = calls[0][0], calls[0][3]
    PROFILE.append((("bf16" if a0.dtype == torch.bfloat16 else "f32", int(kw0.get("a_trans", False)),
                     int(kw0.get("b_trans", False)), 0), flops, dense, alg, e0, e1))
    if PROFILE_DESC is not None:
        PROFILE_DESC.append("group " + " + ".join("M%d N%d K%d" % (kw["M"], kw["N"], kw["K"]) for _, _, _, kw in calls))


def _written_cols(keep_n, N, n_period, BN=128):
    """Per sample: the output columns a launch under SKIP_WRITES_BIT actually stores -- the BN-wide column tiles that hold a kept
    column for the sample's architecture GROUP (gemm_ntk.hip: a tile beyond the group's width is not written; a DropPath-dropped
    sample, marked -(k + 2), still gets its group's width k: its zeros below k are stored)."""
    gw = keep_n.detach().to("cpu", torch.int64)
    gw = torch.where(gw < 0, -gw - 2, gw)
    out = torch.zeros(gw.shape, dtype=torch.float64)
    for w in set(gw.tolist()):
        cols = 0
        for n0 in range(0, N, BN):
            ln = min(BN, N - n0)
            if w <= 0:
                live = False
            elif n_period <= 0:
                live = n0 < w
            else:
                r = n0 % n_period
                live = r < w or r + ln > n_period
            cols += ln if live else 0
        out[gw == w] = cols
    return out


def _gemm_work(a, out, M, N, K, a_trans, rows_in, keep_k, keep_n, k_period, n_period, out2=None, resid=None, dact_u=None,
               pos=None, bias=None, atomic=False, skip_writes=False):
    """(kept FLOPs, algorithmic HBM bytes) of one vr_gemm launch, both from the KEPT (un-masked) widths of the samples it
    touches -- skipped work is never counted, and neither are operand bytes the masks let the kernel skip.
      forward / dgrad (C[M,N] = A[M,K] B[N,K]^T): sample b reads rows_in x kk[b] of A; the weights are read once for the widest
        sample: max(kk) x max(kn).  Outputs and epilogue side tensors (fp32 residual, saved gelu', second output): a launch that
        stores zeros in masked columns (downstream kernels read whole rows) counts rows_in x N per sample; a launch under
        SKIP_WRITES_BIT (skip_writes: bf16 result, no residual) counts only the 128-column tiles its architecture group keeps
        (_written_cols) -- a fully masked tile is neither written nor is its saved gelu' read, a dropped layer counts nothing;
      wgrad (C[M,N] += A[T,M]^T B[T,N], a_trans): sample b reads rows_in x (kr[b] + kc[b]) channels; C (fp32) is read-modified-
        written once over its widest kept block."""
    esz, osz = a.element_size(), out.element_size()

    def kept(keep, dim, period):
        if keep is None:
            return None
        k = keep.detach().to("cpu", torch.float64)
        return torch.clamp(k, min=0, max=period) * (dim // period) if period else torch.clamp(k, min=0, max=dim)
    if a_trans:                                   # wgrad: keep_k bounds output rows (M), keep_n output columns (N); K = tokens
        kr, kc = kept(keep_k, M, k_period), kept(keep_n, N, n_period)
        if kr is None and kc is None:
            return 2.0 * M * N * K, float(K * (M + N) * esz + 2 * M * N * 4)
        nb = len(kr if kr is not None else kc)
        kr = kr if kr is not None else torch.full((nb,), float(M), dtype=torch.float64)
        kc = kc if kc is not None else torch.full((nb,), float(N), dtype=torch.float64)
        flops = float((2.0 * rows_in * kr * kc).sum())
        return flops, float((rows_in * (kr + kc)).sum() * esz + 2 * float(kr.max()) * float(kc.max()) * 4)
    kk, kn = kept(keep_k, K, k_period), kept(keep_n, N, n_period)
    per_elem = osz + (osz if out2 is not None else 0) + (4 if resid is not None else 0) + (esz if dact_u is not None else 0)
    if kk is None and kn is None:
        return 2.0 * M * N * K, float((M * K + N * K) * esz + M * N * per_elem)
    nb = len(kk if kk is not None else kn)
    kk = kk if kk is not None else torch.full((nb,), float(K), dtype=torch.float64)
    kn = kn if kn is not None else torch.full((nb,), float(N), dtype=torch.float64)
    flops = float((2.0 * rows_in * kk * kn).sum())
    if skip_writes and keep_n is not None and rows_in > 0 and resid is None and osz == 2:
        out_bytes = float((rows_in * _written_cols(keep_n, N, n_period)).sum()) * per_elem
    else:
        out_bytes = float(M) * N * per_elem
    return flops, float((rows_in * kk).sum() * esz + float(kk.max()) * float(kn.max()) * esz + out_bytes)


def zero_ranges(buf, ranges):
    """buf[lo:hi] = 0 for every (lo, hi) of `ranges` (fp32 buffer; one launch per 24 ranges) -- vr_zero_ranges."""
    ranges = [(int(lo), int(hi)) for lo, hi in ranges if hi > lo]
    for i in range(0, len(ranges), _lib.MAX_ZERO_RANGES):
        chunk = ranges[i:i + _lib.MAX_ZERO_RANGES]
        zr = _lib.ZeroRanges()
        zr.n = len(chunk)
        for j, (lo, hi) in enumerate(chunk):
            zr.lo[j], zr.count[j] = lo, hi - lo
        _lib.check(_lib.lib().vr_zero_ranges(_p(buf), ctypes.byref(zr), _stream()), "vr_zero_ranges")
    return buf


def zero_(t):
    """t[...] = 0 for a contiguous fp32 / bf16 tensor (vr_zero_ranges on its storage viewed as fp32 words)."""
    if t.numel() == 0:
        return t
    nbytes = t.numel() * t.element_size()
    if not t.is_contiguous() or nbytes % 4 or t.data_ptr() % 4:
        raise ValueError("zero_: contiguous, 4-byte sized and aligned tensors only")
    zr = _lib.ZeroRanges()
    zr.n, zr.lo[0], zr.count[0] = 1, 0, nbytes // 4
    _lib.check(_lib.lib().vr_zero_ranges(_p(t), ctypes.byref(zr), _stream()), "vr_zero_ranges")
    return t


def relayout(src, dst, A, B, C, dst_ld=None, src_ld=None):
    """dst[a, c, b] = src[a, b, c] (flat: dst[a * dst_ld + c * B + b] = src[a * src_ld + b * C + c]); fp32 / bf16 either side --
    vr_relayout.  Pad columns (dst_ld > B * C) keep their contents."""
    dst_ld = B * C if dst_ld is None else dst_ld
    src_ld = B * C if src_ld is None else src_ld
    _lib.check(_lib.lib().vr_relayout(_p(src), _p(dst), A, B, C, src_ld, dst_ld, _dt(src), _dt(dst), _stream()), "vr_relayout")
    return dst


def copy_i32_from_pinned(host, dst):
    """dst (device int32) = host (pinned int32) by a KERNEL on the current stream -- vr_relayout reading the mapped host pages -- instead
    of a hipMemcpyAsync: the copy engine's hand-off to and from the compute queue cost ~25 us of idle GPU per training step."""
    assert host.is_pinned() and host.dtype == torch.int32 and dst.dtype == torch.int32 and dst.is_cuda and host.numel() == dst.numel()
    n = host.numel()
    _lib.check(_lib.lib().vr_relayout(host.data_ptr(), dst.data_ptr(), 1, 1, n, n, n, VR_F32, VR_F32, _stream()), "vr_relayout")
    return dst


def cast_bf16(src, dst):
    _lib.check(_lib.lib().vr_cast_f32_bf16(_p(src), _p(dst), src.numel(), _stream()), "vr_cast_f32_bf16")
    return dst


def tr_descs(entries, device):
    """entries: [(src_off, dst_off, rows, cols, ld_dst)] -> (device descriptor array for cast_transpose_batch, max tiles)."""
    import numpy as np
    arr = np.zeros(len(entries), dtype=np.dtype([("src_off", "<i8"), ("dst_off", "<i8"), ("rows", "<i4"), ("cols", "<i4"),
                                                 ("ld_dst", "<i4"), ("reserved", "<i4")]))
    for i, (so, do, r, c, ld) in enumerate(entries):
        arr[i] = (so, do, r, c, ld, 0)
    max_tiles = max(((r + 63) // 64) * ((c + 63) // 64) for _, _, r, c, _ in entries)
    return torch.from_numpy(arr.view(np.uint8).copy()).to(device), max_tiles, list(entries)


def cast_transpose_batch(src, dst, descs):
    """dst (bf16, pre-zeroed pads) <- transposed bf16 copies of the fp32 matrices described by tr_descs()."""
    dev, max_tiles, entries = descs
    _lib.check(_lib.lib().vr_cast_transpose_batch(_p(src), _p(dst), _p(dev), len(entries), max_tiles, _stream()),
               "vr_cast_transpose_batch")
    return dst


def ln_fwd(x, w, b, keep, rows_per_sample, eps, out_dtype):
    M, C = x.numel() // x.shape[-1], x.shape[-1]
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().vr_ln_fwd(_p(x), _p(w), _p(b), _p(y), _p(mean), _p(rstd), _p(keep), M, C, rows_per_sample,
                                    eps, _dtcode(out_dtype), _stream()), "vr_ln_fwd")
    return y, mean, rstd


def ln_bwd(dy, x, w, mean, rstd, keep, rows_per_sample, dx_in, dw, db, next_cast=None, copies=1):
    """next_cast = (scale or None, keep or None): also return scale_mask_cast(dx, scale, keep) in dy's dtype (the gradient
    entering the next backward branch), produced in the same pass.
    copies > 1: dw / db are [copies, C] rows of partial sums that ln_grad_reduce folds into the parameter gradients."""
    M, C = x.numel() // x.shape[-1], x.shape[-1]
    dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    gt = torch.empty(x.shape, dtype=dy.dtype, device=x.device) if next_cast is not None else None
    sc, kp = next_cast if next_cast is not None else (None, None)
    _lib.check(_lib.lib().vr_ln_bwd(_p(dy), _p(x), _p(w), _p(mean), _p(rstd), _p(keep), _p(dx_in), _p(dx), _p(dw), _p(db),
                                    _p(gt), _p(sc), _p(kp), M, C, rows_per_sample, _dt(dy), copies, _stream()), "vr_ln_bwd")
    return dx if next_cast is None else (dx, gt)


def ln_grad_reduce(slots, copies):
    """slots: (part_w [copies, C], part_b [copies, C], dw [C], db [C]) per LayerNorm -- dw += part_w.sum(0), db likewise, and
    the partial rows are zero again (vr_ln_grad_reduce)."""
    if not slots:
        return
    arr = (_lib.LnGradSlot * len(slots))()
    for i, (pw, pb, dw, db) in enumerate(slots):
        C = dw.numel()
        assert pw.numel() == copies * C and pb.numel() == copies * C and db.numel() == C
        assert pw.is_contiguous() and pb.is_contiguous() and dw.is_contiguous() and db.is_contiguous()
        arr[i].part_w, arr[i].part_b, arr[i].dw, arr[i].db, arr[i].C = _p(pw), _p(pb), _p(dw), _p(db), C
    _lib.check(_lib.lib().vr_ln_grad_reduce(arr, len(slots), copies, _stream()), "vr_ln_grad_reduce")


def _attn_profiled(call, keep_hd, B, N, H, D, passes):
    """bench.py's `roofline.blocks_mfma_util`: kept FLOPs (4 N^2 per kept head channel and pass pair) and HIP events of one launch."""
    kept = float(keep_hd.clamp(min=0).sum().item()) if keep_hd is not None else float(B * H * D)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    call()
    e1.record()
    PROFILE_ATTN.append((passes * 2.0 * N * N * kept, e0, e1))


def attn_fwd(qkv, keep_hd, B, N, H, D, scale):
    o = torch.empty((B, N, H * D), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
    call = lambda: _lib.check(_lib.lib().vr_attn_fwd(_p(qkv), _p(o), _p(lse), _p(keep_hd), B, N, H, D, scale, _dt(qkv), _stream()),
                              "vr_attn_fwd")
    if PROFILE_ATTN is None:
        call()
    else:
        _attn_profiled(call, keep_hd, B, N, H, D, 2)                # Q K^T and P V
    return o, lse


def attn_bwd(qkv, o, d_o, lse, keep_hd, B, N, H, D, scale):
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
    call = lambda: _lib.check(_lib.lib().vr_attn_bwd(_p(qkv), _p(o), _p(d_o), _p(lse), _p(delta), _p(dqkv), _p(keep_hd), B, N, H, D,
                                                     scale, _dt(qkv), _stream()), "vr_attn_bwd")
    if PROFILE_ATTN is None:
        call()
    else:
        _attn_profiled(call, keep_hd, B, N, H, D, 5)                # S recomputed, dP, dV, dQ, dK
    return dqkv


def softce(logits, target, gscale, want_grad=True):
    K = logits.shape[-1]
    R = logits.numel() // K
    loss_rows = torch.empty(R, dtype=torch.float32, device=logits.device)
    dlogits = torch.empty_like(logits) if want_grad else None
    _lib.check(_lib.lib().vr_softce(_p(logits), _p(target), _p(loss_rows), _p(dlogits), R, K, gscale, _stream()),
               "vr_softce")
    return loss_rows, dlogits


def softce_train(logits, target, sample_map, rows_per_sample, loss_acc, grad_dtype):
    """dlogits (grad_dtype, row pitch = classes rounded up to 8, pad zeroed) of mean soft-target CE; loss_acc += the mean."""
    K = logits.shape[-1]
    R = logits.numel() // K
    ld = (K + 7) // 8 * 8
    d = torch.empty((R, ld), dtype=grad_dtype, device=logits.device)
    _lib.check(_lib.lib().vr_softce_train(_p(logits), _p(target), _p(sample_map), rows_per_sample, _p(loss_acc), _p(d), ld,
                                          _dtcode(grad_dtype), R, K, 1.0 / R, 1.0 / R, _stream()), "vr_softce_train")
    return d


def colsum(x, out, M, N, ld, row_map=None):
    _lib.check(_lib.lib().vr_colsum(_p(x), _p(out), M, N, ld, _dt(x), _rm(row_map), _stream()), "vr_colsum")
    return out


def scale_mask_cast(x, scale, keep, rows_per_sample, out_dtype):
    C = x.shape[-1]
    M = x.numel() // C
    out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    _lib.check(_lib.lib().vr_scale_mask_cast(_p(x), _p(out), _p(scale), _p(keep), M, C, rows_per_sample,
                                             _dtcode(out_dtype), _stream()), "vr_scale_mask_cast")
    return out


def token_mean(y, first):
    """[B, N, C] -> [B, C]: mean over tokens first.. (patch_output_type='avg')."""
    B, N, C = y.shape
    out = torch.empty((B, C), dtype=y.dtype, device=y.device)
    _lib.check(_lib.lib().vr_token_mean(_p(y), _p(out), B, N, C, first, _dt(y), _stream()), "vr_token_mean")
    return out


def token_mean_bwd(dmean, dy, first):
    B, N, C = dy.shape
    _lib.check(_lib.lib().vr_token_mean_bwd(_p(dmean), _p(dy), B, N, C, first, _dt(dy), _stream()), "vr_token_mean_bwd")
    return dy


def batchsum(x, out):
    B = x.shape[0]
    inner = x.numel() // B
    _lib.check(_lib.lib().vr_batchsum(_p(x), _p(out), B, inner, _stream()), "vr_batchsum")
    return out


def im2col_patch(img, P, ldk, out_dtype, sample_map=None, out=None):
    """sample_map (int64 [B], device): output sample b is image sample_map[b]; out: a preallocated col matrix."""
    B, Cin, H, W = img.shape
    col = out if out is not None else torch.empty((B * (H // P) * (W // P), ldk), dtype=out_dtype, device=img.device)
    _lib.check(_lib.lib().vr_im2col_patch_map(_p(img), _p(col), _p(sample_map), B, Cin, H, W, P, ldk, _dtcode(out_dtype),
                                              _stream()), "vr_im2col_patch_map")
    return col


def embed_cls(tokens, pos, x, keep, num_tokens=1):
    """token rows 0..num_tokens-1 of x [B, N, C] <- tokens + pos (masked)."""
    B, N, C = x.shape
    _lib.check(_lib.lib().vr_embed_cls(_p(tokens), _p(pos), _p(x), _p(keep), B, N, C, num_tokens, _stream()), "vr_embed_cls")
    return x


def sr_im2col(y, B, g, C, num_tokens=1):
    col = torch.empty((B * (g // 2) ** 2, 9 * C), dtype=y.dtype, device=y.device)
    _lib.check(_lib.lib().vr_sr_im2col(_p(y), _p(col), B, g, C, num_tokens, _dt(y), _stream()), "vr_sr_im2col")
    return col


def sr_col2im(dcol, dy, B, g, C, num_tokens=1):
    _lib.check(_lib.lib().vr_sr_col2im(_p(dcol), _p(dy), B, g, C, num_tokens, _dt(dcol), _stream()), "vr_sr_col2im")
    return dy


def sr_resid(x, B, g, cin, cout, num_tokens=1):
    out = torch.empty((B, num_tokens + (g // 2) ** 2, cout), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().vr_sr_resid(_p(x), _p(out), B, g, cin, cout, num_tokens, _stream()), "vr_sr_resid")
    return out


def sr_resid_bwd(dout, B, g, cin, cout, num_tokens=1):
    dx = torch.empty((B, num_tokens + g * g, cin), dtype=torch.float32, device=dout.device)
    _lib.check(_lib.lib().vr_sr_resid_bwd(_p(dout), _p(dx), B, g, cin, cout, 0, num_tokens, _stream()), "vr_sr_resid_bwd")
    return dx


def mask_rows(x, keep, rows_per_sample):
    C = x.shape[-1]
    M = x.numel() // C
    _lib.check(_lib.lib().vr_mask_rows(_p(x), _p(keep), M, C, rows_per_sample, _stream()), "vr_mask_rows")
    return x


# ---- convolutional patch embedding (stem.hip) ---------------------------------------------------------------
def im2col3x3_image(img, stride, ld, out_dtype):
    B, C, H, W = img.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    col = torch.empty((B * Ho * Wo, ld), dtype=out_dtype, device=img.device)
    _lib.check(_lib.lib().vr_im2col3x3(_p(img), _p(col), B, H, W, C, stride, 1, ld, _dtcode(out_dtype), _stream()),
               "vr_im2col3x3")
    return col


def im2col3x3(a, B, H, W, C):
    col = torch.empty((B * H * W, 9 * C), dtype=a.dtype, device=a.device)
    _lib.check(_lib.lib().vr_im2col3x3(_p(a), _p(col), B, H, W, C, 1, 0, 9 * C, _dt(a), _stream()), "vr_im2col3x3")
    return col


def conv3x3_supported(a, Cin, Cout):
    return a.dtype == torch.bfloat16 and Cin in (16, 24, 32) and Cout <= 32 and Cout % 4 == 0


def conv3x3(a, w, B, H, W, Cin, Cout, out_dtype):
    """Direct 3x3 / stride 1 / pad 1 convolution: a bf16 NHWC [B*H*W, Cin], w bf16 [Cout, 9*Cin] (kh, kw, ci) -> [B*H*W, Cout]."""
    out = torch.empty((B * H * W, Cout), dtype=out_dtype, device=a.device)
    _lib.check(_lib.lib().vr_conv3x3(_p(a), _p(w), _p(out), B, H, W, Cin, Cout, _dtcode(out_dtype), _stream()), "vr_conv3x3")
    return out


def conv3x3_res(a, w, res, B, H, W, Cin, Cout, out_dtype):
    """conv3x3(a, w) + res (bf16 [B*H*W, Cout]) in one pass -- vr_conv3x3_res."""
    out = torch.empty((B * H * W, Cout), dtype=out_dtype, device=a.device)
    _lib.check(_lib.lib().vr_conv3x3_res(_p(a), _p(w), _p(res), _p(out), B, H, W, Cin, Cout, _dtcode(out_dtype), _stream()),
               "vr_conv3x3_res")
    return out


def conv3x3_res_patch(a, w, res_col, B, H, W, Cin, Cout, res_patch, out_dtype):
    """conv3x3_res with `res` in patch order ([B*(H/p)*(W/p), p*p*Cout], patch_unfold's layout) -- vr_conv3x3_res_patch."""
    out = torch.empty((B * H * W, Cout), dtype=out_dtype, device=a.device)
    _lib.check(_lib.lib().vr_conv3x3_res_patch(_p(a), _p(w), _p(res_col), _p(out), B, H, W, Cin, Cout, _dtcode(out_dtype), res_patch,
                                               _stream()), "vr_conv3x3_res_patch")
    return out


def conv_w_flip(w, out_dtype):
    """Weights of the data-gradient convolution of a 3x3 / stride 1 / pad 1 Conv2d: [Ci, (kh, kw, co)] = w[co, ci, 2-kh, 2-kw]."""
    Co, Ci = w.shape[0], w.shape[1]
    out = torch.empty((Ci, 9 * Co), dtype=out_dtype, device=w.device)
    _lib.check(_lib.lib().vr_conv_w_flip(_p(w), _p(out), Co, Ci, _dtcode(out_dtype), _stream()), "vr_conv_w_flip")
    return out


def bn_finalize(sq, n, bn, momentum, update_running):
    """(scale, shift, mean, rstd) fp32 [C] of a train-mode BatchNorm2d from its (sum, sum of squares) [2, C]; updates the module's
    running statistics and batch counter in place (vr_bn_finalize)."""
    C = sq.shape[1]
    out = torch.empty((4, C), dtype=torch.float32, device=sq.device)
    rm, rv, nbt = (bn.running_mean, bn.running_var, bn.num_batches_tracked) if update_running else (None, None, None)
    _lib.check(_lib.lib().vr_bn_finalize(_p(sq[0]), _p(sq[1]), int(n), _p(bn.weight.detach()), _p(bn.bias.detach()), float(bn.eps),
                                         float(momentum), _p(rm), _p(rv), _p(nbt), _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]),
                                         C, _stream()), "vr_bn_finalize")
    return out[0], out[1], out[2], out[3]


def conv1_direct_supported(img, w, Cout):
    return img.dtype == torch.float32 and img.shape[1] == 3 and w.dtype == torch.bfloat16 and w.shape[1] == 32 and \
        Cout <= 32 and Cout % 4 == 0


def conv1_direct(img, w, bias, relu, out_dtype):
    """3x3 / stride 2 / pad 1 convolution of the fp32 NCHW image (3 channels): w bf16 [Cout, 32] (kh, kw, c; zero padded)
    -> NHWC [B*Ho*Wo, Cout], optionally relu(. + bias) (vr_conv1_direct)."""
    B, _, H, W = img.shape
    Cout = w.shape[0]
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((B * Ho * Wo, Cout), dtype=out_dtype, device=img.device)
    _lib.check(_lib.lib().vr_conv1_direct(_p(img), _p(w), _p(bias), _p(out), B, H, W, Cout, int(bool(relu)), _dtcode(out_dtype),
                                          _stream()), "vr_conv1_direct")
    return out


def conv3x3_bias_relu(a, w, bias, res, B, H, W, Cin, Cout, out_dtype):
    """relu(conv3x3(a, w) + bias[co]) (+ res): the evaluation form with BatchNorm folded into w / bias (vr_conv3x3_bias_relu)."""
    out = torch.empty((B * H * W, Cout), dtype=out_dtype, device=a.device)
    _lib.check(_lib.lib().vr_conv3x3_bias_relu(_p(a), _p(w), _p(bias), _p(res), _p(out), B, H, W, Cin, Cout, _dtcode(out_dtype),
                                               _stream()), "vr_conv3x3_bias_relu")
    return out


def conv3x3_bias_relu_patch(a, w, bias, res, B, H, W, Cin, Cout, patch, out_dtype):
    """conv3x3_bias_relu with the result laid out as patch_unfold would: [B*(H/patch)*(W/patch), patch*patch*Cout]."""
    out = torch.empty((B * (H // patch) * (W // patch), patch * patch * Cout), dtype=out_dtype, device=a.device)
    _lib.check(_lib.lib().vr_conv3x3_bias_relu_patch(_p(a), _p(w), _p(bias), _p(res), _p(out), B, H, W, Cin, Cout, patch,
                                                     _dtcode(out_dtype), _stream()), "vr_conv3x3_bias_relu_patch")
    return out


def conv3x3_wgrad_supported(a, Cin, Cout):
    return a.dtype == torch.bfloat16 and Cin == Cout and Cin in (16, 24, 32)


def conv3x3_wgrad(a, dz, dw, B, H, W, Cin, Cout):
    """dw fp32 [Cout, 9*Cin] (kh, kw, ci) += weight gradient of the 3x3 / stride 1 / pad 1 convolution (a, dz bf16 NHWC)."""
    _lib.check(_lib.lib().vr_conv3x3_wgrad(_p(a), _p(dz), _p(dw), B, H, W, Cin, Cout, _stream()), "vr_conv3x3_wgrad")
    return dw


def col2im3x3(dcol, B, H, W, C):
    d = torch.empty((B * H * W, C), dtype=dcol.dtype, device=dcol.device)
    _lib.check(_lib.lib().vr_col2im3x3(_p(dcol), _p(d), B, H, W, C, _dt(dcol), _stream()), "vr_col2im3x3")
    return d


def bn_stats(z, s, q):
    R, C = z.shape
    _lib.check(_lib.lib().vr_bn_stats(_p(z), _p(s), _p(q), R, C, _dt(z), _stream()), "vr_bn_stats")


def bn_relu(z, scale, shift, res, out_dtype):
    R, C = z.shape
    out = torch.empty((R, C), dtype=out_dtype, device=z.device)
    _lib.check(_lib.lib().vr_bn_relu(_p(z), _p(scale), _p(shift), _p(res), _p(out), R, C, _dtcode(out_dtype), _dt(z), _stream()),
               "vr_bn_relu")
    return out


def bn_bwd(da, z, scale, shift, mean, rstd, sg, sgz, training):
    R, C = z.shape
    dz = torch.empty((R, C), dtype=da.dtype, device=z.device)
    _lib.check(_lib.lib().vr_bn_bwd(_p(da), _p(z), _p(scale), _p(shift), _p(mean), _p(rstd), _p(sg), _p(sgz), _p(dz), R, C,
                                    int(training), _dt(da), _dt(z), _stream()), "vr_bn_bwd")
    return dz


def bn_relu_patch(z, scale, shift, res, B, H, W, patch, out_dtype):
    """bn_relu with the result written as the patchify operand [B*(H/patch)*(W/patch), patch*patch*C] (patch_unfold's layout)."""
    C = z.shape[-1]
    col = torch.empty((B * (H // patch) * (W // patch), patch * patch * C), dtype=out_dtype, device=z.device)
    _lib.check(_lib.lib().vr_bn_relu_patch(_p(z), _p(scale), _p(shift), _p(res), _p(col), B, H, W, patch, C, _dtcode(out_dtype),
                                           _dt(z), _stream()), "vr_bn_relu_patch")
    return col


def bn_bwd_patch(dcol, z, scale, shift, mean, rstd, sg, sgz, training, B, H, W, patch):
    """bn_bwd whose incoming gradient is in patch order (the projection's data gradient as its GEMM leaves it); dz is NHWC."""
    R, C = z.shape
    dz = torch.empty((R, C), dtype=dcol.dtype, device=z.device)
    _lib.check(_lib.lib().vr_bn_bwd_patch(_p(dcol), _p(z), _p(scale), _p(shift), _p(mean), _p(rstd), _p(sg), _p(sgz), _p(dz), B, H, W,
                                          patch, C, int(training), _dt(dcol), _dt(z), _stream()), "vr_bn_bwd_patch")
    return dz


def patch_unfold(a, B, gh, gw, P, C):
    col = torch.empty((B * gh * gw, P * P * C), dtype=a.dtype, device=a.device)
    _lib.check(_lib.lib().vr_patch_unfold(_p(a), _p(col), B, gh, gw, P, C, 0, _dt(a), _stream()), "vr_patch_unfold")
    return col


def patch_fold(col, B, gh, gw, P, C):
    a = torch.empty((B * gh * P * gw * P, C), dtype=col.dtype, device=col.device)
    _lib.check(_lib.lib().vr_patch_unfold(_p(a), _p(col), B, gh, gw, P, C, 1, _dt(col), _stream()), "vr_patch_unfold")
    return a
