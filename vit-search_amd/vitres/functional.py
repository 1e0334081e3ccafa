"""Forward / backward of the ViT-Res layers as explicit sequences of HIP kernel calls.

Every `*_fwd` returns (output, saved) and every `*_bwd` consumes `saved` and the incoming gradient,
accumulates parameter gradients into the caller's gradient views and returns the gradient of its input.
The residual stream (x, and its gradient g) is fp32 [B, N, C]; branch activations are in the compute
dtype (bf16 fast mode / fp32 parity mode).  Prefix masks are int32 keep-count vectors on the device.

Reference op sequences replaced (file:line in /root/reference):
  attn_branch : nets/supernet_blocks.py:214-245 (norm1 -> Attention.forward :100-120 -> drop_path -> masks -> +x)
  mlp_branch  : nets/supernet_blocks.py:247-253 (norm2 -> Mlp.forward :37-52 -> drop_path -> mask -> +x)
  sr_block    : nets/vit_sr_supernet.py:114-172
  patch_embed : timm PatchEmbed / nets/patch_conv.py:63-72 + vit_sr_supernet.py:398-407
  head        : nets/vit_sr_supernet.py:420-428,440-449
"""
import os as _os

import torch

from . import kernels as K

# ---- two-stream execution of independent GEMMs ----------------------------------------------------------------
# The weight-gradient GEMM of a Linear and the data-gradient GEMM that feeds the next backward op are independent.
# At ViT-Res shapes each alone leaves most of the chip idle in its prologue / epilogue / tail phases, so they are
# issued on two streams (parallel branches once the step is captured into a hipGraph) with `sched=1` launches
# (one workgroup per tile) so the hardware interleaves both kernels' workgroups on every CU.
OVERLAP = _os.environ.get('VITRES_OVERLAP', '3') != '0'
DEFER_JOIN = _os.environ.get('VITRES_OVERLAP', '3') == '2'
JOIN_PER_BLOCK = _os.environ.get('VITRES_OVERLAP', '3') == '3'
FUSE_CAST = _os.environ.get('VITRES_FUSE_CAST', '1') != '0'      # LayerNorm backward also emits the next branch's gradient
# vr_gemm_ln (gemm_nt_ln.hip): bit 0 = LayerNorm forward behind the Linear that produces its input, bit 1 = LayerNorm backward
# behind the data-gradient GEMM that produces its gradient; whole-row tiles, so only widths <= VITRES_FUSE_LN_MAXN.  Round 2
# (kernel rewritten as GEMM K loop + the LayerNorm kernels' row loop glued through LDS) measured inside the sr_tiny step:
# backward fusion at width 256: 51 us against 28 + 36..50 us for the two kernels (8.34 -> 8.20 ms); forward fusion: proj (K <= 256)
# 28 against 19.5 + 13.6 us, fc2 (K = 768: twelve single-buffered 64 x 256 slices) 45..49 against 32 + 13.6 us (a wash per kernel;
# the step still prefers it, 8.12 against 8.17 ms, one launch less per block; VITRES_FUSE_LN_FWD_MAXK bounds the K it applies to);
# width 512 (32 x 512 tiles, 260 of them at M = 8320) loses in both directions.
FUSE_LN = int(_os.environ.get('VITRES_FUSE_LN', '3'))
FUSE_LN_MAXN = int(_os.environ.get('VITRES_FUSE_LN_MAXN', '256'))
FUSE_LN_FWD_MAXK = int(_os.environ.get('VITRES_FUSE_LN_FWD_MAXK', '4096'))
# LayerNorm weight / bias gradients: the workgroups of a LayerNorm backward add their column sums into LN_COPIES rows of
# partial sums (grads["<weight key>.part"], [2, LN_COPIES, C], kept zero between backwards) instead of all of them into
# the same 2C addresses; flush_ln_grads() folds the rows of every LayerNorm of a backward part in one launch.  vr_ln_bwd at
# (128 x 257, 256): 33 -> 23 us.  0 / 1 = accumulate straight into the parameter gradients.
LN_COPIES = int(_os.environ.get('VITRES_LN_COPIES', '64'))
_DBG_SKIP_WGRAD = _os.environ.get('VITRES_DBG_SKIP_WGRAD', '0') != '0'
_DBG_WGRAD_SCHED = int(_os.environ.get('VITRES_DBG_WGRAD_SCHED', '0'), 0)      # dev aid: OR-ed into the weight gradients' sched (64: 4-wave group kernel)
_ln_pending = []


def _ln_dst(grads, wk, bk):
    """(dw, db, copies) for the LayerNorm backward kernels."""
    part = grads.get(wk + ".part") if LN_COPIES > 1 else None
    if part is None:
        return grads[wk], grads[bk], 1
    _ln_pending.append((part[0], part[1], grads[wk], grads[bk]))
    return part[0], part[1], part.shape[1]


def flush_ln_grads():
    """Fold the partial rows written since the last flush into the parameter gradients (current stream)."""
    if _ln_pending:
        slots = list(_ln_pending)
        _ln_pending.clear()
        K.ln_grad_reduce(slots, slots[0][0].shape[0])


_pending_owner = [None]          # weakref of the model whose backward filled _ln_pending / _block_wgrads last


def claim_pending(model):
    """Called at the start of a model's backward: the module-global pending lists are its own from here on."""
    import weakref
    _pending_owner[0] = weakref.ref(model)


def reset_ln_grads(model=None):
    """Drop slots left behind by a backward that raised; returns True if there were any (their rows need re-zeroing).  The
    weight-gradient calls collected for a block that was never flushed go with them: launched by the next backward they
    would add a dead step's gradients and keep its tensors alive.  The lists are shared by every model of the process: when they
    belong to ANOTHER model that is between the parts of a split backward (engine.GraphedTrainStep(split_for_sync), resume_backward),
    they are that model's live state and are left alone."""
    owner = _pending_owner[0]() if _pending_owner[0] is not None else None
    if model is not None and owner is not None and owner is not model and getattr(owner, "_bwd_state", None) is not None:
        return False
    dirty = bool(_ln_pending)
    _ln_pending.clear()
    del _block_wgrads[:]
    return dirty


STEM_SIDE = _os.environ.get('VITRES_STEM_SIDE', '1') != '0'      # conv-stem weight gradients on the side stream (+2 % since SIDE_DEFER)
_side_streams = {}


_pending = []            # tensors the side stream may still be reading (current generation)
_lagged = []             # [(event recorded on the side stream, tensors)]: generations whose join was postponed
JOIN_LAG = int(_os.environ.get('VITRES_JOIN_LAG', '2'))   # branches a join may trail behind (0: join at once)


N_SIDE = max(1, int(_os.environ.get('VITRES_SIDE_STREAMS', '1')))   # side streams used round-robin
_rr = [0]


def _sides(dev):
    side = _side_streams.get(dev)
    if side is None:
        side = _side_streams[dev] = [torch.cuda.Stream(device=dev) for _ in range(N_SIDE)]
    return side


# VITRES_SIDE_DEFER=1: the side launch is ENQUEUED after the next kernel(s) of the main chain (its dependency -- an event recorded
# now -- is unchanged).  In a captured graph the fork node then lists the main chain's successor first; whether the hipGraph
# executor keeps the first successor on the parent's hardware queue decides if the main chain pays a cross-queue handoff
# (12-17 us measured) at every fork.
SIDE_DEFER = _os.environ.get('VITRES_SIDE_DEFER', '1') != '0'
_deferred = []           # [(event on the main stream, side stream, fn)]


def _run_side(side, fn):
    # (launches on a side stream overlap the main chain's: they get their own stream-K workspace -- kernels.ws_role)
    sides = _side_streams.get(side.device, ())
    role = 1 + (sides.index(side) if side in sides else 0)
    with torch.cuda.stream(side), K.ws_role(role):
        fn()


def _wait_other_sides(side):
    # (N_SIDE > 1 only) work that consumes what EVERY side stream produced -- the in-graph AdamW range update reads gradients whose
    # weight-gradient groups went round-robin over the side streams -- is ordered behind all of them, not just its own stream
    for other in _side_streams.get(side.device, ()):
        if other is not side:
            side.wait_stream(other)


def flush_side():
    while _deferred:
        ev, side, fn, after_all = _deferred.pop(0)
        side.wait_event(ev)
        if after_all:
            _wait_other_sides(side)
        _run_side(side, fn)


def on_side(fn, *keepalive, defer=True, after_all_sides=False):
    """Run fn() on a side stream after everything queued so far on the current stream; the tensors it reads are kept
    alive (so the caching allocator cannot hand them out again) until join_side().  defer=False: launch at once (callers whose
    fn() must have RUN on the host when on_side returns, or with no main kernel following before the join).
    after_all_sides: fn() also waits for everything already issued on the OTHER side streams (VITRES_SIDE_STREAMS > 1)."""
    main = torch.cuda.current_stream()
    sides = _sides(main.device)
    side = sides[_rr[0] % len(sides)]
    _rr[0] += 1
    if SIDE_DEFER and defer:
        flush_side()                       # the previous one: at least one main kernel has been enqueued since
        _deferred.append((main.record_event(), side, fn, after_all_sides))
    else:
        flush_side()
        side.wait_stream(main)
        if after_all_sides:
            _wait_other_sides(side)
        _run_side(side, fn)
    _pending.extend(keepalive)


# Auxiliary streams (round 4): named branches besides the weight-gradient side stream -- "tail": the patch-embedding weight
# gradient and the small reductions that close a backward run beside the first block's weight-gradient group instead of behind
# one another on the main queue.  (Measured and dropped: AdamW over finished arena ranges on an "opt" stream OF ITS OWN beside the
# rest of the backward -- 7.52 -> 7.98 ms however the update was throttled.  What ships is the same update issued IN ORDER on the
# weight gradients' side stream: engine.GraphedTrainStep, VITRES_OPT_OVERLAP, 7.47 -> 7.37 ms.)
_aux_streams = {}
_aux_used = set()
_AUX_ROLE = {}           # stream-K workspace role of each auxiliary stream (kernels.ws_role)


def aux_stream(dev, name):
    st = _aux_streams.get((dev, name))
    if st is None:
        st = _aux_streams[(dev, name)] = torch.cuda.Stream(device=dev)
    return st


def on_aux(name, fn, *keepalive, after_side=False):
    """Run fn() on the auxiliary stream `name` after everything queued so far on the current stream (and, after_side, on the
    side streams: weight gradients already launched there).  Joined by join_side(); keepalive as on_side."""
    flush_side()
    main = torch.cuda.current_stream()
    st = aux_stream(main.device, name)
    st.wait_stream(main)
    if after_side:
        for side in _side_streams.get(main.device, ()):
            st.wait_stream(side)
    with torch.cuda.stream(st), K.ws_role(_AUX_ROLE.setdefault(name, 8 + len(_AUX_ROLE))):
        fn()
    _aux_used.add((main.device, name))
    _pending.extend(keepalive)


def join_side():
    """Current stream waits for ALL side-stream work issued so far (and releases every tensor kept for it)."""
    if _block_wgrads:
        flush_wgrads()
    flush_side()
    if _pending or _side_streams or _aux_used:         # (CPU runs never get here: nothing was ever put on a side stream)
        main = torch.cuda.current_stream()
        for side in _side_streams.get(main.device, ()):
            main.wait_stream(side)
        for key in [k for k in _aux_used if k[0] == main.device]:
            main.wait_stream(_aux_streams[key])
            _aux_used.discard(key)
    _pending.clear()
    _lagged.clear()


def join_side_lagged():
    """End of a backward branch: instead of waiting for THIS branch's weight gradients (a cross-queue dependency the main
    chain would sit idle behind: ~18 us of nothing running per join in the captured graph), record where the side stream is
    and wait for the point recorded JOIN_LAG branches ago -- long finished by now.  The tensors of the postponed generations
    stay alive meanwhile (one or two branches' worth, so buffers still recycle through the Infinity Cache)."""
    if JOIN_LAG <= 0:
        return join_side()
    flush_side()
    main = torch.cuda.current_stream()
    sides = _side_streams.get(main.device)
    if sides is None:
        _pending.clear()
        return
    _lagged.append(([sd.record_event() for sd in sides], list(_pending)))
    _pending.clear()
    while len(_lagged) > JOIN_LAG:
        evs, _keep = _lagged.pop(0)
        for ev in evs:
            main.wait_event(ev)


def _overlap(x):
    return OVERLAP and x.is_cuda


class Weights:
    """Per-forward view of one Linear-like parameter pair in the compute dtype."""
    __slots__ = ("w", "b", "w_c", "ld", "w_t", "ld_t")

    def __init__(self, w, b, w_c, ld, w_t=None, ld_t=0):
        self.w, self.b, self.w_c, self.ld = w, b, w_c, ld   # fp32 param, fp32 bias, compute-dtype matrix [out, ld]
        self.w_t, self.ld_t = w_t, ld_t                     # transposed bf16 shadow [in, ld_t] (None: fp32 / absent)


def linear_dgrad(dy, W, dx, M, N_in, K_out, lddy, lddx, **kw):
    """dx[M, N_in] = epilogue(dy[M, K_out] @ W[K_out, N_in]).  With the transposed shadow both operands are
    K-contiguous (the LDS-DMA kernel); otherwise W is read contraction-major by the general kernel."""
    if W.w_t is not None and lddy % 8 == 0:
        K.gemm(dy, W.w_t, dx, M=M, N=N_in, K=K_out, lda=lddy, ldb=W.ld_t, ldc=lddx, **kw)
    else:
        K.gemm(dy, W.w_c, dx, M=M, N=N_in, K=K_out, lda=lddy, ldb=W.ld, ldc=lddx, b_trans=True, **kw)


def linear_wgrad(dy, x, dw, M, N_out, K_in, lddy, ldx, ldw=None, a_map=None, b_map=None, db=None, keep_rows=None,
                 keep_cols=None, row_period=0, tokens_per_sample=0, sched=0, collect=None, store=False, split=0, m_groups=None):
    """dw[N_out, K_in] += dy[M, N_out]^T @ x[M, K_in]  (fp32 atomics, split over tokens; db[N_out] += colsum(dy)).
    keep_rows / keep_cols: per-sample kept prefix of dy's / x's channels -> fully masked tiles are skipped.
    collect: a list -> the call is appended to it instead of being launched (K.gemm_group launches the list as one kernel).
    store: dw and db are OVERWRITTEN (vr_gemm atomic == 2: one workgroup per tile over all tokens, plain stores; measured round 3 inside
    the sr_tiny step: 7.85 -> 8.1 ms, so no caller asks for it).
    split: explicit token split (workgroups per output tile); 0 = the kernel's rule (32 slices of 64 tokens per workgroup)."""
    kw = dict(M=N_out, N=K_in, K=M, lda=lddy, ldb=ldx, ldc=ldw or K_in, a_trans=True, b_trans=True,
              atomic=(2 if store else True), split_k=(1 if store else split), a_map=a_map, b_map=b_map, bias_grad=db,
              keep_k=keep_rows, keep_n=keep_cols, k_period=row_period, rows_in=tokens_per_sample, sched=sched,
              m_groups=(K.M_GROUPS[0] if m_groups is None else m_groups))       # (now: the call may be launched after the backward has returned)
    if collect is not None:
        collect.append((dy, x, dw, kw))
    else:
        K.gemm(dy, x, dw, **kw)


# VITRES_WGRAD_GROUP=1: the four weight gradients of a transformer block (fc2, fc1, proj, qkv) are issued as ONE vr_gemm_group
# launch on the side stream once the last of their operands (dqkv) exists, instead of four launches of ~200 workgroups each.
WGRAD_GROUP = _os.environ.get('VITRES_WGRAD_GROUP', '1') != '0'
_block_wgrads = []       # collected (dy, x, dw, kwargs) of the block being walked backwards

def flush_wgrads():
    """Launch the collected weight gradients as one group on the side stream (no-op when nothing is pending)."""
    if not _block_wgrads:
        return
    calls = list(_block_wgrads)
    del _block_wgrads[:]
    if _DBG_SKIP_WGRAD:                                     # timing experiment only (wrong gradients): the main chain alone
        if _os.environ.get('VITRES_DBG_SKIP_WGRAD') == '2':  # ... with the fork / join edges kept (a one-element kernel on the side)
            t0 = calls[0][0]
            on_side(lambda: t0.view(-1)[:1].add_(0), t0)
        return
    if not OVERLAP:                                         # single-stream runs (profiling passes): same kernel, in line
        return K.gemm_group(calls)
    on_side(lambda: K.gemm_group(calls), *[t_ for c_ in calls for t_ in c_[:2]])


# --------------------------------------------------------------------------------------------------
# transformer block halves
# --------------------------------------------------------------------------------------------------
def _ln_fusable(a, C, next_ln):
    return next_ln is not None and (FUSE_LN & 1) and C <= FUSE_LN_MAXN and a.shape[-1] <= FUSE_LN_FWD_MAXK and \
        K.gemm_ln_supported(a, C, C)


def attn_branch_fwd(x, p, cfg, embed_keep, attn_keep, out_keep, scale, save, pre=None, next_ln=None):
    """pre: (y, mean, rstd) of norm1(x) when the producer of x already computed it; next_ln = (w, b, keep, eps) of the
    LayerNorm that consumes this branch's output -> third result (y, mean, rstd) of it, or None when it was not fused."""
    B, N, C = x.shape
    M = B * N
    H, D = cfg["heads"], cfg["head_dim"]
    HD = H * D
    dt = cfg["dtype"]
    y, mean, rstd = pre if pre is not None else K.ln_fwd(x, p["n1w"], p["n1b"], embed_keep, N, cfg["eps"], dt)
    qkv = torch.empty((B, N, 3 * HD), dtype=dt, device=x.device)
    K.gemm(y, p["qkv"].w_c, qkv, M=M, N=3 * HD, K=C, lda=C, ldb=p["qkv"].ld, ldc=3 * HD, bias=p["qkv"].b, rows_in=N,
           keep_k=embed_keep, keep_n=attn_keep, n_period=HD)
    o, lse = K.attn_fwd(qkv, attn_keep, B, N, H, D, cfg["scale"])
    x1 = torch.empty_like(x)
    post = None
    if _ln_fusable(o, C, next_ln):
        post = K.gemm_ln_fwd(o, p["proj"].w_c, x1, next_ln[0], next_ln[1], next_ln[2], next_ln[3], M=M, N=C, K=HD, lda=HD,
                             ldb=p["proj"].ld, ldc=C, bias=p["proj"].b, scale=scale, keep_n=out_keep, resid=x, rows_in=N,
                             keep_k=attn_keep)
    elif next_ln is not None:
        # wider rows (stages 2 - 3): the LayerNorm folded into the tiled kernel's launch (vr_gemm_ln_fold), or None
        post = K.gemm_ln_fold_fwd(o, p["proj"].w_c, x1, next_ln[0], next_ln[1], next_ln[2], next_ln[3], M=M, N=C, K=HD, lda=HD,
                                  ldb=p["proj"].ld, ldc=C, bias=p["proj"].b, scale=scale, keep_n=out_keep, resid=x, rows_in=N,
                                  keep_k=attn_keep)
    if post is None and not _ln_fusable(o, C, next_ln):
        K.gemm(o, p["proj"].w_c, x1, M=M, N=C, K=HD, lda=HD, ldb=p["proj"].ld, ldc=C, bias=p["proj"].b, scale=scale,
               keep_n=out_keep, resid=x, rows_in=N, keep_k=attn_keep)
    saved = (x, mean, rstd, y, qkv, o, lse) if save else None
    return x1, saved, post


def attn_branch_bwd(g, saved, p, grads, cfg, embed_keep, attn_keep, out_keep, scale, gt=None, next_cast=None):
    """gt: scale_mask_cast(g, scale, out_keep) when the producer of g already made it (ln_bwd's fused second output);
    next_cast: (scale, keep) of the branch that will consume this function's result -> returns (g_out, gt_next)."""
    x, mean, rstd, y, qkv, o, lse = saved
    B, N, C = x.shape
    M = B * N
    H, D = cfg["heads"], cfg["head_dim"]
    HD = H * D
    dt = cfg["dtype"]
    ov = _overlap(g)
    sch = 1 if ov else 0
    wsch = sch | _DBG_WGRAD_SCHED
    if gt is None:
        gt = K.scale_mask_cast(g, scale, out_keep, N, dt)                   # d(branch output), compute dtype

    grp = _block_wgrads if (WGRAD_GROUP and g.is_cuda) else None
    mg = K.M_GROUPS[0]                 # (now: the closures below may run after the backward has returned)

    def wgrad_proj():
        linear_wgrad(gt, o, grads["proj.w"], M, C, HD, C, HD, db=grads["proj.b"], keep_rows=out_keep, keep_cols=attn_keep,
                     tokens_per_sample=N, sched=wsch, collect=grp, m_groups=mg)
    if grp is not None or not ov:
        wgrad_proj()
    else:
        on_side(wgrad_proj, gt)
    d_o = torch.empty((B, N, HD), dtype=dt, device=x.device)
    linear_dgrad(gt, p["proj"], d_o, M, HD, C, C, HD, keep_n=attn_keep, rows_in=N, keep_k=out_keep, sched=sch)
    dqkv = K.attn_bwd(qkv, o, d_o, lse, attn_keep, B, N, H, D, cfg["scale"])

    def wgrad_qkv():
        linear_wgrad(dqkv, y, grads["qkv.w"], M, 3 * HD, C, 3 * HD, C, db=grads["qkv.b"], keep_rows=attn_keep,
                     keep_cols=embed_keep, row_period=HD, tokens_per_sample=N, sched=wsch, collect=grp, m_groups=mg)
    if grp is not None:
        wgrad_qkv()
        flush_wgrads()                                      # fc2, fc1 (MLP branch), proj, qkv: every operand exists now
    elif ov:
        on_side(wgrad_qkv, dqkv)
    else:
        wgrad_qkv()
    if (FUSE_LN & 2) and C <= FUSE_LN_MAXN and p["qkv"].w_t is not None and K.gemm_ln_supported(dqkv, C, C):
        dw, db, cp = _ln_dst(grads, "n1w", "n1b")
        out = K.gemm_ln_bwd(dqkv, p["qkv"].w_t, x, p["n1w"], mean, rstd, embed_keep, g, dw, db,
                            next_cast=next_cast, M=M, N=C, K=3 * HD, lda=3 * HD, ldb=p["qkv"].ld_t, rows_in=N,
                            keep_k=attn_keep, k_period=HD, copies=cp)
    else:
        dy = torch.empty((B, N, C), dtype=dt, device=x.device)
        linear_dgrad(dqkv, p["qkv"], dy, M, C, 3 * HD, 3 * HD, C, rows_in=N, keep_k=attn_keep, k_period=HD, keep_n=embed_keep,
                     sched=sch)
        dw, db, cp = _ln_dst(grads, "n1w", "n1b")
        out = K.ln_bwd(dy, x, p["n1w"], mean, rstd, embed_keep, N, g, dw, db, next_cast=next_cast, copies=cp)
    if ov and not DEFER_JOIN:
        join_side_lagged()
    return out


def mlp_branch_fwd(x, p, cfg, embed_keep, mlp_keep, out_keep, scale, save, pre=None, next_ln=None):
    B, N, C = x.shape
    M = B * N
    F = cfg["hidden"]
    dt = cfg["dtype"]
    y, mean, rstd = pre if pre is not None else K.ln_fwd(x, p["n2w"], p["n2b"], embed_keep, N, cfg["eps"], dt)
    h = torch.empty((B, N, F), dtype=dt, device=x.device)
    if save:
        # bf16: the tensor kept for the backward is gelu'(u), not u (act = 2): fc1's epilogue has the erf terms at hand, and fc2's
        # data gradient becomes a plain multiply (its erf / exp per element made that kernel VALU-bound); fp32 parity mode keeps u
        u = torch.empty((B, N, F), dtype=dt, device=x.device)
        K.gemm(y, p["fc1"].w_c, u, out2=h, M=M, N=F, K=C, lda=C, ldb=p["fc1"].ld, ldc=F, bias=p["fc1"].b,
               act=(2 if dt == torch.bfloat16 else 1), keep_n=mlp_keep, rows_in=N, keep_k=embed_keep)
    else:                                       # forward-only: the pre-activation is not kept, fc1 writes gelu(u) alone
        u = None
        K.gemm(y, p["fc1"].w_c, h, M=M, N=F, K=C, lda=C, ldb=p["fc1"].ld, ldc=F, bias=p["fc1"].b, act=1,
               keep_n=mlp_keep, rows_in=N, keep_k=embed_keep)
    x2 = torch.empty_like(x)
    post = None
    if _ln_fusable(h, C, next_ln):
        post = K.gemm_ln_fwd(h, p["fc2"].w_c, x2, next_ln[0], next_ln[1], next_ln[2], next_ln[3], M=M, N=C, K=F, lda=F,
                             ldb=p["fc2"].ld, ldc=C, bias=p["fc2"].b, scale=scale, keep_n=out_keep, resid=x, rows_in=N,
                             keep_k=mlp_keep, sched=K.reads_skipped())
    elif next_ln is not None:
        post = K.gemm_ln_fold_fwd(h, p["fc2"].w_c, x2, next_ln[0], next_ln[1], next_ln[2], next_ln[3], M=M, N=C, K=F, lda=F,
                                  ldb=p["fc2"].ld, ldc=C, bias=p["fc2"].b, scale=scale, keep_n=out_keep, resid=x, rows_in=N,
                                  keep_k=mlp_keep, sched=K.reads_skipped())
    if post is None and not _ln_fusable(h, C, next_ln):
        K.gemm(h, p["fc2"].w_c, x2, M=M, N=C, K=F, lda=F, ldb=p["fc2"].ld, ldc=C, bias=p["fc2"].b, scale=scale,
               keep_n=out_keep, resid=x, rows_in=N, keep_k=mlp_keep, sched=K.reads_skipped())
    saved = (x, mean, rstd, y, u, h) if save else None
    return x2, saved, post


def mlp_branch_bwd(g, saved, p, grads, cfg, embed_keep, mlp_keep, out_keep, scale, gt=None, next_cast=None):
    x, mean, rstd, y, u, h = saved
    B, N, C = x.shape
    M = B * N
    F = cfg["hidden"]
    dt = cfg["dtype"]
    ov = _overlap(g)
    sch = 1 if ov else 0
    wsch = sch | _DBG_WGRAD_SCHED
    if gt is None:
        gt = K.scale_mask_cast(g, scale, out_keep, N, dt)

    grp = _block_wgrads if (WGRAD_GROUP and g.is_cuda) else None
    rsk, mg = K.reads_skipped(), K.M_GROUPS[0]            # (now: the closures below may run after the backward has returned)

    def wgrad_fc2():
        linear_wgrad(gt, h, grads["fc2.w"], M, C, F, C, F, db=grads["fc2.b"], keep_rows=out_keep, keep_cols=mlp_keep,
                     tokens_per_sample=N, sched=wsch | rsk, collect=grp, m_groups=mg)
    if ov and grp is None:
        on_side(wgrad_fc2, gt)
    else:
        wgrad_fc2()
    du = torch.empty((B, N, F), dtype=dt, device=x.device)
    linear_dgrad(gt, p["fc2"], du, M, F, C, C, F, dact_u=u, ldu=F, keep_n=mlp_keep, rows_in=N, keep_k=out_keep, sched=sch,
                 act=(2 if dt == torch.bfloat16 else 0))

    def wgrad_fc1():
        linear_wgrad(du, y, grads["fc1.w"], M, F, C, F, C, db=grads["fc1.b"], keep_rows=mlp_keep, keep_cols=embed_keep,
                     tokens_per_sample=N, sched=wsch | rsk, collect=grp, m_groups=mg)
    if ov and grp is None:
        on_side(wgrad_fc1, du)
    else:
        wgrad_fc1()
    if (FUSE_LN & 2) and C <= FUSE_LN_MAXN and p["fc1"].w_t is not None and K.gemm_ln_supported(du, C, C):
        dw, db, cp = _ln_dst(grads, "n2w", "n2b")
        out = K.gemm_ln_bwd(du, p["fc1"].w_t, x, p["n2w"], mean, rstd, embed_keep, g, dw, db,
                            next_cast=next_cast, M=M, N=C, K=F, lda=F, ldb=p["fc1"].ld_t, rows_in=N, keep_k=mlp_keep, copies=cp,
                            sched=K.reads_skipped())
    else:
        dy = torch.empty((B, N, C), dtype=dt, device=x.device)
        linear_dgrad(du, p["fc1"], dy, M, C, F, F, C, rows_in=N, keep_k=mlp_keep, keep_n=embed_keep, sched=sch | K.reads_skipped())
        dw, db, cp = _ln_dst(grads, "n2w", "n2b")
        out = K.ln_bwd(dy, x, p["n2w"], mean, rstd, embed_keep, N, g, dw, db, next_cast=next_cast, copies=cp)
    if ov and not DEFER_JOIN and not JOIN_PER_BLOCK:
        join_side_lagged()
    return out


# --------------------------------------------------------------------------------------------------
# spatial-reduction block
# --------------------------------------------------------------------------------------------------
def sr_fwd(x, p, cfg, embed_keep, new_keep, save, pre=None):
    B, Ni, C = x.shape
    g = cfg["grid"]
    go = g // 2
    T = cfg.get("tokens", 1)                                                 # leading token rows: class (+ distillation) token
    No = T + go * go
    Co = cfg["cout"]
    dt = cfg["dtype"]
    y, mean, rstd = pre if pre is not None else K.ln_fwd(x, p["nw"], p["nb"], embed_keep, Ni, cfg["eps"], dt)
    out = K.sr_resid(x, B, g, C, Co, T)
    col = K.sr_im2col(y, B, g, C, T)
    K.gemm(col, p["reduce"].w_c, out, M=B * go * go, N=Co, K=9 * C, lda=9 * C, ldb=p["reduce"].ld, ldc=Co,
           bias=p["reduce"].b, pos=p["pos"], resid=out, rows_in=go * go, c_map=(go * go, No, T), keep_k=embed_keep,
           k_period=C)
    K.gemm(y, p["token"].w_c, out, M=B * T, N=Co, K=C, lda=C, ldb=p["token"].ld, ldc=Co, bias=p["token"].b, resid=out,
           rows_in=T, a_map=(T, Ni, 0), c_map=(T, No, 0))
    if new_keep is not None:
        K.mask_rows(out, new_keep, No)
    saved = (x, mean, rstd, y, col) if save else None
    return out, saved


def sr_bwd(gout, saved, p, grads, cfg, embed_keep, new_keep, gt=None, next_cast=None):
    flush_wgrads()
    x, mean, rstd, y, col = saved
    B, Ni, C = x.shape
    g = cfg["grid"]
    go = g // 2
    P = go * go
    T = cfg.get("tokens", 1)
    No = T + P
    Co = cfg["cout"]
    dt = cfg["dtype"]
    if gt is None:
        gt = K.scale_mask_cast(gout, None, new_keep, No, dt)                # [B, No, Co]
    # token_transform (the T token rows of every sample)
    def wgrads():
        linear_wgrad(gt, y, grads["token.w"], B * T, Co, C, Co, C, a_map=(T, No, 0), b_map=(T, Ni, 0), db=grads["token.b"])
        # patch_reduce (rows T..)
        K.batchsum(gout, grads["pos_sum"])                                  # [No, Co]; rows T.. are d pos_embed
        linear_wgrad(gt, col, grads["reduce.w"], B * P, Co, 9 * C, Co, 9 * C, a_map=(P, No, T), db=grads["reduce.b"],
                     sched=1 if _overlap(gout) else 0)
        if "finish" in grads:
            grads["finish"]()                                               # re-layout of the conv weight gradient
    if _overlap(gout):
        on_side(wgrads, gt, gout, grads["reduce.w"], grads["pos_sum"], y, col)     # kept alive until the (lagged) join
    else:
        wgrads()
    dcol = torch.empty((B * P, 9 * C), dtype=dt, device=x.device)
    linear_dgrad(gt, p["reduce"], dcol, B * P, 9 * C, Co, Co, 9 * C, a_map=(P, No, T), rows_in=P)
    dy = torch.empty((B, Ni, C), dtype=dt, device=x.device)
    K.sr_col2im(dcol, dy, B, g, C, T)
    linear_dgrad(gt, p["token"], dy, B * T, C, Co, Co, C, a_map=(T, No, 0), c_map=(T, Ni, 0), rows_in=T)
    gres = K.sr_resid_bwd(gout, B, g, C, Co, T)
    dw, db, cp = _ln_dst(grads, "nw", "nb")
    out = K.ln_bwd(dy, x, p["nw"], mean, rstd, embed_keep, Ni, gres, dw, db, next_cast=next_cast, copies=cp)
    if _overlap(gout):
        join_side_lagged()                 # (a full join here stalled the main chain ~200 us behind the conv weight gradient)
    return out


# --------------------------------------------------------------------------------------------------
# patch embedding (type 0: timm PatchEmbed == one patchify GEMM)
# --------------------------------------------------------------------------------------------------
def embed0_fwd(img, p, cfg, keep, save, sample_map=None, col=None):
    """sample_map: internal sample b is the caller's image sample_map[b] (arch-grouped execution order);
    col: the gathered patch matrix when the caller already made it (engine.GraphedTrainStep gathers outside the graph)."""
    B = img.shape[0]
    T = cfg.get("tokens", 1)
    P, C, N = cfg["patches"], cfg["dim"], cfg["patches"] + T
    dt = cfg["dtype"]
    ldk = p["proj"].ld
    if col is None:
        col = K.im2col_patch(img, cfg["patch"], ldk, dt, sample_map=sample_map)
    x = torch.empty((B, N, C), dtype=torch.float32, device=img.device)
    K.gemm(col, p["proj"].w_c, x, M=B * P, N=C, K=ldk, lda=ldk, ldb=ldk, ldc=C, bias=p["proj"].b, pos=p["pos"][0, T:],
           keep_n=keep, rows_in=P, c_map=(P, N, T))
    K.embed_cls(p["tokens"], p["pos"], x, keep, T)
    return x, ((col,) if save else None)


# Token split of the patch-embedding weight gradient (2 x 5 tiles of 128^2 over 32 768 tokens): the kernel's coarse rule gives it
# 160 workgroups walking 32 slices each (103 us, and it was the first kernel of a serial step tail); EMBED_WGRAD_SLICES slices
# per workgroup -> 4x the workgroups, each paying |tile| x 4 B of atomics (42 MB in all at 8).
EMBED_WGRAD_SLICES = int(_os.environ.get('VITRES_EMBED_WGRAD_SLICES', '8'))
# The end of a backward -- patch-embedding weight gradient, positional-embedding sums, LayerNorm partial-row folds -- runs on the
# auxiliary stream "tail" beside the first block's weight-gradient group (they feed nothing but the optimizer); in line on the main
# stream it was 0.15 ms of kernels one after another with nothing beside them (round 3).  On THREE branches (the sums on the main
# stream) the third lands on the weight gradients' hardware queue in front of the last two groups: +0.1 ms (round 4).
TAIL_AUX = True


def embed0_bwd(g, saved, p, grads, cfg, keep, gt=None, wgrad=True, pos=True):
    """wgrad / pos: which half to run (the step tail issues the projection's weight gradient on an auxiliary stream and the
    positional-embedding sums on the main one: nothing orders them)."""
    flush_wgrads()
    (col,) = saved
    B, N, C = g.shape
    T = cfg.get("tokens", 1)
    P = N - T
    dt = cfg["dtype"]
    ldk = p["proj"].ld
    if gt is None:
        gt = K.scale_mask_cast(g, None, keep, N, dt)
    split = 0
    if EMBED_WGRAD_SLICES > 0 and g.is_cuda and dt == torch.bfloat16:
        split = max(1, -(-(B * P) // (64 * EMBED_WGRAD_SLICES)))
    if wgrad:
        linear_wgrad(gt, col, grads["proj.w"], B * P, C, ldk, C, ldk, a_map=(P, N, T), db=grads["proj.b"], split=split)
    if pos:
        K.batchsum(g, grads["pos"])                                         # d pos_embed [N, C] (rows 0..T-1 also = d tokens)


# --------------------------------------------------------------------------------------------------
# final norm + heads
# --------------------------------------------------------------------------------------------------
def head_fwd(x, p, cfg, keep, with_patch, save, pre=None):
    B, N, C = x.shape
    dt = cfg["dtype"]
    nc = cfg["classes"]
    y, mean, rstd = pre if pre is not None else K.ln_fwd(x, p["nw"], p["nb"], keep, N, cfg["eps"], dt)
    T = cfg.get("tokens", 1)
    cls = torch.empty((B, nc), dtype=torch.float32, device=x.device)
    K.gemm(y, p["cls"].w_c, cls, M=B, N=nc, K=C, lda=C, ldb=p["cls"].ld, ldc=nc, bias=p["cls"].b, a_map=(1, N, 0))
    pat, ym = None, None
    if with_patch == 3:                 # two-token variants: second result = dst_head(distillation token) (:455-458)
        pat = torch.empty((B, nc), dtype=torch.float32, device=x.device)
        K.gemm(y, p["dst"].w_c, pat, M=B, N=nc, K=C, lda=C, ldb=p["dst"].ld, ldc=nc, bias=p["dst"].b, a_map=(1, N, 1))
    elif with_patch == 2:               # patch_output_type='avg': patch head on the mean patch token (:447-449)
        ym = K.token_mean(y, T)
        pat = torch.empty((B, nc), dtype=torch.float32, device=x.device)
        K.gemm(ym, p["patch"].w_c, pat, M=B, N=nc, K=C, lda=C, ldb=p["patch"].ld, ldc=nc, bias=p["patch"].b)
    elif with_patch:
        pat = torch.empty((B, N - T, nc), dtype=torch.float32, device=x.device)
        K.gemm(y, p["patch"].w_c, pat, M=B * (N - T), N=nc, K=C, lda=C, ldb=p["patch"].ld, ldc=nc, bias=p["patch"].b,
               a_map=(N - T, N, T))
    saved = (x, mean, rstd, y, ym) if save else None
    return cls, pat, saved


def head_bwd(dcls, dpat, saved, p, grads, cfg, keep, next_cast=None, ready=False):
    """ready: dcls / dpat are already [rows, classes rounded up to 8] in the compute dtype with zeroed pad columns
    (kernels.softce_train) -- no padding / cast glue."""
    x, mean, rstd, y, ym = saved
    B, N, C = x.shape
    dt = cfg["dtype"]
    nc = cfg["classes"]
    T_ = cfg.get("tokens", 1)
    # every token row is written by a head's data gradient when the class head and the per-patch head are both active (and one
    # class token): no zero fill then
    covered = dcls is not None and dpat is not None and "dst" not in p and ym is None and T_ == 1
    dy = torch.empty((B, N, C), dtype=dt, device=x.device)
    if not covered:
        K.zero_(dy)
    ldp = (nc + 7) // 8 * 8                                                 # logits-gradient rows zero-padded to 16 B

    def padded(d2):                                                         # tiny [rows, classes] tensors: torch glue
        if ready:
            return d2.view(-1, ldp)
        out = torch.zeros((d2.shape[0], ldp), dtype=dt, device=x.device)
        out[:, :nc] = d2
        return out
    ov = _overlap(x)
    side = []                                                               # weight gradients of the heads: beside the dgrad chain

    def wg(fn, *keep):
        if ov:
            side.append((fn, keep))
        else:
            fn()
    if dcls is not None:
        gc = padded(dcls if ready else dcls.reshape(B, nc))
        wg(lambda: linear_wgrad(gc, y, grads["cls.w"], B, nc, C, ldp, C, b_map=(1, N, 0), db=grads["cls.b"]), gc)
        linear_dgrad(gc, p["cls"], dy, B, C, nc, ldp, C, c_map=(1, N, 0))
    T = cfg.get("tokens", 1)
    if dpat is not None and "dst" in p:                                      # distillation head on token row 1
        gd = padded(dpat if ready else dpat.reshape(B, nc))
        wg(lambda: linear_wgrad(gd, y, grads["dst.w"], B, nc, C, ldp, C, b_map=(1, N, 1), db=grads["dst.b"]), gd)
        linear_dgrad(gd, p["dst"], dy, B, C, nc, ldp, C, c_map=(1, N, 1))
    elif dpat is not None and ym is not None:                                # 'avg'
        gp = padded(dpat if ready else dpat.reshape(B, nc))
        wg(lambda: linear_wgrad(gp, ym, grads["patch.w"], B, nc, C, ldp, C, db=grads["patch.b"]), gp, ym)
        dmean = torch.empty((B, C), dtype=dt, device=x.device)
        linear_dgrad(gp, p["patch"], dmean, B, C, nc, ldp, C)
        K.token_mean_bwd(dmean, dy, T)
    elif dpat is not None:
        R = B * (N - T)
        gp = padded(dpat if ready else dpat.reshape(R, nc))
        wg(lambda: linear_wgrad(gp, y, grads["patch.w"], R, nc, C, ldp, C, b_map=(N - T, N, T), db=grads["patch.b"]), gp)
        linear_dgrad(gp, p["patch"], dy, R, C, nc, ldp, C, c_map=(N - T, N, T))
    if side:
        on_side(lambda: [fn() for fn, _ in side], y, *[t_ for _, keep in side for t_ in keep])
    dw, db, cp = _ln_dst(grads, "nw", "nb")
    return K.ln_bwd(dy, x, p["nw"], mean, rstd, keep, N, None, dw, db, next_cast=next_cast, copies=cp)
