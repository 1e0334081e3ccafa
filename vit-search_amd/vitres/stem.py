"""Convolutional patch embedding (network_def embed types 4 and 5) on the HIP path.

Reference: nets/patch_conv.py:23-73 -- conv1 (3x3/s2, 3->m) + BN + ReLU, conv2/conv3 (3x3, m->m) + BN + ReLU,
residual from conv1's output, conv_proj (7x7/s7, m->C) -> tokens; then tokens/pos_embed/ChannelDrop as in
nets/vit_sr_supernet.py:398-407.  Activations are NHWC [B*112*112, m] in the compute dtype, pre-BN conv outputs
fp32; every convolution is a gather (vr_im2col3x3 / vr_patch_unfold) + vr_gemm.  BatchNorm uses batch statistics
in training (and updates the running buffers exactly like nn.BatchNorm2d: momentum 0.1, unbiased running_var).
"""
import torch

from . import functional as Fn
from . import kernels as K


def _perm_weight(conv, dt, ld=None):
    """[out, in, kh, kw] -> [out, (kh, kw, in)] zero-padded to ld columns, compute dtype."""
    w = conv.weight.detach()
    o, k = w.shape[0], w.shape[1] * w.shape[2] * w.shape[3]
    ld = ld or k
    out = torch.empty((o, ld), dtype=dt, device=w.device)
    if ld != k:
        K.zero_(out)
    return K.relayout(w, out, o, w.shape[1], w.shape[2] * w.shape[3], ld)


def embed_params(model, need_bwd=True):
    pe, dt = model.patch_embed, model.compute_dtype
    m = pe.mid_chans
    if m % 8:
        raise NotImplementedError('conv patch embedding needs mid_chans % 8 == 0 on the HIP path (got %d)' % m)
    wp = _perm_weight(pe.conv_proj, dt)

    def flipped(conv):
        """Weights of the data-gradient convolution: Wt[ci, (kh, kw, co)] = W[co, ci, 2-kh, 2-kw] (3x3, stride 1, pad 1)."""
        w = conv.weight.detach()
        if w.dtype == torch.float32 and w.is_contiguous():
            return K.conv_w_flip(w, dt)                                # one launch (was flip + permute + cast + copy)
        return w.flip(2, 3).permute(1, 2, 3, 0).reshape(w.shape[1], 9 * w.shape[0]).to(dt).contiguous()
    return {"w1": _perm_weight(pe.conv1.conv, dt, ld=32), "w2": _perm_weight(pe.conv2.conv, dt),
            "w3": _perm_weight(pe.conv3.conv, dt),
            "w2t": flipped(pe.conv2.conv) if need_bwd else None, "w3t": flipped(pe.conv3.conv) if need_bwd else None,
            "proj": Fn.Weights(pe.conv_proj.weight, pe.conv_proj.bias.detach(), wp, wp.shape[1]),
            "pos": model.pos_embed.detach(), "tokens": model.tokens.detach()}


def _bn_affine(z, bn, training):
    """Folded BatchNorm: returns (scale, shift, mean, rstd) fp32 [C]; updates running stats in training."""
    C = z.shape[1]
    if training:
        sq = K.zero_(torch.empty((2, C), dtype=torch.float32, device=z.device))
        K.bn_stats(z, sq[0], sq[1])
        n = z.shape[0]
        if (bn.momentum is not None and bn.track_running_stats and bn.running_mean is not None and bn.running_mean.dtype == torch.float32
                and bn.weight.dtype == torch.float32):
            with torch.no_grad():
                return K.bn_finalize(sq, n, bn, bn.momentum, True)      # one launch (was a dozen elementwise ones per BatchNorm)
        mean = sq[0] / n
        var = (sq[1] / n - mean * mean).clamp_min_(0.)
        with torch.no_grad():
            mom = bn.momentum if bn.momentum is not None else 0.1
            bn.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
            bn.running_var.mul_(1 - mom).add_(var * (n / max(n - 1, 1)), alpha=mom)
            bn.num_batches_tracked += 1
    else:
        mean, var = bn.running_mean.detach().float(), bn.running_var.detach().float()
    rstd = torch.rsqrt(var + bn.eps)
    scale = bn.weight.detach() * rstd
    shift = bn.bias.detach() - mean * scale
    return scale.contiguous(), shift.contiguous(), mean.contiguous(), rstd.contiguous()


# Evaluation (frozen weights, running statistics): BatchNorm folds into the convolution in front of it -- the weights are scaled
# per output channel by gamma / sqrt(var + eps), the shift becomes a bias, ReLU (and conv3's residual) move into the kernel's
# epilogue: three bn_relu passes and the fp32 pre-BatchNorm tensors disappear (15 % of the candidate-scoring step was the stem).
FOLD_BN = __import__("os").environ.get("VITRES_STEM_FOLD_BN", "1") != "0"
# conv1 straight from the NCHW image (vr_conv1_direct) instead of im2col + GEMM; in training the im2col matrix is only built
# in the backward, beside the data-gradient chain, for conv1's weight gradient
DIRECT_CONV1 = __import__("os").environ.get("VITRES_STEM_DIRECT_CONV1", "1") != "0"
# Opt-in (bf16 mode): store the pre-BatchNorm convolution outputs z in bf16 instead of fp32 -- they are read by four passes each
# (statistics, normalise + ReLU, two in the backward; all sums stay fp32).  Measured: ref_tiny step 6.75 -> 6.58 ms, sr_tiny_mh
# 9.89 -> 9.76 ms; but the worst parameter-gradient error of the 56-px conv-stem test nets against the fp32 reference grows
# from 0.07 to 0.09 (bf16 has 3 mantissa bits fewer than the fp16 autocast gives the reference's convolutions), so the default
# keeps fp32.
Z_BF16 = __import__("os").environ.get("VITRES_STEM_Z_BF16", "0") != "0"
# VITRES_STEM_PATCH_DIRECT=0: keep the unfold / fold passes around the 7 x 7 / stride-7 projection in training (measurement)
PATCH_DIRECT = __import__("os").environ.get("VITRES_STEM_PATCH_DIRECT", "1") != "0"


def drop_fold(model):
    """Forget the BatchNorm-folded evaluation weights (anything that may have changed a stem parameter or running statistic calls
    this: a training-mode forward of the stem, a replay of a captured training step, an optimizer step, invalidate_shadow)."""
    model._stem_fold = None


def _folded_params(model):
    """bf16 [(w1', t1), (w2', t2), (w3', t3)] with BatchNorm folded in.  Cached between evaluation forwards; the cache is dropped
    by drop_fold() -- Tensor._version alone is NOT a valid key: FlatAdamW updates the parameters through raw pointers and a
    replayed hipGraph updates the running statistics without touching either version counter (both are still compared, for
    in-place edits through torch)."""
    pe = model.patch_embed
    convs = (pe.conv1, pe.conv2, pe.conv3)
    ver = tuple(t._version for c in convs for t in (c.conv.weight, c.bn.weight, c.bn.bias, c.bn.running_mean, c.bn.running_var))
    ptr = tuple(c.conv.weight.data_ptr() for c in convs)
    cache = getattr(model, "_stem_fold", None)
    if cache is not None and cache[0] == (ver, ptr):
        return cache[1]
    out = []
    for i, c in enumerate(convs):
        bn = c.bn
        scale = bn.weight.detach().float() * torch.rsqrt(bn.running_var.detach().float() + bn.eps)
        shift = (bn.bias.detach().float() - bn.running_mean.detach().float() * scale).contiguous()
        w = c.conv.weight.detach().float()
        o, k = w.shape[0], w.shape[1] * w.shape[2] * w.shape[3]
        wp = w.permute(0, 2, 3, 1).reshape(o, k) * scale[:, None]
        ld = 32 if i == 0 else k
        wf = torch.zeros((o, ld), dtype=torch.bfloat16, device=w.device)
        wf[:, :k] = wp
        out.append((wf, shift))
    model._stem_fold = ((ver, ptr), out)
    return out


def _embed_conv_eval(model, x, p, cfg, keep):
    pe, dt = model.patch_embed, cfg["dtype"]
    B, _, H, W = x.shape
    m, C, P = pe.mid_chans, cfg["dim"], cfg["patches"]
    Hm, Wm = H // 2, W // 2
    R = B * Hm * Wm
    g = Hm // (model.patch_size // 2)
    T = cfg.get("tokens", 1)
    N = P + T
    (w1, t1), (w2, t2), (w3, t3) = _folded_params(model)
    if DIRECT_CONV1 and K.conv1_direct_supported(x, w1, m):
        a1 = K.conv1_direct(x, w1, t1, True, dt)
    else:
        col1 = K.im2col3x3_image(x, 2, 32, dt)
        a1 = torch.empty((R, m), dtype=dt, device=x.device)
        K.gemm(col1, w1, a1, M=R, N=m, K=32, lda=32, ldb=32, ldc=m, bias=t1, act=3)
    a2 = K.conv3x3_bias_relu(a1, w2, t2, None, B, Hm, Wm, m, m, dt)
    ps = model.patch_size // 2
    if Hm % ps == 0 and Wm % ps == 0:          # the last ReLU writes the projection's patchify operand itself
        colp = K.conv3x3_bias_relu_patch(a2, w3, t3, a1, B, Hm, Wm, m, m, ps, dt)
    else:
        a3 = K.conv3x3_bias_relu(a2, w3, t3, a1, B, Hm, Wm, m, m, dt)
        colp = K.patch_unfold(a3, B, g, g, ps, m)
    ldk = ps * ps * m
    out = torch.empty((B, N, C), dtype=torch.float32, device=x.device)
    K.gemm(colp, p["proj"].w_c, out, M=B * P, N=C, K=ldk, lda=ldk, ldb=ldk, ldc=C, bias=p["proj"].b,
           pos=p["pos"][0, T:], keep_n=keep, rows_in=P, c_map=(P, N, T))
    K.embed_cls(p["tokens"], p["pos"], out, keep, T)
    return out, None


def embed_conv_fwd(model, x, p, cfg, keep, save):
    pe, dt = model.patch_embed, cfg["dtype"]
    if FOLD_BN and not save and not model.training and dt == torch.bfloat16 and K.conv3x3_supported(x.new_empty(0, dtype=dt), pe.mid_chans, pe.mid_chans):
        return _embed_conv_eval(model, x, p, cfg, keep)
    B, _, H, W = x.shape
    m, C, P = pe.mid_chans, cfg["dim"], cfg["patches"]
    Hm, Wm = H // 2, W // 2
    R = B * Hm * Wm
    g = Hm // (model.patch_size // 2)
    T = cfg.get("tokens", 1)
    N = P + T
    tr = model.training
    if tr:
        drop_fold(model)                   # this forward moves the running statistics

    zdt = dt if (Z_BF16 and dt == torch.bfloat16) else torch.float32

    def conv(col, w, ld):
        z = torch.empty((R, m), dtype=zdt, device=x.device)
        K.gemm(col, w, z, M=R, N=m, K=ld, lda=ld, ldb=ld, ldc=m)
        return z
    if DIRECT_CONV1 and K.conv1_direct_supported(x, p["w1"], m):
        col1 = None                                            # (built in the backward from the saved image)
        z1 = K.conv1_direct(x, p["w1"], None, False, zdt)
    else:
        col1 = K.im2col3x3_image(x, 2, 32, dt)
        z1 = conv(col1, p["w1"], 32)
    bn1 = _bn_affine(z1, pe.conv1.bn, tr)
    a1 = K.bn_relu(z1, bn1[0], bn1[1], None, dt)
    # conv2 / conv3: direct MFMA convolution when the shape is covered (bf16, m in {16,24,32}); the im2col matrix (9x the
    # activation) is then only built in backward, for the weight gradient
    direct = K.conv3x3_supported(a1, m, m)
    if direct:
        col2, z2 = a1, K.conv3x3(a1, p["w2"], B, Hm, Wm, m, m, zdt)
    else:
        col2 = K.im2col3x3(a1, B, Hm, Wm, m)
        z2 = conv(col2, p["w2"], 9 * m)
    bn2 = _bn_affine(z2, pe.conv2.bn, tr)
    a2 = K.bn_relu(z2, bn2[0], bn2[1], None, dt)
    if direct:
        col3, z3 = a2, K.conv3x3(a2, p["w3"], B, Hm, Wm, m, m, zdt)
    else:
        col3 = K.im2col3x3(a2, B, Hm, Wm, m)
        z3 = conv(col3, p["w3"], 9 * m)
    bn3 = _bn_affine(z3, pe.conv3.bn, tr)
    ps = model.patch_size // 2
    # PATCH_DIRECT (round 4): the last BatchNorm + ReLU (+ skip) writes the projection's patchify operand itself and the backward
    # reads the projection's data gradient where its GEMM leaves it (vr_bn_relu_patch / vr_bn_bwd_patch / vr_conv3x3_res_patch):
    # no patch_unfold / patch_fold pass (410 MB each way at B = 128).  Fast path only (direct convolutions, bf16).
    pdirect = PATCH_DIRECT and direct and dt == torch.bfloat16 and Hm == g * ps and Wm == g * ps
    if pdirect:
        colp = K.bn_relu_patch(z3, bn3[0], bn3[1], a1, B, Hm, Wm, ps, dt)
    else:
        a3 = K.bn_relu(z3, bn3[0], bn3[1], a1, dt)
        colp = K.patch_unfold(a3, B, g, g, ps, m)
    ldk = ps * ps * m
    out = torch.empty((B, N, C), dtype=torch.float32, device=x.device)
    K.gemm(colp, p["proj"].w_c, out, M=B * P, N=C, K=ldk, lda=ldk, ldb=ldk, ldc=C, bias=p["proj"].b,
           pos=p["pos"][0, T:], keep_n=keep, rows_in=P, c_map=(P, N, T))
    K.embed_cls(p["tokens"], p["pos"], out, keep, T)
    saved = (col1 if col1 is not None else ("image", x), z1, bn1, col2, z2, bn2, col3, z3, bn3, colp,
             (B, Hm, Wm, m, g, ps, tr, direct, pdirect)) if save else None
    return out, saved


def embed_conv_bwd(model, gx, saved, p, cfg, keep, gv, gt=None):
    pe, dt = model.patch_embed, cfg["dtype"]
    col1, z1, bn1, col2, z2, bn2, col3, z3, bn3, colp, (B, Hm, Wm, m, g, ps, tr, direct, pdirect) = saved
    dev = gx.device
    _, N, C = gx.shape
    T = cfg.get("tokens", 1)
    P = N - T
    R = B * Hm * Wm
    ldk = ps * ps * m
    if gt is None:
        gt = K.scale_mask_cast(gx, None, keep, N, dt)
    # conv_proj
    ov = Fn._overlap(gx) and Fn.STEM_SIDE
    wtmp = K.zero_(torch.empty((C, ldk), dtype=torch.float32, device=dev))

    def wgrad_proj():
        Fn.linear_wgrad(gt, colp, wtmp, B * P, C, ldk, C, ldk, a_map=(P, N, T), db=gv(pe.conv_proj.bias),
                        sched=1 if ov else 0)
        K.relayout(wtmp, gv(pe.conv_proj.weight), C, ps * ps, m)          # [C, (kh, kw), m] -> [C, m, kh, kw]
        K.batchsum(gx, gv(model.pos_embed))
    # weight gradients run beside the data-gradient chain (functional.on_side); joined at the end of this function
    Fn.on_side(wgrad_proj, gt, wtmp) if ov else wgrad_proj()
    dcolp = torch.empty((B * P, ldk), dtype=dt, device=dev)
    K.gemm(gt, p["proj"].w_c, dcolp, M=B * P, N=ldk, K=C, lda=C, ldb=ldk, ldc=ldk, b_trans=True, a_map=(P, N, T))
    da3 = dcolp if pdirect else K.patch_fold(dcolp, B, g, g, ps, m)     # d(relu(bn3) + a1); pdirect: still in patch order

    acc_in_place = True      # (every backward starts from a zeroed gradient arena: vit_sr_supernet._run_backward / _zero_grad_arena)

    def conv_bwd(da, z, bn, col, w, conv_mod, ld, need_dx, wt=None, res=None, da_patch=False, res_patch=False):
        gbw, gbb = gv(conv_mod.bn.weight), gv(conv_mod.bn.bias)
        if da_patch:                            # (pdirect implies acc_in_place's fast path)
            dz = K.bn_bwd_patch(da, z, bn[0], bn[1], bn[2], bn[3], gbb, gbw, tr, B, Hm, Wm, ps)
        elif acc_in_place:                      # the gradient arena is zero here: vr_bn_bwd's sums land where they belong
            dz = K.bn_bwd(da, z, bn[0], bn[1], bn[2], bn[3], gbb, gbw, tr)
        else:
            sg = K.zero_(torch.empty((2, m), dtype=torch.float32, device=dev))
            dz = K.bn_bwd(da, z, bn[0], bn[1], bn[2], bn[3], sg[0], sg[1], tr)
            gbw.copy_(sg[1])
            gbb.copy_(sg[0])
        wg = K.zero_(torch.empty((m, ld), dtype=torch.float32, device=dev))
        cin = conv_mod.conv.weight.shape[1]
        direct_w = wt is not None and K.conv3x3_wgrad_supported(dz, cin, m)
        if wt is not None and not direct_w:     # forward ran the direct convolution and saved the activation, not its im2col
            col = K.im2col3x3(col, B, Hm, Wm, m)

        def wgrad():
            if direct_w:                        # `col` is the saved NHWC activation: direct weight-gradient kernel
                K.conv3x3_wgrad(col, dz, wg, B, Hm, Wm, cin, m)
            else:
                c_ = col
                if isinstance(c_, tuple):       # conv1 ran straight from the image: its im2col matrix is built here
                    c_ = K.im2col3x3_image(c_[1], 2, ld, dt)
                Fn.linear_wgrad(dz, c_, wg, R, m, ld, m, ld, sched=1 if ov else 0)
            K.relayout(wg, gv(conv_mod.conv.weight), m, 9, cin, src_ld=ld)     # [m, (kh, kw), ci] -> [m, ci, kh, kw]
        Fn.on_side(wgrad, dz, wg) if ov else wgrad()
        if not need_dx:
            return None
        if wt is not None:                      # data gradient = direct convolution of dz with the flipped weights
            if res is not None:                 # (+ the gradient arriving over the skip connection, in the same pass)
                if res_patch:
                    return K.conv3x3_res_patch(dz, wt, res, B, Hm, Wm, m, m, ps, dt)
                return K.conv3x3_res(dz, wt, res, B, Hm, Wm, m, m, dt)
            return K.conv3x3(dz, wt, B, Hm, Wm, m, m, dt)
        dcol = torch.empty((R, ld), dtype=dt, device=dev)
        K.gemm(dz, w, dcol, M=R, N=ld, K=m, lda=m, ldb=ld, ldc=ld, b_trans=True)
        return K.col2im3x3(dcol, B, Hm, Wm, m)
    da2 = conv_bwd(da3, z3, bn3, col3, p["w3"], pe.conv3, 9 * m, True, p["w3t"] if direct else None, da_patch=pdirect)
    fuse_res = direct and da3.dtype == torch.bfloat16
    da1 = conv_bwd(da2, z2, bn2, col2, p["w2"], pe.conv2, 9 * m, True, p["w2t"] if direct else None, res=da3 if fuse_res else None,
                   res_patch=pdirect)
    if not fuse_res:
        da1 = da1 + da3                                             # residual branch (patch_conv.py:69)
    conv_bwd(da1, z1, bn1, col1, p["w1"], pe.conv1, 32, False)
    if ov:
        Fn.join_side()
