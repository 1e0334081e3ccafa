"""TEST-ONLY torch emulation of the C-ABI kernels (semantics of include/vitres_hip.h), used to validate
the host-side orchestration (vitres/functional.py, nets/vit_sr_supernet.py) on a machine without a GPU.

Never imported by the product.  `install(monkeypatch)` swaps the functions of vitres.kernels for these;
the GPU tests compare the real kernels against the same oracle instead.
"""
import math

import torch
import torch.nn.functional as F


def _rows(n, m):
    idx = torch.arange(n)
    if not m or m[0] == 0:
        return idx
    rpi, rps, off = m
    return (idx // rpi) * rps + off + idx % rpi


def _flat2d(t, ld):
    return t.reshape(-1)[: (t.numel() // ld) * ld].view(-1, ld)


def gemm(a, b, out, *, M, N, K, lda, ldb, ldc, a_trans=False, b_trans=False, out2=None, bias=None, pos=None,
         scale=None, keep_n=None, resid=None, dact_u=None, ldu=0, act=0, atomic=False, split_k=1, rows_in=0,
         a_map=None, b_map=None, c_map=None, bias_grad=None, keep_k=None, n_period=0, k_period=0, sched=0, ws=None, ring=0,
         k_shares=0, m_groups=None):
    # keep_k / k_period are pure work-skipping hints (the skipped operands are zero by contract): ignored here
    A2, B2 = _flat2d(a, lda), _flat2d(b, ldb)
    if not a_trans:
        Am = A2[_rows(M, a_map)][:, :K].float()
    else:
        Am = A2[_rows(K, a_map)][:, :M].float().t()
    if not b_trans:
        Bm = B2[_rows(N, None)][:, :K].float()
    else:
        Bm = B2[_rows(K, b_map if a_trans else None)][:, :N].float().t()
    v = Am @ Bm.t()
    if bias_grad is not None:
        if atomic == 2:                     # store form of the weight gradient: C and the bias gradient are overwritten
            bias_grad.copy_(Am.sum(dim=1))
        else:
            bias_grad += Am.sum(dim=1)
    if atomic and a_trans:                  # wgrad form: rows_in / keep_* describe the contraction tokens (hints only)
        rows_in, keep_n, scale = 0, None, None
    m_idx = torch.arange(M)
    sample = (m_idx // rows_in) if rows_in > 0 else torch.zeros(M, dtype=torch.long)
    mloc = (m_idx % rows_in) if rows_in > 0 else m_idx
    if bias is not None:
        v = v + bias.view(1, N)
    if pos is not None:
        v = v + _flat2d(pos, N)[mloc]
    keep = keep_n.long()[sample] if keep_n is not None else torch.full((M,), N)
    ncol = torch.arange(N) % n_period if n_period > 0 else torch.arange(N)
    nmask = ncol[None, :] < keep[:, None]
    orow = _rows(M, c_map)
    C2d = _flat2d(out, ldc)
    if act == 3:                                           # relu, single store
        C2d[orow, :N] = torch.where(nmask, torch.relu(v), torch.zeros_like(v)).to(out.dtype)
        return out
    if act in (1, 2) and dact_u is None:
        v = torch.where(nmask, v, torch.zeros_like(v))
        h = torch.where(nmask, F.gelu(v), torch.zeros_like(v))
        if out2 is None:                                   # forward-only: gelu(u) alone
            C2d[orow, :N] = h.to(out.dtype)
            return out
        if act == 2:                                       # C = gelu'(u) instead of u
            cdf = 0.5 * (1 + torch.erf(v / math.sqrt(2.0)))
            pdf = torch.exp(-0.5 * v * v) / math.sqrt(2 * math.pi)
            v = torch.where(nmask, cdf + v * pdf, torch.zeros_like(v))
        C2d[orow, :N] = v.to(out.dtype)
        _flat2d(out2, ldc)[orow, :N] = h.to(out.dtype)
        return out
    if dact_u is not None:
        u = _flat2d(dact_u, ldu)[orow, :N].float()
        if act == 2:                                       # dact_u is the derivative itself
            v = v * u
        else:
            cdf = 0.5 * (1 + torch.erf(u / math.sqrt(2.0)))
            pdf = torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi)
            v = v * (cdf + u * pdf)
    v = torch.where(nmask, v, torch.zeros_like(v))
    if scale is not None:
        v = v * scale[sample].view(M, 1)
    if atomic == 2:
        C2d[orow, :N] = v
        return out
    if atomic:
        C2d[orow, :N] += v
        return out
    if resid is not None:
        v = v + _flat2d(resid, ldc)[orow, :N]
    C2d[orow, :N] = v.to(out.dtype)
    return out


def cast_bf16(src, dst):
    dst.copy_(src)
    return dst


def cast_transpose_batch(src, dst, descs):
    _, _, entries = descs
    for so, do, r, c, ld in entries:
        w = src[so:so + r * c].view(r, c)
        dst[do:do + c * ld].view(c, ld)[:, :r] = w.t().to(dst.dtype)
    return dst


def _keep_mask(keep, M, C, rps):
    if keep is None:
        return torch.ones(M, C, dtype=torch.bool), torch.full((M,), C, dtype=torch.float32)
    k = keep.long()[torch.arange(M) // rps]
    return torch.arange(C)[None, :] < k[:, None], k.float()


def ln_fwd(x, w, b, keep, rows_per_sample, eps, out_dtype):
    C = x.shape[-1]
    X = x.reshape(-1, C).float()
    M = X.shape[0]
    mask, kc = _keep_mask(keep, M, C, rows_per_sample)
    Xm = X * mask
    mu = Xm.sum(1) / kc
    if keep is None:
        var = ((X - mu[:, None]) ** 2).mean(1)
    else:
        var = (Xm * Xm).sum(1) / kc - mu * mu
    rstd = 1.0 / torch.sqrt(var + eps)
    y = (w * ((Xm - mu[:, None]) * rstd[:, None]) + b) * mask
    return y.view(x.shape).to(out_dtype), mu, rstd


def ln_bwd(dy, x, w, mean, rstd, keep, rows_per_sample, dx_in, dw, db, next_cast=None, copies=1):
    C = x.shape[-1]
    X = x.reshape(-1, C).float()
    G = dy.reshape(-1, C).float()
    M = X.shape[0]
    mask, kc = _keep_mask(keep, M, C, rows_per_sample)
    G = G * mask
    z = (X - mean[:, None]) * rstd[:, None] * mask
    dz = G * w
    s1 = dz.sum(1) / kc
    s2 = (dz * z).sum(1) / kc
    dx = (dz - (s1[:, None] + z * s2[:, None])) * rstd[:, None]
    if dx_in is not None:
        dx = dx + dx_in.reshape(-1, C)
    dx = dx * mask
    # copies > 1: [copies, C] rows of partial sums (the kernels spread their workgroups over them) -- one row will do here
    (dw[0] if copies > 1 else dw).add_((G * z).sum(0))
    (db[0] if copies > 1 else db).add_(G.sum(0))
    dx = dx.view(x.shape)
    if next_cast is None:
        return dx
    return dx, scale_mask_cast(dx, next_cast[0], next_cast[1], rows_per_sample, dy.dtype)


def gemm_group(calls):
    for a, b, out, kw in calls:
        gemm(a, b, out, **kw)


def gemm_ln_supported(a, N, ldc):
    return a.dtype == torch.bfloat16 and N == ldc and N % 8 == 0 and N <= 512


def gemm_ln_fwd(a, b, out, ln_w, ln_b, ln_keep, eps, *, M, N, K, lda, ldb, ldc, bias=None, scale=None, keep_n=None,
                resid=None, rows_in=0, keep_k=None, k_period=0, sched=0):
    gemm(a, b, out, M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=ldc, bias=bias, scale=scale, keep_n=keep_n, resid=resid,
         rows_in=rows_in, keep_k=keep_k, k_period=k_period)
    return ln_fwd(out, ln_w, ln_b, ln_keep, rows_in or M, eps, torch.bfloat16)


def gemm_ln_bwd(du, wt, x, ln_w, mean, rstd, ln_keep, dx_in, dw, db, next_cast=None, *, M, N, K, lda, ldb, rows_in=0,
                keep_k=None, k_period=0, copies=1, sched=0):
    dy = torch.empty(x.shape, dtype=torch.float32)
    gemm(du.float(), wt.float(), dy, M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=N, rows_in=rows_in, keep_k=keep_k, k_period=k_period)
    out = ln_bwd(dy, x, ln_w, mean, rstd, ln_keep, rows_in or M, dx_in, dw, db, next_cast=next_cast, copies=copies)
    return out if next_cast is None else (out[0], out[1].to(torch.bfloat16))


def ln_grad_reduce(slots, copies):
    for pw, pb, dw, db in slots:
        assert pw.shape == (copies, dw.numel()) and pb.shape == (copies, db.numel())
        dw += pw.sum(0)
        db += pb.sum(0)
        pw.zero_()
        pb.zero_()


def attn_fwd(qkv, keep_hd, B, N, H, D, scale):
    q, k, v = qkv.float().view(B, N, 3, H, D).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-2, -1)) * scale
    lse = torch.logsumexp(s, dim=-1)
    o = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, N, H * D)
    if keep_hd is not None:
        hm = (torch.arange(H)[None, :] * D) < keep_hd.long()[:, None]
        o = o * hm.repeat_interleave(D, dim=1)[:, None, :]
        lse = lse * hm[:, :, None]
    return o.to(qkv.dtype), lse


def attn_bwd(qkv, o, d_o, lse, keep_hd, B, N, H, D, scale):
    qkv32 = qkv.float().detach().requires_grad_(True)
    with torch.enable_grad():
        out, _ = attn_fwd(qkv32, keep_hd, B, N, H, D, scale)
        g = d_o.float()
        (dq,) = torch.autograd.grad(out, qkv32, g)
    if keep_hd is not None:
        hm = ((torch.arange(H)[None, :] * D) < keep_hd.long()[:, None]).repeat_interleave(D, dim=1)
        dq = dq.view(B, N, 3, H * D) * hm[:, None, None, :]
    return dq.reshape(qkv.shape).to(qkv.dtype)


def softce(logits, target, gscale, want_grad=True):
    K = logits.shape[-1]
    x, t = logits.reshape(-1, K).float(), target.reshape(-1, K).float()
    lsm = F.log_softmax(x, dim=-1)
    loss = -(t * lsm).sum(-1)
    d = None
    if want_grad:
        d = (gscale * (lsm.exp() * t.sum(-1, keepdim=True) - t)).view(logits.shape)
    return loss, d


def softce_train(logits, target, sample_map, rows_per_sample, loss_acc, grad_dtype):
    K = logits.shape[-1]
    X = logits.reshape(-1, K).float()
    R = X.shape[0]
    s = torch.arange(R) // rows_per_sample
    src = (sample_map.long()[s] if sample_map is not None else s) * rows_per_sample + torch.arange(R) % rows_per_sample
    T = target.reshape(-1, K).float()[src]
    lsm = torch.log_softmax(X, -1)
    loss_acc += (-(T * lsm).sum(-1)).sum() / R
    ld = (K + 7) // 8 * 8
    d = torch.zeros(R, ld)
    d[:, :K] = (torch.softmax(X, -1) * T.sum(-1, keepdim=True) - T) / R
    return d.to(grad_dtype)


def colsum(x, out, M, N, ld, row_map=None):
    out += _flat2d(x, ld)[_rows(M, row_map)][:, :N].float().sum(0)
    return out


def scale_mask_cast(x, scale, keep, rows_per_sample, out_dtype):
    C = x.shape[-1]
    X = x.reshape(-1, C).float()
    M = X.shape[0]
    mask, _ = _keep_mask(keep, M, C, rows_per_sample)
    if scale is not None:
        X = X * scale[torch.arange(M) // rows_per_sample].view(M, 1)
    return (X * mask).view(x.shape).to(out_dtype)


def token_mean(y, first):
    return y[:, first:].float().mean(1).to(y.dtype)


def token_mean_bwd(dmean, dy, first):
    dy[:, first:] = (dmean.float() / (dy.shape[1] - first)).to(dy.dtype)[:, None, :]
    return dy


def conv3x3_supported(a, Cin, Cout):
    return a.dtype == torch.bfloat16 and Cin in (16, 24, 32) and Cout <= 32 and Cout % 4 == 0


def conv3x3(a, w, B, H, W, Cin, Cout, out_dtype):
    x = a.float().view(B, H, W, Cin).permute(0, 3, 1, 2)
    wt = w.float().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    y = torch.nn.functional.conv2d(x, wt, padding=1)
    return y.permute(0, 2, 3, 1).reshape(B * H * W, Cout).to(out_dtype)


def conv1_direct_supported(img, w, Cout):
    return img.dtype == torch.float32 and img.shape[1] == 3 and w.dtype == torch.bfloat16 and w.shape[1] == 32


def conv1_direct(img, w, bias, relu, out_dtype):
    B, _, H, W = img.shape
    Cout = w.shape[0]
    wt = w.float()[:, :27].view(Cout, 3, 3, 3).permute(0, 3, 1, 2)          # (kh, kw, c) -> [co, c, kh, kw]
    y = torch.nn.functional.conv2d(img.to(torch.bfloat16).float(), wt, stride=2, padding=1)
    y = y.permute(0, 2, 3, 1).reshape(-1, Cout)
    if bias is not None:
        y = y + bias[None, :]
    if relu:
        y = torch.relu(y)
    return y.to(out_dtype)


def conv3x3_bias_relu(a, w, bias, res, B, H, W, Cin, Cout, out_dtype):
    y = torch.relu(conv3x3(a, w, B, H, W, Cin, Cout, torch.float32) + bias[None, :])
    if res is not None:
        y = y + res.float()
    return y.to(out_dtype)


def conv3x3_bias_relu_patch(a, w, bias, res, B, H, W, Cin, Cout, patch, out_dtype):
    y = conv3x3_bias_relu(a, w, bias, res, B, H, W, Cin, Cout, out_dtype)
    return patch_unfold(y, B, H // patch, W // patch, patch, Cout)


def conv3x3_res(a, w, res, B, H, W, Cin, Cout, out_dtype):
    return (conv3x3(a, w, B, H, W, Cin, Cout, torch.float32) + res.float()).to(out_dtype)


def conv_w_flip(w, out_dtype):
    return w.flip(2, 3).permute(1, 2, 3, 0).reshape(w.shape[1], 9 * w.shape[0]).to(out_dtype).contiguous()


def bn_finalize(sq, n, bn, momentum, update_running):
    mean = sq[0] / n
    var = (sq[1] / n - mean * mean).clamp_min(0.)
    if update_running:
        bn.running_mean.mul_(1 - momentum).add_(mean, alpha=momentum)
        bn.running_var.mul_(1 - momentum).add_(var * (n / max(n - 1, 1)), alpha=momentum)
        bn.num_batches_tracked += 1
    rstd = torch.rsqrt(var + bn.eps)
    scale = bn.weight.detach() * rstd
    return scale, bn.bias.detach() - mean * scale, mean, rstd


def conv3x3_wgrad_supported(a, Cin, Cout):
    return a.dtype == torch.bfloat16 and Cin == Cout and Cin in (16, 24, 32)


def conv3x3_wgrad(a, dz, dw, B, H, W, Cin, Cout):
    x = a.float().view(B, H, W, Cin).permute(0, 3, 1, 2)
    g = dz.float().view(B, H, W, Cout).permute(0, 3, 1, 2)
    gw = torch.nn.grad.conv2d_weight(x, (Cout, Cin, 3, 3), g, padding=1)           # [Cout, Cin, 3, 3]
    dw += gw.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin)
    return dw


def batchsum(x, out):
    out.add_(x.sum(0).view(out.shape))
    return out


def im2col_patch(img, P, ldk, out_dtype, sample_map=None, out=None):
    if sample_map is not None:
        img = img.index_select(0, sample_map)
    B, Cin, H, W = img.shape
    col = F.unfold(img, kernel_size=P, stride=P).transpose(1, 2).reshape(B * (H // P) * (W // P), Cin * P * P)
    res = torch.zeros(col.shape[0], ldk, dtype=out_dtype) if out is None else out.zero_()
    res[:, :col.shape[1]] = col.to(out_dtype)
    return res


def embed_cls(tokens, pos, x, keep, num_tokens=1):
    B, N, C = x.shape
    T = num_tokens
    rows = (tokens.reshape(-1, C)[:T] + pos.reshape(-1, C)[:T])[None].expand(B, T, C)
    if keep is not None:
        rows = rows * (torch.arange(C)[None, :] < keep.long()[:, None])[:, None, :]
    x[:, :T, :] = rows
    return x


def sr_im2col(y, B, g, C, num_tokens=1):
    T = num_tokens
    img = y.view(B, T + g * g, C)[:, T:, :].float().transpose(1, 2).reshape(B, C, g, g)
    u = F.unfold(img, kernel_size=3, stride=2, padding=1)                    # [B, C*9, go*go], k = (c, kh, kw)
    go = g // 2
    u = u.view(B, C, 9, go * go).permute(0, 3, 2, 1).reshape(B * go * go, 9 * C)   # k = (tap, c)
    return u.to(y.dtype)


def sr_col2im(dcol, dy, B, g, C, num_tokens=1):
    go, T = g // 2, num_tokens
    u = dcol.float().view(B, go * go, 9, C).permute(0, 3, 2, 1).reshape(B, C * 9, go * go)
    img = F.fold(u, output_size=(g, g), kernel_size=3, stride=2, padding=1)  # [B, C, g, g]
    dy.view(B, T + g * g, C)[:, T:, :] = img.flatten(2).transpose(1, 2).to(dy.dtype)
    return dy


def sr_resid(x, B, g, cin, cout, num_tokens=1):
    go, T = g // 2, num_tokens
    out = torch.zeros(B, T + go * go, cout)
    out[:, :T, :cin] = x[:, :T, :]
    img = x[:, T:, :].transpose(1, 2).reshape(B, cin, g, g)
    out[:, T:, :cin] = F.avg_pool2d(img, 2, 2).flatten(2).transpose(1, 2)
    return out


def sr_resid_bwd(dout, B, g, cin, cout, num_tokens=1):
    go, T = g // 2, num_tokens
    dx = torch.zeros(B, T + g * g, cin)
    dx[:, :T, :] = dout[:, :T, :cin]
    gi = dout[:, T:, :cin].transpose(1, 2).reshape(B, cin, go, go)
    gi = gi.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3) * 0.25
    dx[:, T:, :] = gi.flatten(2).transpose(1, 2)
    return dx


def mask_rows(x, keep, rows_per_sample):
    C = x.shape[-1]
    X = x.view(-1, C)
    mask, _ = _keep_mask(keep, X.shape[0], C, rows_per_sample)
    X.mul_(mask)
    return x


def im2col3x3_image(img, stride, ld, out_dtype):
    B, C, H, W = img.shape
    u = F.unfold(img, kernel_size=3, stride=stride, padding=1)               # [B, C*9, L], k = (c, tap)
    L = u.shape[-1]
    u = u.view(B, C, 9, L).permute(0, 3, 2, 1).reshape(B * L, 9 * C)
    out = torch.zeros(B * L, ld, dtype=out_dtype)
    out[:, :9 * C] = u.to(out_dtype)
    return out


def im2col3x3(a, B, H, W, C):
    img = a.view(B, H, W, C).float().permute(0, 3, 1, 2)
    u = F.unfold(img, kernel_size=3, stride=1, padding=1).view(B, C, 9, H * W).permute(0, 3, 2, 1)
    return u.reshape(B * H * W, 9 * C).to(a.dtype)


def col2im3x3(dcol, B, H, W, C):
    u = dcol.float().view(B, H * W, 9, C).permute(0, 3, 2, 1).reshape(B, C * 9, H * W)
    img = F.fold(u, output_size=(H, W), kernel_size=3, stride=1, padding=1)
    return img.permute(0, 2, 3, 1).reshape(B * H * W, C).to(dcol.dtype)


def bn_stats(z, s, q):
    z = z.float()
    s += z.sum(0)
    q += (z * z).sum(0)


def bn_relu(z, scale, shift, res, out_dtype):
    v = torch.relu(z.float() * scale + shift)
    if res is not None:
        v = v + res.float()
    return v.to(out_dtype)


def bn_bwd(da, z, scale, shift, mean, rstd, sg, sgz, training):
    z = z.float()
    g = da.float() * ((z * scale + shift) > 0)
    zh = (z - mean) * rstd
    sg += g.sum(0)
    sgz += (g * zh).sum(0)
    inv_n = 1.0 / z.shape[0] if training else 0.0
    return (scale * (g - sg * inv_n - zh * sgz * inv_n)).to(da.dtype)


def bn_relu_patch(z, scale, shift, res, B, H, W, patch, out_dtype):
    return patch_unfold(bn_relu(z, scale, shift, res, out_dtype), B, H // patch, W // patch, patch, z.shape[-1])


def bn_bwd_patch(dcol, z, scale, shift, mean, rstd, sg, sgz, training, B, H, W, patch):
    return bn_bwd(patch_fold(dcol, B, H // patch, W // patch, patch, z.shape[-1]), z, scale, shift, mean, rstd, sg, sgz, training)


def conv3x3_res_patch(a, w, res_col, B, H, W, Cin, Cout, res_patch, out_dtype):
    return conv3x3_res(a, w, patch_fold(res_col, B, H // res_patch, W // res_patch, res_patch, Cout), B, H, W, Cin, Cout, out_dtype)


def patch_unfold(a, B, gh, gw, P, C):
    x = a.view(B, gh, P, gw, P, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(B * gh * gw, P * P * C).clone()


def patch_fold(col, B, gh, gw, P, C):
    x = col.view(B, gh, gw, P, P, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(B * gh * P * gw * P, C).clone()


ALL = ["im2col3x3_image", "im2col3x3", "col2im3x3", "bn_stats", "bn_relu", "bn_bwd", "patch_unfold", "patch_fold", "gemm", "gemm_group", "cast_bf16", "cast_transpose_batch", "ln_fwd", "ln_bwd", "ln_grad_reduce", "gemm_ln_supported", "gemm_ln_fwd", "gemm_ln_bwd", "attn_fwd", "attn_bwd", "softce", "softce_train", "colsum", "scale_mask_cast",
       "batchsum", "conv3x3", "conv1_direct", "conv1_direct_supported", "conv3x3_bias_relu", "conv3x3_supported", "conv3x3_wgrad", "conv3x3_wgrad_supported", "token_mean", "token_mean_bwd", "im2col_patch", "embed_cls", "sr_im2col", "sr_col2im", "sr_resid", "sr_resid_bwd", "mask_rows", "zero_ranges", "zero_", "relayout", "conv3x3_res", "conv_w_flip", "bn_finalize", "conv3x3_bias_relu_patch", "bn_relu_patch", "bn_bwd_patch", "conv3x3_res_patch"]


def zero_ranges(buf, ranges):
    for lo, hi in ranges:
        buf[int(lo):int(hi)] = 0
    return buf


def zero_(t):
    return t.zero_()


def relayout(src, dst, A, B, C, dst_ld=None, src_ld=None):
    dst_ld = B * C if dst_ld is None else dst_ld
    src_ld = B * C if src_ld is None else src_ld
    sidx = (torch.arange(A)[:, None] * src_ld + torch.arange(B * C)[None, :]).reshape(-1)
    v = src.reshape(-1)[sidx].view(A, B, C).permute(0, 2, 1).reshape(A, B * C)
    didx = (torch.arange(A)[:, None] * dst_ld + torch.arange(B * C)[None, :]).reshape(-1)
    dst.reshape(-1)[didx] = v.reshape(-1).to(dst.dtype)
    return dst


def install(monkeypatch):
    import vitres.kernels as K
    g = globals()
    for name in ALL:
        monkeypatch.setattr(K, name, g[name])
    # the model refuses CPU tensors; tests lift that guard explicitly
    import vitres.nets.vit_sr_supernet as V
    monkeypatch.setattr(V, "_REQUIRE_CUDA", False, raising=False)
