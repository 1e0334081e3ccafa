/*
 * vitres_hip.h -- C ABI of libvitres_hip.so, the MI355X (gfx950) kernel library behind the
 * ViT-Res (super)network hot path of yilunliao/vit-search.
 *
 * The reference has NO native code and no FFI: the seam the hot path sits behind is the timm model
 * registry + nn.Module protocol (reference main.py:329-348).  This header is
 * therefore the boundary a maintainer of the reference would bind (with ctypes)
 * to replace the ATen op sequences listed next to each entry point.
 *
 * Conventions (all entry points):
 *   - extern "C", plain device pointers and sizes; no torch types.
 *   - The caller owns every buffer (inputs, outputs, workspaces); the library never allocates,
 *     frees or retains pointers and keeps no global mutable state; re-entrant.
 *   - All work is enqueued asynchronously on `stream`; no hidden synchronisation.
 *   - Return value: 0 = ok; > 0 = hipError_t of the launch; < 0 = argument validation
 *     (VR_EINVAL -1, VR_EALIGN -2, VR_EUNSUPPORTED -3).  No exceptions cross the ABI.
 *   - dtype codes: VR_F32 = 0 (exact-fp32 parity mode), VR_BF16 = 1 (raw bf16 bits, fast mode).
 *   - `keep` arrays: int32[B], active channel-prefix length of each sample for that tensor
 *     (the reference's ChannelDrop prefix masks, nets/channel_drop.py:153-154); NULL = dense.
 *   - "rows_per_sample": number of consecutive rows (tokens) that belong to one sample.
 */
#ifndef VITRES_HIP_H
#define VITRES_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* vr_stream_t; /* == hipStream_t */

#define VR_F32 0
#define VR_BF16 1

/* library / ABI version (major*1000 + minor) */
int vr_version(void);

/* row remap of a token-indexed operand: row(m) = (m / rpi) * rps + off + (m % rpi); rpi==0: identity */
typedef struct vr_rowmap {
    int32_t rpi, rps, off;
} vr_rowmap;

/*
 * Generic fused GEMM:  C[M,N] (+)= epilogue( A[M,K] * B[N,K]^T )
 * Replaces: nn.Linear forward/backward in nets/supernet_blocks.py:37-52,102-119 (qkv, proj, fc1, fc2),
 * cls_head/patch_head (nets/vit_sr_supernet.py:440-446), token_transform (:151), the patchify
 * convolutions expressed as GEMMs (timm PatchEmbed, nets/patch_conv.py:56-58, vit_sr_supernet.py:140),
 * and the `x * mask` / drop_path / residual adds that follow them (supernet_blocks.py:214-253).
 *
 *   a_trans/b_trans = 0: operand stored [rows, K] (K contiguous); 1: stored [K, rows] (rows contiguous).
 *   forward   y = x W^T      : a_trans 0, b_trans 0
 *   dgrad     dx = dy W      : a_trans 0, b_trans 1      (B = W viewed as [K_out rows][N_in contraction])
 *   wgrad     dW = dy^T x    : a_trans 1, b_trans 1, contraction over tokens, split_k > 1, atomic = 1
 * Epilogue, in this order:  v = acc (+ bias[n]) (+ pos[m % rows_in][n]);
 *   act==1: C gets the pre-activation u=v, C2 gets gelu(u), both 0 where masked by keep_n;   (Mlp.forward)
 *   dact_u != NULL: v *= gelu'(u[m][n]);                                          (fc2 dgrad -> du)
 *   keep_n: v = 0 where n >= keep_n[sample(m)];  scale: v *= scale[sample(m)];   (ChannelDrop, DropPath)
 *   resid != NULL: v += resid[out_row][n] (fp32, ldc layout);  atomic: atomicAdd into fp32 C.
 * Masked-work skipping (the supernet's sampled sub-networks): keep_k[s] promises that A[m, k] == 0 for
 * rows m of sample s and (k % k_period) >= keep_k[s]; K slices with no kept k for any sample of the tile are not
 * loaded, and output tiles with no kept column (keep_n) skip their whole K loop (they still store the masked value).
 * In wgrad form the samples are those of the token range of the split: keep_k bounds the kept output ROWS, keep_n the
 * kept output COLUMNS (rows/columns beyond are exactly zero gradients); fully masked tiles are not computed.
 */
typedef struct vr_gemm_args {
    const void* A;
    const void* B;
    void* C;
    void* C2;            /* act==1: second output gelu(u) (same dtype/ld as C) while C gets u; NULL with act==1: C gets
                            gelu(u) alone (forward-only evaluation: the pre-activation is not kept) */
    const float* bias;   /* [N] or NULL */
    const float* pos;    /* [rows_in, N] or NULL */
    const float* scale;  /* [batch] or NULL */
    const int32_t* keep_n; /* [batch] or NULL */
    const float* resid;  /* fp32 [*, ldc] or NULL (may alias C for in-place accumulate) */
    const void* dact_u;  /* pre-activation for gelu' (dtype = in_dtype, leading dim ldu) or NULL */
    const int32_t* keep_k; /* [batch] or NULL: work-skipping hint, see below */
    float* bias_grad;    /* wgrad only (a_trans && atomic): bias_grad[m] += sum_k A[k][m]  (the Linear's bias gradient,
                            fused so that dY is read once), or NULL */
    int32_t M, N, K;
    int32_t lda, ldb, ldc, ldu;
    int32_t a_trans, b_trans;
    int32_t in_dtype;    /* dtype of A and B (and dact_u) */
    int32_t out_dtype;   /* dtype of C/C2 */
    int32_t act;         /* 0 none; 1 gelu (dual store with C2: C = u, C2 = gelu(u); single store without: C = gelu(u)); 2 the
                            saved-derivative form of the same pair: forward (no dact_u) C = gelu'(u), C2 = gelu(u); data gradient
                            (dact_u given) multiplies by dact_u as it is -- no transcendental in the backward epilogue;
                            3 relu, single store: C = max(u, 0) (no C2 / dact_u) */
    int32_t atomic;      /* 1: atomicAdd into fp32 C; 2 (weight-gradient form, bf16 operands, 8-element aligned leading dimensions):
                            C and bias_grad are OVERWRITTEN by plain stores -- every output tile is computed by one workgroup over
                            the whole token range (split_k <= 1), fully masked tiles write zeros, the destination need not be
                            zero-filled; VR_EUNSUPPORTED where only the atomic kernels cover the form */
    int32_t split_k;     /* >= 1 */
    int32_t rows_in;     /* rows per sample of the M index (0: single sample); wgrad: tokens per sample of the K index */
    int32_t n_period;    /* > 0: column n is kept iff (n % n_period) < keep_n[s] (per-head prefixes of the qkv layout) */
    int32_t k_period;    /* same for keep_k */
    int32_t sched;       /* scheduling hints (bit mask, 0 = default).  None changes results beyond fp32 summation order.
                            1 = launched beside another kernel on a second stream (general kernel: 128x128 tile, one workgroup per
                                tile instead of persistent workgroups);
                            2 = keep the hardware's round-robin workgroup -> XCD order (default: tiles remapped so that an XCD owns
                                a contiguous run);
                            4 = always use the general kernel (gemm.hip) -- measurement aid;
                            8, 16, 32 = measurement aids of gemm_nt.hip (force its kernels, record stamps in ws);
                            64 (vr_gemm_group, first problem) = tn_body's instruction stream for the weight-gradient group instead of
                                the lean LDS-DMA one (tests); 0x10000 (same place) = the 8-wave, double-buffered group kernel -- one
                                workgroup per CU, one token split for the whole group (faster alone, +0.05 ms inside the step: opt-in);
                            128 (vr_gemm_group, first problem) = the group may fill the chip (default: at most two resident
                                workgroups per CU);
                            0x100 = gemm_nt.hip's kernels instead of the lean-loop ones (gemm_ntk.hip);
                            0x600 / 0x1800 = slice buffers (1 - 3; see `ring`) / tile (1: 128 x 128, 2: 64 x 128, 3: 64 x 64) of the
                                lean-loop kernels instead of their grid-size rule (tests);
                            0x40000 = the caller vouches that every reader of C / C2 is a kernel that tiles group by group (m_groups
                                below) and skips the masked channels of a row's group: bf16 outputs without a residual then leave
                                tiles that are masked (keep_n) for every row UNWRITTEN instead of storing zeros;
                            0x80000 = an operand of this launch was produced that way -- the launch runs on the group-by-group
                                kernels or fails with VR_EUNSUPPORTED, it is never handed to a kernel that tiles across groups;
                            0x200000 = the panel-resident kernel (gemm_panel.hip: K <= 320, bf16 result, K-contiguous weight; the A
                                panel in LDS, weight strips in registers) wherever it covers the form, 0x800000 with it: its 80-row /
                                4-wave form; 0x400000 / 0x1000000 / 0x2000000 with it: its measurement forms (no MFMA / no weight
                                loads: results are NOT the GEMM's).  Opt-in: measured slower than the tiled kernels */
    vr_rowmap a_map;     /* remap of A's token rows (M index if a_trans==0, K index if a_trans==1) */
    vr_rowmap b_map;     /* remap of B's token rows (only meaningful when b_trans==1 && a_trans==1) */
    vr_rowmap c_map;     /* remap of output rows */
    int32_t m_groups;    /* > 1: the token index (M rows; wgrad: K tokens) consists of this many equal, contiguous groups of samples
                            with different keep_k / keep_n each -- the architecture groups of the supernet's multi-arch step
                            (engine.py:119-165, channel_drop.py:101-105).  The kernels interleave the groups' row tiles (wgrad:
                            token splits) in their XCD-contiguous workgroup order, so that no XCD is dealt only the sparsest (or
                            only the densest) architecture; when m_groups divides the rows, the bf16 kernels (gemm_ntk / gemm_nt_ln
                            / gemm_tn) also cut the grid per group -- no tile or token split holds rows of two groups (the last
                            tile of a group is short).  Results do not depend on it.  A keep value -(k + 2) marks a sample that is
                            masked on its own inside a group of width k (DropPath): every kernel reads it as 0.  0 / 1: one group */
    void* ws;            /* optional workspace of the K-split kernels (k_shares below): the workgroups that share a tile exchange fp32
                            partial accumulators through it.  >= vr_gemm_ws_bytes() bytes, 16-byte aligned, ZERO before its first
                            use (the kernels leave its tickets at zero), never shared by launches that may run concurrently (one per
                            stream).  NULL: every tile is computed by one workgroup. */
    int64_t ws_bytes;
    int32_t ring;        /* lean-loop bf16 kernels (gemm_ntk.hip): slice buffers of the K loop's ring (1 - 6; one slice is multiplied while
                            ring - 1 are in flight).  0 = the library's rule: as deep as LDS allows without lowering the number of
                            workgroups the grid gives a CU */
    int32_t k_shares;    /* lean-loop bf16 kernels: workgroups that share a tile's K slices (needs ws; the partial fp32 tiles meet in ws by
                            plain write-through stores, the last arriver sums them in share order and runs the epilogue -- results do
                            not depend on arrival order, no fp32 atomics touch the output).  0 / 1 = one workgroup per tile (no rule
                            turns the split on: every real split measured slower than the unsplit kernel on the step's shapes,
                            DESIGN.md), 2 - 4 = that many shares wherever the form has a split kernel and ws holds tiles x shares
                            slabs of the tile's fp32 size */
} vr_gemm_args;

/* bytes of vr_gemm_args.ws that enable tile sharing on the current device */
int vr_gemm_ws_bytes(void);

int vr_gemm(const vr_gemm_args* args, vr_stream_t stream);

/*
 * vr_gemm_group: exactly `for (i < count) vr_gemm(&args[i], stream)`, but 2..4 bf16 weight-gradient problems of the split-token
 * form (a_trans && b_trans && atomic, split_k == 0, un-mapped rows) -- the four Linears of one transformer block, autograd of
 * F.linear at nets/supernet_blocks.py:36,48,102,118 -- run as ONE launch whose workgroups share the CUs.  Any other mix is
 * issued problem by problem.  Outputs must not alias between problems.
 */
int vr_gemm_group(const vr_gemm_args* args, int32_t count, vr_stream_t stream);

/*
 * vr_gemm with the adjacent LayerNorm fused into its epilogue (workgroups own whole output rows; bf16 operands, fp32 output,
 * N % 8 == 0, N <= 512, no n_period / pos / act / row maps on the output; ldc is also the row stride of x, resid, gt_out).
 *   mode 0 -- forward of attention `proj` / Mlp `fc2` (nets/supernet_blocks.py:214-253) and the LayerNorm that consumes the
 *     updated residual stream (norm2 of the block, norm1 of the next block):
 *       C = resid + scale[s] * mask_{keep_n}(A B^T + bias)                     exactly vr_gemm
 *       (y, mean, rstd) = MaskedLayerNorm(C; w, b, keep, eps)                  exactly vr_ln_fwd with bf16 output, y is [M, N]
 *   mode 1 -- data gradient of `qkv` / `fc1` and the backward of the LayerNorm that fed them (nets/masked_layer_norm.py:55-88):
 *       dy = A B^T  (A = dU, B = W^T K-contiguous; never written),
 *       C = (resid ? resid : 0) + dLN/dx(dy);  dw += sum dy*xhat;  db += sum dy;
 *       gt_out[m,c] = c < gt_keep[s] ? C[m,c] * gt_scale[s] : 0 (bf16, optional)  exactly vr_ln_bwd on a fp32 dy
 *     (args.bias / scale / keep_n must be NULL; keep_k / k_period / rows_in keep their vr_gemm meaning).
 * Kernel: 64 x 256 / 64 x 320 / 32 x 512 tiles, two or three workgroups per CU (gemm_nt_ln.hip).
 */
typedef struct vr_ln_epilogue {
    int32_t mode;
    float eps;               /* mode 0 */
    const float* w;          /* LayerNorm weight [N] */
    const float* b;          /* LayerNorm bias [N] (mode 0) */
    const int32_t* keep;     /* [batch] kept prefix of the LayerNorm, NULL = all N channels */
    void* y;                 /* mode 0 out: bf16 [M, N] */
    float* mean;             /* mode 0 out / mode 1 in: [M] */
    float* rstd;
    const float* x;          /* mode 1: the LayerNorm's input saved by the forward, fp32 [M, ldc] */
    float* dw;               /* mode 1: fp32 [max(grad_copies,1)][N], accumulated with atomics (see vr_ln_bwd) */
    float* db;
    void* gt_out;            /* mode 1, optional: bf16 [M, ldc] */
    const float* gt_scale;   /* [batch] or NULL */
    const int32_t* gt_keep;  /* [batch] or NULL */
    int32_t grad_copies;     /* mode 1: rows of partial sums behind dw / db, 0 or 1 = accumulate into dw / db themselves */
} vr_ln_epilogue;
int vr_gemm_ln(const vr_gemm_args* args, const vr_ln_epilogue* ln, vr_stream_t stream);
int vr_gemm_ln_supported(int32_t N);

/*
 * vr_gemm_ln_fold: mode 0 of vr_gemm_ln for widths where a row spans several tiles (N up to 2048; the Linear + LayerNorm pairs of
 * stages 2 and 3: nets/supernet_blocks.py:214-253 followed by nets/masked_layer_norm.py:113-125) --
 *   C = resid + scale[s] * mask_{keep_n}(A B^T + bias)   (fp32, exactly vr_gemm),   (y, mean, rstd) = MaskedLayerNorm(C; w, b, keep, eps)
 * in ONE launch of the tiled lean-loop kernel: every 64 x 128 tile stores its part of C write-through and takes the ticket of its row
 * block; the workgroup that holds a block's last ticket runs vr_ln_fwd's row routine on the block's rows (same arithmetic: results
 * equal vr_gemm followed by vr_ln_fwd up to nothing -- the same fp32 operations in the same order per row).
 * bf16 operands, fp32 C with bias and residual, ldc == N, N % 4 == 0, un-mapped output rows, args->ws >= 16 KB with zeroed tickets (the
 * kernels leave them at zero; never shared by launches that may run concurrently).  VR_EUNSUPPORTED (nothing launched) otherwise.
 */
int vr_gemm_ln_fold(const vr_gemm_args* args, const vr_ln_epilogue* ln, vr_stream_t stream);

/* fp32 -> bf16 (round to nearest even), n elements.  Replaces autocast's per-op weight casts (engine.py:112). */
int vr_cast_f32_bf16(const float* src, void* dst, int64_t n, vr_stream_t stream);

/*
 * Transposed bf16 shadows of a batch of fp32 matrices in one launch: for every descriptor d,
 *   dst[d.dst_off + c * d.ld_dst + r] = bf16(src[d.src_off + r * d.cols + c])   r < rows, c < cols
 * (offsets in elements; ld_dst >= rows; columns rows..ld_dst of dst are not written -- keep them zero).
 * The data-gradient GEMM of a Linear (autograd of F.linear, nets/supernet_blocks.py:41,110) then reads W^T K-contiguous
 * like the forward reads W.  descs is a DEVICE array of n descriptors; max_tiles = max over d of
 * ceil(rows/64) * ceil(cols/64).
 */
typedef struct vr_tr_desc {
    int64_t src_off;
    int64_t dst_off;
    int32_t rows;
    int32_t cols;
    int32_t ld_dst;
    int32_t reserved;
} vr_tr_desc;
int vr_cast_transpose_batch(const float* src, void* dst, const vr_tr_desc* descs, int32_t n, int32_t max_tiles,
                            vr_stream_t stream);

/*
 * AdamW over the flat fp32 parameter arena (timm create_optimizer -> torch.optim.AdamW, main.py:385; decoupled weight decay,
 * bias-corrected moments), one pass, fused with: the bf16 shadow of the updated parameters (NULL to skip), gradient scaling
 * (grad_scale, e.g. 1/world after a sum all-reduce) and the ModelEmaV2 update ema = d*ema + (1-d)*p (main.py:357-363;
 * NULL to skip).  Hyper-parameters come per GROUP: group_of_8[i] (device, one byte per 8 consecutive elements; arena
 * parameters start at multiples of 8) selects groups[...]; 255 = elements that are not updated (padding, frozen).
 *   p *= 1 - lr*wd ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr/bias_c1 * m / (sqrt(v)/sqrt_bias_c2 + eps)
 * with bias_c1 = 1 - b1^t, sqrt_bias_c2 = sqrt(1 - b2^t) computed by the caller for step t.  n % 8 == 0.
 */
#define VR_ADAMW_MAX_GROUPS 16
typedef struct vr_adamw_group {
    float lr, beta1, beta2, eps, weight_decay, bias_c1, sqrt_bias_c2, grad_scale;
} vr_adamw_group;
int vr_adamw_flat(float* p, const float* g, float* m, float* v, void* shadow, float* ema, float ema_decay,
                  const uint8_t* group_of_8, const vr_adamw_group* groups, int32_t n_groups, int64_t n,
                  vr_stream_t stream);
/* Same pass with `groups_dev` in DEVICE memory (read by the kernel at run time): a launch captured into a hipGraph then follows the
 * schedule -- the caller rewrites the n_groups structs before every replay.  All array arguments may point into the middle of
 * the arena (a range of it, offsets multiples of 8 elements; group_of_8 advanced by offset / 8).  A group whose bias_c1 is 0
 * (never the case for a real step) is skipped: an all-zero block turns a captured launch into a no-op for that replay. */
int vr_adamw_flat_dev(float* p, const float* g, float* m, float* v, void* shadow, float* ema, float ema_decay,
                      const uint8_t* group_of_8, const vr_adamw_group* groups_dev, int32_t n_groups, int64_t n,
                      vr_stream_t stream);
/* ... launched with at most max_blocks workgroups (0: the default grid): the update of a finished arena range that runs beside the
 * rest of the backward on the weight gradients' stream (round 4). */
int vr_adamw_flat_dev_capped(float* p, const float* g, float* m, float* v, void* shadow, float* ema, float ema_decay,
                             const uint8_t* group_of_8, const vr_adamw_group* groups_dev, int32_t n_groups, int64_t n,
                             int32_t max_blocks, vr_stream_t stream);


/*
 * Masked LayerNorm forward (nets/masked_layer_norm.py:23-50,113-125).  x fp32 [M,C] -> y (dtype) [M,C];
 * mean/rstd fp32 [M] saved for backward.  keep NULL -> plain LayerNorm (F.layer_norm path :118-122).
 */
int vr_ln_fwd(const float* x, const float* w, const float* b, void* y, float* mean, float* rstd,
              const int32_t* keep, int32_t M, int32_t C, int32_t rows_per_sample, float eps,
              int32_t out_dtype, vr_stream_t stream);

/*
 * Masked LayerNorm backward (nets/masked_layer_norm.py:55-88).  dx_out = (dx_in ? dx_in : 0) + dLN/dx;
 * dw/db (fp32 [C]) are accumulated with atomics (caller zeroes them).  dy has dtype `dy_dtype`.
 * grad_copies > 1: dw and db are [grad_copies][C] rows of partial sums, workgroup i adds into row i % grad_copies (hundreds
 *   of workgroups hitting the same 2C addresses cost a third of the kernel); vr_ln_grad_reduce folds the rows into the
 *   parameter gradients afterwards.
 * gt_out (optional, dtype `dy_dtype`, [M,C]): the gradient entering the NEXT backward branch, written in the same pass --
 *   gt_out[m,c] = c < gt_keep[s] ? dx_out[m,c] * gt_scale[s] : 0   (s = m / rows_per_sample; NULL scale = 1, NULL keep = C)
 * i.e. what vr_scale_mask_cast would produce from dx_out (DropPath nets/drop.py:11-26 + ChannelDrop mask backward).
 */
int vr_ln_bwd(const void* dy, const float* x, const float* w, const float* mean, const float* rstd,
              const int32_t* keep, const float* dx_in, float* dx_out, float* dw, float* db,
              void* gt_out, const float* gt_scale, const int32_t* gt_keep,
              int32_t M, int32_t C, int32_t rows_per_sample, int32_t dy_dtype, int32_t grad_copies, vr_stream_t stream);

/*
 * Fold the partial rows written by vr_ln_bwd / vr_gemm_ln (mode 1) with grad_copies = `copies` into the LayerNorm parameter
 * gradients: dw[c] += sum_k part_w[k][c], db[c] += sum_k part_b[k][c]; the partial rows are zero again afterwards (the
 * caller allocates them zero-filled once).  One launch per 32 slots.
 */
typedef struct vr_ln_grad_slot {
    float* part_w;           /* [copies][C] */
    float* part_b;           /* [copies][C] */
    float* dw;               /* [C] */
    float* db;               /* [C] */
    int32_t C;
    int32_t reserved;
} vr_ln_grad_slot;
int vr_ln_grad_reduce(const vr_ln_grad_slot* slots, int32_t count, int32_t copies, vr_stream_t stream);

/*
 * Multi-head self-attention core (nets/supernet_blocks.py:105-109): qkv [B,N,3,H,D] -> o [B,N,H*D],
 * lse fp32 [B,H,N].  Heads h >= keep_hd[b]/D are written as zeros (Attention's ChannelDrop, :111-112).
 */
int vr_attn_fwd(const void* qkv, void* o, float* lse, const int32_t* keep_hd, int32_t B, int32_t N,
                int32_t H, int32_t D, float scale, int32_t dtype, vr_stream_t stream);
int vr_attn_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta /* [B,H,N] scratch */,
                void* dqkv, const int32_t* keep_hd, int32_t B, int32_t N, int32_t H, int32_t D, float scale,
                int32_t dtype, vr_stream_t stream);

/*
 * Soft-target cross entropy, forward + gradient in one pass (timm SoftTargetCrossEntropy as used by
 * engine.py:153-157): loss_rows[r] = sum_k -t[r,k] * log_softmax(x[r])[k];
 * dlogits[r,k] = gscale * (softmax(x[r])[k] * sum_k t[r,k] - t[r,k]).
 */
int vr_softce(const float* logits, const float* target, float* loss_rows, float* dlogits, int32_t R,
              int32_t K, float gscale, vr_stream_t stream);

/*
 * The loss step of the training loop in one pass (engine.py:153-157 with timm's SoftTargetCrossEntropy): row r of `logits`
 * [R, K] belongs to internal sample s = r / rows_per_sample and is scored against target row
 * (sample_map ? sample_map[s] : s) * rows_per_sample + r % rows_per_sample  (the model runs a batch grouped by architecture,
 * targets stay in the caller's order);  *loss_acc += loss_scale * loss_row (atomics; caller zeroes it);
 * dlogits[r, k] = gscale * (softmax(x[r])[k] * sum_k t[.,k] - t[.,k]) in `grad_dtype` with row pitch ld_grad >= K, columns
 * K..ld_grad-1 zeroed -- exactly what the head's backward GEMMs read.
 */
int vr_softce_train(const float* logits, const float* target, const int64_t* sample_map, int32_t rows_per_sample,
                    float* loss_acc, void* dlogits, int32_t ld_grad, int32_t grad_dtype, int32_t R, int32_t K,
                    float gscale, float loss_scale, vr_stream_t stream);

/* out[n] += sum_m in[map(m), n]  (bias gradients; atomics, caller zeroes out).  in: dtype [*, ld]. */
int vr_colsum(const void* in, float* out, int32_t M, int32_t N, int32_t ld, int32_t dtype, vr_rowmap map, vr_stream_t stream);

/*
 * Gradient entering a masked / drop-path'd residual branch (autograd of supernet_blocks.py:243-253 and
 * drop.py:24-25): out[m,c] = dtype(in[m,c] * scale[sample]) for c < keep[sample], else 0.  in fp32 [M,C].
 */
int vr_scale_mask_cast(const float* in, void* out, const float* scale, const int32_t* keep, int32_t M, int32_t C,
                       int32_t rows_per_sample, int32_t out_dtype, vr_stream_t stream);

/*
 * SwitchTokenMix (token_mixup.py:39-162), device half: the caller has drawn, with the reference's generators and order,
 * partner[b] (global sample index each sample is mixed with: the two torch.randperm of :104 and :126 offset into their half),
 * the patch box [y0,y1) x [x0,x1) in units of the patch_len x patch_len grid, lam_patch = 1 - box/grid, lam_img ~ Beta(.8,.8)
 * (each lam and 1 - lam rounded to fp32 from the double, as torch does for `tensor * python_float`), and the smoothed
 * one-hot values on/off (:22-23).  Rows [0, half): out = partner's pixels inside the box; targets = y*lam_p + y[partner]*(1-lam_p);
 * patch_targets[b, (i,j)] = y[partner] inside the box, y outside.  Rows [half, B): out = x*lam_i + x[partner]*(1-lam_i);
 * targets likewise; every patch target = the mixed target.  samples/out fp32 [B,C,H,W] (out != samples: the reference's
 * in-place update reads a gathered COPY), labels int64 [B], targets fp32 [B,K], patch_targets fp32 [B, patch_len^2, K].
 * Bit-exact with the reference (separately rounded products, then one add).
 */
int vr_token_mix(const float* samples, float* out, const int64_t* labels, const int64_t* partner, float* targets,
                 float* patch_targets, int32_t B, int32_t C, int32_t H, int32_t W, int32_t num_classes, int32_t patch_len,
                 int32_t half, int32_t y0, int32_t y1, int32_t x0, int32_t x1, float lam_patch, float oml_patch,
                 float lam_img, float oml_img, float on_value, float off_value, vr_stream_t stream);

/*
 * patch_output_type == 'avg' (nets/vit_sr_supernet.py:447-449: patch_features.mean(dim=1) before patch_head):
 * out[b, c] = mean over tokens n in [first, N) of y[b, n, c]; backward writes dy[b, n, c] = dmean[b, c] / (N - first) for those
 * tokens.  All tensors in `dtype`, fp32 accumulation.
 */
int vr_token_mean(const void* y, void* out, int32_t B, int32_t N, int32_t C, int32_t first, int32_t dtype, vr_stream_t stream);
int vr_token_mean_bwd(const void* dmean, void* dy, int32_t B, int32_t N, int32_t C, int32_t first, int32_t dtype,
                      vr_stream_t stream);

/* out[r, c] += sum_b in[b, r, c]   (pos_embed / tokens gradients: sum over the batch; fp32 atomics -- the caller
 * zero-initialises out, as the gradient arena is). */
int vr_batchsum(const float* in, float* out, int32_t B, int64_t inner, vr_stream_t stream);

/*
 * timm PatchEmbed as a GEMM operand (vit_sr_supernet.py:227,234): image fp32 NCHW [B,Cin,H,W] ->
 * col (dtype) [B*gh*gw, ldk] with k = (c, i, j), zero padded up to ldk.
 */
int vr_im2col_patch(const float* img, void* col, int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t P,
                    int32_t ldk, int32_t dtype, vr_stream_t stream);
/* The same with output sample b read from image sample_map[b] (NULL: identity): the batch re-ordering of the arch-grouped
 * execution order folded into the gather (one pass over the images instead of index_select + im2col). */
int vr_im2col_patch_map(const float* img, void* col, const int64_t* sample_map, int32_t B, int32_t Cin, int32_t H, int32_t W,
                        int32_t P, int32_t ldk, int32_t dtype, vr_stream_t stream);

/* token rows of the embedding: x[b,t,c] = (tokens[t,c] + pos[t,c]) masked by keep, t < num_tokens (1: class token; 2: class +
 * distillation token of the *_distill_* factories) (vit_sr_supernet.py:399-407) */
int vr_embed_cls(const float* tokens, const float* pos, float* x, const int32_t* keep, int32_t B, int32_t N,
                 int32_t C, int32_t num_tokens, vr_stream_t stream);

/*
 * SpatialReductionPatchEmbedding pieces (nets/vit_sr_supernet.py:114-172), grid g x g -> g/2 x g/2:
 * T = num_tokens leading token rows per sample (1 or 2), then the g*g patch rows:
 *  vr_sr_im2col : y (dtype) [B,T+g*g,C] -> col (dtype) [B*(g/2)^2, 9*C], k = (kh,kw,c), 3x3 stride 2 pad 1
 *  vr_sr_col2im : dcol -> dy rows T.. (gather form, no atomics); token rows left untouched
 *  vr_sr_resid  : out fp32 [B,T+(g/2)^2,Cout] = zero-padded residual (token rows copied; 2x2 avg-pool of patches)
 *  vr_sr_resid_bwd : dx[b,t,:C] (+)= dout[b,t,:C] for t < T; dx[b,T+p,:C] (+)= 0.25*dout[b,T+p/2..]
 */
int vr_sr_im2col(const void* y, void* col, int32_t B, int32_t g, int32_t C, int32_t num_tokens, int32_t dtype, vr_stream_t stream);
int vr_sr_col2im(const void* dcol, void* dy, int32_t B, int32_t g, int32_t C, int32_t num_tokens, int32_t dtype, vr_stream_t stream);
int vr_sr_resid(const float* x, float* out, int32_t B, int32_t g, int32_t Cin, int32_t Cout, int32_t num_tokens, vr_stream_t stream);
int vr_sr_resid_bwd(const float* dout, float* dx, int32_t B, int32_t g, int32_t Cin, int32_t Cout,
                    int32_t accumulate, int32_t num_tokens, vr_stream_t stream);

/*
 * Zero-fill ranges [lo[i], lo[i] + count[i]) (elements) of one fp32 buffer in one launch: optimizer.zero_grad() of the flat
 * gradient arena (reference engine.py:175) minus the spans whose weight gradients are written in store form (vr_gemm atomic == 2).
 */
#define VR_MAX_ZERO_RANGES 24
typedef struct vr_range_list {
    int64_t lo[VR_MAX_ZERO_RANGES];
    int64_t count[VR_MAX_ZERO_RANGES];
    int32_t n;
    int32_t reserved;
} vr_range_list;
int vr_zero_ranges(float* base, const vr_range_list* ranges, vr_stream_t stream);

/*
 * dst[a * dst_ld + c * B + b] = src[a * src_ld + b * C + c]  (dtype codes as everywhere; pad columns of dst untouched).
 * Replaces the `.permute(0, 2, 3, 1).reshape(...)` / `.permute(0, 3, 1, 2)` copies between the reference's convolution weights
 * [out, in, kh, kw] (nets/patch_conv.py:26-27, vit_sr_supernet.py:96-97) and the [out, (kh, kw, in)] form their GEMMs read, and the
 * row copy of the timm PatchEmbed weight into 16-byte-aligned rows (B = 1).
 */
int vr_relayout(const void* src, void* dst, int32_t A, int32_t B, int32_t C, int64_t src_ld, int64_t dst_ld, int32_t src_dtype,
                int32_t dst_dtype, vr_stream_t stream);

/* x[m, c] = 0 for c >= keep[sample(m)]  (ChannelDrop.forward `x * mask`, nets/channel_drop.py:82) */
int vr_mask_rows(float* x, const int32_t* keep, int32_t M, int32_t C, int32_t rows_per_sample, vr_stream_t stream);

/*
 * Convolutional patch embedding (reference nets/patch_conv.py:23-73) as NHWC gathers around vr_gemm.
 *  vr_im2col3x3   : 3x3 / pad 1 window gather -> col [B*Ho*Wo, ld], k = (kh, kw, c).  src_nchw_f32 = 1: src is the fp32
 *                   NCHW image (conv1, any stride, ld >= 9*C zero padded); 0: src is NHWC in `dtype` (stride 1, C % 8 == 0,
 *                   ld == 9*C).
 *  vr_col2im3x3   : its adjoint for the NHWC / stride-1 case (gather form, no atomics).
 *  vr_bn_stats    : sum[c] += sum_r z[r,c], sumsq[c] += sum_r z[r,c]^2  (train-mode BatchNorm2d statistics; caller zeroes).
 *  vr_bn_relu     : out = relu(z * scale[c] + shift[c]) (+ res)          (ConvBnAct.forward :34-38 with folded BN affine).
 *  vr_bn_bwd      : g = da * [bn(z) > 0]; sg[c] += sum g; sgz[c] += sum g*zhat (caller zeroes); then
 *                   dz = scale * (g - sg/R - zhat*sgz/R) (training) or scale * g (eval).  d gamma = sgz, d beta = sg.
 *  z (the pre-BatchNorm convolution output) is fp32 or, with bf16 activations, bf16 (`z_dtype`; what autocast gives the reference's
 *  convolutions): statistics and gradients are accumulated in fp32 either way.
 *  vr_patch_unfold: non-overlapping P x P patches, a NHWC [B, gh*P, gw*P, C] <-> col [B*gh*gw, (i, j, c)] (fold != 0: col -> a).
 */
int vr_im2col3x3(const void* src, void* col, int32_t B, int32_t H, int32_t W, int32_t C, int32_t stride,
                 int32_t src_nchw_f32, int32_t ld, int32_t dtype, vr_stream_t stream);
int vr_col2im3x3(const void* dcol, void* dsrc, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype, vr_stream_t stream);

/*
 * Direct 3x3 / stride 1 / pad 1 convolution on MFMA (bf16 fast path of patch_conv.py's conv2 / conv3 and, with the weights
 * flipped and transposed by the caller, of their data gradients): a bf16 NHWC [B,H,W,Cin], w bf16 [Cout, (kh,kw,ci)],
 * out [B*H*W, Cout] in out_dtype.  No im2col matrix is materialised.  Cin in {16, 24, 32}, Cout <= 32, Cout % 4 == 0;
 * VR_EUNSUPPORTED otherwise (callers fall back to vr_im2col3x3 + vr_gemm).
 */
int vr_conv3x3(const void* a, const void* w, void* out, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
               int32_t out_dtype, vr_stream_t stream);
/*
 * The same convolution with BatchNorm folded in for evaluation (nets/patch_conv.py:23-73 in eval mode: the caller scales the
 * weights by gamma / sqrt(var + eps) per output channel): out = relu(conv(a, w) + bias[co]) (+ res[pixel, co], bf16).
 */
int vr_conv3x3_bias_relu(const void* a, const void* w, const float* bias, const void* res, void* out, int32_t B, int32_t H,
                         int32_t W, int32_t Cin, int32_t Cout, int32_t out_dtype, vr_stream_t stream);
/* The same with the result written as the patchify operand of the projection behind it ([B*(H/patch)*(W/patch), (i, j, co)], the layout of
 * vr_patch_unfold; H, W multiples of patch): the last ReLU of the stem and the unfold in one pass (evaluation). */
int vr_conv3x3_bias_relu_patch(const void* a, const void* w, const float* bias, const void* res, void* out, int32_t B, int32_t H,
                               int32_t W, int32_t Cin, int32_t Cout, int32_t patch, int32_t out_dtype, vr_stream_t stream);
/* out = conv(a, w) + res[pixel, co] (bf16): the data gradient of conv2 plus the gradient arriving over the stem's skip connection
 * (autograd of `x = conv3(conv2(a1)) + a1`, nets/patch_conv.py:69) without a separate add pass. */
int vr_conv3x3_res(const void* a, const void* w, const void* res, void* out, int32_t B, int32_t H, int32_t W, int32_t Cin,
                   int32_t Cout, int32_t out_dtype, vr_stream_t stream);
/* the same with `res` stored in PATCH order (vr_patch_unfold's layout, windows of res_patch x res_patch pixels): the projection's data
 * gradient as its GEMM writes it -- no fold pass between the 7 x 7 / stride-7 projection and the stem's backward (round 4). */
int vr_conv3x3_res_patch(const void* a, const void* w, const void* res, void* out, int32_t B, int32_t H, int32_t W, int32_t Cin,
                         int32_t Cout, int32_t out_dtype, int32_t res_patch, vr_stream_t stream);
/* Weights of that data-gradient convolution: dst[ci, (kh, kw, co)] = src[co, ci, 2 - kh, 2 - kw] (src fp32 [Co, Ci, 3, 3]). */
int vr_conv_w_flip(const float* src, void* dst, int32_t Co, int32_t Ci, int32_t dst_dtype, vr_stream_t stream);

/*
 * conv1 of the conv patch embedding (nets/patch_conv.py:63: 3x3 / stride 2 / pad 1, 3 -> Cout <= 32 channels) straight from
 * the fp32 NCHW image: out [B*Ho*Wo, Cout] (NHWC, fp32 or bf16) = conv(img, w) (+ bias, relu: the evaluation form with
 * BatchNorm folded in).  w bf16 [Cout, 32]: k = (kh, kw, c) in the first 27 columns, zeros behind.  No im2col matrix.
 */
int vr_conv1_direct(const float* img, const void* w, const float* bias, void* out, int32_t B, int32_t H, int32_t W, int32_t Cout,
                    int32_t relu, int32_t out_dtype, vr_stream_t stream);
/* Its weight gradient: dw[co, (kh,kw,ci)] (fp32, [Cout, 9*Cin]) += sum over pixels of dz[p, co] * a[p + (kh-1, kw-1), ci]; a, dz
 * bf16 NHWC.  Covered: Cin == Cout in {16, 24, 32} (the stem's conv2 / conv3); VR_EUNSUPPORTED otherwise. */
int vr_conv3x3_wgrad(const void* a, const void* dz, float* dw, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                     vr_stream_t stream);
int vr_bn_stats(const void* z, float* sum, float* sumsq, int64_t R, int32_t C, int32_t z_dtype, vr_stream_t stream);
/* Between vr_bn_stats and vr_bn_relu in training (torch.nn.BatchNorm2d, nets/patch_conv.py:23-38): mean = sum / n, var = sumsq / n -
 * mean^2 (>= 0); running_mean / running_var (optional, both or none) <- (1 - momentum) * running + momentum * (mean | var * n / (n - 1));
 * *num_batches_tracked += 1 (optional); rstd = 1 / sqrt(var + eps), scale = weight * rstd, shift = bias - mean * scale. */
int vr_bn_finalize(const float* sum, const float* sumsq, int64_t n, const float* weight, const float* bias, float eps, float momentum,
                   float* running_mean, float* running_var, int64_t* num_batches_tracked, float* scale, float* shift, float* mean,
                   float* rstd, int32_t C, vr_stream_t stream);
int vr_bn_relu(const void* z, const float* scale, const float* shift, const void* res, void* out, int64_t R, int32_t C,
               int32_t dtype, int32_t z_dtype, vr_stream_t stream);
int vr_bn_bwd(const void* da, const void* z, const float* scale, const float* shift, const float* mean, const float* rstd,
              float* sg, float* sgz, void* dz, int64_t R, int32_t C, int32_t training, int32_t dtype, int32_t z_dtype,
              vr_stream_t stream);
int vr_patch_unfold(void* a, void* col, int32_t B, int32_t gh, int32_t gw, int32_t P, int32_t C, int32_t fold, int32_t dtype,
                    vr_stream_t stream);
/* Training-mode patchify without the unfold / fold passes (nets/patch_conv.py:56-58,70-72; round 4): vr_bn_relu_patch writes
 * relu(bn(z)) (+ res, NHWC) straight into the patchify operand col [B*(H/patch)*(W/patch), (i, j, c)] of the projection GEMM;
 * vr_bn_bwd_patch reads `da` in that layout (the projection's data gradient as its GEMM leaves it), z and dz stay NHWC. */
int vr_bn_relu_patch(const void* z, const float* scale, const float* shift, const void* res, void* out, int32_t B, int32_t H,
                     int32_t W, int32_t patch, int32_t C, int32_t dtype, int32_t z_dtype, vr_stream_t stream);
int vr_bn_bwd_patch(const void* da, const void* z, const float* scale, const float* shift, const float* mean, const float* rstd,
                    float* sg, float* sgz, void* dz, int32_t B, int32_t H, int32_t W, int32_t patch, int32_t C, int32_t training,
                    int32_t dtype, int32_t z_dtype, vr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VITRES_HIP_H */
