/*
 * Entry points that exist only in `make -C vit-search_amd/csrc EXPERIMENTAL=1` builds (vr_experimental() == 1): kernel forms that
 * are parity-tested but measured slower than the default path inside the workloads of bench.py.  The
 * product library (vit-search_amd/lib/libvitres_hip.so) does not export them and nothing in vitres/ calls them by default.
 */
#ifndef VITRES_HIP_EXPERIMENTAL_H
#define VITRES_HIP_EXPERIMENTAL_H
#include "vitres_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/*
 * Fused forward MLP of a transformer block for the forward-only paths (engine.py:194-261 evaluate, evolutionary-search candidate
 * scoring): Mlp.forward (nets/supernet_blocks.py:37-52) + the block's DropPath / channel masks / residual add (:247-253) in ONE
 * kernel -- the hidden tensor [rows, F] is never written:
 *     out[r, :] = resid[r, :] + scale[s] * mask_{keep_out[s]}( mask_{keep_hid[s]}( gelu(y[r, :] W1^T + b1) ) W2^T + b2 )
 * y: bf16 [rows, ldy] (the block's norm2 output; columns >= keep_in[s] are zero), W1: bf16 [F, ldw1], W2: bf16 [C, ldw2],
 * resid / out: fp32 [rows, ldo].  Row m of the problem (m < M, sample s = m / rows_in) is row map(m) of y, resid and out.
 * Same results as vr_gemm(act = 1, single store) followed by vr_gemm(resid, scale, keep_n) up to the bf16 rounding of the hidden
 * activations (identical) and fp32 summation order.  C <= 320 (the first stage of every shipped search space), C % 8 == 0,
 * F <= 2048: vr_mlp_fwd_supported(C, F).
 */
typedef struct vr_mlp_args {
    const void* y;
    const void* w1;
    const float* b1;            /* [F] or NULL */
    const void* w2;
    const float* b2;            /* [C] or NULL */
    const float* resid;
    float* out;
    const float* scale;         /* [batch] or NULL */
    const int32_t* keep_in;     /* [batch] or NULL: kept prefix of y's columns (work skipping) */
    const int32_t* keep_hid;    /* [batch] or NULL: kept prefix of the hidden units */
    const int32_t* keep_out;    /* [batch] or NULL: kept prefix of the output columns */
    int32_t M, C, F;
    int32_t ldy, ldw1, ldw2, ldo;
    int32_t rows_in;            /* rows per sample of the m index (0: one sample) */
    vr_rowmap map;
} vr_mlp_args;
int vr_mlp_fwd(const vr_mlp_args* args, vr_stream_t stream);
int vr_mlp_fwd_supported(int32_t C, int32_t F);

#ifdef __cplusplus
}
#endif
#endif
