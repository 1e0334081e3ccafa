// Step tail on the flat parameter arena: AdamW as timm 0.3.2's create_optimizer builds it for the
// reference (main.py:385; torch.optim.AdamW semantics: decoupled weight decay, bias-corrected moments), fused with what
// otherwise are separate passes over the 70-144 M parameters: the bf16 shadow the next forward's GEMMs read
// (vr_cast_f32_bf16), the 1/world averaging of the all-reduced gradient, and the ModelEmaV2 update (main.py:357-363,
// ema = d * ema + (1 - d) * p).  One streaming pass, 16-byte accesses, HBM-bound: 28 B/param (+2 shadow, +8 EMA).
#include <cstdlib>

#include "common.h"
#include "../../include/vitres_hip.h"

namespace {

struct Groups {
    vr_adamw_group g[VR_ADAMW_MAX_GROUPS];
};

// DEV: the per-group hyper-parameters are read from device memory (groups_dev) instead of the launch arguments, so that a launch
// captured into a hipGraph follows the learning-rate schedule and the bias corrections of every replayed step.
template <bool DEV>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ shadow, float* __restrict__ ema,
                                                    float ema_decay, const uint8_t* __restrict__ group_of_8, Groups groups,
                                                    const vr_adamw_group* __restrict__ groups_dev, long long n8) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const int gi = group_of_8[i];
        if (gi == 255) continue;                                   // padding / frozen parameters
        const vr_adamw_group h = DEV ? groups_dev[gi] : groups.g[gi];
        if (DEV && h.bias_c1 == 0.f) continue;                     // (1 - beta1^t is never 0: an all-zero group = "no update this replay")
        const long long e = i * 8;
        float pv[8], gv[8], mv[8], vv[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 a = *reinterpret_cast<const float4*>(p + e + 4 * q), b = *reinterpret_cast<const float4*>(g + e + 4 * q);
            const float4 c = *reinterpret_cast<const float4*>(m + e + 4 * q), d = *reinterpret_cast<const float4*>(v + e + 4 * q);
            pv[4 * q] = a.x; pv[4 * q + 1] = a.y; pv[4 * q + 2] = a.z; pv[4 * q + 3] = a.w;
            gv[4 * q] = b.x; gv[4 * q + 1] = b.y; gv[4 * q + 2] = b.z; gv[4 * q + 3] = b.w;
            mv[4 * q] = c.x; mv[4 * q + 1] = c.y; mv[4 * q + 2] = c.z; mv[4 * q + 3] = c.w;
            vv[4 * q] = d.x; vv[4 * q + 1] = d.y; vv[4 * q + 2] = d.z; vv[4 * q + 3] = d.w;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float gr = gv[k] * h.grad_scale;
            float x = pv[k] * (1.0f - h.lr * h.weight_decay);
            mv[k] = h.beta1 * mv[k] + (1.0f - h.beta1) * gr;
            vv[k] = h.beta2 * vv[k] + (1.0f - h.beta2) * gr * gr;
            const float denom = sqrtf(vv[k]) / h.sqrt_bias_c2 + h.eps;
            x -= (h.lr / h.bias_c1) * (mv[k] / denom);
            pv[k] = x;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            *reinterpret_cast<float4*>(p + e + 4 * q) = make_float4(pv[4 * q], pv[4 * q + 1], pv[4 * q + 2], pv[4 * q + 3]);
            *reinterpret_cast<float4*>(m + e + 4 * q) = make_float4(mv[4 * q], mv[4 * q + 1], mv[4 * q + 2], mv[4 * q + 3]);
            *reinterpret_cast<float4*>(v + e + 4 * q) = make_float4(vv[4 * q], vv[4 * q + 1], vv[4 * q + 2], vv[4 * q + 3]);
        }
        if (shadow)
            *reinterpret_cast<uint4*>(shadow + e) = make_uint4(pack_bf2(pv[0], pv[1]), pack_bf2(pv[2], pv[3]),
                                                              pack_bf2(pv[4], pv[5]), pack_bf2(pv[6], pv[7]));
        if (ema) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float4 a = *reinterpret_cast<const float4*>(ema + e + 4 * q);
                a.x = ema_decay * a.x + (1.0f - ema_decay) * pv[4 * q];
                a.y = ema_decay * a.y + (1.0f - ema_decay) * pv[4 * q + 1];
                a.z = ema_decay * a.z + (1.0f - ema_decay) * pv[4 * q + 2];
                a.w = ema_decay * a.w + (1.0f - ema_decay) * pv[4 * q + 3];
                *reinterpret_cast<float4*>(ema + e + 4 * q) = a;
            }
        }
    }
}

}  // namespace

static int adamw_launch(float* p, const float* g, float* m, float* v, void* shadow, float* ema, float ema_decay,
                        const uint8_t* group_of_8, const vr_adamw_group* groups, bool on_device, int32_t n_groups, int64_t n,
                        vr_stream_t stream, int32_t max_blocks = 0) {
    if (!p || !g || !m || !v || !group_of_8 || !groups || n <= 0 || n_groups <= 0) return VR_EINVAL;
    if (n_groups > VR_ADAMW_MAX_GROUPS) return VR_EUNSUPPORTED;
    if (n % 8 || ((uintptr_t)p & 15) || ((uintptr_t)g & 15) || ((uintptr_t)m & 15) || ((uintptr_t)v & 15) ||
        (shadow && ((uintptr_t)shadow & 15)) || (ema && ((uintptr_t)ema & 15)))
        return VR_EALIGN;
    Groups gs = {};
    if (!on_device)
        for (int i = 0; i < n_groups; ++i) gs.g[i] = groups[i];
    const long long n8 = n / 8;
    long long blocks = (n8 + 255) / 256;
    const long long cap = max_blocks > 0 ? max_blocks : 8192;
    if (blocks > cap) blocks = cap;
    if (on_device)
        hipLaunchKernelGGL(adamw_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)shadow,
                           ema, ema_decay, group_of_8, gs, groups, n8);
    else
        hipLaunchKernelGGL(adamw_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)shadow,
                           ema, ema_decay, group_of_8, gs, nullptr, n8);
    VR_CHECK_LAUNCH();
    return VR_OK;
}

extern "C" int vr_adamw_flat(float* p, const float* g, float* m, float* v, void* shadow, float* ema, float ema_decay,
                             const uint8_t* group_of_8, const vr_adamw_group* groups, int32_t n_groups, int64_t n,
                             vr_stream_t stream) {
    return adamw_launch(p, g, m, v, shadow, ema, ema_decay, group_of_8, groups, false, n_groups, n, stream);
}

extern "C" int vr_adamw_flat_dev(float* p, const float* g, float* m, float* v, void* shadow, float* ema, float ema_decay,
                                 const uint8_t* group_of_8, const vr_adamw_group* groups_dev, int32_t n_groups, int64_t n,
                                 vr_stream_t stream) {
    return adamw_launch(p, g, m, v, shadow, ema, ema_decay, group_of_8, groups_dev, true, n_groups, n, stream);
}

// The same launch capped at `max_blocks` resident workgroups (grid-stride): an update of a finished arena range that runs on the
// weight gradients' stream beside the rest of the backward must not take the chip (engine.GraphedTrainStep, VITRES_OPT_OVERLAP).
extern "C" int vr_adamw_flat_dev_capped(float* p, const float* g, float* m, float* v, void* shadow, float* ema, float ema_decay,
                                        const uint8_t* group_of_8, const vr_adamw_group* groups_dev, int32_t n_groups, int64_t n,
                                        int32_t max_blocks, vr_stream_t stream) {
    if (max_blocks < 0) return VR_EINVAL;
    return adamw_launch(p, g, m, v, shadow, ema, ema_decay, group_of_8, groups_dev, true, n_groups, n, stream, max_blocks);
}
