// Masked LayerNorm forward / backward for gfx950 (reference nets/masked_layer_norm.py:23-50,55-88,113-125).
// One wave64 per token row, float4 loads, shuffle reductions; HBM-bound by design (one read of x, one
// write of y; backward: read dy,x once, write dx once).  The prefix mask of each sample is an int keep
// count: statistics and outputs use only channels c < keep (channels beyond are exactly zero upstream).
#include "common.h"
#include <cstdlib>
#include <type_traits>
#include "../../include/vitres_hip.h"

namespace {

constexpr int MAXV_LIMIT = 8;  // float4 per lane -> C <= 2048 (kernels are templated on the actual count)

// A wave owns RU consecutive rows (4 / 2 for <= 2 / <= 4 float4 per lane when M fills the chip that way, else 1): all their loads are issued before the
// first reduction, so that a wave has 2 - 5 KB in flight instead of one row's (C = 320 fp32 rows at batch 256: 36 -> 27 us).
template <typename TO, int MAXV, int RU>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, TO* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     const int* __restrict__ keep, int M, int C, int rps, float eps, int xcd) {
    const int lane = threadIdx.x & 63;
    const int m0 = (xcd_block((int)blockIdx.x, (int)gridDim.x, xcd) * 4 + (threadIdx.x >> 6)) * RU;
    if (m0 >= M) return;
    float4 v[RU][MAXV];
    int kcs[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
        const int m = min(m0 + u, M - 1);                                       // (rows past M: loaded again, never stored)
        kcs[u] = keep ? keep[m / rps] : C;
        const float* xr = x + (long long)m * C;
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            const int c = (lane + 64 * j) * 4;
            v[u][j] = *reinterpret_cast<const float4*>(xr + (c < C ? c : 0));   // clamped, never branched around
        }
    }
    float4 ww[MAXV], bb[MAXV];
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = (lane + 64 * j) * 4;
        ww[j] = *reinterpret_cast<const float4*>(w + (c < C ? c : 0));
        bb[j] = *reinterpret_cast<const float4*>(b + (c < C ? c : 0));
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
        const int m = m0 + u;
        if (m >= M) break;
        const int kc = kcs[u];
        float s = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            const int c = (lane + 64 * j) * 4;
            if (c >= C) v[u][j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < C) {
                if (c + 0 >= kc) v[u][j].x = 0.f;
                if (c + 1 >= kc) v[u][j].y = 0.f;
                if (c + 2 >= kc) v[u][j].z = 0.f;
                if (c + 3 >= kc) v[u][j].w = 0.f;
                s += v[u][j].x + v[u][j].y + v[u][j].z + v[u][j].w;
                s2 += v[u][j].x * v[u][j].x + v[u][j].y * v[u][j].y + v[u][j].z * v[u][j].z + v[u][j].w * v[u][j].w;
            }
        }
        s = wave_sum(s);
        const float inv_n = kc > 0 ? 1.0f / (float)kc : 0.f;
        const float mu = s * inv_n;
        float var;
        if (keep) {
            // masked path: var = E[x^2]/p - mu^2  (masked_layer_norm.py:38-40)
            s2 = wave_sum(s2);
            var = s2 * inv_n - mu * mu;
        } else {
            // plain F.layer_norm path (:118-122): two-pass variance
            float d2 = 0.f;
#pragma unroll
            for (int j = 0; j < MAXV; ++j) {
                const int c = (lane + 64 * j) * 4;
                if (c < C) {
                    const float a = v[u][j].x - mu, b1 = v[u][j].y - mu, c1 = v[u][j].z - mu, d1 = v[u][j].w - mu;
                    d2 += a * a + b1 * b1 + c1 * c1 + d1 * d1;
                }
            }
            var = wave_sum(d2) * inv_n;
        }
        const float rs = 1.0f / sqrtf(var + eps);
        if (lane == 0) {
            mean[m] = mu;
            rstd[m] = rs;
        }
        TO* yr = y + (long long)m * C;
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            const int c = (lane + 64 * j) * 4;
            if (c < C) {
                float o0 = (c + 0 < kc) ? ww[j].x * ((v[u][j].x - mu) * rs) + bb[j].x : 0.f;
                float o1 = (c + 1 < kc) ? ww[j].y * ((v[u][j].y - mu) * rs) + bb[j].y : 0.f;
                float o2 = (c + 2 < kc) ? ww[j].z * ((v[u][j].z - mu) * rs) + bb[j].z : 0.f;
                float o3 = (c + 3 < kc) ? ww[j].w * ((v[u][j].w - mu) * rs) + bb[j].w : 0.f;
                if constexpr (sizeof(TO) == 4) {
                    *reinterpret_cast<float4*>(yr + c) = make_float4(o0, o1, o2, o3);
                } else {
                    *reinterpret_cast<uint2*>(yr + c) = make_uint2(pack_bf2(o0, o1), pack_bf2(o2, o3));
                }
            }
        }
    }
}

template <typename TI> __device__ __forceinline__ float4 load4(const TI* p);
template <> __device__ __forceinline__ float4 load4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 load4<bf16_t>(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}

// rows per workgroup are chosen per launch: more rows = fewer dgamma/dbeta atomics, fewer rows = more workgroups

template <typename TI, int MAXV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TI* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ w, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const int* __restrict__ keep,
                                                     const float* __restrict__ dx_in, float* __restrict__ dx_out,
                                                     float* __restrict__ dw, float* __restrict__ db, TI* __restrict__ gt_out,
                                                     const float* __restrict__ gt_scale, const int* __restrict__ gt_keep,
                                                     int M, int C, int rps, int BWD_ROWS, int copies, int xcd) {
    __shared__ float red[2][4][64 * 4];  // [dw|db][wave][lane*4+e], reused per j
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 gw[MAXV], gb[MAXV], ww[MAXV];
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        gw[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        gb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c = (lane + 64 * j) * 4;
        ww[j] = (c < C) ? *reinterpret_cast<const float4*>(w + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int mbeg = xcd_block((int)blockIdx.x, (int)gridDim.x, xcd) * BWD_ROWS;
    constexpr int RU = MAXV == 1 ? 4 : (MAXV <= 2 ? 2 : 1);           // rows in flight per wave: all their loads are issued before any reduction
    for (int rr = wave; rr < BWD_ROWS; rr += 4 * RU) {
        float4 gv[RU][MAXV], xv[RU][MAXV], rv[RU][MAXV];
        int kc[RU], k2[RU];
        float mu[RU], rs[RU], sc2[RU];
        bool rok[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int m = mbeg + rr + 4 * u;
            rok[u] = (rr + 4 * u < BWD_ROWS) && (m < M);
            const int mc = rok[u] ? m : 0;
            kc[u] = keep ? keep[mc / rps] : C;
            k2[u] = (gt_out && gt_keep) ? gt_keep[mc / rps] : C;
            sc2[u] = (gt_out && gt_scale) ? gt_scale[mc / rps] : 1.0f;
            mu[u] = mean[mc];
            rs[u] = rstd[mc];
#pragma unroll
            for (int j = 0; j < MAXV; ++j) {
                const int c = (lane + 64 * j) * 4;
                const int cc = c < C ? c : 0;                        // clamped: loads are never branched around
                gv[u][j] = load4<TI>(dy + (long long)mc * C + cc);
                xv[u][j] = *reinterpret_cast<const float4*>(x + (long long)mc * C + cc);
                rv[u][j] = dx_in ? *reinterpret_cast<const float4*>(dx_in + (long long)mc * C + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const float inv_n = kc[u] > 0 ? 1.0f / (float)kc[u] : 0.f;
            float4 g[MAXV], z[MAXV];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < MAXV; ++j) {
                const int c = (lane + 64 * j) * 4;
                float4 a = gv[u][j], xx = xv[u][j];
                const bool in = (c < C) && rok[u];
                if (!(in && c + 0 < kc[u])) { a.x = 0.f; xx.x = mu[u]; }
                if (!(in && c + 1 < kc[u])) { a.y = 0.f; xx.y = mu[u]; }
                if (!(in && c + 2 < kc[u])) { a.z = 0.f; xx.z = mu[u]; }
                if (!(in && c + 3 < kc[u])) { a.w = 0.f; xx.w = mu[u]; }
                z[j] = make_float4((xx.x - mu[u]) * rs[u], (xx.y - mu[u]) * rs[u], (xx.z - mu[u]) * rs[u], (xx.w - mu[u]) * rs[u]);
                gw[j].x += a.x * z[j].x; gw[j].y += a.y * z[j].y; gw[j].z += a.z * z[j].z; gw[j].w += a.w * z[j].w;
                gb[j].x += a.x; gb[j].y += a.y; gb[j].z += a.z; gb[j].w += a.w;
                g[j] = make_float4(a.x * ww[j].x, a.y * ww[j].y, a.z * ww[j].z, a.w * ww[j].w);  // dz
                s1 += g[j].x + g[j].y + g[j].z + g[j].w;
                s2 += g[j].x * z[j].x + g[j].y * z[j].y + g[j].z * z[j].z + g[j].w * z[j].w;
            }
            s1 = wave_sum(s1) * inv_n;
            s2 = wave_sum(s2) * inv_n;
            const int m = mbeg + rr + 4 * u;
#pragma unroll
            for (int j = 0; j < MAXV; ++j) {
                const int c = (lane + 64 * j) * 4;
                if (c < C && rok[u]) {
                    // channels >= keep get exactly 0 (also for the pass-through residual gradient): the reference lets
                    // garbage flow there until the stage's `x * mask` kills it (nets/channel_drop.py:82); same param grads.
                    const float4 r = rv[u][j];
                    float4 o;
                    o.x = (c + 0 < kc[u]) ? (g[j].x - (s1 + z[j].x * s2)) * rs[u] + r.x : 0.f;
                    o.y = (c + 1 < kc[u]) ? (g[j].y - (s1 + z[j].y * s2)) * rs[u] + r.y : 0.f;
                    o.z = (c + 2 < kc[u]) ? (g[j].z - (s1 + z[j].z * s2)) * rs[u] + r.z : 0.f;
                    o.w = (c + 3 < kc[u]) ? (g[j].w - (s1 + z[j].w * s2)) * rs[u] + r.w : 0.f;
                    *reinterpret_cast<float4*>(dx_out + (long long)m * C + c) = o;
                    if (gt_out) {       // the gradient entering the next backward branch: DropPath scale, prefix mask, cast
                        float4 t;
                        t.x = (c + 0 < k2[u]) ? o.x * sc2[u] : 0.f;
                        t.y = (c + 1 < k2[u]) ? o.y * sc2[u] : 0.f;
                        t.z = (c + 2 < k2[u]) ? o.z * sc2[u] : 0.f;
                        t.w = (c + 3 < k2[u]) ? o.w * sc2[u] : 0.f;
                        if constexpr (sizeof(TI) == 4) *reinterpret_cast<float4*>(gt_out + (long long)m * C + c) = t;
                        else *reinterpret_cast<uint2*>(gt_out + (long long)m * C + c) = make_uint2(pack_bf2(t.x, t.y), pack_bf2(t.z, t.w));
                    }
                }
            }
        }
    }
    // cross-wave reduction of dgamma / dbeta, one float4 column group at a time
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        if (64 * 4 * j >= C) break;
        __syncthreads();
        *reinterpret_cast<float4*>(&red[0][wave][lane * 4]) = gw[j];
        *reinterpret_cast<float4*>(&red[1][wave][lane * 4]) = gb[j];
        __syncthreads();
        // 256 threads: thread t handles element t of the 256-wide group, for dw; same for db
        const int e = threadIdx.x;
        const int c = 64 * 4 * j + e;
        if (c < C) {
            const float a = red[0][0][e] + red[0][1][e] + red[0][2][e] + red[0][3][e];
            const float bsum = red[1][0][e] + red[1][1][e] + red[1][2][e] + red[1][3][e];
            // `copies` rows of partial sums (vr_ln_grad_reduce folds them): the atomics of ~500..2000 workgroups on the same
            // 2C addresses were a third of this kernel's time
            const long long row = (long long)(blockIdx.x % (unsigned)copies) * C;
            atomicAdd(dw + row + c, a);
            atomicAdd(db + row + c, bsum);
        }
    }
}

struct GradSlots {
    static constexpr int MAX = 32;
    vr_ln_grad_slot s[MAX];
};

// dw[c] += sum_k part_w[k][c] (same for db) and the partial rows go back to zero for the next backward
__global__ __launch_bounds__(256) void ln_grad_reduce_kernel(GradSlots slots, int copies) {
    const vr_ln_grad_slot& sl = slots.s[blockIdx.y];
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int C = sl.C;
    if (e >= 2 * C) return;
    float* part = e < C ? sl.part_w + e : sl.part_b + (e - C);
    float* dst = e < C ? sl.dw + e : sl.db + (e - C);
    // sixteen independent loads in flight per round (one dependent load per partial row made this a 11 us kernel)
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    int k = 0;
    for (; k + 16 <= copies; k += 16) {
        float t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = part[(long long)(k + u) * C];
#pragma unroll
        for (int u = 0; u < 16; ++u) part[(long long)(k + u) * C] = 0.f;
#pragma unroll
        for (int u = 0; u < 16; u += 4) { acc0 += t[u]; acc1 += t[u + 1]; acc2 += t[u + 2]; acc3 += t[u + 3]; }
    }
    for (; k < copies; ++k) {
        acc0 += part[(long long)k * C];
        part[(long long)k * C] = 0.f;
    }
    *dst += (acc0 + acc1) + (acc2 + acc3);
}

}  // namespace

extern "C" int vr_ln_fwd(const float* x, const float* w, const float* b, void* y, float* mean, float* rstd,
                         const int32_t* keep, int32_t M, int32_t C, int32_t rows_per_sample, float eps,
                         int32_t out_dtype, vr_stream_t stream) {
    if (!x || !w || !b || !y || !mean || !rstd || M <= 0 || C <= 0) return VR_EINVAL;
    if (C % 4 || C > 64 * 4 * MAXV_LIMIT) return VR_EUNSUPPORTED;
    if (out_dtype != VR_F32 && out_dtype != VR_BF16) return VR_EUNSUPPORTED;
    if (rows_per_sample <= 0) rows_per_sample = M;
    constexpr int knob_xcd = 1;                     // XCD-contiguous row ranges (common.h xcd_block), forward and backward
    const int nv = (C + 255) / 256;
    int ru = nv <= 2 ? 4 : (nv <= 4 ? 2 : 1);
    while (ru > 1 && M < 4 * ru * 1024) ru >>= 1;                       // (fewer than ~4 workgroups per CU: one row per wave)
    dim3 grid((M + 4 * ru - 1) / (4 * ru));
#define VR_LN_FWD2(NV, R)                                                                                              \
    if (out_dtype == VR_F32)                                                                                           \
        hipLaunchKernelGGL((ln_fwd_kernel<float, NV, R>), grid, dim3(256), 0, (hipStream_t)stream, x, w, b, (float*)y, mean, \
                           rstd, keep, M, C, rows_per_sample, eps, knob_xcd);                                                    \
    else                                                                                                               \
        hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, NV, R>), grid, dim3(256), 0, (hipStream_t)stream, x, w, b, (bf16_t*)y,  \
                           mean, rstd, keep, M, C, rows_per_sample, eps, knob_xcd);
#define VR_LN_FWD(NV)                                                                                                  \
    if (ru == 4 && NV <= 2) { VR_LN_FWD2(NV, (NV <= 2 ? 4 : 1)) }                                                      \
    else if (ru >= 2 && NV <= 4) { VR_LN_FWD2(NV, (NV <= 4 ? 2 : 1)) }                                                 \
    else { VR_LN_FWD2(NV, 1) }
    switch (nv) {
        case 1: VR_LN_FWD(1) break;
        case 2: VR_LN_FWD(2) break;
        case 3: VR_LN_FWD(3) break;
        case 4: VR_LN_FWD(4) break;
        case 5: VR_LN_FWD(5) break;
        default: VR_LN_FWD(8) break;
    }
#undef VR_LN_FWD2
#undef VR_LN_FWD
    VR_CHECK_LAUNCH();
    return VR_OK;
}

extern "C" int vr_ln_bwd(const void* dy, const float* x, const float* w, const float* mean, const float* rstd,
                         const int32_t* keep, const float* dx_in, float* dx_out, float* dw, float* db, void* gt_out,
                         const float* gt_scale, const int32_t* gt_keep, int32_t M, int32_t C, int32_t rows_per_sample,
                         int32_t dy_dtype, int32_t grad_copies, vr_stream_t stream) {
    if (!dy || !x || !w || !mean || !rstd || !dx_out || !dw || !db || M <= 0 || C <= 0 || grad_copies < 0) return VR_EINVAL;
    const int copies = grad_copies > 1 ? grad_copies : 1;
    if (C % 4 || C > 64 * 4 * MAXV_LIMIT) return VR_EUNSUPPORTED;
    if (dy_dtype != VR_F32 && dy_dtype != VR_BF16) return VR_EUNSUPPORTED;
    if (rows_per_sample <= 0) rows_per_sample = M;
    constexpr int knob_xcd = 1;
    // rows per workgroup: with partial rows the atomics no longer collide, so many small workgroups (better tail) win
    const int BWD_ROWS = copies > 1 ? (M >= 8192 ? 16 : (M >= 2048 ? 8 : 4)) : (M >= 32768 ? 64 : (M >= 8192 ? 32 : (M >= 2048 ? 8 : 4)));
    dim3 grid((M + BWD_ROWS - 1) / BWD_ROWS);
    const int nv = (C + 255) / 256;
#define VR_LN_BWD(NV)                                                                                                  \
    if (dy_dtype == VR_F32)                                                                                            \
        hipLaunchKernelGGL((ln_bwd_kernel<float, NV>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)dy, x, w, \
                           mean, rstd, keep, dx_in, dx_out, dw, db, (float*)gt_out, gt_scale, gt_keep, M, C, rows_per_sample, BWD_ROWS, copies, knob_xcd); \
    else                                                                                                               \
        hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, NV>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, x, \
                           w, mean, rstd, keep, dx_in, dx_out, dw, db, (bf16_t*)gt_out, gt_scale, gt_keep, M, C, rows_per_sample, BWD_ROWS, copies, knob_xcd);
    switch (nv) {
        case 1: VR_LN_BWD(1) break;
        case 2: VR_LN_BWD(2) break;
        case 3: VR_LN_BWD(3) break;
        case 4: VR_LN_BWD(4) break;
        case 5: VR_LN_BWD(5) break;
        default: VR_LN_BWD(8) break;
    }
#undef VR_LN_BWD
    VR_CHECK_LAUNCH();
    return VR_OK;
}

extern "C" int vr_ln_grad_reduce(const vr_ln_grad_slot* slots, int32_t count, int32_t copies, vr_stream_t stream) {
    if (count < 0 || copies < 1 || (count > 0 && !slots)) return VR_EINVAL;
    for (int i = 0; i < count; ++i)
        if (!slots[i].part_w || !slots[i].part_b || !slots[i].dw || !slots[i].db || slots[i].C <= 0) return VR_EINVAL;
    for (int first = 0; first < count; first += GradSlots::MAX) {
        GradSlots g;
        const int n = count - first < GradSlots::MAX ? count - first : GradSlots::MAX;
        int cmax = 0;
        for (int i = 0; i < n; ++i) {
            g.s[i] = slots[first + i];
            cmax = g.s[i].C > cmax ? g.s[i].C : cmax;
        }
        hipLaunchKernelGGL(ln_grad_reduce_kernel, dim3((2 * cmax + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, g, copies);
        VR_CHECK_LAUNCH();
    }
    return VR_OK;
}
