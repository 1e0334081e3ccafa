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
                                                     const int* __restrict__ keep, int M, int C, int rps, float eps) {
    const int lane = threadIdx.x & 63;
    const int m0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RU;
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
                                                     int M, int C, int rps, int BWD_ROWS, int copies) {
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
    const int mbeg = blockIdx.x * BWD_ROWS;
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

// Lean form (round 4): the LayerNorm backward of the 512 - 1280 wide stages runs beside the weight-gradient group of the block it
// closes, whose workgroups hold half of every SIMD's registers for their whole life -- at 120 - 164 VGPRs the kernel above then
// fits 1 - 2 waves per SIMD instead of 3 - 4 and takes 46 - 64 us instead of 11 - 16 (profiles/r03_a_step_timeline.txt: every
// co-running launch ends with the weight-gradient group it started beside).  Here nothing row-invariant lives in registers: the
// dgamma / dbeta column sums are accumulated with LDS atomics (ds_add_f32, 2 C per row: ~10 % of a row's load latency), gamma
// is re-read from LDS per row, dy stays packed (bf16), z and dz are recomputed after the two row reductions instead of kept
// across them: <= 64 VGPRs at C = 1024, so 4+ waves per SIMD fit beside two 128-register GEMM waves.
template <typename TI, int MAXV>
__global__ __launch_bounds__(256) void ln_bwd_lean_kernel(const TI* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ w, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const int* __restrict__ keep,
                                                          const float* __restrict__ dx_in, float* __restrict__ dx_out,
                                                          float* __restrict__ dw, float* __restrict__ db, TI* __restrict__ gt_out,
                                                          const float* __restrict__ gt_scale, const int* __restrict__ gt_keep,
                                                          int M, int C, int rps, int BWD_ROWS, int copies) {
    __shared__ __attribute__((aligned(16))) float sw[MAXV * 256], sdw[MAXV * 256], sdb[MAXV * 256];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // rows are wave-uniform: scalar row bases, 32-bit lane offsets
    for (int c = threadIdx.x; c < MAXV * 256; c += 256) {
        sw[c] = c < C ? w[c] : 0.f;
        sdw[c] = 0.f;
        sdb[c] = 0.f;
    }
    __syncthreads();
    const int mbeg = blockIdx.x * BWD_ROWS;
    for (int rr = wave; rr < BWD_ROWS; rr += 4) {
        const int m = mbeg + rr;
        if (m >= M) break;
        const int kc = keep ? keep[m / rps] : C;
        const float mu = mean[m], rs = rstd[m];
        TI const* dyr = dy + (long long)m * C;
        const float* xr = x + (long long)m * C;
        float4 xv[MAXV], rv[MAXV];
        typename std::conditional<sizeof(TI) == 4, float4, uint2>::type gq[MAXV];        // dy as loaded (bf16 stays packed)
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            const int c = (lane + 64 * j) * 4;
            const int cc = c < C ? c : 0;                                                // clamped: loads are never branched around
            if constexpr (sizeof(TI) == 4) gq[j] = *reinterpret_cast<const float4*>(dyr + cc);
            else gq[j] = *reinterpret_cast<const uint2*>(dyr + cc);
            xv[j] = *reinterpret_cast<const float4*>(xr + cc);
            rv[j] = dx_in ? *reinterpret_cast<const float4*>(dx_in + (long long)m * C + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        auto dyv = [&](int j) -> float4 {
            if constexpr (sizeof(TI) == 4) return gq[j];
            else return make_float4(__uint_as_float(gq[j].x << 16), __uint_as_float(gq[j].x & 0xffff0000u),
                                    __uint_as_float(gq[j].y << 16), __uint_as_float(gq[j].y & 0xffff0000u));
        };
        const float inv_n = kc > 0 ? 1.0f / (float)kc : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            const int c = (lane + 64 * j) * 4;
            if (c < C) {
                float4 a = dyv(j);
                const float4 xx = xv[j];
                if (!(c + 0 < kc)) a.x = 0.f;
                if (!(c + 1 < kc)) a.y = 0.f;
                if (!(c + 2 < kc)) a.z = 0.f;
                if (!(c + 3 < kc)) a.w = 0.f;
                const float4 z = make_float4((xx.x - mu) * rs, (xx.y - mu) * rs, (xx.z - mu) * rs, (xx.w - mu) * rs);
                const float4 wv = *reinterpret_cast<const float4*>(&sw[c]);
                const float4 g = make_float4(a.x * wv.x, a.y * wv.y, a.z * wv.z, a.w * wv.w);
                s1 += g.x + g.y + g.z + g.w;
                s2 += g.x * z.x + g.y * z.y + g.z * z.z + g.w * z.w;
                // (masked channels: a == 0 -> both products vanish whatever z is)
                atomicAdd(&sdw[c + 0], a.x * z.x); atomicAdd(&sdw[c + 1], a.y * z.y);
                atomicAdd(&sdw[c + 2], a.z * z.z); atomicAdd(&sdw[c + 3], a.w * z.w);
                atomicAdd(&sdb[c + 0], a.x); atomicAdd(&sdb[c + 1], a.y); atomicAdd(&sdb[c + 2], a.z); atomicAdd(&sdb[c + 3], a.w);
            }
        }
        s1 = wave_sum(s1) * inv_n;
        s2 = wave_sum(s2) * inv_n;
        // the second pass RECOMPUTES dz and z from the loaded values: without this fence the compiler keeps the first pass's
        // products alive across the reductions (common subexpressions) and the kernel is back at 170 registers
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            asm volatile("" : "+v"(xv[j].x), "+v"(xv[j].y), "+v"(xv[j].z), "+v"(xv[j].w));
            if constexpr (sizeof(TI) == 4) asm volatile("" : "+v"(gq[j].x), "+v"(gq[j].y), "+v"(gq[j].z), "+v"(gq[j].w));
            else asm volatile("" : "+v"(gq[j].x), "+v"(gq[j].y));
        }
        const int k2 = (gt_out && gt_keep) ? gt_keep[m / rps] : C;
        const float sc2 = (gt_out && gt_scale) ? gt_scale[m / rps] : 1.0f;
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            const int c = (lane + 64 * j) * 4;
            if (c < C) {
                const float4 a = dyv(j);
                const float4 xx = xv[j];
                const float4 wv = *reinterpret_cast<const float4*>(&sw[c]);
                const float4 r = rv[j];
                float4 o;
                o.x = (c + 0 < kc) ? (a.x * wv.x - (s1 + (xx.x - mu) * rs * s2)) * rs + r.x : 0.f;
                o.y = (c + 1 < kc) ? (a.y * wv.y - (s1 + (xx.y - mu) * rs * s2)) * rs + r.y : 0.f;
                o.z = (c + 2 < kc) ? (a.z * wv.z - (s1 + (xx.z - mu) * rs * s2)) * rs + r.z : 0.f;
                o.w = (c + 3 < kc) ? (a.w * wv.w - (s1 + (xx.w - mu) * rs * s2)) * rs + r.w : 0.f;
                *reinterpret_cast<float4*>(dx_out + (long long)m * C + c) = o;
                if (gt_out) {
                    float4 t;
                    t.x = (c + 0 < k2) ? o.x * sc2 : 0.f;
                    t.y = (c + 1 < k2) ? o.y * sc2 : 0.f;
                    t.z = (c + 2 < k2) ? o.z * sc2 : 0.f;
                    t.w = (c + 3 < k2) ? o.w * sc2 : 0.f;
                    if constexpr (sizeof(TI) == 4) *reinterpret_cast<float4*>(gt_out + (long long)m * C + c) = t;
                    else *reinterpret_cast<uint2*>(gt_out + (long long)m * C + c) = make_uint2(pack_bf2(t.x, t.y), pack_bf2(t.z, t.w));
                }
            }
        }
    }
    __syncthreads();
    const long long row = (long long)(blockIdx.x % (unsigned)copies) * C;
    for (int c = threadIdx.x; c < C; c += 256) {
        atomicAdd(dw + row + c, sdw[c]);
        atomicAdd(db + row + c, sdb[c]);
    }
}

// Column-owned form (round 4, the default): a thread owns 4 J channels of the row for the workgroup's whole life, so gamma and the
// dgamma / dbeta column sums are 12 J registers and need neither atomics nor a cross-wave pass per column group; a row is spread
// over WPR waves (C = 512: two) and RG rows run side by side in a workgroup; R rows per thread are in flight per round (dy packed,
// x, and the pass-through gradient requested before the row reduction).  The 2 R row sums are reduced by a transposing butterfly
// (10 shuffles for 8 values instead of 48) and exchanged between the waves of a row through 256 B of LDS.  <= 64 VGPRs at J = 1:
// the kernel keeps 4+ waves per SIMD beside two 128-register GEMM workgroups (the wave-per-row kernel above: 119 - 164 VGPRs, one
// or two waves per SIMD next to the weight-gradient group it always starts beside -- 31 us in the step for 11 - 16 us alone).
template <typename TI, int J, int R>
__global__ __launch_bounds__(256) void ln_bwd_col_kernel(const TI* __restrict__ dy, const float* __restrict__ x,
                                                         const float* __restrict__ w, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, const int* __restrict__ keep,
                                                         const float* __restrict__ dx_in, float* __restrict__ dx_out,
                                                         float* __restrict__ dw, float* __restrict__ db, TI* __restrict__ gt_out,
                                                         const float* __restrict__ gt_scale, const int* __restrict__ gt_keep,
                                                         int M, int C, int rps, int rows_per_wg, int copies, int WPR) {
    static_assert(R == 4 || R == 2, "row sums are reduced 2 R at a time");
    __shared__ __attribute__((aligned(16))) float red[2][4][2 * R];
    __shared__ __attribute__((aligned(16))) float acc[3 * 2 * 256 * 4];           // column sums of row groups 1.. (J == 1 only)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const int RG = nw / WPR;
    const int rg = wave / WPR, wr = wave - rg * WPR;
    int col[J];
    float4 ww[J], gw[J], gb[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        col[j] = ((wr * 64 + lane) + 64 * WPR * j) * 4;
        ww[j] = col[j] < C ? *reinterpret_cast<const float4*>(w + col[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
        gw[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        gb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int mbeg = blockIdx.x * rows_per_wg;
    const int mend = min(M, mbeg + rows_per_wg);
    int par = 0;
    for (int m0 = mbeg; m0 < mend; m0 += RG * R, par ^= 1) {
        typename std::conditional<sizeof(TI) == 4, float4, uint2>::type gq[R][J];     // dy as loaded (bf16 stays packed)
        float4 xv[R][J], rv[R][J];
        int kc[R], mrow[R];
        float mu[R], rs[R];
        bool rok[R];
        // rows are wave-uniform: scalar row bases + 32-bit lane offsets (no 64-bit address per load in vector registers)
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int m = __builtin_amdgcn_readfirstlane(m0 + u * RG + rg);
            rok[u] = m < mend;
            mrow[u] = rok[u] ? m : mbeg;
            kc[u] = rok[u] ? (keep ? keep[mrow[u] / rps] : C) : 0;                     // (a row past the end: everything masked)
            mu[u] = mean[mrow[u]];
            rs[u] = rstd[mrow[u]];
            const TI* dyr = dy + (long long)mrow[u] * C;
            const float* xr = x + (long long)mrow[u] * C;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const unsigned cc = col[j] < C ? (unsigned)col[j] : 0u;                                // clamped: loads are never branched around
                if constexpr (sizeof(TI) == 4) gq[u][j] = *reinterpret_cast<const float4*>(dyr + cc);
                else gq[u][j] = *reinterpret_cast<const uint2*>(dyr + cc);
                xv[u][j] = *reinterpret_cast<const float4*>(xr + cc);
            }
        }
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const float* rr = dx_in ? dx_in + (long long)mrow[u] * C : nullptr;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const unsigned cc = col[j] < C ? (unsigned)col[j] : 0u;
                rv[u][j] = dx_in ? *reinterpret_cast<const float4*>(rr + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        // masks are prefixes: nv = how many of this thread's four channels are kept; masked dy -> 0, masked x -> mu (z = 0), once
        int nv[R][J];
#pragma unroll
        for (int u = 0; u < R; ++u)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int lim = min(kc[u], C) - col[j];
                const int n = nv[u][j] = lim < 0 ? 0 : (lim > 4 ? 4 : lim);
                if constexpr (sizeof(TI) == 4) {
                    if (n < 1) gq[u][j].x = 0.f;
                    if (n < 2) gq[u][j].y = 0.f;
                    if (n < 3) gq[u][j].z = 0.f;
                    if (n < 4) gq[u][j].w = 0.f;
                } else {
                    gq[u][j].x &= (n < 1 ? 0u : 0xffffu) | (n < 2 ? 0u : 0xffff0000u);
                    gq[u][j].y &= (n < 3 ? 0u : 0xffffu) | (n < 4 ? 0u : 0xffff0000u);
                }
                if (n < 1) xv[u][j].x = mu[u];
                if (n < 2) xv[u][j].y = mu[u];
                if (n < 3) xv[u][j].z = mu[u];
                if (n < 4) xv[u][j].w = mu[u];
                __builtin_amdgcn_sched_barrier(0);           // one row at a time: interleaving the rows doubles the live registers
            }
        auto dyv = [&](int u, int j) -> float4 {
            if constexpr (sizeof(TI) == 4) return gq[u][j];
            else return make_float4(__uint_as_float(gq[u][j].x << 16), __uint_as_float(gq[u][j].x & 0xffff0000u),
                                    __uint_as_float(gq[u][j].y << 16), __uint_as_float(gq[u][j].y & 0xffff0000u));
        };
        auto zv = [&](int u, int j) -> float4 {
            const float4 xx = xv[u][j];
            return make_float4((xx.x - mu[u]) * rs[u], (xx.y - mu[u]) * rs[u], (xx.z - mu[u]) * rs[u], (xx.w - mu[u]) * rs[u]);
        };
        float ps[2 * R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const float4 a = dyv(u, j), z = zv(u, j);
                gw[j].x += a.x * z.x; gw[j].y += a.y * z.y; gw[j].z += a.z * z.z; gw[j].w += a.w * z.w;
                gb[j].x += a.x; gb[j].y += a.y; gb[j].z += a.z; gb[j].w += a.w;
                const float4 g = make_float4(a.x * ww[j].x, a.y * ww[j].y, a.z * ww[j].z, a.w * ww[j].w);
                s1 += g.x + g.y + g.z + g.w;
                s2 += g.x * z.x + g.y * z.y + g.z * z.z + g.w * z.w;
            }
            ps[u] = s1;
            ps[R + u] = s2;
            __builtin_amdgcn_sched_barrier(0);
        }
        // transposing butterfly: after log2(2R) halving steps a lane holds ONE of the 2R sums (over 2R lanes), then plain xor steps
        int idx = 0;
#pragma unroll
        for (int h = R, bit = 1; h >= 1; h >>= 1, bit <<= 1) {
            const bool up = (lane & bit) != 0;
#pragma unroll
            for (int i = 0; i < h; ++i) {
                const float send = up ? ps[i] : ps[i + h];
                const float kept = up ? ps[i + h] : ps[i];
                ps[i] = kept + __shfl_xor(send, bit, 64);
            }
            idx += up ? h : 0;
        }
        float tot = ps[0];
#pragma unroll
        for (int o = 2 * R; o < 64; o <<= 1) tot += __shfl_xor(tot, o, 64);
        if (lane < 2 * R) red[par][wave][idx] = tot;
        __syncthreads();
        // the second pass RECOMPUTES dy's and z's floats from the loaded registers: without this fence the compiler keeps the first
        // pass's values alive across the reduction (common subexpressions)
#pragma unroll
        for (int u = 0; u < R; ++u)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                asm volatile("" : "+v"(xv[u][j].x), "+v"(xv[u][j].y), "+v"(xv[u][j].z), "+v"(xv[u][j].w));
                if constexpr (sizeof(TI) == 4) asm volatile("" : "+v"(gq[u][j].x), "+v"(gq[u][j].y), "+v"(gq[u][j].z), "+v"(gq[u][j].w));
                else asm volatile("" : "+v"(gq[u][j].x), "+v"(gq[u][j].y));
            }
        float s1r[R], s2r[R];
#pragma unroll
        for (int u = 0; u < R; ++u) { s1r[u] = 0.f; s2r[u] = 0.f; }
#pragma unroll 1
        for (int q = 0; q < WPR; ++q) {
            const float* rr_ = red[par][rg * WPR + q];
#pragma unroll
            for (int u = 0; u < R; ++u) { s1r[u] += rr_[u]; s2r[u] += rr_[R + u]; }
        }
#pragma unroll
        for (int u = 0; u < R; ++u) {
            if (!rok[u]) continue;
            const int m = mrow[u];
            const float inv_n = kc[u] > 0 ? 1.0f / (float)kc[u] : 0.f;
            const float s1 = s1r[u] * inv_n, s2 = s2r[u] * inv_n;
            const int k2 = (gt_out && gt_keep) ? gt_keep[m / rps] : C;
            const float sc2 = (gt_out && gt_scale) ? gt_scale[m / rps] : 1.0f;
            float* dxr = dx_out + (long long)m * C;
            TI* gtr = gt_out ? gt_out + (long long)m * C : nullptr;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int c = col[j];
                const unsigned cu = (unsigned)c;
                if (c < C) {
                    const float4 a = dyv(u, j), z = zv(u, j), r = rv[u][j];
                    const int n = nv[u][j];
                    float4 o;
                    o.x = (n > 0) ? (a.x * ww[j].x - (s1 + z.x * s2)) * rs[u] + r.x : 0.f;
                    o.y = (n > 1) ? (a.y * ww[j].y - (s1 + z.y * s2)) * rs[u] + r.y : 0.f;
                    o.z = (n > 2) ? (a.z * ww[j].z - (s1 + z.z * s2)) * rs[u] + r.z : 0.f;
                    o.w = (n > 3) ? (a.w * ww[j].w - (s1 + z.w * s2)) * rs[u] + r.w : 0.f;
                    *reinterpret_cast<float4*>(dxr + cu) = o;
                    if (gt_out) {
                        float4 t;
                        t.x = (c + 0 < k2) ? o.x * sc2 : 0.f;
                        t.y = (c + 1 < k2) ? o.y * sc2 : 0.f;
                        t.z = (c + 2 < k2) ? o.z * sc2 : 0.f;
                        t.w = (c + 3 < k2) ? o.w * sc2 : 0.f;
                        if constexpr (sizeof(TI) == 4) *reinterpret_cast<float4*>(gtr + cu) = t;
                        else *reinterpret_cast<uint2*>(gtr + cu) = make_uint2(pack_bf2(t.x, t.y), pack_bf2(t.z, t.w));
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // column sums: row groups 1.. hand theirs to group 0 through LDS (J == 1 whenever RG > 1), group 0 adds them to its partial row
    if (RG > 1) {
        const int tcol = (wr * 64 + lane) * 4;                                        // < 256 * 4 / RG ... * WPR: within acc's row
        const int rowlen = WPR * 256;
        if (rg > 0) {
            *reinterpret_cast<float4*>(&acc[((rg - 1) * 2 + 0) * rowlen + tcol]) = gw[0];
            *reinterpret_cast<float4*>(&acc[((rg - 1) * 2 + 1) * rowlen + tcol]) = gb[0];
        }
        __syncthreads();
        if (rg == 0) {
            for (int q = 1; q < RG; ++q) {
                const float4 a = *reinterpret_cast<const float4*>(&acc[((q - 1) * 2 + 0) * rowlen + tcol]);
                const float4 b = *reinterpret_cast<const float4*>(&acc[((q - 1) * 2 + 1) * rowlen + tcol]);
                gw[0].x += a.x; gw[0].y += a.y; gw[0].z += a.z; gw[0].w += a.w;
                gb[0].x += b.x; gb[0].y += b.y; gb[0].z += b.z; gb[0].w += b.w;
            }
        }
    }
    if (rg == 0) {
        const long long row = (long long)(blockIdx.x % (unsigned)copies) * C;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int c = col[j];
            if (c < C) {
                atomicAdd(dw + row + c + 0, gw[j].x); atomicAdd(dw + row + c + 1, gw[j].y);
                atomicAdd(dw + row + c + 2, gw[j].z); atomicAdd(dw + row + c + 3, gw[j].w);
                atomicAdd(db + row + c + 0, gb[j].x); atomicAdd(db + row + c + 1, gb[j].y);
                atomicAdd(db + row + c + 2, gb[j].z); atomicAdd(db + row + c + 3, gb[j].w);
            }
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
    const int nv = (C + 255) / 256;
    int ru = nv <= 2 ? 4 : (nv <= 4 ? 2 : 1);
    while (ru > 1 && M < 4 * ru * 1024) ru >>= 1;                       // (fewer than ~4 workgroups per CU: one row per wave)
    dim3 grid((M + 4 * ru - 1) / (4 * ru));
#define VR_LN_FWD2(NV, R)                                                                                              \
    if (out_dtype == VR_F32)                                                                                           \
        hipLaunchKernelGGL((ln_fwd_kernel<float, NV, R>), grid, dim3(256), 0, (hipStream_t)stream, x, w, b, (float*)y, mean, \
                           rstd, keep, M, C, rows_per_sample, eps);                                                    \
    else                                                                                                               \
        hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, NV, R>), grid, dim3(256), 0, (hipStream_t)stream, x, w, b, (bf16_t*)y,  \
                           mean, rstd, keep, M, C, rows_per_sample, eps);
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
    // rows per workgroup: with partial rows the atomics no longer collide, so many small workgroups (better tail) win
    static const int knob_rows = std::getenv("VITRES_LN_BWD_ROWS") ? std::atoi(std::getenv("VITRES_LN_BWD_ROWS")) : 0;
    const int BWD_ROWS = knob_rows > 0 ? knob_rows
                         : copies > 1  ? (M >= 8192 ? 16 : (M >= 2048 ? 8 : 4))
                                       : (M >= 32768 ? 64 : (M >= 8192 ? 32 : (M >= 2048 ? 8 : 4)));
    dim3 grid((M + BWD_ROWS - 1) / BWD_ROWS);
    const int nv = (C + 255) / 256;
    // lean form: widths of the later stages (>= 384), where the kernel runs beside a weight-gradient group (VITRES_LN_BWD_LEAN: 0 never,
    // 1 C >= 384, 2 always)
    static const int knob_lean = std::getenv("VITRES_LN_BWD_LEAN") ? std::atoi(std::getenv("VITRES_LN_BWD_LEAN")) : 0;
    if (knob_lean == 3 || (knob_lean == 4 && C >= 384)) {
        // column-owned kernel: J float4 per thread, WPR waves per row, RG rows side by side, R rows per thread in flight
        const int J = C > 1024 ? 2 : 1;
        const int WPR = (C + 256 * J - 1) / (256 * J);
        const int RG = 4 / WPR >= 1 ? 4 / WPR : 1;
        static const int knob_r = std::getenv("VITRES_LN_BWD_R") ? std::atoi(std::getenv("VITRES_LN_BWD_R")) : 0;
        int R = (long long)M >= 1024LL * RG * 4 ? 4 : 2;                     // >= ~4 workgroups per CU before rows are batched deeper
        if (knob_r == 2 || knob_r == 4) R = knob_r;
        const int rows = knob_rows > 0 ? ((knob_rows + RG * R - 1) / (RG * R)) * RG * R : RG * R;
        dim3 cgrid((M + rows - 1) / rows), cblock(64 * WPR * RG);
#define VR_LN_BWDC2(TT, JJ, RR)                                                                                         \
        hipLaunchKernelGGL((ln_bwd_col_kernel<TT, JJ, RR>), cgrid, cblock, 0, (hipStream_t)stream, (const TT*)dy, x, w, mean, rstd, \
                           keep, dx_in, dx_out, dw, db, (TT*)gt_out, gt_scale, gt_keep, M, C, rows_per_sample, rows, copies, WPR);
#define VR_LN_BWDC(TT)                                                                                                  \
        if (J == 1 && R == 4) { VR_LN_BWDC2(TT, 1, 4) } else if (J == 1) { VR_LN_BWDC2(TT, 1, 2) }                      \
        else if (R == 4) { VR_LN_BWDC2(TT, 2, 4) } else { VR_LN_BWDC2(TT, 2, 2) }
        if (dy_dtype == VR_F32) { VR_LN_BWDC(float) } else { VR_LN_BWDC(bf16_t) }
#undef VR_LN_BWDC
#undef VR_LN_BWDC2
        VR_CHECK_LAUNCH();
        return VR_OK;
    }
    if ((knob_lean == 1 && C >= 384) || knob_lean >= 2) {
#define VR_LN_BWDL(NV)                                                                                                 \
    if (dy_dtype == VR_F32)                                                                                            \
        hipLaunchKernelGGL((ln_bwd_lean_kernel<float, NV>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)dy, x, w, \
                           mean, rstd, keep, dx_in, dx_out, dw, db, (float*)gt_out, gt_scale, gt_keep, M, C, rows_per_sample, BWD_ROWS, copies); \
    else                                                                                                               \
        hipLaunchKernelGGL((ln_bwd_lean_kernel<bf16_t, NV>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, x, \
                           w, mean, rstd, keep, dx_in, dx_out, dw, db, (bf16_t*)gt_out, gt_scale, gt_keep, M, C, rows_per_sample, BWD_ROWS, copies);
        switch (nv) {
            case 1: VR_LN_BWDL(1) break;
            case 2: VR_LN_BWDL(2) break;
            case 3: VR_LN_BWDL(3) break;
            case 4: VR_LN_BWDL(4) break;
            case 5: VR_LN_BWDL(5) break;
            default: VR_LN_BWDL(8) break;
        }
#undef VR_LN_BWDL
        VR_CHECK_LAUNCH();
        return VR_OK;
    }
#define VR_LN_BWD(NV)                                                                                                  \
    if (dy_dtype == VR_F32)                                                                                            \
        hipLaunchKernelGGL((ln_bwd_kernel<float, NV>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)dy, x, w, \
                           mean, rstd, keep, dx_in, dx_out, dw, db, (float*)gt_out, gt_scale, gt_keep, M, C, rows_per_sample, BWD_ROWS, copies); \
    else                                                                                                               \
        hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, NV>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, x, \
                           w, mean, rstd, keep, dx_in, dx_out, dw, db, (bf16_t*)gt_out, gt_scale, gt_keep, M, C, rows_per_sample, BWD_ROWS, copies);
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
