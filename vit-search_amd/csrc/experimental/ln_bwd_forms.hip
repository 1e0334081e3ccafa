// LayerNorm backward, column-owned form (round 4; EXPERIMENTAL builds only).  Measured (profiles/r04_two_chain_and_ln_col.txt): alone
// 23.4 / 16.4 / 16.1 us at (32 896, 256) / (8 320, 512) / (2 176, 1024) against 32.5 / 23.7 / 41.0 us for the wave-per-row kernel of
// ln.hip -- and 7.95 - 8.0 ms per training step against 7.55: beside a weight-gradient group its 256-thread workgroups (a
// __syncthreads per row round, 100 - 154 VGPRs) wait for slots the group's long-lived workgroups hold (34 - 81 us per launch).
// A second lean form (dgamma / dbeta through LDS atomics, 60 - 142 VGPRs) was slower still (8.7 ms) and is not kept.
#include "../common.h"
#include <cstdlib>
#include <type_traits>
#include "../../../include/vitres_hip.h"

namespace {

// Column-owned form: a thread owns 4 J channels of the row for the workgroup's whole life, so gamma and the
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


}  // namespace

bool vr_ln_bwd_col_launch(const void* dy, const float* x, const float* w, const float* mean, const float* rstd, const int32_t* keep,
                          const float* dx_in, float* dx_out, float* dw, float* db, void* gt_out, const float* gt_scale,
                          const int32_t* gt_keep, int32_t M, int32_t C, int32_t rows_per_sample, int32_t dy_dtype, int32_t copies,
                          int32_t knob_rows, hipStream_t stream) {
    {
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
    }
    return true;
}
