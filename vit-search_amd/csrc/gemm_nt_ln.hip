// bf16 GEMM whose workgroups own WHOLE output rows, with the LayerNorm that follows (forward) or precedes (backward) the
// Linear fused behind it:
//
//   mode 0 (forward; attention `proj` and Mlp `fc2`, reference nets/supernet_blocks.py:214-253):
//       x1 = resid + scale[s] * mask_keep_n(A W^T + bias)          -> C       (fp32 residual stream, as vr_gemm)
//       (y, mean, rstd) = MaskedLayerNorm(x1; w, b, keep)           -> ln.y bf16, ln.mean, ln.rstd   (as vr_ln_fwd)
//     i.e. the norm2 of the same block / norm1 of the next block: the residual stream is not read back from HBM and one
//     launch per LayerNorm disappears.
//   mode 1 (backward; data gradient of `qkv` / `fc1` followed by the backward of the LayerNorm that fed them,
//           nets/masked_layer_norm.py:55-88):
//       dy = dU W            (fp32, never written)
//       C = resid + dLN/dx(dy; x, w, mean, rstd, keep);  dw/db += column sums;  gt_out = cast(mask(C) * gt_scale)   (as vr_ln_bwd)
//
// Structure (round 2 rewrite): the K loop of gemm_nt.hip on a BM x BN tile (64 x 256 for N <= 256, 32 x 512 for N <= 512; four
// waves side by side along N, 64 accumulator registers each), then the LayerNorm kernels' own row loop on the tile: the
// accumulators are parked in the (idle) slice buffer as fp32 rows, half a tile at a time, and every wave walks whole rows -- a
// lane owns 4 (8) consecutive columns of a row, row statistics are two wave reductions, all loads of several rows are in flight
// before the first reduction.  The first version kept the row values in dead accumulator registers across a workgroup barrier
// and merged per-wave partial statistics through LDS: its backward form needed 168 registers, spilled and ran 3x slower than
// the two separate kernels.  This form is the two kernels it replaces glued together through LDS: nothing is live across the
// glue but the accumulators.
#include <cstdlib>

#include "common.h"
#include "../../include/vitres_hip.h"
#include "gemm_shared.h"

namespace vr_gemm_ntln {
using namespace vr_gemm_shared;

typedef __bf16 bfv8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

constexpr int BK = 64, NTHR = 256;

__device__ const uint4 zero_chunk[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};

struct RowMeta {
    int keep;      // forward: kept output-column prefix of the GEMM (1 << 30: dense); backward: gt_keep
    float scale;   // forward: DropPath scale of the row's sample; backward: gt_scale
    int orow;      // output row, -1: row >= M
    int lnkeep;    // kept prefix of the LayerNorm (N: dense)
    float mu, rs;  // backward: saved statistics of the row
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int MI, int NJ, int MODE, bool KTAIL>
__global__ __launch_bounds__(NTHR, NJ > 5 ? 2 : 3) void ntln_kernel(   // (32 x 512 tiles: 68 KB of LDS, two per CU)
   const vr_gemm_args p, const vr_ln_epilogue f) {
    constexpr int BM = 16 * MI, BN = 64 * NJ, WCOLS = 16 * NJ;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    constexpr int AP = BM / 32, BP = BN / 32;                      // LDS-DMA pieces (8 rows x 128 B) per wave
    constexpr int NV = (BN + 255) / 256;                           // float4 column groups per lane in the row loop (the last one
                                                                   // partly outside the tile at BN = 320: cin[] / reads of dead slots)
    constexpr int HM = MI / 2, PR = 16 * HM;                       // 16-row fragments / rows parked per half tile
    constexpr int SLOTS = BN / 4;                                  // 16-byte slots per parked row
    static_assert(AP >= 1 && PR * BN * 4 <= A_BYTES + B_BYTES, "tile shape");
    __shared__ __attribute__((aligned(1024))) char smem[A_BYTES + B_BYTES];
    __shared__ RowMeta rowmeta[BM];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tiles_m = group_tiles(p.M, BM, p.m_groups);
    int tile = blockIdx.x;
    if (tiles_m >= 16) {                 // XCD-aware order (see gemm_nt.hip): an XCD owns a contiguous run of row tiles
        const int xq = tiles_m >> 3, xr = tiles_m & 7, x = tile & 7;
        tile = x * xq + min(x, xr) + (tile >> 3);
    }
    // every architecture group of a multi-arch batch on every XCD, and no tile with rows of two groups (gemm_shared.h): [m0, mend)
    int m0, mend;
    group_tile_rows(tile, p.M, BM, p.m_groups, m0, mend);
    const RowMap amap = {p.a_map.rpi, p.a_map.rps, p.a_map.off};

    // ---- masked-work skipping (rules of the general kernel) ----
    const int ntiles = (p.K + BK - 1) / BK;
    int kmax = 1 << 30;
    bool n_any = true;
    if (p.keep_k || p.keep_n) {
        int s_lo = 0, s_hi = 0;
        if (p.rows_in > 0) { s_lo = m0 / p.rows_in; s_hi = (min(m0 + BM, mend) - 1) / p.rows_in; }
        kmax = max_keep(p.keep_k, s_lo, s_hi, 1 << 30);
        n_any = max_keep(p.keep_n, s_lo, s_hi, 1 << 30) > 0;
    }
    LiveSlices live;                 // cursor over the slices with kept k (gemm_shared.h)
    live.init(p.keep_k, p.k_period, 0, ntiles, kmax, n_any);

    // ---- LDS-DMA source addressing: piece = 8 tile rows, lane -> (row, slot); slot p of row r holds k-chunk p ^ ((r >> 1) & 7) ----
    const char* gA[AP];
    const char* gB[BP];
    int chunkA[AP], chunkB[BP];
    {
        const int rb = wave * (8 * BP) + (lane >> 3);
        if (BN <= p.N) {                                        // every weight row of the tile exists: affine addresses
            const char* b0 = reinterpret_cast<const char*>(p.B) + (long long)rb * p.ldb * 2;
            const long long step = (long long)p.ldb * 16;
#pragma unroll
            for (int h = 0; h < BP; ++h) {
                const int c = (lane & 7) ^ (((rb + 8 * h) >> 1) & 7);
                gB[h] = b0 + h * step + c * 16;
                chunkB[h] = c * 8;
            }
        } else {
#pragma unroll
            for (int h = 0; h < BP; ++h) {
                const int r = rb + h * 8;
                const int c = (lane & 7) ^ ((r >> 1) & 7);
                const int nb = min(r, p.N - 1);
                gB[h] = reinterpret_cast<const char*>(p.B) + ((long long)nb * p.ldb + c * 8) * 2;
                chunkB[h] = c * 8;
            }
        }
        const int ra = wave * (8 * AP) + (lane >> 3);
        if (amap.rpi == 0 && m0 + BM <= mend) {
            const char* a0 = reinterpret_cast<const char*>(p.A) + (long long)(m0 + ra) * p.lda * 2;
            const long long step = (long long)p.lda * 16;
#pragma unroll
            for (int h = 0; h < AP; ++h) {
                const int c = (lane & 7) ^ (((ra + 8 * h) >> 1) & 7);
                gA[h] = a0 + h * step + c * 16;
                chunkA[h] = c * 8;
            }
        } else {
#pragma unroll
            for (int h = 0; h < AP; ++h) {
                const int r = ra + h * 8;
                const int c = (lane & 7) ^ ((r >> 1) & 7);
                const int ma = min(m0 + r, mend - 1);
                gA[h] = reinterpret_cast<const char*>(p.A) + (map_row(amap, ma) * (long long)p.lda + c * 8) * 2;
                chunkA[h] = c * 8;
            }
        }
    }
    const char* zero = reinterpret_cast<const char*>(zero_chunk);

    const int frow = lane & 15, fswz = (frow >> 1) & 7;
    const int slot0 = (((lane >> 4)) ^ fswz) << 4, slot1 = ((4 + (lane >> 4)) ^ fswz) << 4;
    const char* As = smem + frow * 128;
    const char* Bs = smem + A_BYTES + (wave * WCOLS + frow) * 128;

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto issue = [&](int kt) {
        const int k0 = kt * BK;
        const long long kb = (long long)k0 * 2;
#pragma unroll
        for (int h = 0; h < AP; ++h) {
            const char* sa = gA[h] + kb;
            if constexpr (KTAIL) sa = (k0 + chunkA[h] < p.K) ? sa : zero;
            __builtin_amdgcn_global_load_lds((glb_void*)sa, (lds_void*)(smem + (wave * AP + h) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int h = 0; h < BP; ++h) {
            const char* sb = gB[h] + kb;
            if constexpr (KTAIL) sb = (k0 + chunkB[h] < p.K) ? sb : zero;
            __builtin_amdgcn_global_load_lds((glb_void*)sb, (lds_void*)(smem + A_BYTES + (wave * BP + h) * 1024), 16, 0, 0);
        }
    };
    auto compute = [&]() {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int so = s == 0 ? slot0 : slot1;
            bfv8 a[MI], b[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const bfv8*>(As + i * 2048 + so);
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[j] = *reinterpret_cast<const bfv8*>(Bs + j * 2048 + so);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
    };

    int kt = live.take();
    if (kt < ntiles) issue(kt);
    if (t < BM) {                          // per-row metadata of the row loop; its loads overlap the first slice
        const int m = m0 + t;
        RowMeta rm;
        rm.keep = 1 << 30; rm.scale = 1.0f; rm.orow = -1; rm.lnkeep = p.N; rm.mu = 0.f; rm.rs = 0.f;
        if (m < mend) {
            const int sample = p.rows_in > 0 ? m / p.rows_in : 0;
            rm.orow = m;
            if (f.keep) rm.lnkeep = min(f.keep[sample], p.N);
            if constexpr (MODE == 0) {
                if (p.scale) rm.scale = p.scale[sample];
                if (p.keep_n) rm.keep = p.keep_n[sample];
            } else {
                if (f.gt_scale) rm.scale = f.gt_scale[sample];
                if (f.gt_keep) rm.keep = f.gt_keep[sample];
                rm.mu = f.mean[m];
                rm.rs = f.rstd[m];
            }
        }
        rowmeta[t] = rm;
    }
    if (kt >= ntiles) __syncthreads();
    while (kt < ntiles) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        compute();
        __syncthreads();
        kt = live.take();
        if (kt < ntiles) issue(kt);
    }

    // ---- row loop: the tile is parked half by half ([PR rows][BN columns] fp32 in the slice buffer; 16-byte slot s of row r
    //      sits at slot s ^ (r & 7): the 8 rows a ds_write_b128 lane group writes at one column fall on 8 different bank
    //      groups, a row read back by the 64 lanes of a wave is a permutation of its 64 slots) and walked as whole rows ----
    float* park = reinterpret_cast<float*>(smem);
    const int g4 = lane >> 4, c16 = lane & 15;
    // this lane's columns in the row loop: float4 group v covers columns 4 (lane + 64 v) .. +3
    float4 lw[NV], lb[NV], bv[NV];
    bool cin[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = 4 * (lane + 64 * v);
        cin[v] = c < p.N;                                      // N % 8 == 0 -> a group is whole or outside
        const int cc = cin[v] ? c : 0;
        lw[v] = ld4(f.w + cc);
        lb[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        bv[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (MODE == 0) {
            lb[v] = ld4(f.b + cc);
            if (p.bias) bv[v] = ld4(p.bias + cc);
        }
    }
    float4 gw[NV], gb[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) { gw[v] = make_float4(0.f, 0.f, 0.f, 0.f); gb[v] = gw[v]; }
    const bool has_res = p.resid != nullptr;
    constexpr int RPW = PR / 4;                                // rows per wave and half tile
    constexpr int RU = (NV == 1 && RPW >= 4) ? 4 : 2;         // rows in flight per wave (register budget: 168)

#pragma unroll
    for (int half = 0; half < 2; ++half) {
        // park rows [half * PR, half * PR + PR): lane holds C[16 i + c16][wave * WCOLS + 16 j + 4 g4 + 0..3]
#pragma unroll
        for (int ih = 0; ih < HM; ++ih) {
            const int i = half * HM + ih;
            const int r = 16 * ih + c16;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int slot = (wave * WCOLS + 16 * j + 4 * g4) >> 2;
                *reinterpret_cast<f32x4*>(park + ((size_t)r * SLOTS + (slot ^ (r & 7))) * 4) = acc[i][j];
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int r0 = 0; r0 < RPW; r0 += RU) {
            RowMeta rm[RU];
            float4 dv[RU][NV], xv[RU][NV], rv[RU][NV];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int r = 4 * (r0 + u) + wave;                 // row inside the parked half
                rm[u] = rowmeta[half * PR + r];
                const long long orow = rm[u].orow < 0 ? 0 : rm[u].orow;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int slot = lane + 64 * v;
                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(park + ((size_t)r * SLOTS + (slot ^ (r & 7))) * 4);
                    dv[u][v] = make_float4(a4[0], a4[1], a4[2], a4[3]);
                    const int cc = cin[v] ? 4 * slot : 0;
                    rv[u][v] = has_res ? ld4(p.resid + orow * p.ldc + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (MODE == 1) xv[u][v] = ld4(f.x + orow * p.ldc + cc);
                }
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const bool rok = rm[u].orow >= 0;
                const int lk = rm[u].lnkeep;
                const float inv_n = lk > 0 ? 1.0f / (float)lk : 0.f;
                if constexpr (MODE == 0) {
                    // x1 = resid + scale * mask(acc + bias); LayerNorm over the first lk channels (vr_ln_fwd)
                    float4 x1[NV];
                    float s = 0.f, s2 = 0.f;
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const int c = 4 * (lane + 64 * v);
                        const int kn = rm[u].keep - c;
                        const float sc = rm[u].scale;
                        float4 a = dv[u][v];
                        a.x = (0 < kn) ? (a.x + bv[v].x) * sc : 0.f;
                        a.y = (1 < kn) ? (a.y + bv[v].y) * sc : 0.f;
                        a.z = (2 < kn) ? (a.z + bv[v].z) * sc : 0.f;
                        a.w = (3 < kn) ? (a.w + bv[v].w) * sc : 0.f;
                        a.x += rv[u][v].x; a.y += rv[u][v].y; a.z += rv[u][v].z; a.w += rv[u][v].w;
                        if (rok && cin[v]) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (long long)rm[u].orow * p.ldc + c) = a;
                        const int kl = cin[v] ? lk - c : 0;
                        a.x = (0 < kl) ? a.x : 0.f; a.y = (1 < kl) ? a.y : 0.f; a.z = (2 < kl) ? a.z : 0.f; a.w = (3 < kl) ? a.w : 0.f;
                        x1[v] = a;
                        s += a.x + a.y + a.z + a.w;
                        s2 += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
                    }
                    s = wave_sum(s);
                    const float mu = s * inv_n;
                    float var;
                    if (f.keep) {                                  // masked path: var = E[x^2] / p - mu^2 (masked_layer_norm.py:38-40)
                        var = wave_sum(s2) * inv_n - mu * mu;
                    } else {                                       // F.layer_norm: two-pass variance
                        float d2 = 0.f;
#pragma unroll
                        for (int v = 0; v < NV; ++v) {
                            if (cin[v]) {
                                const float a = x1[v].x - mu, b2 = x1[v].y - mu, c2 = x1[v].z - mu, d = x1[v].w - mu;
                                d2 += a * a + b2 * b2 + c2 * c2 + d * d;
                            }
                        }
                        var = wave_sum(d2) * inv_n;
                    }
                    const float rs = 1.0f / sqrtf(var + f.eps);
                    if (rok) {
                        if (lane == 0) { f.mean[rm[u].orow] = mu; f.rstd[rm[u].orow] = rs; }
#pragma unroll
                        for (int v = 0; v < NV; ++v) {
                            const int c = 4 * (lane + 64 * v);
                            if (cin[v]) {
                                const int kl = lk - c;
                                const float o0 = (0 < kl) ? lw[v].x * ((x1[v].x - mu) * rs) + lb[v].x : 0.f;
                                const float o1 = (1 < kl) ? lw[v].y * ((x1[v].y - mu) * rs) + lb[v].y : 0.f;
                                const float o2 = (2 < kl) ? lw[v].z * ((x1[v].z - mu) * rs) + lb[v].z : 0.f;
                                const float o3 = (3 < kl) ? lw[v].w * ((x1[v].w - mu) * rs) + lb[v].w : 0.f;
                                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(f.y) + (long long)rm[u].orow * p.N + c) =
                                    make_uint2(pack_bf2(o0, o1), pack_bf2(o2, o3));
                            }
                        }
                    }
                } else {
                    // dLN/dx of dy (vr_ln_bwd): dz = dy * w; dx = (dz - (mean(dz) + z mean(z dz))) * rstd + resid
                    float4 gz[NV], zz[NV];
                    float s1 = 0.f, s2 = 0.f;
                    const float mu = rm[u].mu, rs = rm[u].rs;
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const int c = 4 * (lane + 64 * v);
                        const int kl = (rok && cin[v]) ? lk - c : 0;
                        float4 a = dv[u][v], xx = xv[u][v];
                        if (!(0 < kl)) { a.x = 0.f; xx.x = mu; }
                        if (!(1 < kl)) { a.y = 0.f; xx.y = mu; }
                        if (!(2 < kl)) { a.z = 0.f; xx.z = mu; }
                        if (!(3 < kl)) { a.w = 0.f; xx.w = mu; }
                        const float4 z = make_float4((xx.x - mu) * rs, (xx.y - mu) * rs, (xx.z - mu) * rs, (xx.w - mu) * rs);
                        gw[v].x += a.x * z.x; gw[v].y += a.y * z.y; gw[v].z += a.z * z.z; gw[v].w += a.w * z.w;
                        gb[v].x += a.x; gb[v].y += a.y; gb[v].z += a.z; gb[v].w += a.w;
                        const float4 g = make_float4(a.x * lw[v].x, a.y * lw[v].y, a.z * lw[v].z, a.w * lw[v].w);
                        s1 += g.x + g.y + g.z + g.w;
                        s2 += g.x * z.x + g.y * z.y + g.z * z.z + g.w * z.w;
                        gz[v] = g;
                        zz[v] = z;
                    }
                    s1 = wave_sum(s1) * inv_n;
                    s2 = wave_sum(s2) * inv_n;
                    if (rok) {
#pragma unroll
                        for (int v = 0; v < NV; ++v) {
                            const int c = 4 * (lane + 64 * v);
                            if (cin[v]) {
                                const int kl = lk - c, kg = rm[u].keep - c;
                                const float4 r = rv[u][v];
                                float4 o;
                                o.x = (0 < kl) ? (gz[v].x - (s1 + zz[v].x * s2)) * rs + r.x : 0.f;
                                o.y = (1 < kl) ? (gz[v].y - (s1 + zz[v].y * s2)) * rs + r.y : 0.f;
                                o.z = (2 < kl) ? (gz[v].z - (s1 + zz[v].z * s2)) * rs + r.z : 0.f;
                                o.w = (3 < kl) ? (gz[v].w - (s1 + zz[v].w * s2)) * rs + r.w : 0.f;
                                const long long oidx = (long long)rm[u].orow * p.ldc + c;
                                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + oidx) = o;
                                if (f.gt_out) {
                                    const float sc = rm[u].scale;
                                    const float t0 = (0 < kg) ? o.x * sc : 0.f, t1 = (1 < kg) ? o.y * sc : 0.f;
                                    const float t2 = (2 < kg) ? o.z * sc : 0.f, t3 = (3 < kg) ? o.w * sc : 0.f;
                                    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(f.gt_out) + oidx) =
                                        make_uint2(pack_bf2(t0, t1), pack_bf2(t2, t3));
                                }
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();                                       // the park area is rewritten by the second half / reused below
    }

    if constexpr (MODE == 1) {
        // LayerNorm weight / bias gradients: every wave holds partial column sums over its rows -> cross-wave sum through LDS
        float* red = reinterpret_cast<float*>(smem);               // [2][4 waves][BN]
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (4 * (lane + 64 * v) < BN) {
                *reinterpret_cast<float4*>(red + (0 * 4 + wave) * BN + 4 * (lane + 64 * v)) = gw[v];
                *reinterpret_cast<float4*>(red + (1 * 4 + wave) * BN + 4 * (lane + 64 * v)) = gb[v];
            }
        }
        __syncthreads();
        const long long grow = (long long)(blockIdx.x % (unsigned)(f.grad_copies > 1 ? f.grad_copies : 1)) * p.N;
        for (int c = t; c < BN; c += NTHR) {
            if (c < p.N) {
                const float a = red[0 * BN + c] + red[1 * BN + c] + red[2 * BN + c] + red[3 * BN + c];
                const float b2 = red[4 * BN + c] + red[5 * BN + c] + red[6 * BN + c] + red[7 * BN + c];
                atomicAdd(f.dw + grow + c, a);                 // partial row of this workgroup (vr_ln_bwd's grad_copies)
                atomicAdd(f.db + grow + c, b2);
            }
        }
    }
}

template <int MI, int NJ> int launch(const vr_gemm_args& a, const vr_ln_epilogue& f, hipStream_t stream) {
    const unsigned tiles = (unsigned)vr_gemm_shared::group_tiles(a.M, 16 * MI, a.m_groups);
    const bool ktail = (a.K % BK) != 0;
    if (f.mode == 0) {
        if (ktail) hipLaunchKernelGGL((ntln_kernel<MI, NJ, 0, true>), dim3(tiles), dim3(NTHR), 0, stream, a, f);
        else hipLaunchKernelGGL((ntln_kernel<MI, NJ, 0, false>), dim3(tiles), dim3(NTHR), 0, stream, a, f);
    } else {
        if (ktail) hipLaunchKernelGGL((ntln_kernel<MI, NJ, 1, true>), dim3(tiles), dim3(NTHR), 0, stream, a, f);
        else hipLaunchKernelGGL((ntln_kernel<MI, NJ, 1, false>), dim3(tiles), dim3(NTHR), 0, stream, a, f);
    }
    VR_CHECK_LAUNCH();
    return VR_OK;
}

}  // namespace vr_gemm_ntln

extern "C" int vr_gemm_ln_supported(int32_t N) { return N > 0 && N % 8 == 0 && N <= 512; }

extern "C" int vr_gemm_ln(const vr_gemm_args* g, const vr_ln_epilogue* ln, vr_stream_t stream) {
    using namespace vr_gemm_ntln;
    if (!g || !ln || !g->A || !g->B || !g->C || !ln->w || g->M <= 0 || g->N <= 0 || g->K <= 0) return VR_EINVAL;
    const vr_gemm_args& a = *g;
    if (a.in_dtype != VR_BF16 || a.out_dtype != VR_F32 || a.a_trans || a.b_trans || a.atomic || a.split_k > 1 || a.bias_grad ||
        a.act || a.dact_u || a.pos || a.C2 || a.n_period > 0 || a.c_map.rpi != 0)
        return VR_EUNSUPPORTED;
    if (!vr_gemm_ln_supported(a.N) || a.ldc % 8 || a.lda % 8 || a.ldb % 8 || a.ldc < a.N) return VR_EUNSUPPORTED;
    // sched bit 0x80000: masked tiles of A may be unwritten -- readable only when no row tile straddles two architecture groups
    if ((a.sched & 0x80000) && a.keep_k && a.m_groups > 1 && !group_pure(a.M, a.m_groups)) return VR_EUNSUPPORTED;
    if (ln->mode == 0) {
        if (!ln->b || !ln->y || !ln->mean || !ln->rstd) return VR_EINVAL;
    } else if (ln->mode == 1) {
        if (!ln->x || !ln->mean || !ln->rstd || !ln->dw || !ln->db || a.bias || a.scale || a.keep_n) return VR_EINVAL;
    } else {
        return VR_EINVAL;
    }
    if (a.N <= 256) return launch<4, 4>(a, *ln, (hipStream_t)stream);
    if (a.N <= 320) return launch<4, 5>(a, *ln, (hipStream_t)stream);      // sr_small's first stage: 64 x 320 tiles
    return launch<2, 8>(a, *ln, (hipStream_t)stream);
}
