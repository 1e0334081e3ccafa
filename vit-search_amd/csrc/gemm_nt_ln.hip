// bf16 GEMM whose workgroups own WHOLE output rows, with the LayerNorm that follows (forward) or precedes (backward) the
// Linear fused into the epilogue:
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
// Tile: BM rows x BN >= N columns (64 x 256 for N <= 256, 32 x 512 for N <= 512), four waves side by side along N, each
// BM x BN/4 = MI x NJ v_mfma_f32_16x16x32_bf16 tiles (64 accumulators).  K loop, LDS image and LDS-DMA addressing are those of
// gemm_nt.hip (single slice buffer; the CU's other workgroups hide the load latency).  Row statistics are combined across the
// four waves through LDS with one workgroup barrier: forward merges per-wave (sum, M2 about the wave's own mean) pairs (Chan),
// so the variance is as robust as a two-pass one.
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

template <int LPR> __device__ __forceinline__ float row_sum(float v) {     // over the LPR consecutive lanes of a row
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) v += __shfl_xor(v, o);
    return v;
}
template <int LPR> __device__ __forceinline__ float col_sum(float v) {     // over the 64 / LPR lanes sharing a column group
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) v += __shfl_xor(v, o);
    return v;
}

template <int MI, int NJ, int MODE>
__global__ __launch_bounds__(NTHR, 3) void ntln_kernel(const vr_gemm_args p, const vr_ln_epilogue f) {
    constexpr int BM = 16 * MI, BN = 64 * NJ, WCOLS = 16 * NJ;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    constexpr int AP = BM / 32, BP = BN / 32;                      // LDS-DMA pieces (8 rows x 128 B) per wave
    constexpr int LPR = 2 * NJ, RPP = 64 / LPR, NQ = 16 / RPP;     // epilogue: lanes per row, rows per pass, passes per 16 rows
    constexpr int PARK = 16 * WCOLS * 4;                           // bytes a wave parks per 16-row round
    constexpr int CW = 8;
    static_assert(AP >= 1 && 4 * PARK <= A_BYTES + B_BYTES, "tile shape");
    __shared__ __attribute__((aligned(1024))) char smem[A_BYTES + B_BYTES];
    __shared__ RowMeta rowmeta[BM];
    __shared__ float2 stats[BM][4];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tiles_m = (p.M + BM - 1) / BM;
    int tile = blockIdx.x;
    if (tiles_m >= 16) {                 // XCD-aware order (see gemm_nt.hip): an XCD owns a contiguous run of row tiles
        const int xq = tiles_m >> 3, xr = tiles_m & 7, x = tile & 7;
        tile = x * xq + min(x, xr) + (tile >> 3);
    }
    const int m0 = tile * BM;
    const RowMap amap = {p.a_map.rpi, p.a_map.rps, p.a_map.off};

    // ---- masked-work skipping (rules of the general kernel) ----
    const int ntiles = (p.K + BK - 1) / BK;
    int kmax = 1 << 30;
    bool n_any = true;
    if (p.keep_k || p.keep_n) {
        int s_lo = 0, s_hi = 0;
        if (p.rows_in > 0) { s_lo = m0 / p.rows_in; s_hi = (min(m0 + BM, p.M) - 1) / p.rows_in; }
        kmax = max_keep(p.keep_k, s_lo, s_hi, 1 << 30);
        n_any = max_keep(p.keep_n, s_lo, s_hi, 1 << 30) > 0;
    }
    auto slice_live = [&](int kt) -> bool {
        return n_any && (p.keep_k == nullptr || range_has_kept(kt * BK, BK, p.k_period, kmax));
    };
    auto next_live = [&](int kt) -> int {
        while (kt < ntiles && !slice_live(kt)) ++kt;
        return kt;
    };

    // ---- LDS-DMA source addressing: piece = 8 tile rows, lane -> (row, slot); slot p of row r holds k-chunk p ^ ((r >> 1) & 7) ----
    const char* gA[AP];
    const char* gB[BP];
    int chunkA[AP], chunkB[BP];
#pragma unroll
    for (int h = 0; h < BP; ++h) {
        const int r = (wave * BP + h) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        const int nb = min(r, p.N - 1);
        gB[h] = reinterpret_cast<const char*>(p.B) + ((long long)nb * p.ldb + c * 8) * 2;
        chunkB[h] = c * 8;
    }
#pragma unroll
    for (int h = 0; h < AP; ++h) {
        const int r = (wave * AP + h) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        const int ma = min(m0 + r, p.M - 1);
        gA[h] = reinterpret_cast<const char*>(p.A) + (map_row(amap, ma) * (long long)p.lda + c * 8) * 2;
        chunkA[h] = c * 8;
    }
    const char* zero = reinterpret_cast<const char*>(zero_chunk);
    const bool ktail = (p.K % BK) != 0;

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
            const char* sa = (!ktail || (k0 + chunkA[h] < p.K)) ? gA[h] + kb : zero;
            __builtin_amdgcn_global_load_lds((glb_void*)sa, (lds_void*)(smem + (wave * AP + h) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int h = 0; h < BP; ++h) {
            const char* sb = (!ktail || (k0 + chunkB[h] < p.K)) ? gB[h] + kb : zero;
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

    int kt = next_live(0);
    if (kt < ntiles) issue(kt);
    if (t < BM) {                          // per-row epilogue metadata; its loads overlap the first slice
        const int m = m0 + t;
        RowMeta rm;
        rm.keep = 1 << 30; rm.scale = 1.0f; rm.orow = -1; rm.lnkeep = p.N; rm.mu = 0.f; rm.rs = 0.f;
        if (m < p.M) {
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
        kt = next_live(kt + 1);
        if (kt < ntiles) issue(kt);
    }

    // ---- epilogue: lane owns C[m = 16 i + (lane & 15)][n = 16 j + 4 (lane >> 4) + 0..3] of the wave's BM x WCOLS; a round
    //      parks 16 rows in the wave's own piece of the slice buffer and reads them back as rows: lane -> row lane / LPR
    //      (+ RPP q), columns 8 (lane % LPR) .. +8 ----
    float* park = reinterpret_cast<float*>(smem + wave * PARK);
    const int cg = lane % LPR, rsub = lane / LPR;
    const int n = wave * WCOLS + cg * CW;
    const bool live = n < p.N;                               // N % 8 == 0: the group is whole or outside
    const int nc = live ? n : 0;
    const RowMeta* meta = rowmeta + rsub;
    float lw[CW];
    loadw<float, CW>(f.w, nc, lw, true, CW);
    // row-layout values kept across the workgroup barrier (forward: x1, LayerNorm-masked; backward: dy, masked) live in the
    // accumulator registers of their round, which are dead once the round is parked: value (i, q, e) -> acc[i][(8 q + e) / 4][(8 q + e) % 4]
#define XV(i, q, e) acc[i][((q) * CW + (e)) >> 2][((q) * CW + (e)) & 3]
    const int wlo = wave * WCOLS;

    auto park_round = [&](int i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int slot = (4 * j + (lane >> 4)) ^ ((lane & 15) & (4 * NJ - 1));
            *reinterpret_cast<f32x4*>(park + (lane & 15) * WCOLS + slot * 4) = acc[i][j];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    auto unpark = [&](int q, float (&v)[CW]) {
        const int rl = q * RPP + rsub;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int slot = (2 * cg + h) ^ (rl & (4 * NJ - 1));
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(park + rl * WCOLS + slot * 4);
            v[4 * h] = a4[0]; v[4 * h + 1] = a4[1]; v[4 * h + 2] = a4[2]; v[4 * h + 3] = a4[3];
        }
    };

    if constexpr (MODE == 0) {
        float bv[CW], lb[CW];
#pragma unroll
        for (int e = 0; e < CW; ++e) bv[e] = 0.f;
        if (p.bias) loadw<float, CW>(p.bias, nc, bv, true, CW);
        loadw<float, CW>(f.b, nc, lb, true, CW);
        const bool has_res = p.resid != nullptr;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            park_round(i);
            RowMeta rm[NQ];
            float rv[NQ][CW];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                rm[q] = meta[i * 16 + q * RPP];
                const long long orow = rm[q].orow < 0 ? 0 : rm[q].orow;
#pragma unroll
                for (int e = 0; e < CW; ++e) rv[q][e] = 0.f;
                if (has_res) loadw<float, CW>(p.resid, orow * p.ldc + nc, rv[q], true, CW);
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                float v[CW];
                unpark(q, v);
                const int kn = live ? rm[q].keep - n : 0, kl = live ? rm[q].lnkeep - n : 0;
                const float sc = rm[q].scale;
#pragma unroll
                for (int e = 0; e < CW; ++e) {
                    v[e] += bv[e];
                    v[e] = (e < kn) ? v[e] * sc : 0.f;
                    v[e] += rv[q][e];
                    XV(i, q, e) = (e < kl) ? v[e] : 0.f;
                }
                if (rm[q].orow >= 0 && live) storew<float, CW>(p.C, (long long)rm[q].orow * p.ldc + nc, v, true, true, CW);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // per-wave partial statistics of every row: (sum, M2 about the wave's own mean) over its kept columns
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int row = i * 16 + q * RPP + rsub;
                const int lk = rowmeta[row].lnkeep;
                const int nw = min(max(lk - wlo, 0), WCOLS);
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < CW; ++e) s += XV(i, q, e);
                s = row_sum<LPR>(s);
                const float mw = nw > 0 ? s / (float)nw : 0.f;
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < CW; ++e) {
                    const float u = XV(i, q, e) - mw;
                    d += (n + e < lk) ? u * u : 0.f;
                }
                d = row_sum<LPR>(d);
                if (cg == 0) stats[row][wave] = make_float2(s, d);
            }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int row = i * 16 + q * RPP + rsub;
                const RowMeta rm = rowmeta[row];
                const int lk = rm.lnkeep;
                const float inv_n = lk > 0 ? 1.0f / (float)lk : 0.f;
                float2 st[4];
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) { st[w] = stats[row][w]; tot += st[w].x; }
                const float mu = tot * inv_n;
                float m2 = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const int nw = min(max(lk - w * WCOLS, 0), WCOLS);
                    const float dm = nw > 0 ? st[w].x / (float)nw - mu : 0.f;
                    m2 += st[w].y + (float)nw * dm * dm;
                }
                const float rs = 1.0f / sqrtf(m2 * inv_n + f.eps);
                if (rm.orow >= 0) {
                    if (wave == 0 && cg == 0) { f.mean[rm.orow] = mu; f.rstd[rm.orow] = rs; }
                    if (live) {
                        float o[CW];
#pragma unroll
                        for (int e = 0; e < CW; ++e) o[e] = (n + e < lk) ? lw[e] * ((XV(i, q, e) - mu) * rs) + lb[e] : 0.f;
                        storew<bf16_t, CW>(f.y, (long long)rm.orow * p.N + nc, o, true, true, CW);
                    }
                }
            }
    } else {
        float gwp[CW], gbp[CW];
#pragma unroll
        for (int e = 0; e < CW; ++e) { gwp[e] = 0.f; gbp[e] = 0.f; }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            park_round(i);
            RowMeta rm[NQ];
            float xx[NQ][CW];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                rm[q] = meta[i * 16 + q * RPP];
                const long long orow = rm[q].orow < 0 ? 0 : rm[q].orow;
                loadw<float, CW>(f.x, orow * p.ldc + nc, xx[q], true, CW);
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                float v[CW];
                unpark(q, v);
                const int kl = (rm[q].orow >= 0 && live) ? rm[q].lnkeep - nc : 0;
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int e = 0; e < CW; ++e) {
                    const bool in = e < kl;
                    const float a = in ? v[e] : 0.f;
                    const float z = in ? (xx[q][e] - rm[q].mu) * rm[q].rs : 0.f;
                    XV(i, q, e) = a;
                    gwp[e] += a * z;
                    gbp[e] += a;
                    const float g = a * lw[e];
                    s1 += g;
                    s2 += g * z;
                }
                s1 = row_sum<LPR>(s1);
                s2 = row_sum<LPR>(s2);
                if (cg == 0) stats[i * 16 + q * RPP + rsub][wave] = make_float2(s1, s2);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        const bool has_res = p.resid != nullptr;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            RowMeta rm[NQ];
            float xx[NQ][CW], rv[NQ][CW];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                rm[q] = meta[i * 16 + q * RPP];
                const long long orow = rm[q].orow < 0 ? 0 : rm[q].orow;
                loadw<float, CW>(f.x, orow * p.ldc + nc, xx[q], true, CW);       // second read: L2 hit
#pragma unroll
                for (int e = 0; e < CW; ++e) rv[q][e] = 0.f;
                if (has_res) loadw<float, CW>(p.resid, orow * p.ldc + nc, rv[q], true, CW);
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int row = i * 16 + q * RPP + rsub;
                const int lk = rm[q].lnkeep;
                const float inv_n = lk > 0 ? 1.0f / (float)lk : 0.f;
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) { const float2 st = stats[row][w]; s1 += st.x; s2 += st.y; }
                s1 *= inv_n;
                s2 *= inv_n;
                const int kl = lk - nc, kg = rm[q].keep - nc;
                float o[CW], tg[CW];
#pragma unroll
                for (int e = 0; e < CW; ++e) {
                    const float z = (xx[q][e] - rm[q].mu) * rm[q].rs;
                    const float g = XV(i, q, e) * lw[e];
                    o[e] = (e < kl) ? (g - (s1 + z * s2)) * rm[q].rs + rv[q][e] : 0.f;
                    tg[e] = (e < kg) ? o[e] * rm[q].scale : 0.f;
                }
                if (rm[q].orow >= 0 && live) {
                    const long long oidx = (long long)rm[q].orow * p.ldc + nc;
                    storew<float, CW>(p.C, oidx, o, true, true, CW);
                    if (f.gt_out) storew<bf16_t, CW>(f.gt_out, oidx, tg, true, true, CW);
                }
            }
        }
        // LayerNorm weight / bias gradients: column sums over the tile's rows (this wave owns its columns alone)
#pragma unroll
        for (int e = 0; e < CW; ++e) {
            gwp[e] = col_sum<LPR>(gwp[e]);
            gbp[e] = col_sum<LPR>(gbp[e]);
        }
        if (rsub == 0 && live) {
#pragma unroll
            for (int e = 0; e < CW; ++e) {
                atomicAdd(f.dw + n + e, gwp[e]);
                atomicAdd(f.db + n + e, gbp[e]);
            }
        }
    }
}

template <int MI, int NJ> int launch(const vr_gemm_args& a, const vr_ln_epilogue& f, hipStream_t stream) {
    const unsigned tiles = (unsigned)((a.M + 16 * MI - 1) / (16 * MI));
    if (f.mode == 0) hipLaunchKernelGGL((ntln_kernel<MI, NJ, 0>), dim3(tiles), dim3(NTHR), 0, stream, a, f);
    else hipLaunchKernelGGL((ntln_kernel<MI, NJ, 1>), dim3(tiles), dim3(NTHR), 0, stream, a, f);
    VR_CHECK_LAUNCH();
    return VR_OK;
}

#undef XV

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
    if (ln->mode == 0) {
        if (!ln->b || !ln->y || !ln->mean || !ln->rstd) return VR_EINVAL;
    } else if (ln->mode == 1) {
        if (!ln->x || !ln->mean || !ln->rstd || !ln->dw || !ln->db || a.bias || a.scale || a.keep_n) return VR_EINVAL;
    } else {
        return VR_EINVAL;
    }
    if (a.N <= 256) return launch<4, 4>(a, *ln, (hipStream_t)stream);
    return launch<2, 8>(a, *ln, (hipStream_t)stream);
}
