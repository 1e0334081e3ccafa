// Pieces shared by the GEMM kernels (gemm.hip: all forms, both precisions; gemm_nt.hip: bf16 K-contiguous fast path).
#pragma once
#include "common.h"

namespace vr_gemm_shared {

__device__ __forceinline__ bool kept_col(int n, int period, int keep) { return (period > 0 ? n % period : n) < keep; }
// does [x0, x0 + len) contain an index with (x % period) < keep ?
__device__ __forceinline__ bool range_has_kept(int x0, int len, int period, int keep) {
    if (keep <= 0) return false;
    if (period <= 0) return x0 < keep;
    const int r = x0 % period;
    return r < keep || r + len > period;
}
__device__ __forceinline__ int max_keep(const int* keep, int s_lo, int s_hi, int dense) {
    if (!keep) return dense;
    int mk = 0;
    for (int s = s_lo; s <= s_hi; ++s) mk = max(mk, keep[s]);
    return mk;
}

// Cursor over the K slices (64 wide) of [kb, ke) that hold a kept k for any sample of a tile: take() returns the next live slice
// (ke: none) and moves past it.  The same slices as range_has_kept(kt * 64, 64, k_period, kmax) -- but that test is a modulo per
// call, ~100 scalar instructions per slice on the critical path between a slice's last MFMA and the next slice's loads (PMC, round
// 3: 6.7 SALU instructions per MFMA in the forward / dgrad kernel); here (kt * 64) % k_period is carried along.
struct LiveSlices {
    int nk, ke, nr, period, kmax;
    bool masked, prefix;
    __device__ __forceinline__ void init(const int* keep_k, int k_period, int kb, int ke_, int kmax_, bool any) {
        masked = keep_k != nullptr;
        period = k_period >= 64 ? k_period : 0;        // periods below a slice: every slice holds kept columns
        prefix = masked && k_period <= 0;              // plain prefix: the slices below kmax
        kmax = kmax_;
        ke = ke_;
        nk = (any && (!masked || kmax_ > 0)) ? kb : ke_;
        nr = period > 0 ? (kb * 64) % period : kb * 64;
    }
    __device__ __forceinline__ int take() {
        while (nk < ke) {
            const bool lv = !masked || (period > 0 ? (nr < kmax || nr + 64 > period) : (!prefix || nr < kmax));
            const int cur = nk;
            ++nk;
            nr += 64;
            if (period > 0 && nr >= period) nr -= period;
            if (lv) return cur;
            if (prefix) { nk = ke; break; }            // beyond a prefix nothing is kept
        }
        return ke;
    }
};

// Row tiles / token splits of a multi-architecture batch (vr_gemm_args.m_groups = G contiguous, equal groups of rows, each with its
// own keep values): when G divides the rows, the bf16 kernels lay the grid out PER GROUP -- ceil((rows / G) / BM) tiles for each,
// the last one short -- so that no tile (and no token split of a weight gradient) holds rows of two architectures.  Every consumer
// of a masked activation then reads, for a row, only the channel slices below ITS architecture's keep (rounded up to the 64-wide
// slice); that is what lets the producers leave fully masked tiles unwritten (vr_gemm_args.sched bit 0x40000, gemm_ntk.hip).
__host__ __device__ inline bool group_pure(int rows, int G) { return G > 1 && rows % G == 0; }
__host__ __device__ inline int group_tiles(int rows, int BM, int G) {
    return group_pure(rows, G) ? G * ((rows / G + BM - 1) / BM) : (rows + BM - 1) / BM;
}

// Position p of the workgroup order -> index in [0, n) such that consecutive positions walk the G equal index groups
// round-robin (group g = [g n / G, (g + 1) n / G)): the XCD-contiguous runs of the tile order then hold every architecture
// group of a multi-arch batch in equal parts (vr_gemm_args.m_groups).  A bijection for every n, G.
__device__ __forceinline__ int interleave_groups(int p, int n, int G) {
    if (G <= 1 || n < 2 * G) return p;
    const int t = n / G;                                   // every group has t or t + 1 members
    if (p < t * G) {
        const int g = p % G, r = p / G;
        return (int)((long long)g * n / G) + r;
    }
    int k = p - t * G;                                     // the groups' (t + 1)-th members, in group order
    for (int g = 0; g < G; ++g) {
        const int s0 = (int)((long long)g * n / G), s1 = (int)((long long)(g + 1) * n / G);
        if (s1 - s0 > t) {
            if (k == 0) return s0 + t;
            --k;
        }
    }
    return p;
}

// position q of the row-tile order -> rows [m0, mend) the tile may touch: group-pure layout (consecutive positions walk the groups
// round-robin, like interleave_groups) or the plain one
__device__ __forceinline__ void group_tile_rows(int q, int rows, int BM, int G, int& m0, int& mend) {
    if (group_pure(rows, G)) {
        const int rpg = rows / G, g = q % G, i = q / G;
        m0 = g * rpg + i * BM;
        mend = g * rpg + rpg;
    } else {
        m0 = interleave_groups(q, (rows + BM - 1) / BM, G) * BM;
        mend = rows;
    }
}

// Epilogue flavours (compile-time, keeps every instantiation small enough to unroll fully):
//   EPI_STORE : (+bias)(+pos) -> keep mask -> scale -> (+resid) -> store TO
//   EPI_GELU  : (+bias) -> C = u, C2 = gelu(u) masked by keep            (Mlp.fc1)
//   EPI_DGELU : * gelu'(u) -> keep mask -> store TO                        (fc2 dgrad)
//   EPI_ATOMIC: keep mask/scale -> atomicAdd fp32                          (split-K wgrad)
//   EPI_DMUL  : * saved gelu'(u) (dact_u holds the derivative itself, act == 2) -> keep mask -> store TO
enum { EPI_STORE = 0, EPI_GELU = 1, EPI_DGELU = 2, EPI_ATOMIC = 3, EPI_DMUL = 4 };

// CW consecutive elements (CW = 4 or 8): 16-byte accesses whenever the group is whole and aligned
template <typename TI, int CW> __device__ __forceinline__ void loadw(const void* base, long long idx, float (&v)[CW], bool vec,
                                                                     int nvalid) {
    const TI* p = reinterpret_cast<const TI*>(base) + idx;
    if (vec) {
        if constexpr (sizeof(TI) == 4) {
#pragma unroll
            for (int h = 0; h < CW / 4; ++h) {
                const float4 x = *reinterpret_cast<const float4*>(p + 4 * h);
                v[4 * h] = x.x; v[4 * h + 1] = x.y; v[4 * h + 2] = x.z; v[4 * h + 3] = x.w;
            }
        } else if constexpr (CW == 8) {
            const uint4 u = *reinterpret_cast<const uint4*>(p);
            const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                v[2 * h] = __uint_as_float(w[h] << 16);
                v[2 * h + 1] = __uint_as_float(w[h] & 0xffff0000u);
            }
        } else {
            const uint2 u = *reinterpret_cast<const uint2*>(p);
            v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
            v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
        }
    } else {
#pragma unroll
        for (int e = 0; e < CW; ++e) v[e] = Elem<TI>::ld(p + (e < nvalid ? e : 0));
    }
}
// WT: fp32 stores go out write-through (sc1) -- a workgroup on another XCD reads them back inside the same launch (gemm_ntk.hip LNF)
template <typename TO, int CW, bool WT = false> __device__ __forceinline__ void storew(void* base, long long idx, const float (&v)[CW], bool vec,
                                                                                       bool rowok, int nvalid) {
    TO* p = reinterpret_cast<TO*>(base) + idx;
    if (vec) {
        if constexpr (sizeof(TO) == 4 && WT) {
#pragma unroll
            for (int h = 0; h < CW / 4; ++h) {
                const f32x4 x = {v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]};
                // (s_nop: the store reads its data registers for a few cycles after issue -- a hazard hipcc covers for its own stores only)
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p + 4 * h), "v"(x) : "memory");
            }
        } else if constexpr (sizeof(TO) == 4) {
#pragma unroll
            for (int h = 0; h < CW / 4; ++h)
                *reinterpret_cast<float4*>(p + 4 * h) = make_float4(v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]);
        } else if constexpr (CW == 8) {
            *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]),
                                                      pack_bf2(v[6], v[7]));
        } else {
            *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
        }
    } else {
#pragma unroll
        for (int e = 0; e < CW; ++e)
            if (rowok && e < nvalid) Elem<TO>::st(p + e, v[e]);
    }
}

}  // namespace vr_gemm_shared
