// Pieces shared by the LDS-DMA bf16 forward / data-gradient GEMM kernels (gemm_nt.hip: 4-wave tiles, several workgroups per
// CU; gemm_ntw.hip: 8-wave 256 x 128 tiles, ring-pipelined, stream-K): LDS geometry of the k-major weight slice, the per-row
// epilogue metadata and the epilogue itself (reference nets/supernet_blocks.py:37-52,102-119: bias, GELU, prefix masks, DropPath
// scale, residual add -- everything that follows F.linear there).
#pragma once
#include "common.h"
#include "../../include/vitres_hip.h"
#include "gemm_shared.h"

namespace vr_gemm_nt {
using namespace vr_gemm_shared;

typedef __bf16 bfv8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

constexpr int BK = 64;

// per-row epilogue metadata, computed once per tile while the first slices are in flight
struct RowMeta {
    int keep;      // kept output-column prefix of the row's sample (1 << 30: dense)
    float scale;   // DropPath scale of the row's sample
    int orow;      // output row after c_map, -1: row >= M
    int mloc;      // row index inside its sample (pos-embed row)
};

// BKM (b_trans: data gradients that read the forward's weight W [K = out features][N = in features] as it is): the weight slice
// is staged k-major ([64 k rows][BN columns], the image of gemm_tn.hip: 16-byte slot s of row k holds column chunk s ^ swz(k))
// and its MFMA fragments -- 8 consecutive k of one column -- come from the transposing LDS read ds_read_b64_tr_b16.  No
// transposed bf16 copy of the weights (one batched transposing cast of every Linear per step, 383 MB of traffic) is needed.
template <int BN> struct KMajor {      // geometry of the k-major weight slice (gemm_tn.hip Geo<TW>)
    static constexpr int ROWB = BN * 2, SLOTS = BN / 8, TPP = 1024 / ROWB;
    __device__ static __forceinline__ int swz(int t) {
        if constexpr (BN == 128) return ((t & 3) << 1) ^ (((t >> 3) & 1) << 3);
        else return (((t >> 1) & 1) << 1) | (((t >> 3) & 1) << 2);
    }
};
typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
template <int ROWB> __device__ __forceinline__ bfv8 tr_frag(const char* p) {
    typedef __attribute__((address_space(3))) s4v lds_s4v;
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(p));
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(p + 4 * ROWB));
    const s8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bfv8, v);
}

// Epilogue of one wave's [16 MI rows][16 NJ columns] block of accumulators (the MFMAs computed the transposed tile: a lane owns
// output row 16 i + (lane & 15) and columns 16 j + 4 (lane >> 4) + 0..3 of fragment (i, j)).
//   park  : this wave's own 4 KB of LDS (16 rows x 64 columns of floats, XOR-swizzled; no workgroup barrier is needed)
//   meta  : RowMeta of the wave's first row
//   nw0   : first output column of the wave
// FAST: N % 8 == 0, ldc / ldu % 8 == 0, n_period % 8 == 0 (checked on the host) -- every lane's 8-column group is whole
// or entirely outside the matrix, so the epilogue is branch-free 16-byte accesses.
// FEAT (FAST kernels): which optional epilogue terms exist is a compile-time fact of the kernel -- 0: none, 1: bias, 2: bias +
// residual, 3: bias + residual + DropPath scale, 4: decided per launch from the arguments (pos-embed, any other mix; the only
// form of the non-FAST kernels).  As run-time uniform conditions the compiler if-converts them into a v_cndmask per element
// and term (measured: 890 VALU instructions per tile and wave against 128 MFMAs at K = 256, VALU pipe busy 2x the matrix pipe).
// DEPTH: rounds of side operands in flight.  The fp32 residual costs 16 registers per round; with ONE round in flight (the load
// at the top of its own round) every round exposes a whole HBM latency: stamps put the residual epilogue of a 128 x 128 tile at
// 8 us against 2.5 us for a plain bf16 tile -- 16 KB in flight per workgroup, i.e. latency-bound by registers.  DEPTH = 2: the
// next round's residual is requested as soon as this round's accumulators are parked (their registers are free then: no
// higher peak), a round's worth of work ahead of its use; DEPTH = MI (the 8-wave kernel, two waves per SIMD): everything up front.
// WAITV (gemm_panel.hip): one s_waitcnt vmcnt(0) in front of the tile's first store -- see there.
// WT (gemm_ntk.hip LNF): the fp32 result is stored write-through (gemm_shared.h storew).
template <typename TO, int EPI, bool FAST, int MI, int NJ, int FEAT, int DEPTH = 1, bool WAITV = false, bool WT = false>
__device__ __forceinline__ void epilogue(const vr_gemm_args& p, f32x4 (&acc)[MI][NJ], float* park, const RowMeta* meta0, const int nw0,
                                         const int lane) {
    constexpr int WCOLS = 16 * NJ;
    constexpr int LPR = 2 * NJ, RPP = 64 / LPR, NQ = 16 / RPP;   // lanes per row, rows per pass, passes per 16 rows
    // ---- epilogue: lane owns C[m = 16 i + (lane & 15)][n = 16 j + 4 (lane >> 4) + 0..3] of the wave's 64 x 64 ----
    // Optional terms (bias, pos-embed, prefix mask, DropPath scale, residual) are wave-uniform kernel arguments: each is one
    // scalar branch around its code instead of arithmetic on neutral elements -- for the plain Linear forms the epilogue used
    // to issue 5x the instructions of the K = 256 loop.  The prefix mask is applied only by waves that hold a boundary group.
    constexpr int CW = 8;
    const int n = nw0 + (lane % LPR) * 8;                 // this lane's 8 columns
    const int nvalid = min(CW, p.N - n);
    const int nc = nvalid > 0 ? n : 0;
    const int nv = nvalid > 0 ? nvalid : 1;
    constexpr int OALIGN = sizeof(TO) == 2 ? 8 : 4;
    const bool vec = FAST || (nvalid == CW && (p.ldc % OALIGN == 0) && ((EPI != EPI_DGELU && EPI != EPI_DMUL) || p.ldu % 8 == 0));
    const bool vecb = FAST || (nvalid == CW && (p.N % 4 == 0));
    // prefix masks: the lane's 8 columns sit at ncp.. inside their period (periods are multiples of 8 on this path, so a
    // group never wraps; other periods take the per-element test)
    const bool grp = FAST || p.n_period <= 0 || (p.n_period & 7) == 0;
    const int ncp = p.n_period > 0 ? nc % p.n_period : nc;
    constexpr bool GEN = FEAT == 4;
    const bool has_bias = (EPI == EPI_STORE || EPI == EPI_GELU) && (GEN ? p.bias != nullptr : FEAT >= 1);
    const bool has_pos = (EPI == EPI_STORE) && GEN && p.pos;
    const bool has_res = (EPI == EPI_STORE) && (GEN ? p.resid != nullptr : FEAT >= 2);
    const bool has_mask = p.keep_n != nullptr;
    const bool has_scale = GEN ? p.scale != nullptr : FEAT == 3;
    float bv[CW];
#pragma unroll
    for (int e = 0; e < CW; ++e) bv[e] = 0.f;
    if (has_bias) loadw<float, CW>(p.bias, nc, bv, vecb, nv);
    const bool live = nvalid > 0;
    const RowMeta* meta = meta0 + (lane / LPR);
    // one round per 16-row fragment: park [16 rows][WCOLS columns] (<= 4 KB per wave), read back as rows.
    // Side operands of the epilogue (saved gelu'(u) / pre-activation of the fc2 data gradient, fp32 residual stream) are
    // requested as raw 16-byte loads ahead of their use -- the bf16 one a whole round ahead, the fp32 one (16 registers per
    // round: a second copy would spill) at the top of its round in front of the park / barrier: issued behind the barrier
    // of their own round, four rounds of exposed HBM latency made the fc2 data gradient 1.4x slower than the fc1 forward
    // of the same shape.
    constexpr bool SIDE_D = EPI == EPI_DGELU || EPI == EPI_DMUL;
    constexpr bool SIDE_R = EPI == EPI_STORE && (FEAT == 2 || FEAT == 3);
    constexpr bool PREF = FAST && (SIDE_D || SIDE_R);
    constexpr bool DEEP = DEPTH >= MI;
    constexpr int NSET = DEEP ? MI : (DEPTH > 1 ? DEPTH : 1);
    RowMeta rmn[NSET][NQ];
    uint4 dn[NSET][NQ];
    float4 rn[NSET][NQ][2];
    auto prefetch = [&](int i) {
        const int sidx = i % NSET;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            rmn[sidx][q] = meta[i * 16 + q * RPP];
            const long long row = rmn[sidx][q].orow < 0 ? 0 : rmn[sidx][q].orow;
            if constexpr (SIDE_D)
                dn[sidx][q] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.dact_u) + row * p.ldu + nc);
            if constexpr (SIDE_R) {
                const float* r = p.resid + row * p.ldc + nc;
                rn[sidx][q][0] = *reinterpret_cast<const float4*>(r);
                rn[sidx][q][1] = *reinterpret_cast<const float4*>(r + 4);
            }
        }
    };
    if constexpr (PREF && DEEP) {
#pragma unroll
        for (int i = 0; i < MI; ++i) prefetch(i);
    } else if constexpr (PREF && DEPTH > 1) {
        prefetch(0);
    } else if constexpr (SIDE_D && PREF) {
        prefetch(0);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        if constexpr (SIDE_R && PREF && !DEEP && DEPTH <= 1) prefetch(i);
        RowMeta rm[NQ];
        long long oidx[NQ];
        float rv[NQ][CW], pv[NQ][CW];
        uint4 dc[NQ];
        float4 rc[NQ][2];
        if constexpr (PREF) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                rm[q] = rmn[i % NSET][q];
                dc[q] = dn[i % NSET][q];
                rc[q][0] = rn[i % NSET][q][0];
                rc[q][1] = rn[i % NSET][q][1];
            }
            if constexpr (SIDE_D && !DEEP && DEPTH <= 1) {
                if (i + 1 < MI) prefetch(i + 1);
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int slot = (4 * j + (lane >> 4)) ^ (lane & (4 * NJ - 1));
            *reinterpret_cast<f32x4*>(park + (lane & 15) * WCOLS + slot * 4) = acc[i][j];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if constexpr (PREF && !DEEP && DEPTH > 1) {                // this round's accumulators are parked: their registers take
            if (i + 1 < MI) prefetch(i + 1);                       // the next round's side operands (other slot than rc's source)
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if constexpr (PREF) {
                oidx[q] = (long long)(rm[q].orow < 0 ? 0 : rm[q].orow) * p.ldc + nc;
                if constexpr (SIDE_D) {
                    const uint32_t w[4] = {dc[q].x, dc[q].y, dc[q].z, dc[q].w};
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        rv[q][2 * h] = __uint_as_float(w[h] << 16);
                        rv[q][2 * h + 1] = __uint_as_float(w[h] & 0xffff0000u);
                    }
                } else {
                    rv[q][0] = rc[q][0].x; rv[q][1] = rc[q][0].y; rv[q][2] = rc[q][0].z; rv[q][3] = rc[q][0].w;
                    rv[q][4] = rc[q][1].x; rv[q][5] = rc[q][1].y; rv[q][6] = rc[q][1].z; rv[q][7] = rc[q][1].w;
                }
            } else {
                rm[q] = meta[i * 16 + q * RPP];
                oidx[q] = (long long)(rm[q].orow < 0 ? 0 : rm[q].orow) * p.ldc + nc;
                if constexpr (EPI == EPI_DGELU || EPI == EPI_DMUL) loadw<bf16_t, CW>(p.dact_u, (long long)(rm[q].orow < 0 ? 0 : rm[q].orow) * p.ldu + nc, rv[q], vec, nv);
                if constexpr (EPI == EPI_STORE) {
                    if (has_res) loadw<float, CW>(p.resid, oidx[q], rv[q], vec, nv);
                    if (has_pos) loadw<float, CW>(p.pos, (long long)rm[q].mloc * p.N + nc, pv[q], vecb, nv);
                }
            }
        }
        if constexpr (WAITV) {
            if (i == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int rl = q * RPP + (lane / LPR);
            const bool mok = rm[q].orow >= 0;
            float v[CW];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int slot = (2 * (lane % LPR) + h) ^ (rl & (4 * NJ - 1));
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(park + rl * WCOLS + slot * 4);
                v[4 * h] = a4[0]; v[4 * h + 1] = a4[1]; v[4 * h + 2] = a4[2]; v[4 * h + 3] = a4[3];
            }
            if (has_bias) {
#pragma unroll
                for (int e = 0; e < CW; ++e) v[e] += bv[e];
            }
            if (has_pos) {
#pragma unroll
                for (int e = 0; e < CW; ++e) v[e] += pv[q][e];
            }
            const bool any = mok && live;
            if constexpr (EPI == EPI_DGELU) {
#pragma unroll
                for (int e = 0; e < CW; ++e) v[e] *= dgelu_fast(rv[q][e]);
            }
            if constexpr (EPI == EPI_DMUL) {                       // the forward saved gelu'(u) itself (act == 2)
#pragma unroll
                for (int e = 0; e < CW; ++e) v[e] *= rv[q][e];
            }
            float hh[CW];
            if constexpr (EPI == EPI_GELU) {
                if (p.act == 2) {                                  // C = gelu'(u) instead of u: the backward multiplies by it
#pragma unroll
                    for (int e = 0; e < CW; ++e) {
                        float cdf, pdf;
                        gelu_terms_fast(v[e], cdf, pdf);
                        hh[e] = v[e] * cdf;
                        v[e] = fmaf(v[e], pdf, cdf);
                    }
                } else if (p.act == 3) {                           // ReLU (BatchNorm-folded convolutions of the evaluation stem)
#pragma unroll
                    for (int e = 0; e < CW; ++e) hh[e] = fmaxf(v[e], 0.f);
                } else {
#pragma unroll
                    for (int e = 0; e < CW; ++e) hh[e] = gelu_fast(v[e]);
                }
            }
            if (has_mask) {
                const int kn = rm[q].keep - ncp;                       // kept columns of this lane's group (>= 8: all)
                const bool edge = !grp || kn < CW;
                if (__builtin_amdgcn_ballot_w64(edge) != 0) {          // some lane of the wave holds a mask boundary
#pragma unroll
                    for (int e = 0; e < CW; ++e) {
                        const bool kc = grp ? (e < kn) : kept_col(nc + e, p.n_period, rm[q].keep);
                        v[e] = kc ? v[e] : 0.f;                         // (GELU: masked hidden units: u = 0, gelu(u) = 0)
                        if constexpr (EPI == EPI_GELU) hh[e] = kc ? hh[e] : 0.f;
                    }
                }
            }
            if constexpr (EPI == EPI_GELU) {
                if (any) {
                    if (p.C2) {
                        storew<TO, CW>(p.C, oidx[q], v, vec, mok, nvalid);
                        storew<TO, CW>(p.C2, oidx[q], hh, vec, mok, nvalid);
                    } else {
                        storew<TO, CW>(p.C, oidx[q], hh, vec, mok, nvalid);      // forward-only (evaluation): gelu(u) alone
                    }
                }
            } else {
                if (has_scale) {
                    const float sc = rm[q].scale;
#pragma unroll
                    for (int e = 0; e < CW; ++e) v[e] *= sc;
                }
                if constexpr (EPI == EPI_STORE) {
                    if (has_res) {
#pragma unroll
                        for (int e = 0; e < CW; ++e) v[e] += rv[q][e];
                    }
                }
                if (any) storew<TO, CW, WT && EPI == EPI_STORE>(p.C, oidx[q], v, vec, mok, nvalid);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace vr_gemm_nt
