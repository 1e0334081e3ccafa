// Fused MFMA GEMM for gfx950:  C[M,N] = epilogue(A[M,K] * B[N,K]^T)
//
// One kernel family covers forward (NT), dgrad (B contraction-major) and wgrad (both operands
// contraction-major) of every nn.Linear / patchify-conv on the ViT-Res hot path (reference
// nets/supernet_blocks.py:37-52,102-119; nets/vit_sr_supernet.py:140,151,440-446), in two precisions:
//   bf16 : v_mfma_f32_32x32x16_bf16, fp32 accumulate           (fast path)
//   fp32 : v_mfma_f32_32x32x2_f32, exact fp32 (fmaf chain)     (parity path)
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each 64x64 = 2x2 MFMA tiles of 32x32),
// 128-byte K slices (64 bf16 / 32 fp32) double-buffered in LDS.  LDS rows are padded to 144 bytes: 9 is odd, so
// the 16 lanes of a ds_read_b128 lane group hit 16 distinct 16-B slots (conflict free).
// Contraction-major operands are transposed in registers on the global->LDS path (coalesced dword loads ->
// 16-B LDS rows), so the MFMA side is identical for all forms.
// The MFMAs compute the TRANSPOSED tile (B fragment as the first operand): each lane then owns one output row m
// and 4 consecutive columns n per accumulator quad -> 8/16-byte vector loads+stores in the epilogue and only two
// rows of per-sample metadata per lane.
// Global loads are never predicated (hipcc waits vmcnt(0) around a branched load): addresses are clamped and the
// value is selected afterwards; all addressing is hoisted out of the K loop as 32-bit byte offsets.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "common.h"
#include "../../include/vitres_hip.h"
#include "gemm_shared.h"

bool vr_gemm_nt_launch(const vr_gemm_args& a, hipStream_t stream, int n_cu);   // gemm_nt.hip
bool vr_gemm_tn_launch(const vr_gemm_args& a, hipStream_t stream, int n_cu);   // gemm_tn.hip
bool vr_gemm_tn_group_launch(const vr_gemm_args* args, int count, hipStream_t stream, int n_cu);

namespace {
using namespace vr_gemm_shared;

constexpr int LROW = 144;  // padded LDS row, bytes (128-byte K-slice payload)

// Workgroup tile = WR x WC waves, each wave MI x NI MFMA tiles of 32x32.
//   Std : 2x2 waves of 64x64   -> 128x128, 256 threads, 72 KB LDS (2 workgroups / CU)
//   Big : 2x4 waves of 128x64  -> 256x256, 512 threads, 144 KB LDS (1 workgroup / CU): half the global->LDS bytes per
//         MAC; the K loop of the Std tile is bound by the CU's load path (~17 B/clk measured), not by MFMA or HBM.
template <int WR_, int WC_, int MI_, int NI_> struct TileCfg {
    static constexpr int WR = WR_, WC = WC_, MI = MI_, NI = NI_;
    static constexpr int BM = WR * MI * 32, BN = WC * NI * 32, NTHR = WR * WC * 64;
    static constexpr int A_BYTES = BM * LROW, B_BYTES = BN * LROW, BUF_BYTES = A_BYTES + B_BYTES;
};
typedef TileCfg<2, 2, 2, 2> CfgStd;
typedef TileCfg<2, 4, 4, 2> CfgBig;

template <typename T> struct Cfg;
template <> struct Cfg<bf16_t> {
    static constexpr int BK = 64;  // elements per K slice
    static constexpr int EPC = 8;  // elements per 16-B chunk
};
template <> struct Cfg<float> {
    static constexpr int BK = 32;
    static constexpr int EPC = 4;
};

struct Stage {          // one K slice of one operand in flight: 64 bytes per thread
    uint4 v[4];
};

// ---------------------------------------------------------------------------------------------------------
// K-contiguous operand: tile row = 8 chunks of 16 B; thread t: chunk t&7, rows (t>>3) + 32h.
// ---------------------------------------------------------------------------------------------------------
template <typename T, int ROWS, int NTHR> struct LoaderN {
    static constexpr int RPP = NTHR / 8;   // rows per pass; ROWS / RPP == 4 passes for both tile configs
    static_assert(ROWS / RPP == 4, "loader shape");
    const char* base;
    uint32_t off[4];  // byte offset of (row, this thread's chunk) at k = 0; out-of-range rows are clamped to row 0
    bool rok[4];
    int kchunk;       // element offset of this thread's chunk inside a K slice

    __device__ __forceinline__ void init(const T* p, int ld, const RowMap& rm, int r0, int R, int t) {
        base = reinterpret_cast<const char*>(p);
        kchunk = (t & 7) * Cfg<T>::EPC;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int r = r0 + (t >> 3) + RPP * h;
            rok[h] = r < R;
            off[h] = (uint32_t)((map_row(rm, rok[h] ? r : 0) * (long long)ld + kchunk) * (long long)sizeof(T));
        }
    }
    __device__ __forceinline__ void gload(Stage& s, int k0, int kend, int) const {
        const bool kok = (k0 + kchunk) < kend;
        const uint32_t kb = (uint32_t)k0 * (uint32_t)sizeof(T);
        const uint32_t back = (uint32_t)(kchunk * (int)sizeof(T));
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const uint32_t o = kok ? off[h] + kb : off[h] - back;       // clamp to the row start: always readable
            uint4 x = *reinterpret_cast<const uint4*>(base + o);
            if (!(kok && rok[h])) x = make_uint4(0, 0, 0, 0);
            s.v[h] = x;
        }
    }
    __device__ __forceinline__ void lstore(const Stage& s, char* tile, int t) const {
#pragma unroll
        for (int h = 0; h < 4; ++h)
            *reinterpret_cast<uint4*>(tile + ((t >> 3) + RPP * h) * LROW + (t & 7) * 16) = s.v[h];
    }
};

// ---------------------------------------------------------------------------------------------------------
// contraction-major operand: element (kk, r) at base[map(kk)*ld + r]
//   bf16: thread t owns row pair 2*(t&63), k-octets (t>>6) + 4h (h = 0,1): 16 dword loads, 256 B per wave-load
//   fp32: thread t owns row t&127, k-quads (t>>7) + 2h (h = 0..3): 16 dword loads
// ---------------------------------------------------------------------------------------------------------
template <typename T, int ROWS, int NTHR> struct LoaderT;

template <int ROWS, int NTHR> struct LoaderT<bf16_t, ROWS, NTHR> {
    static constexpr int NP = ROWS / 2;          // row pairs
    static constexpr int OPP = NTHR / NP;        // k-octets per pass (4)
    static_assert(OPP == 4, "loader shape");
    const char* base;
    uint32_t coloff, ldb;
    bool rok, ident;
    RowMap rm;
    __device__ __forceinline__ void init(const bf16_t* p, int ld, const RowMap& m, int r0, int R, int t) {
        base = reinterpret_cast<const char*>(p);
        const int r = r0 + 2 * (t % NP);
        rok = r < R;
        coloff = (uint32_t)((rok ? r : 0) * 2);
        ldb = (uint32_t)ld * 2u;
        rm = m;
        ident = m.rpi == 0;
    }
    __device__ __forceinline__ void gload(Stage& s, int k0, int kend, int t) const {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kb = k0 + 8 * ((t / NP) + OPP * h);
            uint32_t w[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int kk = kb + e;
                const bool ok = rok && (kk < kend);
                const int kc = ok ? kk : 0;
                const uint32_t o = ident ? (uint32_t)kc * ldb : (uint32_t)(map_row(rm, kc) * (long long)ldb);
                const uint32_t x = *reinterpret_cast<const uint32_t*>(base + o + coloff);
                w[e] = ok ? x : 0u;
            }
            s.v[2 * h] = make_uint4((w[0] & 0xffffu) | (w[1] << 16), (w[2] & 0xffffu) | (w[3] << 16),
                                 (w[4] & 0xffffu) | (w[5] << 16), (w[6] & 0xffffu) | (w[7] << 16));
            s.v[2 * h + 1] = make_uint4((w[0] >> 16) | (w[1] & 0xffff0000u), (w[2] >> 16) | (w[3] & 0xffff0000u),
                                 (w[4] >> 16) | (w[5] & 0xffff0000u), (w[6] >> 16) | (w[7] & 0xffff0000u));
        }
    }
    // running sums over the contraction index of this thread's two rows (fused bias gradient of the wgrad)
    __device__ __forceinline__ void rowsum(const Stage& s, float (&rs)[2]) const {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t a[4] = {s.v[2 * h].x, s.v[2 * h].y, s.v[2 * h].z, s.v[2 * h].w};
            const uint32_t b[4] = {s.v[2 * h + 1].x, s.v[2 * h + 1].y, s.v[2 * h + 1].z, s.v[2 * h + 1].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                rs[0] += __uint_as_float(a[e] << 16) + __uint_as_float(a[e] & 0xffff0000u);
                rs[1] += __uint_as_float(b[e] << 16) + __uint_as_float(b[e] & 0xffff0000u);
            }
        }
    }
    // rs -> bias_grad[r0 + ...] : 4 waves hold different k-octets of the same 128 rows
    __device__ __forceinline__ void rowsum_flush(const float (&rs)[2], float* red, float* out, int r0, int R, int t) const {
        red[(t / NP) * ROWS + 2 * (t % NP)] = rs[0];
        red[(t / NP) * ROWS + 2 * (t % NP) + 1] = rs[1];
        __syncthreads();
        if (t < ROWS && r0 + t < R) atomicAdd(out + r0 + t, red[t] + red[ROWS + t] + red[2 * ROWS + t] + red[3 * ROWS + t]);
    }
    __device__ __forceinline__ void lstore(const Stage& s, char* tile, int t) const {
        const int r = 2 * (t % NP);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int o = (t / NP) + OPP * h;
            *reinterpret_cast<uint4*>(tile + r * LROW + o * 16) = s.v[2 * h];
            *reinterpret_cast<uint4*>(tile + (r + 1) * LROW + o * 16) = s.v[2 * h + 1];
        }
    }
};

template <int ROWS, int NTHR> struct LoaderT<float, ROWS, NTHR> {
    static constexpr int QPP = NTHR / ROWS;      // k-quads per pass (2)
    static_assert(QPP == 2, "loader shape");
    const char* base;
    uint32_t coloff, ldb;
    bool rok, ident;
    RowMap rm;
    __device__ __forceinline__ void init(const float* p, int ld, const RowMap& m, int r0, int R, int t) {
        base = reinterpret_cast<const char*>(p);
        const int r = r0 + (t % ROWS);
        rok = r < R;
        coloff = (uint32_t)((rok ? r : 0) * 4);
        ldb = (uint32_t)ld * 4u;
        rm = m;
        ident = m.rpi == 0;
    }
    __device__ __forceinline__ void gload(Stage& s, int k0, int kend, int t) const {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int kb = k0 + 4 * ((t / ROWS) + QPP * h);
            uint32_t w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kk = kb + e;
                const bool ok = rok && (kk < kend);
                const int kc = ok ? kk : 0;
                const uint32_t o = ident ? (uint32_t)kc * ldb : (uint32_t)(map_row(rm, kc) * (long long)ldb);
                const uint32_t x = *reinterpret_cast<const uint32_t*>(base + o + coloff);
                w[e] = ok ? x : 0u;
            }
            s.v[h] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    __device__ __forceinline__ void rowsum(const Stage& s, float (&rs)[2]) const {
#pragma unroll
        for (int h = 0; h < 4; ++h)
            rs[0] += (__uint_as_float(s.v[h].x) + __uint_as_float(s.v[h].y)) + (__uint_as_float(s.v[h].z) + __uint_as_float(s.v[h].w));
    }
    __device__ __forceinline__ void rowsum_flush(const float (&rs)[2], float* red, float* out, int r0, int R, int t) const {
        red[(t / ROWS) * ROWS + (t % ROWS)] = rs[0];
        __syncthreads();
        if (t < ROWS && r0 + t < R) atomicAdd(out + r0 + t, red[t] + red[ROWS + t]);
    }
    __device__ __forceinline__ void lstore(const Stage& s, char* tile, int t) const {
        const int r = t % ROWS;
#pragma unroll
        for (int h = 0; h < 4; ++h) *reinterpret_cast<uint4*>(tile + r * LROW + ((t / ROWS) + QPP * h) * 16) = s.v[h];
    }
};

// ---- LDS -> MFMA (SWAP: transposed tile, first operand = B fragment) ------------------------------------------
template <bool SWAP, int MI, int NI>
__device__ __forceinline__ void mma_tile(f32x16 (&acc)[MI][NI], const char* As, const char* Bs, int wm, int wn, int lane,
                                         bf16_t*) {
    typedef __bf16 bfv8 __attribute__((ext_vector_type(8)));
    const int rr = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        bfv8 a[MI], b[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
            a[i] = *reinterpret_cast<const bfv8*>(As + (wm * MI * 32 + i * 32 + rr) * LROW + (ks * 2 + kh) * 16);
#pragma unroll
        for (int j = 0; j < NI; ++j)
            b[j] = *reinterpret_cast<const bfv8*>(Bs + (wn * NI * 32 + j * 32 + rr) * LROW + (ks * 2 + kh) * 16);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
                acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}
template <bool SWAP, int MI, int NI>
__device__ __forceinline__ void mma_tile(f32x16 (&acc)[MI][NI], const char* As, const char* Bs, int wm, int wn, int lane,
                                         float*) {
    const int rr = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        float a[MI], b[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
            a[i] = *reinterpret_cast<const float*>(As + (wm * MI * 32 + i * 32 + rr) * LROW + (ks * 2 + kh) * 4);
#pragma unroll
        for (int j = 0; j < NI; ++j)
            b[j] = *reinterpret_cast<const float*>(Bs + (wn * NI * 32 + j * 32 + rr) * LROW + (ks * 2 + kh) * 4);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
                acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(b[j], a[i], acc[i][j], 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}

// 4 consecutive output elements
template <typename TO> __device__ __forceinline__ void store4(void* base, long long idx, const float (&v)[4], bool vec,
                                                              const bool (&ok)[4]) {
    TO* p = reinterpret_cast<TO*>(base) + idx;
    if (vec) {
        if constexpr (sizeof(TO) == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        else *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (ok[e]) Elem<TO>::st(p + e, v[e]);
    }
}
template <typename TI> __device__ __forceinline__ void load4(const void* base, long long idx, float (&v)[4], bool vec,
                                                             int nvalid) {
    const TI* p = reinterpret_cast<const TI*>(base) + idx;
    if (vec) {
        if constexpr (sizeof(TI) == 4) {
            const float4 x = *reinterpret_cast<const float4*>(p);
            v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
        } else {
            const uint2 u = *reinterpret_cast<const uint2*>(p);
            v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
            v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = Elem<TI>::ld(p + (e < nvalid ? e : 0));
    }
}

// Epilogue of one lane = 2 output rows (i) x 8 column quads (j, g).  ALL global loads (bias, pos-embed, residual,
// GELU pre-activation) are issued before the first store: on gfx950 vmcnt counts stores too, so a load issued after
// a store cannot be waited for without draining that store (measured: 16 interleaved load/store pairs cost 14 us
// of a 50 us kernel).
template <typename T, typename TO, int EPI, int MI, int NI>
__device__ __forceinline__ void epilogue_all(const vr_gemm_args& p, f32x16 (&acc)[MI][NI], int m0, int n0, int wm, int wn,
                                             int lane, bool first_split) {
    const RowMap cmap = {p.c_map.rpi, p.c_map.rps, p.c_map.off};
    const bool vec_ok = (p.ldc % 4 == 0) && (p.N % 4 == 0) && (EPI != EPI_DGELU || p.ldu % 4 == 0);
    const bool has_bias = (EPI == EPI_STORE || EPI == EPI_GELU) && p.bias && first_split;
    const bool has_pos = (EPI == EPI_STORE) && p.pos && first_split;
    const bool has_res = (EPI == EPI_STORE) && p.resid;
    int nq[NI][4], nvq[NI][4];
    bool vq[NI][4];
    float bv[NI][4][4];
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + wn * NI * 32 + j * 32 + 8 * g + 4 * (lane >> 5);
            const int nvalid = min(4, p.N - n);
            nvq[j][g] = nvalid;
            nq[j][g] = nvalid > 0 ? n : 0;
            vq[j][g] = vec_ok && nvalid == 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[j][g][e] = 0.f;
            if (has_bias) load4<float>(p.bias, nq[j][g], bv[j][g], vq[j][g], nvalid > 0 ? nvalid : 1);
        }
    bool mok[MI];
    int mloc[MI], keep[MI];
    long long orow[MI];
    float sc[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * MI * 32 + i * 32 + (lane & 31);
        mok[i] = m < p.M;
        const int mc = mok[i] ? m : 0;
        const int sample = p.rows_in > 0 ? mc / p.rows_in : 0;
        mloc[i] = p.rows_in > 0 ? mc - sample * p.rows_in : mc;
        orow[i] = map_row(cmap, mc);
        sc[i] = p.scale ? p.scale[sample] : 1.0f;
        keep[i] = p.keep_n ? p.keep_n[sample] : (1 << 30);
    }
    if (has_pos) {                                    // rare (patch embedding / SR): added to the accumulators up front
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float pv[4];
                    load4<float>(p.pos, (long long)mloc[i] * p.N + nq[j][g], pv, vq[j][g], nvq[j][g] > 0 ? nvq[j][g] : 1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] += pv[e];
                }
    }
    // one row (I) at a time: its side-input quads are loaded together, then its quads are stored (one vmcnt drain per
    // row instead of one per quad).  Rows are expanded through an index_sequence so that acc[I] is a compile-time index
    // even when the unroller refuses a plain loop (acc would otherwise spill to scratch).
    auto row = [&](auto Ic) {
        constexpr int i = decltype(Ic)::value;
        float rv[NI][4][4];
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nv = nvq[j][g] > 0 ? nvq[j][g] : 1;
#pragma unroll
                for (int e = 0; e < 4; ++e) rv[j][g][e] = 0.f;
                if constexpr (EPI == EPI_DGELU) load4<T>(p.dact_u, orow[i] * p.ldu + nq[j][g], rv[j][g], vq[j][g], nv);
                if constexpr (EPI == EPI_STORE) {
                    if (has_res) load4<float>(p.resid, orow[i] * p.ldc + nq[j][g], rv[j][g], vq[j][g], nv);
                }
            }
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nc = nq[j][g];
                const bool any = mok[i] && nvq[j][g] > 0;
                const long long oidx = orow[i] * p.ldc + nc;
                float v[4];
                bool ok[4], kc[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[i][j][4 * g + e] + bv[j][g][e];
                    ok[e] = mok[i] && (e < nvq[j][g]);
                    kc[e] = kept_col(nc + e, p.n_period, keep[i]);
                }
                if constexpr (EPI == EPI_GELU) {
                    float h[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = kc[e] ? v[e] : 0.f;       // masked hidden units: u = 0, gelu(u) = 0 (their K loop may be skipped)
                        h[e] = kc[e] ? (p.act == 3 ? fmaxf(v[e], 0.f) : (sizeof(T) == 2 ? gelu_fast(v[e]) : gelu_f(v[e]))) : 0.f;
                        if (p.act == 2) v[e] = kc[e] ? (sizeof(T) == 2 ? dgelu_fast(v[e]) : dgelu_f(v[e])) : 0.f;   // C = gelu'(u)
                    }
                    if (any) {
                        if (p.C2) {
                            store4<TO>(p.C, oidx, v, vq[j][g], ok);
                            store4<TO>(p.C2, oidx, h, vq[j][g], ok);
                        } else {
                            store4<TO>(p.C, oidx, h, vq[j][g], ok);      // forward-only: the activation alone
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if constexpr (EPI == EPI_DGELU) v[e] *= p.act == 2 ? rv[j][g][e] : (sizeof(T) == 2 ? dgelu_fast(rv[j][g][e]) : dgelu_f(rv[j][g][e]));
                        v[e] = kc[e] ? v[e] * sc[i] : 0.f;
                        if constexpr (EPI == EPI_STORE) v[e] += rv[j][g][e];
                    }
                    if (any) store4<TO>(p.C, oidx, v, vq[j][g], ok);
                }
            }
    };
    [&]<int... Is>(std::integer_sequence<int, Is...>) { (row(std::integral_constant<int, Is>{}), ...); }
    (std::make_integer_sequence<int, MI>{});
}

// Epilogue through LDS (128x128 tile): the accumulators are parked in LDS as fp32 [128][132]; every thread then owns a
// fixed group of CW consecutive columns (16 bytes of OUTPUT: 8 bf16 or 4 fp32 -- stores are issue-bound, so every store
// instruction must carry 16 B per lane) and walks down the rows: each wave-instruction reads / writes whole rows of the
// tile, the bias group is loaded once per thread, and all side inputs of a batch of 8 rows are in flight before the
// first store of the batch (vmcnt counts stores: a load issued after a store cannot be waited for without draining it).
constexpr int CROW = 132;  // floats per parked row (528 B = 33 x 16 B, odd -> conflict-free float4 accesses)
template <typename T, typename TO, int EPI>
__device__ __forceinline__ void epilogue_lds(const vr_gemm_args& p, f32x16 (&acc)[2][2], float* Ct, int m0, int n0, int wm,
                                             int wn, int t, bool first_split) {
    constexpr int CW = sizeof(TO) == 2 ? 8 : 4;      // columns per thread
    constexpr int TPR = 128 / CW;                    // threads per tile row
    constexpr int RPP = 256 / TPR;                   // rows per pass
    constexpr int NPASS = 128 / RPP;                 // 8 (bf16) or 16 (fp32) rows per thread
    constexpr int BATCH = 8;
    const int lane = t & 63;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ml = wm * 64 + i * 32 + (lane & 31), nl = wn * 64 + j * 32 + 8 * g + 4 * (lane >> 5);
                *reinterpret_cast<float4*>(Ct + ml * CROW + nl) =
                    make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
            }
    __syncthreads();
    const RowMap cmap = {p.c_map.rpi, p.c_map.rps, p.c_map.off};
    const bool vec_ok = (p.ldc % CW == 0) && (p.N % CW == 0) && (EPI != EPI_DGELU || p.ldu % CW == 0);
    const int nl = CW * (t % TPR);
    const int n = n0 + nl;
    const int nvalid = min(CW, p.N - n);
    const int nc = nvalid > 0 ? n : 0;
    const int nv = nvalid > 0 ? nvalid : 1;
    const bool vec = vec_ok && nvalid == CW;
    float bv[CW];
#pragma unroll
    for (int e = 0; e < CW; ++e) bv[e] = 0.f;
    if ((EPI == EPI_STORE || EPI == EPI_GELU) && p.bias && first_split) loadw<float, CW>(p.bias, nc, bv, vec, nv);
    const bool has_pos = (EPI == EPI_STORE) && p.pos && first_split;
    const bool has_res = (EPI == EPI_STORE) && p.resid;
#pragma unroll
    for (int b0 = 0; b0 < NPASS; b0 += BATCH) {
        bool mok[BATCH];
        int keep[BATCH];
        long long orow[BATCH];
        float sc[BATCH], rv[BATCH][CW], pv[BATCH][CW];
#pragma unroll
        for (int q = 0; q < BATCH; ++q) {
            const int m = m0 + (t / TPR) + RPP * (b0 + q);
            mok[q] = m < p.M;
            const int mc = mok[q] ? m : 0;
            const int sample = p.rows_in > 0 ? mc / p.rows_in : 0;
            const int mloc = p.rows_in > 0 ? mc - sample * p.rows_in : mc;
            orow[q] = map_row(cmap, mc);
            sc[q] = p.scale ? p.scale[sample] : 1.0f;
            keep[q] = p.keep_n ? p.keep_n[sample] : (1 << 30);
#pragma unroll
            for (int e = 0; e < CW; ++e) { rv[q][e] = 0.f; pv[q][e] = 0.f; }
            if constexpr (EPI == EPI_DGELU) loadw<T, CW>(p.dact_u, orow[q] * p.ldu + nc, rv[q], vec, nv);
            if constexpr (EPI == EPI_STORE) {
                if (has_res) loadw<float, CW>(p.resid, orow[q] * p.ldc + nc, rv[q], vec, nv);
                if (has_pos) loadw<float, CW>(p.pos, (long long)mloc * p.N + nc, pv[q], vec, nv);
            }
        }
#pragma unroll
        for (int q = 0; q < BATCH; ++q) {
            const int rl = (t / TPR) + RPP * (b0 + q);
            float v[CW];
#pragma unroll
            for (int h = 0; h < CW / 4; ++h) {
                const float4 a4 = *reinterpret_cast<const float4*>(Ct + rl * CROW + nl + 4 * h);
                v[4 * h] = a4.x; v[4 * h + 1] = a4.y; v[4 * h + 2] = a4.z; v[4 * h + 3] = a4.w;
            }
            bool kc[CW];
#pragma unroll
            for (int e = 0; e < CW; ++e) {
                v[e] += bv[e] + pv[q][e];
                kc[e] = kept_col(nc + e, p.n_period, keep[q]);
            }
            const bool any = mok[q] && nvalid > 0;
            const long long oidx = orow[q] * p.ldc + nc;
            if constexpr (EPI == EPI_GELU) {
                float h[CW];
#pragma unroll
                for (int e = 0; e < CW; ++e) {
                    v[e] = kc[e] ? v[e] : 0.f;       // masked hidden units: u = 0, gelu(u) = 0 (their K loop may be skipped)
                    h[e] = kc[e] ? (p.act == 3 ? fmaxf(v[e], 0.f) : (sizeof(T) == 2 ? gelu_fast(v[e]) : gelu_f(v[e]))) : 0.f;
                    if (p.act == 2) v[e] = kc[e] ? (sizeof(T) == 2 ? dgelu_fast(v[e]) : dgelu_f(v[e])) : 0.f;   // C = gelu'(u)
                }
                if (any) {
                    if (p.C2) {
                        storew<TO, CW>(p.C, oidx, v, vec, mok[q], nvalid);
                        storew<TO, CW>(p.C2, oidx, h, vec, mok[q], nvalid);
                    } else {
                        storew<TO, CW>(p.C, oidx, h, vec, mok[q], nvalid);      // forward-only: the activation alone
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < CW; ++e) {
                    if constexpr (EPI == EPI_DGELU) v[e] *= p.act == 2 ? rv[q][e] : (sizeof(T) == 2 ? dgelu_fast(rv[q][e]) : dgelu_f(rv[q][e]));
                    v[e] = kc[e] ? v[e] * sc[q] : 0.f;
                    if constexpr (EPI == EPI_STORE) v[e] += rv[q][e];
                }
                if (any) storew<TO, CW>(p.C, oidx, v, vec, mok[q], nvalid);
            }
        }
    }
}

// Persistent workgroups: gridDim.x = (resident workgroups per CU) x CUs; each walks the output tiles
// tile, tile + gridDim.x, ...  The first K slice of the NEXT tile is fetched into registers before the epilogue of the
// current one, so its HBM latency hides behind the stores and the chip never runs in lock-step load / store phases.
template <typename T, bool TA, bool TB, typename TO, int EPI, typename TC>
__global__ __launch_bounds__(TC::NTHR, 2) void gemm_kernel(const vr_gemm_args p) {
    constexpr int BM = TC::BM, BN = TC::BN, NTHR = TC::NTHR, MI = TC::MI, NI = TC::NI;
    constexpr int TILE_BYTES = TC::A_BYTES, BUF = TC::BUF_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];  // [A0 B0][A1 B1]
    constexpr int BK = Cfg<T>::BK;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / TC::WC, wn = wave % TC::WC;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int total = tiles_n * tiles_m * p.split_k;
    int kper = p.K;
    if (p.split_k > 1) kper = ((p.K + p.split_k - 1) / p.split_k + BK - 1) / BK * BK;
    const RowMap amap = {p.a_map.rpi, p.a_map.rps, p.a_map.off};
    const RowMap bmap = {p.b_map.rpi, p.b_map.rps, p.b_map.off};

    typename std::conditional<TA, LoaderT<T, BM, NTHR>, LoaderN<T, BM, NTHR>>::type la;
    typename std::conditional<TB, LoaderT<T, BN, NTHR>, LoaderN<T, BN, NTHR>>::type lb;
    Stage sa, sb;
    float rs[2] = {0.f, 0.f};
    // ---- state of the tile being set up / computed ----
    int m0 = 0, n0 = 0, z = 0, kbeg = 0, kend = 0, ntiles = 0, kmax = 0;
    bool n_any = true, want_bg = false;

    // Workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2): give XCD x one contiguous run of the
    // n-fastest tile order, so that the tiles sharing an A row panel run on the same L2 instead of fetching it 8 times.
    const int xq = total >> 3, xr = total & 7;
    const bool xcd_map = !(p.sched & 2) && total >= 16;
    auto setup = [&](int ptile) {
        int tile = ptile;
        if (xcd_map) {
            const int x = ptile & 7, i = ptile >> 3;
            tile = x * xq + min(x, xr) + i;
        }
        const int tn = tile % tiles_n, rest = tile / tiles_n;
        const int tm = rest % tiles_m;
        z = rest / tiles_m;
        m0 = tm * BM;
        n0 = tn * BN;
        kbeg = z * kper;
        kend = min(p.K, kbeg + kper);
        ntiles = kbeg < kend ? (kend - kbeg + BK - 1) / BK : 0;
        // masked-work skipping: which samples does this tile touch, and how much of K / N do they keep
        kmax = 1 << 30;
        n_any = true;
        if ((p.keep_k || p.keep_n) && ntiles > 0) {
            int s_lo = 0, s_hi = 0;
            if (p.rows_in > 0) {
                if constexpr (TA) { s_lo = kbeg / p.rows_in; s_hi = (kend - 1) / p.rows_in; }
                else { s_lo = m0 / p.rows_in; s_hi = (min(m0 + BM, p.M) - 1) / p.rows_in; }
            }
            kmax = max_keep(p.keep_k, s_lo, s_hi, 1 << 30);
            const int nmax = max_keep(p.keep_n, s_lo, s_hi, 1 << 30);
            n_any = range_has_kept(n0, BN, p.n_period, nmax);
            if constexpr (TA) {
                // wgrad: keep_k bounds the kept output rows; a tile without kept rows or columns adds exactly zero
                if (!n_any || !range_has_kept(m0, BM, p.k_period, kmax)) ntiles = 0;
            }
        }
        la.init(reinterpret_cast<const T*>(p.A), p.lda, amap, m0, p.M, t);
        lb.init(reinterpret_cast<const T*>(p.B), p.ldb, bmap, n0, p.N, t);
        rs[0] = 0.f;
        rs[1] = 0.f;
        want_bg = TA && p.bias_grad != nullptr && tn == 0 && ntiles > 0;
    };
    auto tile_live = [&](int kt) -> bool {
        if constexpr (TA) return true;
        else return n_any && (p.keep_k == nullptr || range_has_kept(kbeg + kt * BK, BK, p.k_period, kmax));
    };
    auto next_live = [&](int kt) -> int {
        while (kt < ntiles && !tile_live(kt)) ++kt;
        return kt;
    };
    // slice i is computed from LDS while slice i+1 is in flight into registers (a second slice in flight was measured:
    // no gain -- the K loop is bound by the CU's load path, not by latency -- and it costs 32 VGPRs)
    auto fetch = [&](int kt) {
        la.gload(sa, kbeg + kt * BK, kend, t);
        lb.gload(sb, kbeg + kt * BK, kend, t);
        if constexpr (TA) { if (want_bg) la.rowsum(sa, rs); }
    };

    int tile = blockIdx.x;
    if (tile >= total) return;
    setup(tile);
    int kt = next_live(0);
    if (kt < ntiles) fetch(kt);
    for (;;) {
        f32x16 acc[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        if (kt < ntiles) {
            la.lstore(sa, smem, t);
            lb.lstore(sb, smem + TILE_BYTES, t);
        }
        __syncthreads();
        int cur = 0;
        while (kt < ntiles) {
            const int nxt = next_live(kt + 1);
            const bool more = nxt < ntiles;
            if (more) fetch(nxt);
            const char* As = smem + cur * BUF;
            mma_tile<EPI != EPI_ATOMIC, MI, NI>(acc, As, As + TILE_BYTES, wm, wn, lane, (T*)nullptr);
            if (more) {
                char* Ad = smem + (cur ^ 1) * BUF;
                la.lstore(sa, Ad, t);
                lb.lstore(sb, Ad + TILE_BYTES, t);
            }
            __syncthreads();
            cur ^= 1;
            kt = nxt;
        }
        if constexpr (TA) {
            if (want_bg) la.rowsum_flush(rs, reinterpret_cast<float*>(smem), p.bias_grad, m0, p.M, t);
        }
        // coordinates of the finished tile, then start the next tile's first slice before storing this one
        const int em0 = m0, en0 = n0;
        const bool efirst = z == 0, edone = ntiles > 0 || !TA;
        const int nxt_tile = tile + gridDim.x;
        const bool has_next = nxt_tile < total;
        if (has_next) {
            setup(nxt_tile);
            kt = next_live(0);
            if (kt < ntiles) fetch(kt);
        }
        if constexpr (EPI == EPI_ATOMIC) {
            // natural accumulator layout: a half-wave adds 32 consecutive fp32 of one output row (128 B) per instruction
            if (edone) {
                float* C = reinterpret_cast<float*>(p.C);
                const RowMap cm = {p.c_map.rpi, p.c_map.rps, p.c_map.off};
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = em0 + wm * MI * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        const long long orow = map_row(cm, m < p.M ? m : 0);
#pragma unroll
                        for (int j = 0; j < NI; ++j) {
                            const int n = en0 + wn * NI * 32 + j * 32 + (lane & 31);
                            if (m < p.M && n < p.N) atomicAdd(C + orow * p.ldc + n, acc[i][j][r]);
                        }
                    }
            }
        } else {
            // transposed accumulators: lane owns rows m_i = ... + (lane&31) and, per accumulator quad g, the 4
            // consecutive columns n = n0 + wn*NI*32 + j*32 + 8*g + 4*(lane>>5) + e
            if constexpr (BM == 128 && BN == 128 && sizeof(TO) == 4) {
                // fp32 outputs (residual-stream updates): parked in LDS, whole-row 16-byte accesses (measured 64 -> 46 us);
                // bf16 outputs are faster straight from registers (measured: the LDS round trip costs more than it saves)
                epilogue_lds<T, TO, EPI>(p, acc, reinterpret_cast<float*>(smem), em0, en0, wm, wn, t, efirst);
            } else {
                epilogue_all<T, TO, EPI, MI, NI>(p, acc, em0, en0, wm, wn, lane, efirst);
            }
        }
        if (!has_next) break;
        if constexpr (EPI != EPI_ATOMIC && BM == 128 && BN == 128 && sizeof(TO) == 4) __syncthreads();   // Ct is re-used by the next lstore
        tile = nxt_tile;
    }
}

inline int cu_count() {
    static int n = 0;           // read once per process (one process per GPU)
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}

template <typename T, bool TA, bool TB, typename TO, int EPI>
void launch1(vr_gemm_args a, hipStream_t stream) {
    // measured (tools/gemm_probe.py): the 256x256 tile wins for the split-K weight gradients only -- for the forward /
    // dgrad shapes (K <= 1024) its 1 workgroup / CU residency and tail cost more than the halved load traffic buys
    const long long big_tiles = (long long)((a.M + 255) / 256) * ((a.N + 255) / 256);
    const int want_split = a.atomic ? max(1, min(a.K / 512, 1 << 16)) : 1;
    const bool shared = (a.sched & 1) != 0;
    const bool big = !shared && a.atomic && (a.M >= 192 && a.N >= 192) && big_tiles * want_split >= 192;
    const long long tiles = big ? big_tiles : (long long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    if (a.atomic && a.split_k <= 0) a.split_k = (int)max(1LL, min((long long)want_split, max(1LL, 1024 / tiles)));
    const long long total = tiles * a.split_k;
    if (big) {
        const int grid = (int)min(total, (long long)cu_count());   // persistent workgroups; 144 KB LDS: one per CU
        hipLaunchKernelGGL((gemm_kernel<T, TA, TB, TO, EPI, CfgBig>), dim3(grid), dim3(CfgBig::NTHR), 0, stream, a);
    } else {
        const int grid = (int)(!shared ? min(total, 2LL * cu_count()) : total);        // 72 KB LDS, <= 256 VGPR: two per CU
        hipLaunchKernelGGL((gemm_kernel<T, TA, TB, TO, EPI, CfgStd>), dim3(grid), dim3(CfgStd::NTHR), 0, stream, a);
    }
}

template <typename T>
int launch(const vr_gemm_args& a, hipStream_t stream) {
    const bool of32 = a.out_dtype == VR_F32;
    if (a.atomic) {                                             // weight gradients (fp32 accumulate)
        if (a.a_trans && a.b_trans) launch1<T, true, true, float, EPI_ATOMIC>(a, stream);
        else return VR_EUNSUPPORTED;
    } else if (a.a_trans) {
        return VR_EUNSUPPORTED;
    } else if (a.act == 1 || a.act == 3 || (a.act == 2 && !a.dact_u)) {
        if (a.b_trans || of32 != (sizeof(T) == 4)) return VR_EUNSUPPORTED;
        launch1<T, false, false, T, EPI_GELU>(a, stream);
    } else if (a.dact_u) {
        if (of32 != (sizeof(T) == 4)) return VR_EUNSUPPORTED;
        if (a.b_trans) launch1<T, false, true, T, EPI_DGELU>(a, stream);
        else launch1<T, false, false, T, EPI_DGELU>(a, stream);
    } else if (!a.b_trans) {
        if (of32) launch1<T, false, false, float, EPI_STORE>(a, stream);
        else launch1<T, false, false, bf16_t, EPI_STORE>(a, stream);
    } else {
        if (of32) launch1<T, false, true, float, EPI_STORE>(a, stream);
        else launch1<T, false, true, bf16_t, EPI_STORE>(a, stream);
    }
    VR_CHECK_LAUNCH();
    return VR_OK;
}

}  // namespace

// argument validation shared by vr_gemm and vr_gemm_group (normalises split_k in place)
static int gemm_validate(vr_gemm_args& a) {
    if (!a.A || !a.B || !a.C) return VR_EINVAL;
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return VR_EINVAL;
    if (a.atomic < 0 || a.atomic > 2) return VR_EINVAL;
    if (a.split_k < 0 || (!a.atomic && a.split_k == 0)) a.split_k = 1;   // 0 with atomic = choose automatically
    if (a.atomic == 2) {                                                  // store form of the weight gradient: one workgroup per tile
        if (!(a.a_trans && a.b_trans) || a.split_k > 1) return VR_EINVAL;
    }
    if (a.split_k > 1 && !a.atomic) return VR_EINVAL;
    if (a.atomic && a.out_dtype != VR_F32) return VR_EINVAL;
    if (a.bias_grad && !(a.a_trans && a.atomic)) return VR_EINVAL;
    if (a.act < 0 || a.act > 3 || (a.act == 3 && (a.C2 || a.dact_u))) return VR_EINVAL;
    if (a.in_dtype != VR_F32 && a.in_dtype != VR_BF16) return VR_EUNSUPPORTED;
    if (a.out_dtype != VR_F32 && a.out_dtype != VR_BF16) return VR_EUNSUPPORTED;
    const int epc = a.in_dtype == VR_BF16 ? 8 : 4;
    const int esz = a.in_dtype == VR_BF16 ? 2 : 4;
    // K-contiguous operands are fetched in 16-byte chunks (rows must be readable up to roundup(K, chunk): the caller
    // zero-pads); contraction-major ones in dwords.  Offsets are 32-bit: operands must be < 4 GiB.
    const int kpad = (a.K + epc - 1) / epc * epc;
    if (!a.a_trans) {
        if (a.lda % epc || a.lda < kpad || ((uintptr_t)a.A & 15)) return VR_EALIGN;
    } else {
        if ((a.M * esz) % 4 || (a.lda * esz) % 4 || ((uintptr_t)a.A & 3)) return VR_EALIGN;
    }
    if (!a.b_trans) {
        if (a.ldb % epc || a.ldb < kpad || ((uintptr_t)a.B & 15)) return VR_EALIGN;
    } else {
        if ((a.N * esz) % 4 || (a.ldb * esz) % 4 || ((uintptr_t)a.B & 3)) return VR_EALIGN;
    }
    // the vector epilogue needs 16-byte aligned output / side-input base pointers
    if (((uintptr_t)a.C & 15) || (a.C2 && ((uintptr_t)a.C2 & 15)) || (a.resid && ((uintptr_t)a.resid & 15)) ||
        (a.bias && ((uintptr_t)a.bias & 15)) || (a.pos && ((uintptr_t)a.pos & 15)) ||
        (a.dact_u && ((uintptr_t)a.dact_u & 15)))
        return VR_EALIGN;
    if (a.in_dtype == VR_F32 && a.out_dtype == VR_BF16) return VR_EUNSUPPORTED;
    if (a.ws && (((uintptr_t)a.ws & 15) || a.ws_bytes < 0)) return VR_EALIGN;
    return VR_OK;
}

// tickets (16 KB) + four 128 x 128 fp32 slabs per CU: the workspace of the K-split kernels (gemm_ntk.hip SPLIT, vr_gemm_args.k_shares)
extern "C" int vr_gemm_ws_bytes(void) { return (int)(4096 * 4 + (size_t)cu_count() * 2 * (256 * 128) * sizeof(float)); }

extern "C" int vr_gemm(const vr_gemm_args* args, vr_stream_t stream) {
    if (!args) return VR_EINVAL;
    vr_gemm_args a = *args;
    const int vrc = gemm_validate(a);
    if (vrc != VR_OK) return vrc;
    if (!(a.sched & 4) && vr_gemm_nt_launch(a, (hipStream_t)stream, cu_count())) {
        VR_CHECK_LAUNCH();
        return VR_OK;
    }
    if (!(a.sched & 4) && vr_gemm_tn_launch(a, (hipStream_t)stream, cu_count())) {
        VR_CHECK_LAUNCH();
        return VR_OK;
    }
    if (a.atomic == 2) return VR_EUNSUPPORTED;       // the store form exists on the bf16 LDS-DMA weight-gradient kernel only
    // sched bit 0x80000 (masked tiles of the operands may be unwritten): only the group-pure bf16 kernels may read them
    if ((a.sched & 0x80000) && a.m_groups > 1 && (a.keep_k || (a.a_trans && a.keep_n))) return VR_EUNSUPPORTED;
    if (a.in_dtype == VR_BF16) return launch<bf16_t>(a, (hipStream_t)stream);
    return launch<float>(a, (hipStream_t)stream);
}

bool vr_gemm_ntk_launch(const vr_gemm_args& a, hipStream_t stream, int n_cu, const vr_ln_epilogue* ln);   // gemm_ntk.hip

// vr_gemm with the LayerNorm that consumes its fp32 result folded into the launch (see include/vitres_hip.h)
extern "C" int vr_gemm_ln_fold(const vr_gemm_args* args, const vr_ln_epilogue* ln, vr_stream_t stream) {
    if (!args || !ln) return VR_EINVAL;
    vr_gemm_args a = *args;
    const int vrc = gemm_validate(a);
    if (vrc != VR_OK) return vrc;
    if (a.in_dtype != VR_BF16 || a.a_trans || a.atomic || a.split_k > 1 || a.bias_grad || a.pos) return VR_EUNSUPPORTED;
    if (((uintptr_t)ln->y & 15) || ((uintptr_t)ln->w & 15) || ((uintptr_t)ln->b & 15)) return VR_EALIGN;
    if (!vr_gemm_ntk_launch(a, (hipStream_t)stream, cu_count(), ln)) return VR_EUNSUPPORTED;
    VR_CHECK_LAUNCH();
    return VR_OK;
}

extern "C" int vr_gemm_group(const vr_gemm_args* args, int count, vr_stream_t stream) {
    if (!args || count <= 0) return VR_EINVAL;
    if (count >= 2 && count <= 4) {
        vr_gemm_args v[4];
        bool ok = true;
        for (int i = 0; i < count && ok; ++i) {
            v[i] = args[i];
            const int want_auto = v[i].atomic && v[i].split_k == 0;
            if (gemm_validate(v[i]) != VR_OK || (v[i].sched & 4)) ok = false;
            if (want_auto) v[i].split_k = 0;            // (validation keeps 0 = automatic for atomic forms)
        }
        if (ok && vr_gemm_tn_group_launch(v, count, (hipStream_t)stream, cu_count())) {
            VR_CHECK_LAUNCH();
            return VR_OK;
        }
    }
    for (int i = 0; i < count; ++i) {
        const int rc = vr_gemm(args + i, stream);
        if (rc != VR_OK) return rc;
    }
    return VR_OK;
}
