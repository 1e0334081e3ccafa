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
#include <type_traits>

#include "common.h"
#include "../../include/vitres_hip.h"

namespace {

constexpr int BM = 128, BN = 128, NTHR = 256;
constexpr int LROW = 144;  // padded LDS row, bytes (128-byte K-slice payload)
constexpr int TILE_BYTES = BM * LROW;

template <typename T> struct Cfg;
template <> struct Cfg<bf16_t> {
    static constexpr int BK = 64;  // elements per K slice
    static constexpr int EPC = 8;  // elements per 16-B chunk
};
template <> struct Cfg<float> {
    static constexpr int BK = 32;
    static constexpr int EPC = 4;
};

// ---------------------------------------------------------------------------------------------------------
// K-contiguous operand: tile row = 8 chunks of 16 B; thread t: chunk t&7, rows (t>>3) + 32h.
// ---------------------------------------------------------------------------------------------------------
template <typename T> struct LoaderN {
    const char* base;
    uint32_t off[4];  // byte offset of (row, this thread's chunk) at k = 0; out-of-range rows are clamped to row 0
    bool rok[4];
    int kchunk;       // element offset of this thread's chunk inside a K slice
    uint4 v[4];

    __device__ __forceinline__ void init(const T* p, int ld, const RowMap& rm, int r0, int R, int t) {
        base = reinterpret_cast<const char*>(p);
        kchunk = (t & 7) * Cfg<T>::EPC;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int r = r0 + (t >> 3) + 32 * h;
            rok[h] = r < R;
            off[h] = (uint32_t)((map_row(rm, rok[h] ? r : 0) * (long long)ld + kchunk) * (long long)sizeof(T));
        }
    }
    __device__ __forceinline__ void gload(int k0, int kend, int) {
        const bool kok = (k0 + kchunk) < kend;
        const uint32_t kb = (uint32_t)k0 * (uint32_t)sizeof(T);
        const uint32_t back = (uint32_t)(kchunk * (int)sizeof(T));
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const uint32_t o = kok ? off[h] + kb : off[h] - back;       // clamp to the row start: always readable
            uint4 x = *reinterpret_cast<const uint4*>(base + o);
            if (!(kok && rok[h])) x = make_uint4(0, 0, 0, 0);
            v[h] = x;
        }
    }
    __device__ __forceinline__ void lstore(char* tile, int t) const {
#pragma unroll
        for (int h = 0; h < 4; ++h)
            *reinterpret_cast<uint4*>(tile + ((t >> 3) + 32 * h) * LROW + (t & 7) * 16) = v[h];
    }
};

// ---------------------------------------------------------------------------------------------------------
// contraction-major operand: element (kk, r) at base[map(kk)*ld + r]
//   bf16: thread t owns row pair 2*(t&63), k-octets (t>>6) + 4h (h = 0,1): 16 dword loads, 256 B per wave-load
//   fp32: thread t owns row t&127, k-quads (t>>7) + 2h (h = 0..3): 16 dword loads
// ---------------------------------------------------------------------------------------------------------
template <typename T> struct LoaderT;

template <> struct LoaderT<bf16_t> {
    const char* base;
    uint32_t coloff, ldb;
    bool rok, ident;
    RowMap rm;
    uint4 v[2][2];
    __device__ __forceinline__ void init(const bf16_t* p, int ld, const RowMap& m, int r0, int R, int t) {
        base = reinterpret_cast<const char*>(p);
        const int r = r0 + 2 * (t & 63);
        rok = r < R;
        coloff = (uint32_t)((rok ? r : 0) * 2);
        ldb = (uint32_t)ld * 2u;
        rm = m;
        ident = m.rpi == 0;
    }
    __device__ __forceinline__ void gload(int k0, int kend, int t) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kb = k0 + 8 * ((t >> 6) + 4 * h);
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
            v[h][0] = make_uint4((w[0] & 0xffffu) | (w[1] << 16), (w[2] & 0xffffu) | (w[3] << 16),
                                 (w[4] & 0xffffu) | (w[5] << 16), (w[6] & 0xffffu) | (w[7] << 16));
            v[h][1] = make_uint4((w[0] >> 16) | (w[1] & 0xffff0000u), (w[2] >> 16) | (w[3] & 0xffff0000u),
                                 (w[4] >> 16) | (w[5] & 0xffff0000u), (w[6] >> 16) | (w[7] & 0xffff0000u));
        }
    }
    // running sums over the contraction index of this thread's two rows (fused bias gradient of the wgrad)
    __device__ __forceinline__ void rowsum(float (&rs)[2]) const {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t a[4] = {v[h][0].x, v[h][0].y, v[h][0].z, v[h][0].w};
            const uint32_t b[4] = {v[h][1].x, v[h][1].y, v[h][1].z, v[h][1].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                rs[0] += __uint_as_float(a[e] << 16) + __uint_as_float(a[e] & 0xffff0000u);
                rs[1] += __uint_as_float(b[e] << 16) + __uint_as_float(b[e] & 0xffff0000u);
            }
        }
    }
    // rs -> bias_grad[r0 + ...] : 4 waves hold different k-octets of the same 128 rows
    __device__ __forceinline__ void rowsum_flush(const float (&rs)[2], float* red, float* out, int r0, int R, int t) const {
        red[(t >> 6) * 128 + 2 * (t & 63)] = rs[0];
        red[(t >> 6) * 128 + 2 * (t & 63) + 1] = rs[1];
        __syncthreads();
        if (t < 128 && r0 + t < R) atomicAdd(out + r0 + t, red[t] + red[128 + t] + red[256 + t] + red[384 + t]);
    }
    __device__ __forceinline__ void lstore(char* tile, int t) const {
        const int r = 2 * (t & 63);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int o = (t >> 6) + 4 * h;
            *reinterpret_cast<uint4*>(tile + r * LROW + o * 16) = v[h][0];
            *reinterpret_cast<uint4*>(tile + (r + 1) * LROW + o * 16) = v[h][1];
        }
    }
};

template <> struct LoaderT<float> {
    const char* base;
    uint32_t coloff, ldb;
    bool rok, ident;
    RowMap rm;
    uint4 v[4];
    __device__ __forceinline__ void init(const float* p, int ld, const RowMap& m, int r0, int R, int t) {
        base = reinterpret_cast<const char*>(p);
        const int r = r0 + (t & 127);
        rok = r < R;
        coloff = (uint32_t)((rok ? r : 0) * 4);
        ldb = (uint32_t)ld * 4u;
        rm = m;
        ident = m.rpi == 0;
    }
    __device__ __forceinline__ void gload(int k0, int kend, int t) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int kb = k0 + 4 * ((t >> 7) + 2 * h);
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
            v[h] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    __device__ __forceinline__ void rowsum(float (&rs)[2]) const {
#pragma unroll
        for (int h = 0; h < 4; ++h)
            rs[0] += (__uint_as_float(v[h].x) + __uint_as_float(v[h].y)) + (__uint_as_float(v[h].z) + __uint_as_float(v[h].w));
    }
    __device__ __forceinline__ void rowsum_flush(const float (&rs)[2], float* red, float* out, int r0, int R, int t) const {
        red[(t >> 7) * 128 + (t & 127)] = rs[0];
        __syncthreads();
        if (t < 128 && r0 + t < R) atomicAdd(out + r0 + t, red[t] + red[128 + t]);
    }
    __device__ __forceinline__ void lstore(char* tile, int t) const {
        const int r = t & 127;
#pragma unroll
        for (int h = 0; h < 4; ++h) *reinterpret_cast<uint4*>(tile + r * LROW + ((t >> 7) + 2 * h) * 16) = v[h];
    }
};

// ---- LDS -> MFMA (transposed tile: first operand = B fragment) -------------------------------------------
template <bool SWAP>
__device__ __forceinline__ void mma_tile(f32x16 (&acc)[2][2], const char* As, const char* Bs, int wm, int wn, int lane,
                                         bf16_t*) {
    typedef __bf16 bfv8 __attribute__((ext_vector_type(8)));
    const int rr = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        bfv8 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a[i] = *reinterpret_cast<const bfv8*>(As + (wm * 64 + i * 32 + rr) * LROW + (ks * 2 + kh) * 16);
            b[i] = *reinterpret_cast<const bfv8*>(Bs + (wn * 64 + i * 32 + rr) * LROW + (ks * 2 + kh) * 16);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}
template <bool SWAP>
__device__ __forceinline__ void mma_tile(f32x16 (&acc)[2][2], const char* As, const char* Bs, int wm, int wn, int lane,
                                         float*) {
    const int rr = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        float a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a[i] = *reinterpret_cast<const float*>(As + (wm * 64 + i * 32 + rr) * LROW + (ks * 2 + kh) * 4);
            b[i] = *reinterpret_cast<const float*>(Bs + (wn * 64 + i * 32 + rr) * LROW + (ks * 2 + kh) * 4);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
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

// Epilogue flavours (compile-time, keeps every instantiation small enough to unroll fully):
//   EPI_STORE : (+bias)(+pos) -> keep mask -> scale -> (+resid) -> store TO
//   EPI_GELU  : (+bias) -> C = u, C2 = gelu(u) masked by keep            (Mlp.fc1)
//   EPI_DGELU : * gelu'(u) -> keep mask -> store TO                        (fc2 dgrad)
//   EPI_ATOMIC: keep mask/scale -> atomicAdd fp32                          (split-K wgrad)
enum { EPI_STORE = 0, EPI_GELU = 1, EPI_DGELU = 2, EPI_ATOMIC = 3 };

// epilogue of 4 consecutive columns n..n+3 of one output row
template <typename T, typename TO, int EPI>
__device__ __forceinline__ void epilogue_quad(const vr_gemm_args& p, float (&v)[4], int n, bool mok, int mloc,
                                              long long orow, float sc, int keep, bool first_split, bool vec_ok) {
    const int nvalid = min(4, p.N - n);          // <= 0: nothing to write
    const int nc = nvalid > 0 ? n : 0;
    const bool vec = vec_ok && nvalid == 4;
    const int nv = nvalid > 0 ? nvalid : 1;
    const bool any = mok && nvalid > 0;
    float aux[4];
    bool ok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) ok[e] = mok && (e < nvalid);
    const long long oidx = orow * p.ldc + nc;
    if constexpr (EPI == EPI_STORE || EPI == EPI_GELU) {
        if (p.bias && first_split) {
            load4<float>(p.bias, nc, aux, vec, nv);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += aux[e];
        }
    }
    if constexpr (EPI == EPI_STORE) {
        if (p.pos && first_split) {
            load4<float>(p.pos, (long long)mloc * p.N + nc, aux, vec, nv);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += aux[e];
        }
    }
    if constexpr (EPI == EPI_GELU) {
        float h[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (nc + e < keep) ? gelu_f(v[e]) : 0.f;
        if (any) {
            store4<TO>(p.C, oidx, v, vec, ok);
            store4<TO>(p.C2, oidx, h, vec, ok);
        }
        return;
    }
    if constexpr (EPI == EPI_DGELU) {
        load4<T>(p.dact_u, orow * p.ldu + nc, aux, vec, nv);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= dgelu_f(aux[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (nc + e < keep) ? v[e] * sc : 0.f;
    if constexpr (EPI == EPI_ATOMIC) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (ok[e]) atomicAdd(reinterpret_cast<float*>(p.C) + oidx + e, v[e]);
        return;
    }
    if constexpr (EPI == EPI_STORE) {
        if (p.resid) {
            load4<float>(p.resid, oidx, aux, vec, nv);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += aux[e];
        }
    }
    if (any) store4<TO>(p.C, oidx, v, vec, ok);
}

template <typename T, bool TA, bool TB, typename TO, int EPI>
__global__ __launch_bounds__(NTHR) void gemm_kernel(const vr_gemm_args p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // A0 B0 A1 B1
    constexpr int BK = Cfg<T>::BK;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    int kbeg = 0, kend = p.K;
    if (p.split_k > 1) {
        int per = (p.K + p.split_k - 1) / p.split_k;
        per = (per + BK - 1) / BK * BK;
        kbeg = blockIdx.z * per;
        kend = min(p.K, kbeg + per);
        if (kbeg >= kend) return;
    }
    const int ntiles = (kend - kbeg + BK - 1) / BK;

    const RowMap amap = {p.a_map.rpi, p.a_map.rps, p.a_map.off};
    const RowMap bmap = {p.b_map.rpi, p.b_map.rps, p.b_map.off};
    typename std::conditional<TA, LoaderT<T>, LoaderN<T>>::type la;
    typename std::conditional<TB, LoaderT<T>, LoaderN<T>>::type lb;
    la.init(reinterpret_cast<const T*>(p.A), p.lda, amap, m0, p.M, t);
    lb.init(reinterpret_cast<const T*>(p.B), p.ldb, bmap, n0, p.N, t);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float rs[2] = {0.f, 0.f};
    const bool want_bg = TA && p.bias_grad != nullptr && blockIdx.x == 0;
    la.gload(kbeg, kend, t);
    lb.gload(kbeg, kend, t);
    if constexpr (TA) { if (want_bg) la.rowsum(rs); }
    la.lstore(smem, t);
    lb.lstore(smem + TILE_BYTES, t);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < ntiles; ++kt) {
        const bool more = kt + 1 < ntiles;
        if (more) {
            la.gload(kbeg + (kt + 1) * BK, kend, t);
            lb.gload(kbeg + (kt + 1) * BK, kend, t);
            if constexpr (TA) { if (want_bg) la.rowsum(rs); }
        }
        const char* As = smem + cur * 2 * TILE_BYTES;
        mma_tile<EPI != EPI_ATOMIC>(acc, As, As + TILE_BYTES, wm, wn, lane, (T*)nullptr);
        if (more) {
            char* Ad = smem + (cur ^ 1) * 2 * TILE_BYTES;
            la.lstore(Ad, t);
            lb.lstore(Ad + TILE_BYTES, t);
        }
        __syncthreads();
        cur ^= 1;
    }

    if constexpr (TA) {
        if (want_bg) la.rowsum_flush(rs, reinterpret_cast<float*>(smem), p.bias_grad, m0, p.M, t);
    }
    if constexpr (EPI == EPI_ATOMIC) {
        // natural accumulator layout: a half-wave adds 32 consecutive fp32 of one output row (128 B) per instruction
        float* C = reinterpret_cast<float*>(p.C);
        const RowMap cm = {p.c_map.rpi, p.c_map.rps, p.c_map.off};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const long long orow = map_row(cm, m < p.M ? m : 0);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = n0 + wn * 64 + j * 32 + (lane & 31);
                    if (m < p.M && n < p.N) atomicAdd(C + orow * p.ldc + n, acc[i][j][r]);
                }
            }
        return;
    }
    // ---- epilogue (transposed accumulators): lane owns rows m_i = ... + (lane&31) and, per accumulator quad g,
    // the 4 consecutive columns n = n0 + wn*64 + j*32 + 8*g + 4*(lane>>5) + e
    const RowMap cmap = {p.c_map.rpi, p.c_map.rps, p.c_map.off};
    const bool first_split = blockIdx.z == 0;
    const bool vec_ok = (p.ldc % 4 == 0) && (p.N % 4 == 0) && (EPI != EPI_DGELU || p.ldu % 4 == 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 64 + i * 32 + (lane & 31);
        const bool mok = m < p.M;
        const int mc = mok ? m : 0;
        const int sample = p.rows_in > 0 ? mc / p.rows_in : 0;
        const int mloc = p.rows_in > 0 ? mc - sample * p.rows_in : mc;
        const long long orow = map_row(cmap, mc);
        const float sc = p.scale ? p.scale[sample] : 1.0f;
        const int keep = p.keep_n ? p.keep_n[sample] : p.N;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
                const int n = n0 + wn * 64 + j * 32 + 8 * g + 4 * (lane >> 5);
                epilogue_quad<T, TO, EPI>(p, v, n, mok, mloc, orow, sc, keep, first_split, vec_ok);
            }
        }
    }
}

template <typename T, bool TA, bool TB, typename TO, int EPI>
void launch1(const vr_gemm_args& a, hipStream_t stream) {
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, a.split_k);
    hipLaunchKernelGGL((gemm_kernel<T, TA, TB, TO, EPI>), grid, dim3(NTHR), 0, stream, a);
}

template <typename T>
int launch(const vr_gemm_args& a, hipStream_t stream) {
    const bool of32 = a.out_dtype == VR_F32;
    if (a.atomic) {                                             // weight gradients (fp32 accumulate)
        if (a.a_trans && a.b_trans) launch1<T, true, true, float, EPI_ATOMIC>(a, stream);
        else if (!a.a_trans && !a.b_trans) launch1<T, false, false, float, EPI_ATOMIC>(a, stream);
        else return VR_EUNSUPPORTED;
    } else if (a.a_trans) {
        return VR_EUNSUPPORTED;
    } else if (a.act == 1) {
        if (a.b_trans || of32 != (sizeof(T) == 4)) return VR_EUNSUPPORTED;
        launch1<T, false, false, T, EPI_GELU>(a, stream);
    } else if (a.dact_u) {
        if (!a.b_trans || of32 != (sizeof(T) == 4)) return VR_EUNSUPPORTED;
        launch1<T, false, true, T, EPI_DGELU>(a, stream);
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

extern "C" int vr_gemm(const vr_gemm_args* args, vr_stream_t stream) {
    if (!args || !args->A || !args->B || !args->C) return VR_EINVAL;
    vr_gemm_args a = *args;
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return VR_EINVAL;
    if (a.split_k < 1) a.split_k = 1;
    if (a.split_k > 1 && !a.atomic) return VR_EINVAL;
    if (a.atomic && a.out_dtype != VR_F32) return VR_EINVAL;
    if (a.act == 1 && !a.C2) return VR_EINVAL;
    if (a.bias_grad && !(a.a_trans && a.atomic)) return VR_EINVAL;
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
    if (a.in_dtype == VR_BF16) return launch<bf16_t>(a, (hipStream_t)stream);
    return launch<float>(a, (hipStream_t)stream);
}
