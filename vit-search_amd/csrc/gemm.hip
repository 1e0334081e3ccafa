// Fused MFMA GEMM for gfx950:  C[M,N] = epilogue(A[M,K] * B[N,K]^T)
//
// One kernel family covers forward (NT), dgrad (N,T-of-B) and wgrad (both operands contraction-major)
// of every nn.Linear / patchify-conv on the ViT-Res hot path (reference nets/supernet_blocks.py:37-52,
// 102-119; nets/vit_sr_supernet.py:140,151,440-446), in two precisions:
//   bf16 : v_mfma_f32_32x32x16_bf16, fp32 accumulate           (fast path)
//   fp32 : v_mfma_f32_32x32x2_f32, exact fp32 (fmaf-chain)     (parity path)
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each 64x64 = 2x2 MFMA tiles of
// 32x32), 64-byte K slices (32 bf16 / 16 fp32) double-buffered in LDS.  LDS rows are padded to 80 bytes:
// 5 is odd, so the 16 lanes of a ds_read_b128 lane group hit 16 distinct 16-B slots (conflict free).
// Contraction-major operands (dgrad's W, wgrad's dY and X) are transposed in registers on the
// global->LDS path (8 coalesced dword loads -> two 16-B rows), so the MFMA side is identical for all forms.
#include "common.h"
#include "../../include/vitres_hip.h"

namespace {

constexpr int BM = 128, BN = 128, NTHR = 256;
constexpr int LROW = 80;   // padded LDS row in bytes
constexpr int TILE_BYTES = BM * LROW;

template <typename T> struct Cfg;
template <> struct Cfg<bf16_t> {
    static constexpr int BK = 32;   // elements per K slice
    static constexpr int EPC = 8;   // elements per 16-B chunk
};
template <> struct Cfg<float> {
    static constexpr int BK = 16;
    static constexpr int EPC = 4;
};

struct Stage {
    uint4 v[2];
};

// ---- global -> registers ---------------------------------------------------------------------------
// K-contiguous operand: tile row r (0..127) = 4 chunks of 16 B.  thread t: chunk t&3, rows t>>2 and 64+(t>>2)
template <typename T>
__device__ __forceinline__ void gload_n(Stage& s, const T* __restrict__ base, int ld, const RowMap& rm, int r0,
                                        int R, int k0, int kend, int t) {
    constexpr int EPC = Cfg<T>::EPC;
    const int c = t & 3;
    const int k = k0 + c * EPC;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = r0 + (t >> 2) + h * 64;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r < R && k < kend) v = *reinterpret_cast<const uint4*>(base + map_row(rm, r) * (long long)ld + k);
        s.v[h] = v;
    }
}
__device__ __forceinline__ void lstore_n(const Stage& s, char* tile, int t) {
    const int c = t & 3;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = (t >> 2) + h * 64;
        *reinterpret_cast<uint4*>(tile + r * LROW + c * 16) = s.v[h];
    }
}

// contraction-major bf16 operand: element (kk, r) at base[map(kk)*ld + r].
// thread t: row pair rp = t&63 (rows 2rp, 2rp+1), k-octet o = t>>6; 8 dword loads, each wave-load 256 B contiguous.
__device__ __forceinline__ void gload_t(Stage& s, const bf16_t* __restrict__ base, int ld, const RowMap& rm, int r0,
                                        int R, int k0, int kend, int t) {
    const int r = r0 + 2 * (t & 63);
    const int kb = k0 + 8 * (t >> 6);
    uint32_t w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int kk = kb + e;
        w[e] = 0;
        if (r < R && kk < kend) w[e] = *reinterpret_cast<const uint32_t*>(base + map_row(rm, kk) * (long long)ld + r);
    }
    // row 2rp   <- low halves, row 2rp+1 <- high halves
    s.v[0] = make_uint4((w[0] & 0xffffu) | (w[1] << 16), (w[2] & 0xffffu) | (w[3] << 16),
                        (w[4] & 0xffffu) | (w[5] << 16), (w[6] & 0xffffu) | (w[7] << 16));
    s.v[1] = make_uint4((w[0] >> 16) | (w[1] & 0xffff0000u), (w[2] >> 16) | (w[3] & 0xffff0000u),
                        (w[4] >> 16) | (w[5] & 0xffff0000u), (w[6] >> 16) | (w[7] & 0xffff0000u));
}
__device__ __forceinline__ void lstore_t(const Stage& s, char* tile, int t, bf16_t*) {
    const int r = 2 * (t & 63);
    const int o = t >> 6;
    *reinterpret_cast<uint4*>(tile + r * LROW + o * 16) = s.v[0];
    *reinterpret_cast<uint4*>(tile + (r + 1) * LROW + o * 16) = s.v[1];
}

// contraction-major fp32 operand: thread t: row t&127, k-quads (t>>7) and (t>>7)+2
__device__ __forceinline__ void gload_t(Stage& s, const float* __restrict__ base, int ld, const RowMap& rm, int r0,
                                        int R, int k0, int kend, int t) {
    const int r = r0 + (t & 127);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int kb = k0 + 4 * ((t >> 7) + 2 * h);
        float w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int kk = kb + e;
            w[e] = 0.f;
            if (r < R && kk < kend) w[e] = base[map_row(rm, kk) * (long long)ld + r];
        }
        s.v[h] = make_uint4(__float_as_uint(w[0]), __float_as_uint(w[1]), __float_as_uint(w[2]), __float_as_uint(w[3]));
    }
}
__device__ __forceinline__ void lstore_t(const Stage& s, char* tile, int t, float*) {
    const int r = t & 127;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int q = (t >> 7) + 2 * h;
        *reinterpret_cast<uint4*>(tile + r * LROW + q * 16) = s.v[h];
    }
}

// ---- LDS -> MFMA -----------------------------------------------------------------------------------
__device__ __forceinline__ void mma_tile(f32x16 (&acc)[2][2], const char* As, const char* Bs, int wm, int wn, int lane,
                                         bf16_t*) {
    typedef __bf16 bfv8 __attribute__((ext_vector_type(8)));
    const int rr = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
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
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}
__device__ __forceinline__ void mma_tile(f32x16 (&acc)[2][2], const char* As, const char* Bs, int wm, int wn, int lane,
                                         float*) {
    const int rr = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
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
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}

template <typename TO> __device__ __forceinline__ void store_out(void* p, long long idx, float v);
template <> __device__ __forceinline__ void store_out<float>(void* p, long long idx, float v) {
    reinterpret_cast<float*>(p)[idx] = v;
}
template <> __device__ __forceinline__ void store_out<bf16_t>(void* p, long long idx, float v) {
    reinterpret_cast<bf16_t*>(p)[idx] = f2bf(v);
}

template <typename T, bool TA, bool TB>
__global__ __launch_bounds__(NTHR) void gemm_kernel(const vr_gemm_args p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // A0 B0 A1 B1
    constexpr int BK = Cfg<T>::BK;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    // K range of this split
    int kbeg = 0, kend = p.K;
    if (p.split_k > 1) {
        int per = (p.K + p.split_k - 1) / p.split_k;
        per = (per + BK - 1) / BK * BK;
        kbeg = blockIdx.z * per;
        kend = min(p.K, kbeg + per);
        if (kbeg >= kend) return;
    }
    const int ntiles = (kend - kbeg + BK - 1) / BK;

    const T* A = reinterpret_cast<const T*>(p.A);
    const T* B = reinterpret_cast<const T*>(p.B);
    const RowMap amap = {p.a_map.rpi, p.a_map.rps, p.a_map.off};
    const RowMap bmap = {p.b_map.rpi, p.b_map.rps, p.b_map.off};

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    Stage sa, sb;
    auto gload = [&](int k0) {
        if constexpr (TA) gload_t(sa, A, p.lda, amap, m0, p.M, k0, kend, t);
        else gload_n<T>(sa, A, p.lda, amap, m0, p.M, k0, kend, t);
        if constexpr (TB) gload_t(sb, B, p.ldb, bmap, n0, p.N, k0, kend, t);
        else gload_n<T>(sb, B, p.ldb, bmap, n0, p.N, k0, kend, t);
    };
    auto lstore = [&](int buf) {
        char* As = smem + buf * 2 * TILE_BYTES;
        char* Bs = As + TILE_BYTES;
        if constexpr (TA) lstore_t(sa, As, t, (T*)nullptr);
        else lstore_n(sa, As, t);
        if constexpr (TB) lstore_t(sb, Bs, t, (T*)nullptr);
        else lstore_n(sb, Bs, t);
    };

    gload(kbeg);
    lstore(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < ntiles; ++kt) {
        const bool more = kt + 1 < ntiles;
        if (more) gload(kbeg + (kt + 1) * BK);
        const char* As = smem + cur * 2 * TILE_BYTES;
        mma_tile(acc, As, As + TILE_BYTES, wm, wn, lane, (T*)nullptr);
        if (more) lstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue ----
    const int col = lane & 31, rq = lane >> 5;
    const RowMap cmap = {p.c_map.rpi, p.c_map.rps, p.c_map.off};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * rq;
            if (m >= p.M) continue;
            const int sample = p.rows_in > 0 ? m / p.rows_in : 0;
            const int mloc = p.rows_in > 0 ? m - sample * p.rows_in : m;
            const long long orow = map_row(cmap, m);
            const float sc = p.scale ? p.scale[sample] : 1.0f;
            const int keep = p.keep_n ? p.keep_n[sample] : p.N;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + j * 32 + col;
                if (n >= p.N) continue;
                float v = acc[i][j][r];
                if (p.bias && blockIdx.z == 0) v += p.bias[n];
                if (p.pos && blockIdx.z == 0) v += p.pos[(long long)mloc * p.N + n];
                const long long oidx = orow * p.ldc + n;
                if (p.act == 1) {
                    const float h = (n < keep) ? gelu_f(v) : 0.f;
                    if (p.out_dtype == VR_BF16) {
                        store_out<bf16_t>(p.C, oidx, v);
                        store_out<bf16_t>(p.C2, oidx, h);
                    } else {
                        store_out<float>(p.C, oidx, v);
                        store_out<float>(p.C2, oidx, h);
                    }
                    continue;
                }
                if (p.dact_u) {
                    const float u = Elem<T>::ld(reinterpret_cast<const T*>(p.dact_u) + orow * p.ldu + n);
                    v *= dgelu_f(u);
                }
                if (n >= keep) v = 0.f;
                v *= sc;
                if (p.atomic) {
                    atomicAdd(reinterpret_cast<float*>(p.C) + oidx, v);
                    continue;
                }
                if (p.resid) v += p.resid[oidx];
                if (p.out_dtype == VR_BF16) store_out<bf16_t>(p.C, oidx, v);
                else store_out<float>(p.C, oidx, v);
            }
        }
    }
}

template <typename T>
int launch(const vr_gemm_args& a, hipStream_t stream) {
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, a.split_k);
    if (!a.a_trans && !a.b_trans) hipLaunchKernelGGL((gemm_kernel<T, false, false>), grid, dim3(NTHR), 0, stream, a);
    else if (!a.a_trans && a.b_trans) hipLaunchKernelGGL((gemm_kernel<T, false, true>), grid, dim3(NTHR), 0, stream, a);
    else if (a.a_trans && a.b_trans) hipLaunchKernelGGL((gemm_kernel<T, true, true>), grid, dim3(NTHR), 0, stream, a);
    else hipLaunchKernelGGL((gemm_kernel<T, true, false>), grid, dim3(NTHR), 0, stream, a);
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
    if (a.in_dtype != VR_F32 && a.in_dtype != VR_BF16) return VR_EUNSUPPORTED;
    if (a.out_dtype != VR_F32 && a.out_dtype != VR_BF16) return VR_EUNSUPPORTED;
    const int epc = a.in_dtype == VR_BF16 ? 8 : 4;
    const int esz = a.in_dtype == VR_BF16 ? 2 : 4;
    // K-contiguous operands are fetched in 16-byte chunks (rows must be readable up to roundup(K, chunk): the caller
    // zero-pads); contraction-major ones in dwords
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
    if (a.in_dtype == VR_BF16) return launch<bf16_t>(a, (hipStream_t)stream);
    return launch<float>(a, (hipStream_t)stream);
}
