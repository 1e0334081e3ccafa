// bf16 MFMA multi-head self-attention, forward and backward, for the ViT-Res token counts (N = 257 / 65 / 17 at
// 224 px; any N <= 288) and head dims 32 / 48 / 64 (reference nets/supernet_blocks.py:105-109 and its autograd).
//
// One workgroup per (sample, head): the whole head lives in LDS, no online softmax is needed.
//   forward   : LDS = K, V (chunk-major); each wave owns 16-query tiles.
//               S^T = K Q^T  (v_mfma_f32_16x16x32_bf16, A = K rows from LDS, B = Q rows from global)
//               softmax over keys = registers + two cross-lane shuffles (xor 16, 32)
//               O^T = V^T P^T (A = V^T by transposing reads, B = P straight from the S^T accumulators)
//   backward A: dQ     (LDS = K, V chunk-major)            per 16-query tile, same dataflow as the forward
//   backward B: dK, dV (LDS = Q, dO chunk-major)           per 16-key tile
// Why S^T: the C/D layout of the 16x16 MFMA gives each lane 4 consecutive ROWS of one column; with rows = keys the
// accumulators of two key tiles are exactly an A-operand (i = query, k = 8 key slots) of the next MFMA -- P never
// leaves registers.  Any consistent assignment of contraction slots to keys is valid as long as the B operand uses
// the same one: slot (g, e) of key-pair tile kp is key 32*kp + 16*(e/4) + 4*g + e%4.
// LDS layouts (both conflict free for their read instruction):
//   chunk-major  [D/8][Np+8][8 bf16] : ds_read_b128 fragment (row = lane%16, chunk = 4*dk + lane/16); the SAME image
//   serves the contraction-over-rows operands (V in P V, K in dS K, Q / dO in the dK / dV products) through the gfx950
//   transposing read ds_read_b64_tr_b16: a 16-lane group fetches a [4 rows][16 d] block, lane i supplying the address
//   of row i/4, d 4(i%4)..+3 (8 contiguous bytes inside a chunk).  The chunk stride (Np + 8 rows) is an odd multiple
//   of 128 B, so the two chunks a 32-lane half touches fall on different halves of the 256-B bank row.
#include <cstdlib>

#include "common.h"
#include "../../include/vitres_hip.h"

namespace vr_attn_mfma {

typedef __bf16 bfv8 __attribute__((ext_vector_type(8)));

template <int D> struct AC {
    static constexpr int NCH = D / 8;          // 16-byte chunks per row
    static constexpr int DK = (D + 31) / 32;   // MFMA k-steps over d
    static constexpr int DT = D / 16;          // 16-wide output tiles over d
    static constexpr bool SWZ = (D != 48);
};

__device__ __forceinline__ f32x4 mfma16(bfv8 a, bfv8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ---- staging ---------------------------------------------------------------------------------------------
// rows [N][D] at src (row stride rs elements) -> chunk-major LDS, rows N..Np-1 zero.
// Trip counts are compile-time (MAXNP = 32 * NKP >= Np) and the loop is fully unrolled with every global load issued
// before the first LDS store.  Lane -> element map: a group of 8 consecutive lanes covers 4 rows x 2 adjacent chunks, NCH / 2
// consecutive groups cover the whole width of those 4 rows -- a wave-instruction reads whole 128-byte lines (the previous map,
// consecutive lanes = consecutive rows of ONE chunk, touched 64 different lines per instruction and every line 8 times), and
// the 8 lanes of a ds_write_b128 group fall on 8 different 16-byte bank slots: the chunk-plane stride is 128 (mod 256) bytes,
// so the chunk parity picks the half of the bank row and the 4 consecutive rows fill 64 bytes of it.
template <int D, int MAXNP, int NTHR>
__device__ __forceinline__ void stage_chunked(char* dst, const bf16_t* __restrict__ src, int rs, int N, int Np, int tid) {
    const int NpS = Np + 8;                    // row stride of a chunk plane (see header)
    constexpr int NCH = AC<D>::NCH, HP = NCH / 2;
    constexpr int IT = (MAXNP * NCH + NTHR - 1) / NTHR;
    const int total = Np * NCH;
    uint4 v[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int idx = tid + it * NTHR;
        const int q = idx >> 3, l8 = idx & 7;
        const int rb = q / HP, cp = q - rb * HP;
        const int n = 4 * rb + (l8 >> 1), ch = 2 * cp + (l8 & 1);
        const bool ok = n < N && idx < total;
        v[it] = *reinterpret_cast<const uint4*>(src + (long long)(ok ? n : 0) * rs + (ok ? ch : 0) * 8);
        if (!ok) v[it] = make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int idx = tid + it * NTHR;
        const int q = idx >> 3, l8 = idx & 7;
        const int rb = q / HP, cp = q - rb * HP;
        const int n = 4 * rb + (l8 >> 1), ch = 2 * cp + (l8 & 1);
        if (idx < total) *reinterpret_cast<uint4*>(dst + ((size_t)ch * NpS + n) * 16) = v[it];
    }
}

// ---- fragments -------------------------------------------------------------------------------------------
// 16 rows x 32 d from global rows (lane: row r0 + lane%16, d = dk*32 + 8*(lane/16) ..+7), zero outside [N) x [D)
template <int D>
__device__ __forceinline__ bfv8 gfrag(const bf16_t* __restrict__ src, int rs, int r0, int N, int dk, int lane) {
    const int r = r0 + (lane & 15), d0 = dk * 32 + 8 * (lane >> 4);
    const bool ok = (r < N) && (d0 < D);
    uint4 v = *reinterpret_cast<const uint4*>(src + (long long)(ok ? r : 0) * rs + (ok ? d0 : 0));
    if (!ok) v = make_uint4(0, 0, 0, 0);
    return __builtin_bit_cast(bfv8, v);
}
template <int D>
__device__ __forceinline__ bfv8 cfrag(const char* base, int Np, int n0, int dk, int lane) {
    int ch = dk * 4 + (lane >> 4);
    ch = ch < AC<D>::NCH ? ch : AC<D>::NCH - 1;      // partner operand is zero there (D = 48)
    return *reinterpret_cast<const bfv8*>(base + ((size_t)ch * (Np + 8) + n0 + (lane & 15)) * 16);
}
// contraction slots of pair tile kp (rows 32 kp + 16 (e/4) + 4 g + e%4) for column d = dt*16 + lane%16, read from the
// chunk-major image with two transposing reads (rows +0..3 and +16..19)
typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
template <int D>
__device__ __forceinline__ bfv8 tfrag(const char* base, int Np, int kp, int dt, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const int row = 32 * kp + 4 * g + (i >> 2), d0 = dt * 16 + 4 * (i & 3);
    const char* p = base + ((size_t)(d0 >> 3) * (Np + 8) + row) * 16 + (d0 & 7) * 2;
    typedef __attribute__((address_space(3))) s4v lds_s4v;
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(p));
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(p + 16 * 16));
    const s8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bfv8, v);
}
__device__ __forceinline__ bfv8 pack8(const f32x4& a, const f32x4& b) {
    return __builtin_bit_cast(bfv8, make_uint4(pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]), pack_bf2(b[0], b[1]),
                                               pack_bf2(b[2], b[3])));
}
__device__ __forceinline__ float gmax(float v) {   // over the 4 lanes sharing lane%16
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float gsum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// ==========================================================================================================
// Occupancy plan of the whole-head kernels (N <= 288).  A head of N = 257, D = 64 needs 76 KB of LDS: two workgroups per CU
// at most.  The kernels are therefore written for <= 128 VGPRs with NW = 8 waves (a multiple of the 4 SIMDs: a 6-wave
// workgroup occupies the SIMDs 2/2/1/1 and a second one was not co-scheduled -- measured: 1.2 resident waves per SIMD): two
// co-resident workgroups give every SIMD four waves, and the staging of one head overlaps the MFMA / softmax work of the
// other.  Register pressure is bounded by construction: the forward walks the keys in blocks of PB tile pairs with a
// running (max, sum) and one accumulator rescale per block (the scores of a block, not of the whole row, live in registers);
// the backward kernels hold one key / query tile per wave.  exp(x) is v_exp_f32 on x * log2(e) folded into the scale;
// zero-padded LDS rows make every mask of the backward kernels redundant (an out-of-range key or query contributes exact
// zeros), the forward masks the one partial key tile only.
// The padded row count NP = 32 * NKP is a compile-time constant: every LDS fragment address is one per-lane base (computed
// once per kernel) plus an immediate -- PMC counters of the previous form showed 935 VALU instructions per 16-query tile,
// half of them address arithmetic, on a kernel whose VALU pipe was busy 3.6x longer than its matrix pipe.
// ==========================================================================================================
constexpr float LOG2E = 1.4426950408889634f;
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

template <int D, int NP> struct Img {
    static constexpr int S = (NP + 8) * 16;                   // bytes between chunk planes
    static constexpr int BYTES = AC<D>::NCH * S;               // one staged operand
    // per-lane byte offset of the ds_read_b128 fragment (row lane % 16, chunk 4 dk + lane / 16); + 16 * row0 selects the tile
    __device__ static __forceinline__ int cbase(int dk, int lane) {
        int ch = dk * 4 + (lane >> 4);
        ch = ch < AC<D>::NCH ? ch : AC<D>::NCH - 1;            // partner operand is zero there (D = 48)
        return (ch * (NP + 8) + (lane & 15)) * 16;
    }
    // per-lane byte offset of the transposing read (see tfrag); + (2 dt (NP + 8) + 32 kp) * 16 selects pair tile / d tile
    __device__ static __forceinline__ int tbase(int lane) {
        const int g = lane >> 4, i = lane & 15;
        return ((((i & 3) >> 1)) * (NP + 8) + 4 * g + (i >> 2)) * 16 + (i & 1) * 8;
    }
    // the per-lane base as an opaque 32-bit LDS address: without the barrier the compiler folds the image's offset inside the
    // workgroup's LDS into every fragment offset, the sum no longer fits the 16-bit immediate of ds_read and each read pays
    // a v_or / v_add
    __device__ static __forceinline__ const char* opaque(const char* p) {
        typedef __attribute__((address_space(3))) const char lds_char;
        unsigned a = (unsigned)(unsigned long long)(lds_char*)p;
        asm volatile("" : "+v"(a));
        return (const char*)(lds_char*)(unsigned long long)a;
    }
    __device__ static __forceinline__ bfv8 cread(const char* lane_base, int row0) {
        return *reinterpret_cast<const bfv8*>(lane_base + row0 * 16);
    }
    __device__ static __forceinline__ bfv8 tread(const char* lane_base, int kp, int dt) {
        typedef __attribute__((address_space(3))) s4v lds_s4v;
        const char* p = lane_base + (2 * dt * (NP + 8) + 32 * kp) * 16;
        const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(p));
        const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(p + 16 * 16));
        const s8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bfv8, v);
    }
};

// ==========================================================================================================
// forward
// ==========================================================================================================
template <int D, int NKP, int NW, int PB, int OCC, bool FULL>
__global__ __launch_bounds__(NW * 64, OCC) void fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                           float* __restrict__ lse, const int* __restrict__ keep_hd, int B,
                                                           int N, int H, float scale, int ATTN_XCD) {
    static_assert(NKP % PB == 0, "key-tile pairs must split into whole blocks");
    constexpr int NP = 32 * NKP;
    typedef Img<D, NP> I;
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int bidx = xcd_block((int)blockIdx.x, (int)gridDim.x, ATTN_XCD);   // (sample-major: XCD x gets samples [x B / 8, ..))
    const int b = bidx / H, h = bidx % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int HD = H * D, RS = 3 * HD;
    const bf16_t* base = qkv + (long long)b * N * RS + h * D;
    bf16_t* ob = o + (long long)b * N * HD + h * D;
    float* lb = lse + ((long long)b * H + h) * N;
    if (keep_hd && h * D >= keep_hd[b]) {
        for (int i = tid; i < N * (D / 8); i += NW * 64) {
            const int n = i / (D / 8), ch = i % (D / 8);
            *reinterpret_cast<uint4*>(ob + (long long)n * HD + ch * 8) = make_uint4(0, 0, 0, 0);
        }
        for (int n = tid; n < N; n += NW * 64) lb[n] = 0.f;
        return;
    }
    char* Kc = sm;
    char* Vc = sm + I::BYTES;
    int q0 = wave * 16;
    bfv8 qf[AC<D>::DK];                                   // first query tile: in flight while K / V are staged
#pragma unroll
    for (int dk = 0; dk < AC<D>::DK; ++dk) qf[dk] = gfrag<D>(base, RS, q0, N, dk, lane);
    stage_chunked<D, NP, NW * 64>(Kc, base + HD, RS, N, NP, tid);
    stage_chunked<D, NP, NW * 64>(Vc, base + 2 * HD, RS, N, NP, tid);
    __syncthreads();
    const char* kb[AC<D>::DK];
#pragma unroll
    for (int dk = 0; dk < AC<D>::DK; ++dk) kb[dk] = I::opaque(Kc + I::cbase(dk, lane));
    const char* vb = I::opaque(Vc + I::tbase(lane));
    const float cs = scale * LOG2E;
    while (q0 < N) {
        float m = -INFINITY, sum = 0.f;                   // running raw max (same in the 4 lanes of a query) and per-lane sum
        f32x4 oacc[AC<D>::DT];
#pragma unroll
        for (int dt = 0; dt < AC<D>::DT; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b0 = 0; b0 < NKP; b0 += PB) {
            if (FULL || b0 * 32 < N) {
                f32x4 st[2 * PB];
                float bm = -INFINITY;
#pragma unroll
                for (int j = 0; j < 2 * PB; ++j) {
                    const int k0 = (2 * b0 + j) * 16;
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    if (FULL || k0 < N) {
#pragma unroll
                        for (int dk = 0; dk < AC<D>::DK; ++dk) acc = mfma16(I::cread(kb[dk], k0), qf[dk], acc);
                        // FULL: N > 32 (NKP - 1), only the last pair of tiles can hold rows >= N (compile-time choice of the
                        // masked tiles, no per-tile scalar conditions: those cost more SGPRs than the kernel has)
                        if (FULL ? (2 * b0 + j >= 2 * NKP - 2) : (k0 + 16 > N)) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[r] = (k0 + 4 * g + r) < N ? acc[r] : -INFINITY;
                        }
                    } else {
                        acc = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                    }
                    st[j] = acc;
                    bm = fmaxf(fmaxf(bm, fmaxf(acc[0], acc[1])), fmaxf(acc[2], acc[3]));
                }
                bm = gmax(bm);
                const float mn = fmaxf(m, bm);
                const float alpha = ex2((m - mn) * cs);   // 0 for the first block (m = -inf)
                const float mc = mn * cs;
                float ps = 0.f;
#pragma unroll
                for (int j = 0; j < 2 * PB; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = ex2(fmaf(st[j][r], cs, -mc));
                        st[j][r] = pv;
                        ps += pv;
                    }
                sum = sum * alpha + ps;
                m = mn;
                if (b0 > 0) {
#pragma unroll
                    for (int dt = 0; dt < AC<D>::DT; ++dt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) oacc[dt][r] *= alpha;
                }
#pragma unroll
                for (int jp = 0; jp < PB; ++jp) {
                    if (FULL || (b0 + jp) * 32 < N) {
                        const bfv8 pf = pack8(st[2 * jp], st[2 * jp + 1]);
#pragma unroll
                        for (int dt = 0; dt < AC<D>::DT; ++dt) oacc[dt] = mfma16(I::tread(vb, b0 + jp, dt), pf, oacc[dt]);
                    }
                }
                // keep the blocks apart: in straight-line code the scheduler hoists every LDS read of the tile to the top and
                // spills (87 VGPRs at the 128-register budget); the other three waves of the SIMD cover this wave's latencies
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        sum = gsum(sum);
        // O^T tiles (V^T as the first operand): the lane owns query q0 + c and 4 consecutive d per tile -> 8-byte stores
        const float inv = 1.0f / sum;
        const int qs = q0;
        q0 += NW * 16;
        bfv8 qn[AC<D>::DK];                               // next tile's queries: issued before this tile's stores
        if (q0 < N) {
#pragma unroll
            for (int dk = 0; dk < AC<D>::DK; ++dk) qn[dk] = gfrag<D>(base, RS, q0, N, dk, lane);
        }
        if (qs + c < N) {
#pragma unroll
            for (int dt = 0; dt < AC<D>::DT; ++dt)
                *reinterpret_cast<uint2*>(ob + (long long)(qs + c) * HD + dt * 16 + 4 * g) =
                    make_uint2(pack_bf2(oacc[dt][0] * inv, oacc[dt][1] * inv), pack_bf2(oacc[dt][2] * inv, oacc[dt][3] * inv));
            if (g == 0) lb[qs + c] = m * scale + __logf(sum);
        }
        if (q0 < N) {
#pragma unroll
            for (int dk = 0; dk < AC<D>::DK; ++dk) qf[dk] = qn[dk];
        }
    }
}

// ==========================================================================================================
// backward A: dQ (+ delta = rowsum(dO * O))
// ==========================================================================================================
template <int D, int NKP, int NW, int OCC, bool FULL>
__global__ __launch_bounds__(NW * 64, OCC) void bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                              const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                              float* __restrict__ delta, bf16_t* __restrict__ dqkv,
                                                              const int* __restrict__ keep_hd, int B, int N, int H,
                                                              float scale, int ATTN_XCD) {
    constexpr int NP = 32 * NKP;
    typedef Img<D, NP> I;
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int bidx = xcd_block((int)blockIdx.x, (int)gridDim.x, ATTN_XCD);   // (sample-major: XCD x gets samples [x B / 8, ..))
    const int b = bidx / H, h = bidx % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int HD = H * D, RS = 3 * HD;
    const bf16_t* base = qkv + (long long)b * N * RS + h * D;
    bf16_t* dbase = dqkv + (long long)b * N * RS + h * D;
    if (keep_hd && h * D >= keep_hd[b]) {
        for (int i = tid; i < N * (D / 8); i += NW * 64) {
            const int n = i / (D / 8), ch = i % (D / 8);
            *reinterpret_cast<uint4*>(dbase + (long long)n * RS + ch * 8) = make_uint4(0, 0, 0, 0);
        }
        return;
    }
    const bf16_t* ob = o + (long long)b * N * HD + h * D;
    const bf16_t* gb = d_o + (long long)b * N * HD + h * D;
    const float* lb = lse + ((long long)b * H + h) * N;
    float* db = delta + ((long long)b * H + h) * N;
    char* Kc = sm;
    char* Vc = Kc + I::BYTES;
    int q0 = wave * 16;
    bfv8 qf[AC<D>::DK], gf[AC<D>::DK], of[AC<D>::DK];     // first query tile: in flight while K / V are staged
    float lq = 0.f;
#pragma unroll
    for (int dk = 0; dk < AC<D>::DK; ++dk) {
        qf[dk] = gfrag<D>(base, RS, q0, N, dk, lane);
        gf[dk] = gfrag<D>(gb, HD, q0, N, dk, lane);
        of[dk] = gfrag<D>(ob, HD, q0, N, dk, lane);
    }
    if (q0 + c < N) lq = lb[q0 + c];
    stage_chunked<D, NP, NW * 64>(Kc, base + HD, RS, N, NP, tid);
    stage_chunked<D, NP, NW * 64>(Vc, base + 2 * HD, RS, N, NP, tid);
    __syncthreads();
    const char *kb[AC<D>::DK], *vb[AC<D>::DK];
#pragma unroll
    for (int dk = 0; dk < AC<D>::DK; ++dk) {
        kb[dk] = I::opaque(Kc + I::cbase(dk, lane));
        vb[dk] = I::opaque(Vc + I::cbase(dk, lane));
    }
    const char* kt_b = I::opaque(Kc + I::tbase(lane));
    const float cs = scale * LOG2E;
    while (q0 < N) {
        float d = 0.f;
#pragma unroll
        for (int dk = 0; dk < AC<D>::DK; ++dk)
#pragma unroll
            for (int e = 0; e < 8; ++e) d += (float)gf[dk][e] * (float)of[dk][e];
        const float dl = gsum(d);
        const float l2 = lq * LOG2E;
        if (g == 0 && q0 + c < N) db[q0 + c] = dl;
        f32x4 dq[AC<D>::DT];
#pragma unroll
        for (int dt = 0; dt < AC<D>::DT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        // keys beyond N are zero rows of K and V in LDS: their dS is finite and meets a zero K^T fragment -- no masks
#pragma unroll
        for (int kp = 0; kp < NKP; ++kp) {
            if (FULL || kp * 32 < N) {
                f32x4 ds[2];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int k0 = (2 * kp + tt) * 16;
                    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                    if (FULL || k0 < N) {
#pragma unroll
                        for (int dk = 0; dk < AC<D>::DK; ++dk) {
                            sc = mfma16(I::cread(kb[dk], k0), qf[dk], sc);
                            dp = mfma16(I::cread(vb[dk], k0), gf[dk], dp);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) sc[r] = ex2(fmaf(sc[r], cs, -l2)) * ((dp[r] - dl) * scale);
                    }
                    ds[tt] = sc;
                }
                const bfv8 sf = pack8(ds[0], ds[1]);                                  // dS never leaves registers
#pragma unroll
                for (int dt = 0; dt < AC<D>::DT; ++dt) dq[dt] = mfma16(I::tread(kt_b, kp, dt), sf, dq[dt]);
                if (kp % 3 == 2) __builtin_amdgcn_sched_barrier(0);      // see fwd_kernel: bounds the scheduler's hoisting
            }
        }
        const int qs = q0;
        q0 += NW * 16;
        if (q0 < N) {                                       // next tile's operands: issued before this tile's stores
#pragma unroll
            for (int dk = 0; dk < AC<D>::DK; ++dk) {
                qf[dk] = gfrag<D>(base, RS, q0, N, dk, lane);
                gf[dk] = gfrag<D>(gb, HD, q0, N, dk, lane);
                of[dk] = gfrag<D>(ob, HD, q0, N, dk, lane);
            }
            lq = q0 + c < N ? lb[q0 + c] : 0.f;
        }
        if (qs + c < N) {      // dQ^T tiles: query qs + c, 4 consecutive d per tile
#pragma unroll
            for (int dt = 0; dt < AC<D>::DT; ++dt)
                *reinterpret_cast<uint2*>(dbase + (long long)(qs + c) * RS + dt * 16 + 4 * g) =
                    make_uint2(pack_bf2(dq[dt][0], dq[dt][1]), pack_bf2(dq[dt][2], dq[dt][3]));
        }
    }
}

// ==========================================================================================================
// backward B: dK, dV
// ==========================================================================================================
template <int D, int NKP, int NW, int OCC, bool FULL>
__global__ __launch_bounds__(NW * 64, OCC) void bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ d_o,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               bf16_t* __restrict__ dqkv, const int* __restrict__ keep_hd,
                                                               int B, int N, int H, float scale, int ATTN_XCD) {
    constexpr int NP = 32 * NKP;
    typedef Img<D, NP> I;
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int bidx = xcd_block((int)blockIdx.x, (int)gridDim.x, ATTN_XCD);   // (sample-major: XCD x gets samples [x B / 8, ..))
    const int b = bidx / H, h = bidx % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int HD = H * D, RS = 3 * HD;
    const bf16_t* base = qkv + (long long)b * N * RS + h * D;
    bf16_t* dbase = dqkv + (long long)b * N * RS + h * D;
    if (keep_hd && h * D >= keep_hd[b]) {
        for (int i = tid; i < N * (D / 8); i += NW * 64) {
            const int n = i / (D / 8), ch = i % (D / 8);
            *reinterpret_cast<uint4*>(dbase + (long long)n * RS + HD + ch * 8) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(dbase + (long long)n * RS + 2 * HD + ch * 8) = make_uint4(0, 0, 0, 0);
        }
        return;
    }
    const bf16_t* gb = d_o + (long long)b * N * HD + h * D;
    const int nqp = FULL ? NKP : (N + 31) / 32;
    char* Qc = sm;
    char* Gc = Qc + I::BYTES;
    float* Ls = reinterpret_cast<float*>(Gc + I::BYTES);      // lse * log2(e) per query (0 beyond N)
    float* Ds = Ls + NP;
    int u0 = wave * 16;
    bfv8 kf[AC<D>::DK], vf[AC<D>::DK];                    // first key tile: in flight while Q / dO are staged
#pragma unroll
    for (int dk = 0; dk < AC<D>::DK; ++dk) {
        kf[dk] = gfrag<D>(base + HD, RS, u0, N, dk, lane);
        vf[dk] = gfrag<D>(base + 2 * HD, RS, u0, N, dk, lane);
    }
    for (int n = tid; n < NP; n += NW * 64) {
        Ls[n] = n < N ? lse[((long long)b * H + h) * N + n] * LOG2E : 0.f;
        Ds[n] = n < N ? delta[((long long)b * H + h) * N + n] : 0.f;
    }
    stage_chunked<D, NP, NW * 64>(Qc, base, RS, N, NP, tid);
    stage_chunked<D, NP, NW * 64>(Gc, gb, HD, N, NP, tid);
    __syncthreads();
    const char *qb[AC<D>::DK], *gcb[AC<D>::DK];
#pragma unroll
    for (int dk = 0; dk < AC<D>::DK; ++dk) {
        qb[dk] = I::opaque(Qc + I::cbase(dk, lane));
        gcb[dk] = I::opaque(Gc + I::cbase(dk, lane));
    }
    const char* qt_b = I::opaque(Qc + I::tbase(lane));
    const char* gt_b = I::opaque(Gc + I::tbase(lane));
    const float* lsg = Ls + 4 * g;
    const float* dsg = Ds + 4 * g;
    const float cs = scale * LOG2E;
    while (u0 < N) {
        f32x4 dka[AC<D>::DT], dva[AC<D>::DT];
#pragma unroll
        for (int dt = 0; dt < AC<D>::DT; ++dt) {
            dka[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            dva[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // queries beyond N are zero rows of Q and dO in LDS (and Ls = Ds = 0): P is finite, dS = 0 -- no masks
#pragma unroll(NKP <= 3 ? NKP : 1)
        for (int qp = 0; qp < nqp; ++qp) {
            f32x4 p[2], ds[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int q0 = qp * 32 + tt * 16;
                const float4 l4 = *reinterpret_cast<const float4*>(lsg + q0);
                const float4 d4 = *reinterpret_cast<const float4*>(dsg + q0);
                const float lr[4] = {l4.x, l4.y, l4.z, l4.w}, dr[4] = {d4.x, d4.y, d4.z, d4.w};
                f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dk = 0; dk < AC<D>::DK; ++dk) {
                    sc = mfma16(I::cread(qb[dk], q0), kf[dk], sc);
                    dp = mfma16(I::cread(gcb[dk], q0), vf[dk], dp);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = ex2(fmaf(sc[r], cs, -lr[r]));
                    p[tt][r] = pv;
                    ds[tt][r] = pv * ((dp[r] - dr[r]) * scale);
                }
            }
            const bfv8 pf = pack8(p[0], p[1]), sf = pack8(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < AC<D>::DT; ++dt) {
                dva[dt] = mfma16(I::tread(gt_b, qp, dt), pf, dva[dt]);
                dka[dt] = mfma16(I::tread(qt_b, qp, dt), sf, dka[dt]);
            }
            __builtin_amdgcn_sched_barrier(0);              // (unrolled short loops: see fwd_kernel)
        }
        const int k = u0 + c;
        u0 += NW * 16;
        if (u0 < N) {                                       // next key tile: issued before this tile's stores
#pragma unroll
            for (int dk = 0; dk < AC<D>::DK; ++dk) {
                kf[dk] = gfrag<D>(base + HD, RS, u0, N, dk, lane);
                vf[dk] = gfrag<D>(base + 2 * HD, RS, u0, N, dk, lane);
            }
        }
        if (k < N) {      // dK^T / dV^T tiles: key k, 4 consecutive d per tile
#pragma unroll
            for (int dt = 0; dt < AC<D>::DT; ++dt) {
                bf16_t* dst = dbase + (long long)k * RS + dt * 16 + 4 * g;
                *reinterpret_cast<uint2*>(dst + HD) =
                    make_uint2(pack_bf2(dka[dt][0], dka[dt][1]), pack_bf2(dka[dt][2], dka[dt][3]));
                *reinterpret_cast<uint2*>(dst + 2 * HD) =
                    make_uint2(pack_bf2(dva[dt][0], dva[dt][1]), pack_bf2(dva[dt][2], dva[dt][3]));
            }
        }
    }
}

// ==========================================================================================================
// backward, short sequences (N <= 96: the 65- and 17-token stages): dQ, dK, dV in ONE launch.  Q, K, V and dO of the head are
// staged once (53 KB at N = 65, D = 64 -- three workgroups per CU) and the two phases above run back to back on the staged
// images: the query-tile phase (delta, dQ) reads its Q / dO fragments from LDS instead of global memory and leaves delta in
// LDS for the key-tile phase.  The two-kernel form read q, k, v and dO twice and paid two launch tails on grids of 1024-2048
// small workgroups (22 + 15 us at 128 x 65 x 8 x 64).  At N = 257 the four images (152 KB) would leave one workgroup per
// CU: the long stage keeps the two kernels.
// ==========================================================================================================
template <int D, int MAXNP, int NTHR>
__device__ __forceinline__ void stage_load(uint4 (&v)[(MAXNP * AC<D>::NCH + NTHR - 1) / NTHR], const bf16_t* __restrict__ src, int rs,
                                           int N, int Np, int tid) {
    constexpr int NCH = AC<D>::NCH, HP = NCH / 2;
    constexpr int IT = (MAXNP * NCH + NTHR - 1) / NTHR;
    const int total = Np * NCH;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int idx = tid + it * NTHR;
        const int q = idx >> 3, l8 = idx & 7;
        const int rb = q / HP, cp = q - rb * HP;
        const int n = 4 * rb + (l8 >> 1), ch = 2 * cp + (l8 & 1);
        const bool ok = n < N && idx < total;
        v[it] = *reinterpret_cast<const uint4*>(src + (long long)(ok ? n : 0) * rs + (ok ? ch : 0) * 8);
        if (!ok) v[it] = make_uint4(0, 0, 0, 0);
    }
}
template <int D, int MAXNP, int NTHR>
__device__ __forceinline__ void stage_store(char* dst, const uint4 (&v)[(MAXNP * AC<D>::NCH + NTHR - 1) / NTHR], int Np, int tid) {
    const int NpS = Np + 8;
    constexpr int NCH = AC<D>::NCH, HP = NCH / 2;
    constexpr int IT = (MAXNP * NCH + NTHR - 1) / NTHR;
    const int total = Np * NCH;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int idx = tid + it * NTHR;
        const int q = idx >> 3, l8 = idx & 7;
        const int rb = q / HP, cp = q - rb * HP;
        const int n = 4 * rb + (l8 >> 1), ch = 2 * cp + (l8 & 1);
        if (idx < total) *reinterpret_cast<uint4*>(dst + ((size_t)ch * NpS + n) * 16) = v[it];
    }
}

template <int D, int NKP, int NW, int OCC, bool FULL>
__global__ __launch_bounds__(NW * 64, OCC) void bwd_short_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                                 const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                                 float* __restrict__ delta, bf16_t* __restrict__ dqkv,
                                                                 const int* __restrict__ keep_hd, int B, int N, int H,
                                                                 float scale, int ATTN_XCD) {
    constexpr int NP = 32 * NKP, NT = NW * 64;
    constexpr int IT = (NP * AC<D>::NCH + NT - 1) / NT;
    typedef Img<D, NP> I;
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int bidx = xcd_block((int)blockIdx.x, (int)gridDim.x, ATTN_XCD);   // (sample-major: XCD x gets samples [x B / 8, ..))
    const int b = bidx / H, h = bidx % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int HD = H * D, RS = 3 * HD;
    const bf16_t* base = qkv + (long long)b * N * RS + h * D;
    bf16_t* dbase = dqkv + (long long)b * N * RS + h * D;
    if (keep_hd && h * D >= keep_hd[b]) {
        for (int i = tid; i < N * (D / 8); i += NT) {
            const int n = i / (D / 8), ch = i % (D / 8);
            bf16_t* dst = dbase + (long long)n * RS + ch * 8;
            *reinterpret_cast<uint4*>(dst) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(dst + HD) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(dst + 2 * HD) = make_uint4(0, 0, 0, 0);
        }
        return;
    }
    const bf16_t* ob = o + (long long)b * N * HD + h * D;
    const bf16_t* gb = d_o + (long long)b * N * HD + h * D;
    const float* lb = lse + ((long long)b * H + h) * N;
    float* db = delta + ((long long)b * H + h) * N;
    char* Qc = sm;
    char* Kc = Qc + I::BYTES;
    char* Vc = Kc + I::BYTES;
    char* Gc = Vc + I::BYTES;
    float* Ls = reinterpret_cast<float*>(Gc + I::BYTES);      // lse * log2(e) per query (0 beyond N)
    float* Ds = Ls + NP;                                      // delta per query (0 beyond N), written by the query-tile phase
    int q0 = wave * 16;
    bfv8 of[AC<D>::DK];                                       // O of the first query tile: in flight while the head is staged
#pragma unroll
    for (int dk = 0; dk < AC<D>::DK; ++dk) of[dk] = gfrag<D>(ob, HD, q0, N, dk, lane);
    {
        uint4 vq[IT], vk[IT], vv[IT], vg[IT];                 // every global load of the four images before the first LDS store
        stage_load<D, NP, NT>(vq, base, RS, N, NP, tid);
        stage_load<D, NP, NT>(vk, base + HD, RS, N, NP, tid);
        stage_load<D, NP, NT>(vv, base + 2 * HD, RS, N, NP, tid);
        stage_load<D, NP, NT>(vg, gb, HD, N, NP, tid);
        for (int n = tid; n < NP; n += NT) {
            Ls[n] = n < N ? lb[n] * LOG2E : 0.f;
            Ds[n] = 0.f;
        }
        stage_store<D, NP, NT>(Qc, vq, NP, tid);
        stage_store<D, NP, NT>(Kc, vk, NP, tid);
        stage_store<D, NP, NT>(Vc, vv, NP, tid);
        stage_store<D, NP, NT>(Gc, vg, NP, tid);
    }
    __syncthreads();
    const char *qb[AC<D>::DK], *kb[AC<D>::DK], *vb[AC<D>::DK], *gcb[AC<D>::DK];
#pragma unroll
    for (int dk = 0; dk < AC<D>::DK; ++dk) {
        qb[dk] = I::opaque(Qc + I::cbase(dk, lane));
        kb[dk] = I::opaque(Kc + I::cbase(dk, lane));
        vb[dk] = I::opaque(Vc + I::cbase(dk, lane));
        gcb[dk] = I::opaque(Gc + I::cbase(dk, lane));
    }
    const char* kt_b = I::opaque(Kc + I::tbase(lane));
    const char* qt_b = I::opaque(Qc + I::tbase(lane));
    const char* gt_b = I::opaque(Gc + I::tbase(lane));
    const float cs = scale * LOG2E;
    // a fragment that plays the role the global-memory fragments play in the two-kernel form: D = 48 has 6 chunks, the lanes of
    // k-step 1 that would hold chunks 6 / 7 must supply zeros (cread clamps the chunk and relies on a zero partner)
    auto zfrag = [&](const char* lane_base, int row0, int dk) -> bfv8 {
        bfv8 v = I::cread(lane_base, row0);
        if constexpr (AC<D>::NCH % 4 != 0) {
            if (dk * 4 + (lane >> 4) >= AC<D>::NCH) v = __builtin_bit_cast(bfv8, make_uint4(0, 0, 0, 0));
        }
        return v;
    };

    // ---- phase A: per 16-query tile delta and dQ (bwd_dq_kernel with Q / dO fragments from the staged images) ----
    while (q0 < N) {
        bfv8 qf[AC<D>::DK], gf[AC<D>::DK];
#pragma unroll
        for (int dk = 0; dk < AC<D>::DK; ++dk) {
            qf[dk] = zfrag(qb[dk], q0, dk);
            gf[dk] = zfrag(gcb[dk], q0, dk);
        }
        float d = 0.f;
#pragma unroll
        for (int dk = 0; dk < AC<D>::DK; ++dk)
#pragma unroll
            for (int e = 0; e < 8; ++e) d += (float)gf[dk][e] * (float)of[dk][e];
        const float dl = gsum(d);
        const float l2 = Ls[q0 + c];
        if (g == 0 && q0 + c < N) {
            db[q0 + c] = dl;
            Ds[q0 + c] = dl;
        }
        f32x4 dq[AC<D>::DT];
#pragma unroll
        for (int dt = 0; dt < AC<D>::DT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kp = 0; kp < NKP; ++kp) {
            if (FULL || kp * 32 < N) {
                f32x4 ds[2];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int k0 = (2 * kp + tt) * 16;
                    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                    if (FULL || k0 < N) {
#pragma unroll
                        for (int dk = 0; dk < AC<D>::DK; ++dk) {
                            sc = mfma16(I::cread(kb[dk], k0), qf[dk], sc);
                            dp = mfma16(I::cread(vb[dk], k0), gf[dk], dp);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) sc[r] = ex2(fmaf(sc[r], cs, -l2)) * ((dp[r] - dl) * scale);
                    }
                    ds[tt] = sc;
                }
                const bfv8 sf = pack8(ds[0], ds[1]);
#pragma unroll
                for (int dt = 0; dt < AC<D>::DT; ++dt) dq[dt] = mfma16(I::tread(kt_b, kp, dt), sf, dq[dt]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const int qs = q0;
        q0 += NW * 16;
        if (q0 < N) {
#pragma unroll
            for (int dk = 0; dk < AC<D>::DK; ++dk) of[dk] = gfrag<D>(ob, HD, q0, N, dk, lane);
        }
        if (qs + c < N) {
#pragma unroll
            for (int dt = 0; dt < AC<D>::DT; ++dt)
                *reinterpret_cast<uint2*>(dbase + (long long)(qs + c) * RS + dt * 16 + 4 * g) =
                    make_uint2(pack_bf2(dq[dt][0], dq[dt][1]), pack_bf2(dq[dt][2], dq[dt][3]));
        }
    }
    __syncthreads();                                          // Ds complete

    // ---- phase B: per 16-key tile dK, dV (bwd_dkv_kernel with K / V fragments from the staged images) ----
    const int nqp = FULL ? NKP : (N + 31) / 32;
    const float* lsg = Ls + 4 * g;
    const float* dsg = Ds + 4 * g;
    for (int u0 = wave * 16; u0 < N; u0 += NW * 16) {
        bfv8 kf[AC<D>::DK], vf[AC<D>::DK];
#pragma unroll
        for (int dk = 0; dk < AC<D>::DK; ++dk) {
            kf[dk] = zfrag(kb[dk], u0, dk);
            vf[dk] = zfrag(vb[dk], u0, dk);
        }
        f32x4 dka[AC<D>::DT], dva[AC<D>::DT];
#pragma unroll
        for (int dt = 0; dt < AC<D>::DT; ++dt) {
            dka[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            dva[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int qp = 0; qp < NKP; ++qp) {
            if (qp < nqp) {
                f32x4 p[2], ds[2];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int qq = qp * 32 + tt * 16;
                    const float4 l4 = *reinterpret_cast<const float4*>(lsg + qq);
                    const float4 d4 = *reinterpret_cast<const float4*>(dsg + qq);
                    const float lr[4] = {l4.x, l4.y, l4.z, l4.w}, dr[4] = {d4.x, d4.y, d4.z, d4.w};
                    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int dk = 0; dk < AC<D>::DK; ++dk) {
                        sc = mfma16(I::cread(qb[dk], qq), kf[dk], sc);
                        dp = mfma16(I::cread(gcb[dk], qq), vf[dk], dp);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = ex2(fmaf(sc[r], cs, -lr[r]));
                        p[tt][r] = pv;
                        ds[tt][r] = pv * ((dp[r] - dr[r]) * scale);
                    }
                }
                const bfv8 pf = pack8(p[0], p[1]), sf = pack8(ds[0], ds[1]);
#pragma unroll
                for (int dt = 0; dt < AC<D>::DT; ++dt) {
                    dva[dt] = mfma16(I::tread(gt_b, qp, dt), pf, dva[dt]);
                    dka[dt] = mfma16(I::tread(qt_b, qp, dt), sf, dka[dt]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const int k = u0 + c;
        if (k < N) {
#pragma unroll
            for (int dt = 0; dt < AC<D>::DT; ++dt) {
                bf16_t* dst = dbase + (long long)k * RS + dt * 16 + 4 * g;
                *reinterpret_cast<uint2*>(dst + HD) =
                    make_uint2(pack_bf2(dka[dt][0], dka[dt][1]), pack_bf2(dka[dt][2], dka[dt][3]));
                *reinterpret_cast<uint2*>(dst + 2 * HD) =
                    make_uint2(pack_bf2(dva[dt][0], dva[dt][1]), pack_bf2(dva[dt][2], dva[dt][3]));
            }
        }
    }
}

// ==========================================================================================================
// N > 288 (fine-tuning at 280 / 336 / 392 px: N = 401 / 577 / 785, reference scripts/vit-sr-nas/finetune/*): the head no
// longer fits in LDS.  Same fragments and dataflow, but a workgroup owns 128 queries (or keys) and walks the other
// sequence in blocks of 256 rows staged in LDS one after the other.  The forward makes two passes over the key blocks --
// log-sum-exp first, then P = exp(s - lse) and O += V^T P^T exactly like the backward kernels recompute P -- instead of an
// online softmax with accumulator rescaling; QK^T is computed twice, on a path that is 5 % of the FLOPs.
// ==========================================================================================================
constexpr int LKB = 256, LQB = 128, LNW = 8;              // rows per staged block, rows per workgroup, waves

template <int D>
__global__ __launch_bounds__(LNW * 64) void fwd_long_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                            float* __restrict__ lse, const int* __restrict__ keep_hd, int B,
                                                            int N, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int nqb = (N + LQB - 1) / LQB;
    const int qb = blockIdx.x % nqb, bh = blockIdx.x / nqb, b = bh / H, h = bh % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int HD = H * D, RS = 3 * HD;
    const bf16_t* base = qkv + (long long)b * N * RS + h * D;
    bf16_t* ob = o + (long long)b * N * HD + h * D;
    float* lb = lse + ((long long)b * H + h) * N;
    const int q0 = qb * LQB + wave * 16;
    if (keep_hd && h * D >= keep_hd[b]) {
        for (int i = tid; i < LQB * (D / 8); i += LNW * 64) {
            const int n = qb * LQB + i / (D / 8), ch = i % (D / 8);
            if (n < N) *reinterpret_cast<uint4*>(ob + (long long)n * HD + ch * 8) = make_uint4(0, 0, 0, 0);
        }
        for (int n = qb * LQB + tid; n < min(N, (qb + 1) * LQB); n += LNW * 64) lb[n] = 0.f;
        return;
    }
    char* Kc = sm;
    char* Vc = sm + (size_t)D * (LKB + 8) * 2;
    bfv8 qf[AC<D>::DK];
#pragma unroll
    for (int dk = 0; dk < AC<D>::DK; ++dk) qf[dk] = gfrag<D>(base, RS, q0, N, dk, lane);
    const int nkb = (N + LKB - 1) / LKB;
    // ---- pass 1: log-sum-exp of every query over all key blocks ----
    float mx = -INFINITY, sum = 0.f;
    for (int kb = 0; kb < nkb; ++kb) {
        const int k0 = kb * LKB, nk = min(LKB, N - k0);
        __syncthreads();
        stage_chunked<D, LKB, LNW * 64>(Kc, base + HD + (long long)k0 * RS, RS, nk, LKB, tid);
        __syncthreads();
        f32x4 st[LKB / 16];
        float bm = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < LKB / 16; ++kt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dk = 0; dk < AC<D>::DK; ++dk) acc = mfma16(cfrag<D>(Kc, LKB, kt * 16, dk, lane), qf[dk], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[r] = (kt * 16 + 4 * g + r) < nk ? acc[r] * scale : -INFINITY;
                bm = fmaxf(bm, acc[r]);
            }
            st[kt] = acc;
        }
        bm = gmax(bm);
        const float nm = fmaxf(mx, bm);
        float bs = 0.f;
#pragma unroll
        for (int kt = 0; kt < LKB / 16; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) bs += __expf(st[kt][r] - nm);
        sum = sum * __expf(mx - nm) + gsum(bs);
        mx = nm;
    }
    const float l = mx + __logf(sum);
    if (g == 0 && q0 + c < N) lb[q0 + c] = l;
    // ---- pass 2: O^T += V^T P^T with P = exp(s - lse) ----
    f32x4 oacc[AC<D>::DT];
#pragma unroll
    for (int dt = 0; dt < AC<D>::DT; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < nkb; ++kb) {
        const int k0 = kb * LKB, nk = min(LKB, N - k0);
        __syncthreads();
        stage_chunked<D, LKB, LNW * 64>(Kc, base + HD + (long long)k0 * RS, RS, nk, LKB, tid);
        stage_chunked<D, LKB, LNW * 64>(Vc, base + 2 * HD + (long long)k0 * RS, RS, nk, LKB, tid);
        __syncthreads();
#pragma unroll
        for (int kp = 0; kp < LKB / 32; ++kp) {
            f32x4 pr[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int kt = 2 * kp + tt;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dk = 0; dk < AC<D>::DK; ++dk) acc = mfma16(cfrag<D>(Kc, LKB, kt * 16, dk, lane), qf[dk], acc);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = (kt * 16 + 4 * g + r) < nk ? __expf(acc[r] * scale - l) : 0.f;
                pr[tt] = acc;
            }
            const bfv8 pf = pack8(pr[0], pr[1]);
#pragma unroll
            for (int dt = 0; dt < AC<D>::DT; ++dt) oacc[dt] = mfma16(tfrag<D>(Vc, LKB, kp, dt, lane), pf, oacc[dt]);
        }
    }
    if (q0 + c < N) {
#pragma unroll
        for (int dt = 0; dt < AC<D>::DT; ++dt)
            *reinterpret_cast<uint2*>(ob + (long long)(q0 + c) * HD + dt * 16 + 4 * g) =
                make_uint2(pack_bf2(oacc[dt][0], oacc[dt][1]), pack_bf2(oacc[dt][2], oacc[dt][3]));
    }
}

template <int D>
__global__ __launch_bounds__(LNW * 64) void bwd_dq_long_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                               const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                               float* __restrict__ delta, bf16_t* __restrict__ dqkv,
                                                               const int* __restrict__ keep_hd, int B, int N, int H,
                                                               float scale) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int nqb = (N + LQB - 1) / LQB;
    const int qb = blockIdx.x % nqb, bh = blockIdx.x / nqb, b = bh / H, h = bh % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int HD = H * D, RS = 3 * HD;
    const bf16_t* base = qkv + (long long)b * N * RS + h * D;
    bf16_t* dbase = dqkv + (long long)b * N * RS + h * D;
    if (keep_hd && h * D >= keep_hd[b]) {
        for (int i = tid; i < LQB * (D / 8); i += LNW * 64) {
            const int n = qb * LQB + i / (D / 8), ch = i % (D / 8);
            if (n < N) *reinterpret_cast<uint4*>(dbase + (long long)n * RS + ch * 8) = make_uint4(0, 0, 0, 0);
        }
        return;
    }
    const bf16_t* ob = o + (long long)b * N * HD + h * D;
    const bf16_t* gb = d_o + (long long)b * N * HD + h * D;
    const float* lb = lse + ((long long)b * H + h) * N;
    float* db = delta + ((long long)b * H + h) * N;
    char* Kc = sm;
    char* Vc = Kc + (size_t)D * (LKB + 8) * 2;
    const int q0 = qb * LQB + wave * 16;
    const bool qok = q0 + c < N;
    bfv8 qf[AC<D>::DK], gf[AC<D>::DK];
    float dl = 0.f;
#pragma unroll
    for (int dk = 0; dk < AC<D>::DK; ++dk) {
        qf[dk] = gfrag<D>(base, RS, q0, N, dk, lane);
        gf[dk] = gfrag<D>(gb, HD, q0, N, dk, lane);
        const bfv8 of = gfrag<D>(ob, HD, q0, N, dk, lane);
#pragma unroll
        for (int e = 0; e < 8; ++e) dl += (float)gf[dk][e] * (float)of[e];
    }
    dl = gsum(dl);
    const float l = qok ? lb[q0 + c] : 0.f;
    if (g == 0 && qok) db[q0 + c] = dl;
    f32x4 dq[AC<D>::DT];
#pragma unroll
    for (int dt = 0; dt < AC<D>::DT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nkb = (N + LKB - 1) / LKB;
    for (int kb = 0; kb < nkb; ++kb) {
        const int k0 = kb * LKB, nk = min(LKB, N - k0);
        __syncthreads();
        stage_chunked<D, LKB, LNW * 64>(Kc, base + HD + (long long)k0 * RS, RS, nk, LKB, tid);
        stage_chunked<D, LKB, LNW * 64>(Vc, base + 2 * HD + (long long)k0 * RS, RS, nk, LKB, tid);
        __syncthreads();
#pragma unroll
        for (int kp = 0; kp < LKB / 32; ++kp) {
            f32x4 ds[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int kt = 2 * kp + tt;
                f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dk = 0; dk < AC<D>::DK; ++dk) {
                    sc = mfma16(cfrag<D>(Kc, LKB, kt * 16, dk, lane), qf[dk], sc);
                    dp = mfma16(cfrag<D>(Vc, LKB, kt * 16, dk, lane), gf[dk], dp);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = qok && ((kt * 16 + 4 * g + r) < nk);
                    const float pr = ok ? __expf(sc[r] * scale - l) : 0.f;
                    sc[r] = pr * (dp[r] - dl) * scale;
                }
                ds[tt] = sc;
            }
            const bfv8 sf = pack8(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < AC<D>::DT; ++dt) dq[dt] = mfma16(tfrag<D>(Kc, LKB, kp, dt, lane), sf, dq[dt]);
        }
    }
    if (qok) {
#pragma unroll
        for (int dt = 0; dt < AC<D>::DT; ++dt)
            *reinterpret_cast<uint2*>(dbase + (long long)(q0 + c) * RS + dt * 16 + 4 * g) =
                make_uint2(pack_bf2(dq[dt][0], dq[dt][1]), pack_bf2(dq[dt][2], dq[dt][3]));
    }
}

template <int D>
__global__ __launch_bounds__(LNW * 64) void bwd_dkv_long_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ d_o,
                                                                const float* __restrict__ lse, const float* __restrict__ delta,
                                                                bf16_t* __restrict__ dqkv, const int* __restrict__ keep_hd,
                                                                int B, int N, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int nkbk = (N + LQB - 1) / LQB;
    const int kblk = blockIdx.x % nkbk, bh = blockIdx.x / nkbk, b = bh / H, h = bh % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int HD = H * D, RS = 3 * HD;
    const bf16_t* base = qkv + (long long)b * N * RS + h * D;
    bf16_t* dbase = dqkv + (long long)b * N * RS + h * D;
    if (keep_hd && h * D >= keep_hd[b]) {
        for (int i = tid; i < LQB * (D / 8); i += LNW * 64) {
            const int n = kblk * LQB + i / (D / 8), ch = i % (D / 8);
            if (n < N) {
                *reinterpret_cast<uint4*>(dbase + (long long)n * RS + HD + ch * 8) = make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4*>(dbase + (long long)n * RS + 2 * HD + ch * 8) = make_uint4(0, 0, 0, 0);
            }
        }
        return;
    }
    const bf16_t* gb = d_o + (long long)b * N * HD + h * D;
    const float* lb = lse + ((long long)b * H + h) * N;
    const float* db = delta + ((long long)b * H + h) * N;
    char* Qc = sm;
    char* Gc = Qc + (size_t)D * (LKB + 8) * 2;
    float* Ls = reinterpret_cast<float*>(Gc + (size_t)D * (LKB + 8) * 2);
    float* Ds = Ls + LKB;
    const int k0 = kblk * LQB + wave * 16;
    bfv8 kf[AC<D>::DK], vf[AC<D>::DK];
#pragma unroll
    for (int dk = 0; dk < AC<D>::DK; ++dk) {
        kf[dk] = gfrag<D>(base + HD, RS, k0, N, dk, lane);
        vf[dk] = gfrag<D>(base + 2 * HD, RS, k0, N, dk, lane);
    }
    f32x4 dka[AC<D>::DT], dva[AC<D>::DT];
#pragma unroll
    for (int dt = 0; dt < AC<D>::DT; ++dt) {
        dka[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        dva[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int nqb = (N + LKB - 1) / LKB;
    for (int qb = 0; qb < nqb; ++qb) {
        const int q0b = qb * LKB, nq = min(LKB, N - q0b);
        __syncthreads();
        stage_chunked<D, LKB, LNW * 64>(Qc, base + (long long)q0b * RS, RS, nq, LKB, tid);
        stage_chunked<D, LKB, LNW * 64>(Gc, gb + (long long)q0b * HD, HD, nq, LKB, tid);
        for (int n = tid; n < LKB; n += LNW * 64) {
            Ls[n] = n < nq ? lb[q0b + n] : 0.f;
            Ds[n] = n < nq ? db[q0b + n] : 0.f;
        }
        __syncthreads();
#pragma unroll 1
        for (int qp = 0; qp < LKB / 32; ++qp) {
            f32x4 p[2], ds[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int q0 = qp * 32 + tt * 16;
                f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dk = 0; dk < AC<D>::DK; ++dk) {
                    sc = mfma16(cfrag<D>(Qc, LKB, q0, dk, lane), kf[dk], sc);
                    dp = mfma16(cfrag<D>(Gc, LKB, q0, dk, lane), vf[dk], dp);
                }
                const float4 l4 = *reinterpret_cast<const float4*>(Ls + q0 + 4 * g);
                const float4 d4 = *reinterpret_cast<const float4*>(Ds + q0 + 4 * g);
                const float lr[4] = {l4.x, l4.y, l4.z, l4.w}, dr[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = (q0 + 4 * g + r) < nq;
                    const float pv = ok ? __expf(sc[r] * scale - lr[r]) : 0.f;
                    p[tt][r] = pv;
                    ds[tt][r] = pv * (dp[r] - dr[r]) * scale;
                }
            }
            const bfv8 pf = pack8(p[0], p[1]), sf = pack8(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < AC<D>::DT; ++dt) {
                dva[dt] = mfma16(tfrag<D>(Gc, LKB, qp, dt, lane), pf, dva[dt]);
                dka[dt] = mfma16(tfrag<D>(Qc, LKB, qp, dt, lane), sf, dka[dt]);
            }
        }
    }
    if (k0 + c < N) {
#pragma unroll
        for (int dt = 0; dt < AC<D>::DT; ++dt) {
            bf16_t* dst = dbase + (long long)(k0 + c) * RS + dt * 16 + 4 * g;
            *reinterpret_cast<uint2*>(dst + HD) = make_uint2(pack_bf2(dka[dt][0], dka[dt][1]), pack_bf2(dka[dt][2], dka[dt][3]));
            *reinterpret_cast<uint2*>(dst + 2 * HD) = make_uint2(pack_bf2(dva[dt][0], dva[dt][1]), pack_bf2(dva[dt][2], dva[dt][3]));
        }
    }
}

// ---- host dispatch ---------------------------------------------------------------------------------------
// Raising the dynamic-LDS limit is a per-function, idempotent driver call; it is made once per (kernel, size class)
// so that later launches (e.g. under hipGraph stream capture) are pure stream work.
template <typename K> static int set_lds(K kernel, size_t bytes) {
    static size_t granted = 0;     // one instance per kernel type K
    if (bytes > 64 * 1024 && bytes > granted) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)bytes);
        if (e != hipSuccess) return (int)e;
        granted = bytes;
    }
    return 0;
}

// waves per (batch, head) and workgroups per CU the register budget is set for (a multiple of the 4 SIMDs, see above)
template <int NKP> struct Plan {
    // (NKP = 3, N = 65 = 5 tiles of 16: five waves in one round instead of four waves in two were measured slower, bwd 27 -> 34 us;
    //  N = 257 with four waves -- 17 tiles in 5 rounds instead of 3 rounds of eight -- also: fwd 25.1 -> 28.8, bwd 63 -> 71 us)
    static constexpr int NW = NKP >= 9 ? 8 : 4;
    static constexpr int OCC = 4;                            // NW = 8: two workgroups per CU; NW = 4: four
    static constexpr int PB = NKP >= 9 ? 3 : NKP;
};

// sample-major block order per XCD (common.h xcd_block)
static constexpr int attn_xcd() { return 1; }
template <int D, int NKP>
static int launch_fwd(const bf16_t* qkv, bf16_t* o, float* lse, const int* keep, int B, int N, int H, float scale,
                      hipStream_t st) {
    constexpr int NW = Plan<NKP>::NW, OCC = Plan<NKP>::OCC, PB = Plan<NKP>::PB;
    const size_t lds = (size_t)2 * Img<D, 32 * NKP>::BYTES;
    // FULL: the last pair of key tiles is in use (N = 257 / 65 / 17 and every N > 32 (NKP - 1)): no run-time tile guards
    if (N > 32 * (NKP - 1)) {
        int rc = set_lds(fwd_kernel<D, NKP, NW, PB, OCC, true>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((fwd_kernel<D, NKP, NW, PB, OCC, true>), dim3(B * H), dim3(NW * 64), lds, st, qkv, o, lse, keep, B, N, H,
                           scale, attn_xcd());
    } else {
        int rc = set_lds(fwd_kernel<D, NKP, NW, PB, OCC, false>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((fwd_kernel<D, NKP, NW, PB, OCC, false>), dim3(B * H), dim3(NW * 64), lds, st, qkv, o, lse, keep, B, N, H,
                           scale, attn_xcd());
    }
    return 0;
}
template <int D, int NKP>
static int launch_bwd(const bf16_t* qkv, const bf16_t* o, const bf16_t* d_o, const float* lse, float* delta, bf16_t* dqkv,
                      const int* keep, int B, int N, int H, float scale, hipStream_t st) {
    constexpr int NW = Plan<NKP>::NW, OCC = Plan<NKP>::OCC;
    const size_t l1 = (size_t)2 * Img<D, 32 * NKP>::BYTES, l2 = l1 + 2 * 32 * NKP * sizeof(float);
    {       // one launch for dQ, dK, dV where the four staged images leave a CU two workgroups or more: the short sequences
            // and N = 257 at D = 32 (76 KB: 50.5 -> 41.9 us at 64 x 257 x 8 x 32; D = 64 would be alone on its CU: 63 -> 69 us).
        constexpr size_t L3 = (size_t)4 * Img<D, 32 * NKP>::BYTES + 2 * 32 * NKP * sizeof(float);
        constexpr int SOCC = NKP <= 3 ? 3 : (L3 <= 80 * 1024 ? 2 : 1);
        const bool merged = SOCC >= 2;
        if (merged) {
            const size_t l3 = (size_t)4 * Img<D, 32 * NKP>::BYTES + 2 * 32 * NKP * sizeof(float);
            if (N > 32 * (NKP - 1)) {
                int rc = set_lds(bwd_short_kernel<D, NKP, NW, SOCC, true>, l3);
                if (rc) return rc;
                hipLaunchKernelGGL((bwd_short_kernel<D, NKP, NW, SOCC, true>), dim3(B * H), dim3(NW * 64), l3, st, qkv, o, d_o, lse, delta,
                                   dqkv, keep, B, N, H, scale, attn_xcd());
            } else {
                int rc = set_lds(bwd_short_kernel<D, NKP, NW, SOCC, false>, l3);
                if (rc) return rc;
                hipLaunchKernelGGL((bwd_short_kernel<D, NKP, NW, SOCC, false>), dim3(B * H), dim3(NW * 64), l3, st, qkv, o, d_o, lse, delta,
                                   dqkv, keep, B, N, H, scale, attn_xcd());
            }
            return 0;
        }
    }
    if (N > 32 * (NKP - 1)) {
        int rc = set_lds(bwd_dq_kernel<D, NKP, NW, OCC, true>, l1);
        if (rc) return rc;
        rc = set_lds(bwd_dkv_kernel<D, NKP, NW, OCC, true>, l2);
        if (rc) return rc;
        hipLaunchKernelGGL((bwd_dq_kernel<D, NKP, NW, OCC, true>), dim3(B * H), dim3(NW * 64), l1, st, qkv, o, d_o, lse, delta, dqkv,
                           keep, B, N, H, scale, attn_xcd());
        hipLaunchKernelGGL((bwd_dkv_kernel<D, NKP, NW, OCC, true>), dim3(B * H), dim3(NW * 64), l2, st, qkv, d_o, lse, delta, dqkv,
                           keep, B, N, H, scale, attn_xcd());
    } else {
        int rc = set_lds(bwd_dq_kernel<D, NKP, NW, OCC, false>, l1);
        if (rc) return rc;
        rc = set_lds(bwd_dkv_kernel<D, NKP, NW, OCC, false>, l2);
        if (rc) return rc;
        hipLaunchKernelGGL((bwd_dq_kernel<D, NKP, NW, OCC, false>), dim3(B * H), dim3(NW * 64), l1, st, qkv, o, d_o, lse, delta, dqkv,
                           keep, B, N, H, scale, attn_xcd());
        hipLaunchKernelGGL((bwd_dkv_kernel<D, NKP, NW, OCC, false>), dim3(B * H), dim3(NW * 64), l2, st, qkv, d_o, lse, delta, dqkv,
                           keep, B, N, H, scale, attn_xcd());
    }
    return 0;
}

template <int D>
static int launch_fwd_long(const bf16_t* qkv, bf16_t* o, float* lse, const int* keep, int B, int N, int H, float scale,
                           hipStream_t st) {
    const size_t lds = (size_t)2 * D * (LKB + 8) * 2;
    int rc = set_lds(fwd_long_kernel<D>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((fwd_long_kernel<D>), dim3(B * H * ((N + LQB - 1) / LQB)), dim3(LNW * 64), lds, st, qkv, o, lse, keep, B, N, H,
                       scale);
    return 0;
}
template <int D>
static int launch_bwd_long(const bf16_t* qkv, const bf16_t* o, const bf16_t* d_o, const float* lse, float* delta, bf16_t* dqkv,
                           const int* keep, int B, int N, int H, float scale, hipStream_t st) {
    const size_t l1 = (size_t)2 * D * (LKB + 8) * 2, l2 = l1 + 2 * LKB * sizeof(float);
    int rc = set_lds(bwd_dq_long_kernel<D>, l1);
    if (rc) return rc;
    rc = set_lds(bwd_dkv_long_kernel<D>, l2);
    if (rc) return rc;
    const unsigned grid = (unsigned)(B * H * ((N + LQB - 1) / LQB));
    hipLaunchKernelGGL((bwd_dq_long_kernel<D>), dim3(grid), dim3(LNW * 64), l1, st, qkv, o, d_o, lse, delta, dqkv, keep, B, N, H, scale);
    hipLaunchKernelGGL((bwd_dkv_long_kernel<D>), dim3(grid), dim3(LNW * 64), l2, st, qkv, d_o, lse, delta, dqkv, keep, B, N, H, scale);
    return 0;
}

#define VR_ATTN_DISPATCH(FN, ...)                                            \
    do {                                                                     \
        const int nkp = (N + 31) / 32;                                       \
        if (D == 64) {                                                       \
            if (nkp <= 1) return FN<64, 1>(__VA_ARGS__);                     \
            if (nkp <= 3) return FN<64, 3>(__VA_ARGS__);                     \
            return FN<64, 9>(__VA_ARGS__);                                   \
        } else if (D == 48) {                                                \
            if (nkp <= 1) return FN<48, 1>(__VA_ARGS__);                     \
            if (nkp <= 3) return FN<48, 3>(__VA_ARGS__);                     \
            return FN<48, 9>(__VA_ARGS__);                                   \
        } else {                                                             \
            if (nkp <= 1) return FN<32, 1>(__VA_ARGS__);                     \
            if (nkp <= 3) return FN<32, 3>(__VA_ARGS__);                     \
            return FN<32, 9>(__VA_ARGS__);                                   \
        }                                                                    \
    } while (0)

bool supported(int N, int H, int D) {
    return (D == 32 || D == 48 || D == 64) && N >= 1 && ((H * D) % 8 == 0);
}

int fwd(const void* qkv, void* o, float* lse, const int* keep, int B, int N, int H, int D, float scale, hipStream_t st) {
    if (N > 288) {
        if (D == 64) return launch_fwd_long<64>((const bf16_t*)qkv, (bf16_t*)o, lse, keep, B, N, H, scale, st);
        if (D == 48) return launch_fwd_long<48>((const bf16_t*)qkv, (bf16_t*)o, lse, keep, B, N, H, scale, st);
        return launch_fwd_long<32>((const bf16_t*)qkv, (bf16_t*)o, lse, keep, B, N, H, scale, st);
    }
    VR_ATTN_DISPATCH(launch_fwd, (const bf16_t*)qkv, (bf16_t*)o, lse, keep, B, N, H, scale, st);
}
int bwd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv, const int* keep, int B,
        int N, int H, int D, float scale, hipStream_t st) {
    if (N > 288) {
        const bf16_t *q = (const bf16_t*)qkv, *oo = (const bf16_t*)o, *go = (const bf16_t*)d_o;
        if (D == 64) return launch_bwd_long<64>(q, oo, go, lse, delta, (bf16_t*)dqkv, keep, B, N, H, scale, st);
        if (D == 48) return launch_bwd_long<48>(q, oo, go, lse, delta, (bf16_t*)dqkv, keep, B, N, H, scale, st);
        return launch_bwd_long<32>(q, oo, go, lse, delta, (bf16_t*)dqkv, keep, B, N, H, scale, st);
    }
    VR_ATTN_DISPATCH(launch_bwd, (const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)d_o, lse, delta, (bf16_t*)dqkv, keep, B,
                     N, H, scale, st);
}

}  // namespace vr_attn_mfma
