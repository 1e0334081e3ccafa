// Fused forward MLP of a transformer block for the forward-only paths (evaluate(), candidate scoring):
//     out = resid + scale[s] * mask_{keep_out}( mask_{keep_hid}(gelu(y W1^T + b1)) W2^T + b2 )
// Reference: Mlp.forward + the block's drop_path / channel masks / residual add (nets/supernet_blocks.py:37-52,247-253).
// The hidden tensor [rows, F] never exists in memory: the two-GEMM form writes and re-reads it (252 MB of the pair's 546 MB
// at B = 256, C = 320, F = 960), this kernel moves y (bf16), resid and out (fp32) once and streams the weights from L2.
//
// One 512-thread workgroup per CU walks 128-row tiles.  Waves are 4 (rows) x 2 (columns); a wave owns 32 rows:
//   * its rows of y live in REGISTERS as MFMA A fragments for the whole tile (2 x C/32 fragments: 64 - 80 VGPRs) -- y is
//     read from memory once and never re-read from LDS;
//   * the hidden dimension is walked in chunks of 32: GEMM 1 (wave: 32 rows x 16 hidden, K = C) -> + bias, GELU, hidden keep
//     mask -> bf16 -> the chunk's h tile [128][32] in LDS (8 KB; the two column waves of a row group exchange their halves) ->
//     GEMM 2 (wave: 32 rows x every other 16-column fragment, K = 32) into 2 x C/32 accumulators that stay in registers for the whole tile;
//   * W1[chunk] (32 x C) and W2[:, chunk] (C x 32) arrive by LDS-DMA, double-buffered: chunk c + 1 is requested right after the
//     barrier that opens chunk c, so a chunk's 40 KB ride behind 80 MFMAs per wave.
// LDS reads per MFMA: 0.55 x 1 KB (W1: 1 fragment per 2 MFMAs, W2 / h: 12 per 20) -- about half the LDS read rate when the
// matrix pipes are full.  Masked work is skipped as in vr_gemm: chunks beyond the tile's largest kept hidden prefix, K steps
// beyond its largest kept input prefix, output fragments beyond its largest kept output prefix.
//
// LDS images (all swizzles are applied on the SOURCE side of the LDS-DMA, reads apply the same XOR):
//   W1 chunk : C/64 slices of [32 hidden rows][128 B]; slot p of row r holds k-chunk p ^ ((r >> 1) & 7)      (gemm_nt.hip's image)
//   W2 chunk : [C output rows][64 B]; slot p (16 B) of row n holds hidden chunk p ^ PERM[(n >> 2) & 3], PERM = {0, 3, 2, 1}
//   h chunk  : [128 rows][64 B], same rule as W2 (ds_read_b128 lane groups {0-3, 12-15, 20-27} ... then touch 16 different
//              16-byte slots of the 256-B bank row)
#include <cstdlib>
#include <type_traits>

#include "../common.h"
#include "../gemm_shared.h"
#include "../../../include/vitres_hip_experimental.h"

namespace vr_mlp {
using namespace vr_gemm_shared;

typedef __bf16 bfv8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

constexpr int NTHR = 512, BM = 128, FC = 32;

__device__ const uint4 zero_chunk[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};

__device__ __forceinline__ int perm4(int q) { return (4 - q) & 3; }       // {0, 3, 2, 1}

// KS = MFMA K steps of GEMM 1 = 16-column output fragments per wave of GEMM 2: the kernel covers C <= 32 KS
template <int KS>
__global__ __launch_bounds__(NTHR, 1) void mlp_fwd_kernel(const vr_mlp_args p) {
    constexpr int CP = 32 * KS;                       // padded width
    constexpr int NSL = KS / 2;                       // 64-wide k slices of a W1 chunk
    constexpr int W1_BYTES = NSL * 32 * 128;          // 16 / 20 KB
    constexpr int W2_BYTES = CP * 64;                 // 16 / 20 KB
    constexpr int NP = 2 * KS;                        // 1 KB LDS-DMA pieces per W1 chunk and per W2 chunk
    constexpr int PPW = (NP + 7) / 8;                 // pieces per wave
    __shared__ __attribute__((aligned(1024))) char smem[2 * W1_BYTES + 2 * W2_BYTES + BM * 64];
    __shared__ __attribute__((aligned(16))) float b1s[2048];
    char* const w1buf = smem;
    char* const w2buf = smem + 2 * W1_BYTES;
    char* const hbuf = smem + 2 * W1_BYTES + 2 * W2_BYTES;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, g = lane >> 4;
    const RowMap rmap = {p.map.rpi, p.map.rps, p.map.off};
    const char* zero = reinterpret_cast<const char*>(zero_chunk);

    for (int f = t; f < p.F && f < 2048; f += NTHR) b1s[f] = p.b1 ? p.b1[f] : 0.f;

    // ---- LDS-DMA source addressing (chunk 0; a chunk advances W1 by 32 rows, W2 by 64 bytes) ----
    int oW1[PPW], oW2[PPW];                            // byte offsets from W1 / W2 (chunk 0), -1: this lane's piece is all zero
    int rW1[PPW], kW2[PPW];
#pragma unroll
    for (int h = 0; h < PPW; ++h) {
        const int q = wave + 8 * h;                   // piece
        {   // W1: piece = 8 hidden rows x 128 B of slice q / 4
            const int r = 8 * (q & 3) + (lane >> 3), sl = q >> 2;
            const int ch = (lane & 7) ^ ((r >> 1) & 7);
            const int k = sl * 64 + ch * 8;
            rW1[h] = r;
            oW1[h] = (q < NP && k + 8 <= p.ldw1 && k < p.C) ? (r * p.ldw1 + k) * 2 : -1;
        }
        {   // W2: piece = 16 output rows x 64 B
            const int n = 16 * q + (lane >> 2);
            const int ch = (lane & 3) ^ perm4((n >> 2) & 3);
            kW2[h] = ch * 8;
            oW2[h] = (q < NP && n < p.C) ? (n * p.ldw2 + ch * 8) * 2 : -1;
        }
    }
    const char* const w1g = reinterpret_cast<const char*>(p.w1);
    const char* const w2g = reinterpret_cast<const char*>(p.w2);
    auto issue = [&](int c, int buf) {
        const long long adv1 = (long long)c * FC * p.ldw1 * 2;
#pragma unroll
        for (int h = 0; h < PPW; ++h) {
            const int q = wave + 8 * h;
            if (q < NP) {
                const char* s1 = (oW1[h] >= 0 && FC * c + rW1[h] < p.F) ? w1g + adv1 + oW1[h] : zero;
                __builtin_amdgcn_global_load_lds((glb_void*)s1, (lds_void*)(w1buf + buf * W1_BYTES + q * 1024), 16, 0, 0);
                const char* s2 = (oW2[h] >= 0 && FC * c + kW2[h] + 8 <= p.ldw2 && FC * c + kW2[h] < p.F) ? w2g + c * (FC * 2) + oW2[h] : zero;
                __builtin_amdgcn_global_load_lds((glb_void*)s2, (lds_void*)(w2buf + buf * W2_BYTES + q * 1024), 16, 0, 0);
            }
        }
    };

    // ---- fragment read offsets ----
    const int f1row = 16 * wn + li;                                   // hidden row of this lane's W1 fragment
    const int f1swz = (f1row >> 1) & 7;
    int offW1[2];                                                     // k-chunk g (+ 4) of a slice
    offW1[0] = f1row * 128 + (((g) ^ f1swz) << 4);
    offW1[1] = f1row * 128 + (((4 + g) ^ f1swz) << 4);
    int offH[2], offHw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 32 * wm + 16 * i + li;
        offH[i] = r * 64 + ((g ^ perm4((r >> 2) & 3)) << 4);                              // read: hidden 8 g .. 8 g + 8
        offHw[i] = r * 64 + ((((2 * wn + (g >> 1)) ^ perm4((r >> 2) & 3))) << 4) + (g & 1) * 8;   // write: hidden 16 wn + 4 g .. + 4
    }
    // output fragment j of this wave = columns 32 j + 16 wn .. + 16 (the two column waves interleave, so that a kept prefix of the
    // outputs cuts both waves' work alike); W2 row n = 32 j + 16 wn + li: (n >> 2) & 3 does not depend on j -> offW2 + 2048 j
    const int offW2 = (16 * wn + li) * 64 + ((g ^ perm4((li >> 2) & 3)) << 4);

    const int tiles = (p.M + BM - 1) / BM;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int m0 = tile * BM;
        // ---- masked-work bounds of this tile ----
        int fmax = p.F, kmax = p.C, nmax = p.C;
        if (p.keep_hid || p.keep_in || p.keep_out) {
            int s_lo = 0, s_hi = 0;
            if (p.rows_in > 0) { s_lo = m0 / p.rows_in; s_hi = (min(m0 + BM, p.M) - 1) / p.rows_in; }
            fmax = min(p.F, max_keep(p.keep_hid, s_lo, s_hi, p.F));
            kmax = min(p.C, max_keep(p.keep_in, s_lo, s_hi, p.C));
            nmax = min(p.C, max_keep(p.keep_out, s_lo, s_hi, p.C));
        }
        const int nch = (fmax + FC - 1) / FC;
        const int ksmax = (kmax + 31) / 32;

        // ---- this wave's rows: metadata + y fragments (registers) ----
        long long orow[2];
        int khid[2], kout[2];
        float scl[2];
        bool rok[2];
        bfv8 ya[2][KS];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + 32 * wm + 16 * i + li;
            rok[i] = m < p.M;
            const int mc = rok[i] ? m : p.M - 1;
            const int s = p.rows_in > 0 ? mc / p.rows_in : 0;
            orow[i] = map_row(rmap, mc);
            khid[i] = p.keep_hid ? p.keep_hid[s] : p.F;
            kout[i] = p.keep_out ? p.keep_out[s] : p.C;
            scl[i] = p.scale ? p.scale[s] : 1.0f;
            const bf16_t* yr = reinterpret_cast<const bf16_t*>(p.y) + orow[i] * p.ldy;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int k = 32 * ks + 8 * g;
                const bool ok = rok[i] && ks < ksmax && k + 8 <= p.C;
                const uint4 u = *reinterpret_cast<const uint4*>(ok ? reinterpret_cast<const char*>(yr + k) : zero);
                ya[i][ks] = __builtin_bit_cast(bfv8, u);
            }
        }
        f32x4 acc2[2][KS];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < KS; ++j) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // KA = active K steps of GEMM 1 = active output fragments of GEMM 2 (the kept input / output prefix of the widest sample of
        // the tile, in units of 32): a compile-time bound per case, so that every loop below is straight-line code whose LDS reads
        // the compiler batches ahead of the MFMAs (a uniform branch per step serialised read latency + MFMA per step: 2.1x slower)
        const int ka = min(KS, max(ksmax, (nmax + 31) / 32));
        auto chunks = [&](auto ka_tag) {
            constexpr int KA = decltype(ka_tag)::value;
            __syncthreads();                          // (b1s; the previous tile's last GEMM 2 has left the buffers)
            if (nch > 0) issue(0, 0);
            for (int c = 0; c < nch; ++c) {
                const int buf = c & 1;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();                      // chunk c landed for everyone; everyone is past GEMM 2 of chunk c - 1
                if (c + 1 < nch) issue(c + 1, buf ^ 1);
                // ---- GEMM 1: h_pre[32 rows][16 hidden] ----
                f32x4 acc1[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                const char* w1 = w1buf + buf * W1_BYTES;
#pragma unroll
                for (int ks = 0; ks < KA; ++ks) {
                    const bfv8 b = *reinterpret_cast<const bfv8*>(w1 + (ks >> 1) * 4096 + offW1[ks & 1]);
                    acc1[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, ya[0][ks], acc1[0], 0, 0, 0);
                    acc1[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, ya[1][ks], acc1[1], 0, 0, 0);
                    if (KA > 6 && ks == KA / 2 - 1) __builtin_amdgcn_sched_barrier(0);      // (at most KA / 2 fragments read ahead: registers)
                }
                // ---- bias, GELU, hidden keep mask, bf16, exchange through LDS ----
                const int f0 = FC * c + 16 * wn + 4 * g;
                const float4 bb = *reinterpret_cast<const float4*>(&b1s[f0 < 2048 - 4 ? f0 : 0]);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float v[4] = {acc1[i][0] + bb.x, acc1[i][1] + bb.y, acc1[i][2] + bb.z, acc1[i][3] + bb.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (f0 + r < khid[i]) ? gelu_fast(v[r]) : 0.f;
                    *reinterpret_cast<uint2*>(hbuf + offHw[i]) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                }
                __syncthreads();                      // h chunk complete
                // ---- GEMM 2: out[32 rows][this wave's fragments] += h[32 rows][32] W2[.][32]^T ----
                const bfv8 ah0 = *reinterpret_cast<const bfv8*>(hbuf + offH[0]);
                const bfv8 ah1 = *reinterpret_cast<const bfv8*>(hbuf + offH[1]);
                const char* w2 = w2buf + buf * W2_BYTES;
#pragma unroll
                for (int j = 0; j < KA; ++j) {
                    const bfv8 b = *reinterpret_cast<const bfv8*>(w2 + offW2 + 2048 * j);
                    acc2[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, ah0, acc2[0][j], 0, 0, 0);
                    acc2[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, ah1, acc2[1][j], 0, 0, 0);
                    if (KA > 6 && j == KA / 2 - 1) __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        if (ka >= KS) chunks(std::integral_constant<int, KS>{});
        else if (KS >= 2 && ka == KS - 1) chunks(std::integral_constant<int, (KS >= 2 ? KS - 1 : 1)>{});
        else if (KS >= 3 && ka == KS - 2) chunks(std::integral_constant<int, (KS >= 3 ? KS - 2 : 1)>{});
        else if (KS >= 4 && ka == KS - 3) chunks(std::integral_constant<int, (KS >= 4 ? KS - 3 : 1)>{});
        else chunks(std::integral_constant<int, KS>{});           // narrower prefixes: full width (zeros beyond the prefix)
        // ---- epilogue: lane holds out[row 16 i + li][32 j + 16 wn + 4 g .. + 4] ----
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (!rok[i]) continue;
            const float* rr = p.resid + orow[i] * p.ldo;
            float* orw = p.out + orow[i] * p.ldo;
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                const int n = 32 * j + 16 * wn + 4 * g;
                if (n < p.C) {
                    const float4 b2 = p.b2 ? *reinterpret_cast<const float4*>(p.b2 + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4 rs = *reinterpret_cast<const float4*>(rr + n);
                    float4 o;
                    o.x = rs.x + ((n + 0 < kout[i]) ? (acc2[i][j][0] + b2.x) * scl[i] : 0.f);
                    o.y = rs.y + ((n + 1 < kout[i]) ? (acc2[i][j][1] + b2.y) * scl[i] : 0.f);
                    o.z = rs.z + ((n + 2 < kout[i]) ? (acc2[i][j][2] + b2.z) * scl[i] : 0.f);
                    o.w = rs.w + ((n + 3 < kout[i]) ? (acc2[i][j][3] + b2.w) * scl[i] : 0.f);
                    *reinterpret_cast<float4*>(orw + n) = o;
                }
            }
        }
    }
}

}  // namespace vr_mlp

extern "C" int vr_mlp_fwd_supported(int32_t C, int32_t F) { return (C > 0 && C <= 320 && C % 8 == 0 && F > 0 && F <= 2048 && F % 8 == 0) ? 1 : 0; }

extern "C" int vr_mlp_fwd(const vr_mlp_args* a, vr_stream_t stream) {
    using namespace vr_mlp;
    if (!a || !a->y || !a->w1 || !a->w2 || !a->resid || !a->out || a->M <= 0) return VR_EINVAL;
    if (!vr_mlp_fwd_supported(a->C, a->F)) return VR_EUNSUPPORTED;
    if (a->ldy % 8 || a->ldw1 % 8 || a->ldw2 % 8 || a->ldo % 4 || a->ldy < a->C || a->ldw1 < a->C || a->ldw2 < a->F || a->ldo < a->C)
        return VR_EALIGN;
    if (((uintptr_t)a->y & 15) || ((uintptr_t)a->w1 & 15) || ((uintptr_t)a->w2 & 15) || ((uintptr_t)a->resid & 15) ||
        ((uintptr_t)a->out & 15) || (a->b1 && ((uintptr_t)a->b1 & 15)) || (a->b2 && ((uintptr_t)a->b2 & 15)))
        return VR_EALIGN;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return VR_EINVAL;
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int tiles = (a->M + BM - 1) / BM;
    const dim3 grid((unsigned)(tiles < n_cu ? tiles : n_cu)), block(NTHR);
    const int ks = (a->C + 63) / 64 * 2;
    hipStream_t st = (hipStream_t)stream;
    switch (ks) {
        case 2: hipLaunchKernelGGL(mlp_fwd_kernel<2>, grid, block, 0, st, *a); break;
        case 4: hipLaunchKernelGGL(mlp_fwd_kernel<4>, grid, block, 0, st, *a); break;
        case 6: hipLaunchKernelGGL(mlp_fwd_kernel<6>, grid, block, 0, st, *a); break;
        case 8: hipLaunchKernelGGL(mlp_fwd_kernel<8>, grid, block, 0, st, *a); break;
        case 10: hipLaunchKernelGGL(mlp_fwd_kernel<10>, grid, block, 0, st, *a); break;
        default: return VR_EUNSUPPORTED;
    }
    VR_CHECK_LAUNCH();
    return VR_OK;
}
