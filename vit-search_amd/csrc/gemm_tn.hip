// bf16 weight-gradient GEMM:  C[M,N] += A[T,M]^T * B[T,N]   (fp32 atomics, split over the tokens T)
//                         or  C[M,N]  = A[T,M]^T * B[T,N]   (atomic == 2: one workgroup per output tile, plain stores)
//
// Weight gradient of every nn.Linear on the ViT-Res hot path (autograd of F.linear, reference
// nets/supernet_blocks.py:41,110): A = dY [tokens, out], B = X [tokens, in], both as the forward/backward kernels left
// them (channel-contiguous, token-major).  vr_gemm (gemm.hip) dispatches here for a_trans && b_trans && atomic;
// the meaning of every vr_gemm_args field is unchanged.
//
// Same skeleton as gemm_nt.hip (128x128 tile, 4 waves of 64x64 = 4x4 v_mfma_f32_16x16x32_bf16, one 32 KB slice of
// 64 tokens moved by LDS-DMA, single-buffered, 4 workgroups per CU), but the MFMA fragments need 8 consecutive TOKENS
// of one channel while LDS holds [token][channel]: they are read with the gfx950 transposing LDS read
// ds_read_b64_tr_b16 -- a 16-lane group reads a [4 tokens][16 channels] block, lane i supplying the address of token
// i/4, channels 4 (i%4).., and receives channel i of the four tokens (semantics pinned by tools/probes/tr_probe.hip).
// Two such reads give the 8-token fragment of v_mfma_f32_16x16x32_bf16.
//
// LDS image: a piece of LDS-DMA (1 KB) = 4 tokens x 128 channels; token row = 16 slots of 16 B; slot p of token t holds
// channel chunk p ^ ((t & 3) << 1) ^ (((t >> 3) & 1) << 3) (applied on the SOURCE address, like gemm_nt.hip): the transposing
// read is serviced in two 32-lane halves, each covering tokens {0..3, 8..11} (+ const) x 2 chunks = 16 different slots of
// the 256-B bank row.
//
// The Linear's bias gradient (column sums of dY) is one more MFMA per A fragment against an all-ones B fragment, done
// by the workgroups of the first column tile only.
#include <cstdlib>

#include "common.h"
#include "../../include/vitres_hip.h"
#include "gemm_shared.h"

namespace vr_gemm_tn {
using namespace vr_gemm_shared;

typedef __bf16 bfv8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) s4 lds_s4;
typedef __attribute__((address_space(1))) const void glb_void;

#ifndef TN_CAP10
#define TN_CAP10 20
#endif
constexpr int BT = 64, NTHR = 256;                       // BT: tokens per slice

__device__ const uint4 zero_chunk[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};

// Geometry of a TW-channel wide operand slice in LDS (TW = 128 or 64): token row = TW*2 bytes = SLOTS 16-byte slots; a piece of
// LDS-DMA (1 KB) = TPP token rows; slot p of token t holds channel chunk p ^ swz(t), chosen so that the 8 tokens x 2 chunks a
// 32-lane half of a transposing read touches ({0..3, 8..11} + const) fall on 16 different slots of the 256-B bank row:
//   TW = 128 (one token per bank row)  : swz = ((t & 3) << 1) ^ (((t >> 3) & 1) << 3)
//   TW = 64  (two tokens per bank row, t & 1 picks the half) : swz = (((t >> 1) & 1) << 1) | (((t >> 3) & 1) << 2)
template <int TW> struct Geo {
    static constexpr int ROWB = TW * 2, SLOTS = TW / 8, TPP = 1024 / ROWB, PIECES = BT / TPP, PPW = PIECES / 4;
    static constexpr int TILE_BYTES = BT * ROWB;
    static constexpr int F = TW / 32;                    // 16-channel fragments per wave and operand (waves are 2 x 2)
    __device__ static __forceinline__ int swz(int t) {
        if constexpr (TW == 128) return ((t & 3) << 1) ^ (((t >> 3) & 1) << 3);
        else return (((t >> 1) & 1) << 1) | (((t >> 3) & 1) << 2);
    }
};

template <int ROWB> __device__ __forceinline__ bfv8 tr_frag(const char* p) {
    // tokens +0..3 and +4..7 (4 token rows further) of this lane's channel
    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(p));
    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(p + 4 * ROWB));
    const s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bfv8, v);
}

// MAPPED: token rows of A / B go through vr_rowmap (cls / patch rows of the embedding, spatial-reduction and head GEMMs): the
// row address is recomputed per slice instead of advancing by a constant stride.
// TW: tile = TW x TW outputs.  64 for small weights: the atomic volume (split x |W|) is the same as with 128 x 128 tiles, but
// four times as many workgroups share it -- a 768 x 256 weight is 12 tiles of 128^2, i.e. ~200 workgroups at the coarse token
// split the atomics call for, each running its 32 slices alone on a CU at the full ~1.5 us slice latency.
// STAGES = 3 (opt-in, VITRES_TN_STAGES=3): ring of three slice buffers, two slices in flight (counted vmcnt + raw barrier, as
// gemm_nt.hip).  Faster alone; inside the training step its 96 KB of LDS displace the data-gradient workgroups it runs beside
// (measured -5 %), so the default stays the single buffer.
template <bool BIAS, bool MAPPED, int TW, int STAGES>
__device__ __forceinline__ void tn_body(const vr_gemm_args& p, const int bid) {
    typedef Geo<TW> G;
    constexpr int ROWB = G::ROWB, SLOTS = G::SLOTS, TPP = G::TPP, PPW = G::PPW, TILE_BYTES = G::TILE_BYTES, F = G::F;
    constexpr int STAGE_BYTES = 2 * TILE_BYTES;
    __shared__ __attribute__((aligned(1024))) char smem[STAGES * STAGE_BYTES];   // ring of [A slice][B slice]
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + TW - 1) / TW, tiles_m = (p.M + TW - 1) / TW;
    const int total = tiles_n * tiles_m * p.split_k;
    int tile = bid;
    if (tile >= total) return;      // (grouped launches pad every problem to a multiple of 8 workgroups)
    if (total >= 16) {   // XCD-aware order (see gemm_nt.hip)
        const int xq = total >> 3, xr = total & 7, x = tile & 7;
        tile = x * xq + min(x, xr) + (tile >> 3);
    }
    const int tn = tile % tiles_n, rest = tile / tiles_n;
    const int tm = rest % tiles_m, zq = rest / tiles_m;
    const int m0 = tm * TW, n0 = tn * TW;
    // token range of this split: the splits of a multi-architecture batch never straddle two groups (the launchers make split_k a
    // multiple of m_groups; gemm_shared.h group_pure) and consecutive positions walk the groups round-robin -- every group on
    // every XCD; otherwise equal ranges over all tokens
    int kbeg, kend;
    if (group_pure(p.K, p.m_groups) && p.split_k % p.m_groups == 0) {
        const int G = p.m_groups, kg = p.K / G, spg = p.split_k / G, g = zq % G, zi = zq / G;
        const int kper = ((kg + spg - 1) / spg + BT - 1) / BT * BT;
        kbeg = g * kg + zi * kper;
        kend = min(g * kg + kg, kbeg + kper);
    } else {
        const int z = interleave_groups(zq, p.split_k, p.m_groups);
        const int kper = ((p.K + p.split_k - 1) / p.split_k + BT - 1) / BT * BT;
        kbeg = z * kper;
        kend = min(p.K, kbeg + kper);
    }
    int ntiles = kbeg < kend ? (kend - kbeg + BT - 1) / BT : 0;
    // masked-work skipping: keep_k bounds the kept output rows (dY channels), keep_n the kept columns (X channels) of
    // the samples this token range touches; a tile without kept rows or columns adds exactly zero
    // (kmax / nmax also bound what the epilogue adds: rows / columns beyond the largest kept prefix of this token range hold exact
    // zeros -- round 5: fp32 atomics run at ~1.1 TB/s of payload, 40 - 55 % of a group launch's time, profiles/r05_wgrad_atomics.txt)
    int kmax = 1 << 30, nmax = 1 << 30;
    if ((p.keep_k || p.keep_n) && ntiles > 0) {
        int s_lo = 0, s_hi = 0;
        if (p.rows_in > 0) { s_lo = kbeg / p.rows_in; s_hi = (kend - 1) / p.rows_in; }
        kmax = max_keep(p.keep_k, s_lo, s_hi, 1 << 30);
        nmax = max_keep(p.keep_n, s_lo, s_hi, 1 << 30);
        if (!range_has_kept(n0, TW, p.n_period, nmax) || !range_has_kept(m0, TW, p.k_period, kmax)) ntiles = 0;
    }
    // atomic == 2 (store form, split_k == 1): the tile's one workgroup OVERWRITES the gradient -- no zero-filled destination, no
    // read-modify-write; a fully masked tile must then write its zeros instead of leaving
    const bool store = p.atomic == 2;
    if (ntiles == 0 && !store) return;
    const bool nochunk = (p.sched & 0x8000) != 0;                                // (sched 0x8000 / VITRES_DBG_TN=4: A/B aid)
    const bool skipm = nochunk || !p.keep_k || (p.k_period > 0 && (p.k_period & 7)), skipn = nochunk || !p.keep_n || (p.n_period > 0 && (p.n_period & 7));

    // ---- LDS-DMA source addressing: piece h of this wave = slice tokens (wave*PPW + h)*TPP .. ; lane -> (token, slot) ----
    const char* gA[PPW];
    const char* gB[PPW];
    int tok[PPW];
    const long long strideA = (long long)BT * p.lda * 2, strideB = (long long)BT * p.ldb * 2;
    const char* zero = reinterpret_cast<const char*>(zero_chunk);
#pragma unroll
    for (int h = 0; h < PPW; ++h) {
        const int tk = (wave * PPW + h) * TPP + lane / SLOTS;
        const int c = (lane % SLOTS) ^ G::swz(tk);
        // channel chunks past the matrix edge read the row's own padding (lda, ldb >= roundup(M / N, 8)) or, past that,
        // the zero page; their products only reach outputs that are not stored
        // ... and so do the chunks beyond the largest kept prefix of this token range (round 5): they hold exact zeros by the
        // contract of keep_k / keep_n, and a tile that straddles the prefix would stream them from HBM (a width of 160 kept of 256
        // read as 256: reads were 1.8x the kept-width bytes).  Periods that are not multiples of 8 keep every chunk.
        const bool ak = skipm || kept_col(m0 + c * 8, p.k_period, kmax), bk = skipn || kept_col(n0 + c * 8, p.n_period, nmax);
        const bool aok = m0 + c * 8 + 8 <= p.lda && ak, bok = n0 + c * 8 + 8 <= p.ldb && bk;
        tok[h] = tk;
        const long long ra = MAPPED ? 0 : (long long)(kbeg + tk) * p.lda, rb = MAPPED ? 0 : (long long)(kbeg + tk) * p.ldb;
        gA[h] = aok ? reinterpret_cast<const char*>(p.A) + (ra + m0 + c * 8) * 2 : nullptr;
        gB[h] = bok ? reinterpret_cast<const char*>(p.B) + (rb + n0 + c * 8) * 2 : nullptr;
    }
    const RowMap amap = {p.a_map.rpi, p.a_map.rps, p.a_map.off};
    const RowMap bmap = {p.b_map.rpi, p.b_map.rps, p.b_map.off};

    // ---- fragment addresses: lane (g = lane >> 4, li = lane & 15) -> token 8 g + (li >> 2) (+ 32 s, + 4 for the second
    //      read), channel (TW/2) w + 16 i + 4 (li & 3); slot = chunk ^ swz(token) (the same for both reads and all s) ----
    const int li = lane & 15, g = lane >> 4;
    const int xr2 = G::swz(8 * g + (li >> 2));
    const int rowoff = (8 * g + (li >> 2)) * ROWB + (li & 1) * 8;
    int offA[F], offB[F];
#pragma unroll
    for (int i = 0; i < F; ++i) {
        const int ca = ((TW / 16) * wm + 2 * i + ((li & 3) >> 1)) ^ xr2;
        const int cb = ((TW / 16) * wn + 2 * i + ((li & 3) >> 1)) ^ xr2;
        offA[i] = rowoff + ca * 16;
        offB[i] = TILE_BYTES + rowoff + cb * 16;
    }

    f32x4 acc[F][F];
    f32x4 accb[F];
#pragma unroll
    for (int i = 0; i < F; ++i) {
        accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < F; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const bool want_bg = BIAS && p.bias_grad != nullptr && tn == 0 && wn == 0;
    const s8 ones_bits = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};   // bf16 1.0
    const bfv8 ones = __builtin_bit_cast(bfv8, ones_bits);

    auto issue = [&](int kt, int buf) {
        const int k0 = kbeg + kt * BT;
        char* dst = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int h = 0; h < PPW; ++h) {
            const bool in = k0 + tok[h] < kend;
            const char* sa;
            const char* sb;
            if constexpr (MAPPED) {
                const int tg = in ? k0 + tok[h] : kbeg;
                sa = (in && gA[h]) ? gA[h] + map_row(amap, tg) * (long long)p.lda * 2 : zero;
                sb = (in && gB[h]) ? gB[h] + map_row(bmap, tg) * (long long)p.ldb * 2 : zero;
            } else {
                sa = (in && gA[h]) ? gA[h] + kt * strideA : zero;
                sb = (in && gB[h]) ? gB[h] + kt * strideB : zero;
            }
            __builtin_amdgcn_global_load_lds((glb_void*)sa, (lds_void*)(dst + (wave * PPW + h) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)sb, (lds_void*)(dst + TILE_BYTES + (wave * PPW + h) * 1024), 16, 0, 0);
        }
    };
    auto compute = [&](int buf) {
        const char* sb_ = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bfv8 a[F];
#pragma unroll
            for (int i = 0; i < F; ++i) a[i] = tr_frag<ROWB>(sb_ + offA[i] + s * 32 * ROWB);
            if (want_bg) {
#pragma unroll
                for (int i = 0; i < F; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], ones, accb[i], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < F; ++j) {
                const bfv8 b = tr_frag<ROWB>(sb_ + offB[j] + s * 32 * ROWB);
#pragma unroll
                for (int i = 0; i < F; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b, acc[i][j], 0, 0, 0);
            }
        }
    };
    const int dbg = (p.sched >> 13) & 3;         // measurement aid (VITRES_DBG_TN; wrong results): 1 no epilogue, 2 no K loop
    if (dbg == 2) ntiles = 0;
    if constexpr (STAGES == 1) {
        for (int kt = 0; kt < ntiles; ++kt) {
            issue(kt, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute(0);
            __syncthreads();
        }
    } else {
        if (ntiles > 0) issue(0, 0);
        if (ntiles > 1) issue(1, 1);
        int buf = 0;
        for (int kt = 0; kt < ntiles; ++kt) {
            if (kt + 1 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");   // slice kt landed, kt+1 may fly
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < ntiles) issue(kt + 2, buf == 0 ? 2 : buf - 1);
            compute(buf);
            buf = buf == 2 ? 0 : buf + 1;
        }
    }

    // ---- epilogue: lane holds C[m = 16 i + 4 (lane >> 4) + r][n = 16 j + (lane & 15)]: 16 consecutive columns (64 B)
    //      of 4 rows per atomic instruction ----
    if (dbg == 1) {
        if (acc[0][0][0] == 12345.678f) reinterpret_cast<float*>(p.C)[0] = 1.f;      // (keeps the loop alive)
        return;
    }
    float* C = reinterpret_cast<float*>(p.C);
    const RowMap cm = {p.c_map.rpi, p.c_map.rps, p.c_map.off};
    // masked rows / columns are exact zeros: the atomic form leaves them alone (the store form must write them)
    const bool skipz = !store && !(p.sched & 0x8000);                       // (sched bit 0x8000: add the zeros too -- A/B aid)
    bool ncol[F];
#pragma unroll
    for (int j = 0; j < F; ++j) {
        const int n = n0 + wn * (TW / 2) + 16 * j + li;
        ncol[j] = n < p.N && (!skipz || kept_col(n, p.n_period, nmax));
    }
    // (rotating the row order by the split index -- the splits of a tile finish together and add to the same rows -- changes
    // nothing: the atomics are bound by their volume, ~1.1 TB/s of payload, not by same-line contention; round 5)
#pragma unroll
    for (int i = 0; i < F; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm * (TW / 2) + 16 * i + 4 * g + r;
            if (m >= p.M) continue;
            if (skipz && !kept_col(m, p.k_period, kmax)) continue;
            float* crow = C + map_row(cm, m) * (long long)p.ldc;
#pragma unroll
            for (int j = 0; j < F; ++j) {
                const int n = n0 + wn * (TW / 2) + 16 * j + li;
                if (ncol[j]) {
                    if (store) crow[n] = acc[i][j][r];
                    else atomicAdd(crow + n, acc[i][j][r]);
                }
            }
            if (want_bg && li == 0) {
                if (store) p.bias_grad[m] = accb[i][r];
                else atomicAdd(p.bias_grad + m, accb[i][r]);
            }
        }
}

template <bool BIAS, bool MAPPED, int TW, int STAGES>
__global__ __launch_bounds__(NTHR, 4) void tn_kernel(const vr_gemm_args p) {
    tn_body<BIAS, MAPPED, TW, STAGES>(p, blockIdx.x);
}

// Several weight gradients in ONE launch (vr_gemm_group): the four Linears of a transformer block have 12 + 4 + 12 + 12 tiles
// at 256 x 768 -- alone, at the coarse token split their fp32 atomics call for, each runs ~200 workgroups, one per CU, every
// 64-token slice at the full load latency.  Together they put ~4 workgroups on every CU, which overlap one another's slices
// exactly like the forward kernel's workgroups do, in a quarter of the launches.  Problem k owns workgroups
// [first[k], first[k + 1]) (multiples of 8, so that the XCD-aware tile order of tn_body still sees its own XCD).
constexpr int MAXG = 4;
struct TnGroup {
    vr_gemm_args a[MAXG];
    int first[MAXG + 1];
    int count;
};
// Resident form (round 4): the launch is capped at `cap` workgroups per CU and a workgroup walks work items blockIdx.x,
// blockIdx.x + gridDim.x, ... (gridDim.x a multiple of 8: an item stays on the XCD the tile order gave it).  An uncapped group is
// 680 - 1280 workgroups that live 25 - 50 us each: as the short-lived workgroups of the main chain's kernel retire, the group's
// pending ones take their slots -- up to all four per CU -- and the kernel the group runs beside is left a third of the chip
// (stage-1 data gradients 28.6 -> 51.8 us, LayerNorm backward 16.5 -> 31.7 us: profiles/r04_wgrad_contention.txt).
template <bool MAPPED, int TW, int STAGES = 1>
__global__ __launch_bounds__(NTHR, 4) void tn_group_kernel(const TnGroup g) {
    for (int bid = (int)blockIdx.x; bid < g.first[MAXG]; bid += (int)gridDim.x) {
        int k = 0;
#pragma unroll
        for (int i = 1; i < MAXG; ++i)
            if (i < g.count && bid >= g.first[i]) k = i;
        tn_body<true, MAPPED, TW, STAGES>(g.a[k], bid - g.first[k]);
    }
}

// ---- round 6: the group launch's kernel for the atomic form ---------------------------------------------------------------
// tn_body above is one 4-wave workgroup per item, single slice buffer (issue, wait, multiply: nothing overlaps inside it), two of
// them resident per CU -- and every item pays |tile| x 4 B of fp32 atomics whatever its share of the tokens: 16 token splits x
// 2.6 MB x 2 (read-modify-write) per stage-1 group at ~1.1 TB/s of payload, 40 of 107 us (profiles/r05_wgrad_atomics.txt), 2.16x the
// algorithmic bytes in measured traffic.  tn8_body is ONE 8-wave workgroup per CU on the same 64 KB of LDS: 128 x 128 tile, waves
// 2 (rows) x 4 (columns) of 64 x 32, the slices double-buffered INSIDE the workgroup (LDS-DMA through a buffer descriptor: one
// 32-bit offset per piece, tokens past the split's end and masked channel chunks read as zeros through its range check; counted
// vmcnt, one barrier per slice) -- the second wave per SIMD overlaps what the second workgroup used to.  An item then covers twice
// the tokens at the same residency: half the token splits, half the atomic payload.
constexpr int NTHR8 = 512;
typedef int v4i_tn __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4i_tn tn_rsrc(const void* ptr, unsigned num_records) {
    const unsigned long long a = (unsigned long long)(uintptr_t)ptr;
    v4i_tn r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)num_records);
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ void tn_dma16(unsigned lds, unsigned voff, v4i_tn rsrc) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds), "v"(voff), "s"(rsrc) : "memory", "m0");
}

// NW = 8: as above.  NW = 4: the 4-wave workgroup (waves 2 x 2 of 64 x 64, one slice buffer, two resident per CU) with the same lean
// instruction stream -- what tn_body does in ~8 x (64-bit address + two selects) per piece and slice is one v_add here.
template <int NW>
__device__ __forceinline__ void tn8_body(const vr_gemm_args& p, const int bid, char* smem) {
    typedef Geo<128> G;
    constexpr int TW = 128, ROWB = G::ROWB, SLOTS = G::SLOTS, TPP = G::TPP, TILE_BYTES = G::TILE_BYTES;
    constexpr int STAGE_BYTES = 2 * TILE_BYTES, FM = 4, FN = NW == 8 ? 2 : 4, PPW8 = G::PIECES / NW;       // 16 pieces per operand slice
    constexpr int WCOLS = FN * 16, WNB = NW == 8 ? 2 : 1;                                     // columns per wave, log2(waves along N)
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> WNB, wn = wave & ((1 << WNB) - 1);
    const int tiles_n = (p.N + TW - 1) / TW, tiles_m = (p.M + TW - 1) / TW;
    const int total = tiles_n * tiles_m * p.split_k;
    int tile = bid;
    if (tile >= total) return;
    if (total >= 16) {
        const int xq = total >> 3, xr = total & 7, x = tile & 7;
        tile = x * xq + min(x, xr) + (tile >> 3);
    }
    const int tn = tile % tiles_n, rest = tile / tiles_n;
    const int tm = rest % tiles_m, zq = rest / tiles_m;
    const int m0 = tm * TW, n0 = tn * TW;
    int kbeg, kend;
    if (group_pure(p.K, p.m_groups) && p.split_k % p.m_groups == 0) {
        const int Gr = p.m_groups, kg = p.K / Gr, spg = p.split_k / Gr, g = zq % Gr, zi = zq / Gr;
        const int kper = ((kg + spg - 1) / spg + BT - 1) / BT * BT;
        kbeg = g * kg + zi * kper;
        kend = min(g * kg + kg, kbeg + kper);
    } else {
        const int z = interleave_groups(zq, p.split_k, p.m_groups);
        const int kper = ((p.K + p.split_k - 1) / p.split_k + BT - 1) / BT * BT;
        kbeg = z * kper;
        kend = min(p.K, kbeg + kper);
    }
    int ntiles = kbeg < kend ? (kend - kbeg + BT - 1) / BT : 0;
    int kmax = 1 << 30, nmax = 1 << 30;
    if ((p.keep_k || p.keep_n) && ntiles > 0) {
        int s_lo = 0, s_hi = 0;
        if (p.rows_in > 0) { s_lo = kbeg / p.rows_in; s_hi = (kend - 1) / p.rows_in; }
        kmax = max_keep(p.keep_k, s_lo, s_hi, 1 << 30);
        nmax = max_keep(p.keep_n, s_lo, s_hi, 1 << 30);
        if (!range_has_kept(n0, TW, p.n_period, nmax) || !range_has_kept(m0, TW, p.k_period, kmax)) ntiles = 0;
    }
    if (ntiles == 0) return;
    const bool skipm = !p.keep_k || (p.k_period > 0 && (p.k_period & 7)), skipn = !p.keep_n || (p.n_period > 0 && (p.n_period & 7));

    // ---- LDS-DMA: piece h of this wave = slice tokens (2 wave + h) * 4 .. + 4; lane -> (token, 16-byte slot).  Offsets are relative
    //      to the split's first token row; the descriptors end with its last one (rows beyond read as zeros), a chunk beyond the
    //      row's width or the kept prefix of this token range gets an offset no slice brings back into range ----
    const v4i_tn rsA = tn_rsrc(reinterpret_cast<const bf16_t*>(p.A) + (long long)kbeg * p.lda, (unsigned)((long long)(kend - kbeg) * p.lda * 2));
    const v4i_tn rsB = tn_rsrc(reinterpret_cast<const bf16_t*>(p.B) + (long long)kbeg * p.ldb, (unsigned)((long long)(kend - kbeg) * p.ldb * 2));
    unsigned voA[PPW8], voB[PPW8];
#pragma unroll
    for (int h = 0; h < PPW8; ++h) {
        const int tk = (wave * PPW8 + h) * TPP + lane / SLOTS;
        const int c = (lane % SLOTS) ^ G::swz(tk);
        const bool ak = skipm || kept_col(m0 + c * 8, p.k_period, kmax), bk = skipn || kept_col(n0 + c * 8, p.n_period, nmax);
        const bool aok = m0 + c * 8 + 8 <= p.lda && ak, bok = n0 + c * 8 + 8 <= p.ldb && bk;
        voA[h] = aok ? (unsigned)((tk * p.lda + m0 + c * 8) * 2) : 0x80000000u;
        voB[h] = bok ? (unsigned)((tk * p.ldb + n0 + c * 8) * 2) : 0x80000000u;
    }
    const unsigned stepA = (unsigned)(BT * p.lda * 2), stepB = (unsigned)(BT * p.ldb * 2);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    const int li = lane & 15, g = lane >> 4;
    const int xr2 = G::swz(8 * g + (li >> 2));
    const int rowoff = (8 * g + (li >> 2)) * ROWB + (li & 1) * 8;
    int offA[FM], offB[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) offA[i] = rowoff + ((8 * wm + 2 * i + ((li & 3) >> 1)) ^ xr2) * 16;
#pragma unroll
    for (int j = 0; j < FN; ++j) offB[j] = TILE_BYTES + rowoff + (((WCOLS / 8) * wn + 2 * j + ((li & 3) >> 1)) ^ xr2) * 16;

    f32x4 acc[FM][FN];
    f32x4 accb[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const bool want_bg = p.bias_grad != nullptr && tn == 0 && wn == 0;
    const s8 ones_bits = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};   // bf16 1.0
    const bfv8 ones = __builtin_bit_cast(bfv8, ones_bits);

    auto issue = [&](int kt, int buf) {
        const unsigned dst = lds0 + buf * STAGE_BYTES + wave * (PPW8 * 1024);
#pragma unroll
        for (int h = 0; h < PPW8; ++h) {
            tn_dma16(dst + h * 1024, voA[h] + kt * stepA, rsA);
            tn_dma16(dst + TILE_BYTES + h * 1024, voB[h] + kt * stepB, rsB);
        }
    };
    auto compute = [&](int buf) {
        const char* sb_ = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bfv8 a[FM];
#pragma unroll
            for (int i = 0; i < FM; ++i) a[i] = tr_frag<ROWB>(sb_ + offA[i] + s * 32 * ROWB);
            if (want_bg) {
#pragma unroll
                for (int i = 0; i < FM; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], ones, accb[i], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const bfv8 b = tr_frag<ROWB>(sb_ + offB[j] + s * 32 * ROWB);
#pragma unroll
                for (int i = 0; i < FM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b, acc[i][j], 0, 0, 0);
            }
        }
    };
    __syncthreads();                     // (the previous item's last slice has been read by every wave)
    if constexpr (NW == 8) {
        issue(0, 0);
        for (int kt = 0; kt < ntiles; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // slice kt (the only one in flight) has landed
            __builtin_amdgcn_s_barrier();                             // ... for every wave; the other buffer has been read
            if (kt + 1 < ntiles) issue(kt + 1, (kt + 1) & 1);
            compute(kt & 1);
        }
    } else {
        for (int kt = 0; kt < ntiles; ++kt) {
            issue(kt, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute(0);
            __syncthreads();
        }
    }

    float* C = reinterpret_cast<float*>(p.C);
    bool ncol[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = n0 + wn * WCOLS + 16 * j + li;
        ncol[j] = n < p.N && kept_col(n, p.n_period, nmax);
    }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm * 64 + 16 * i + 4 * g + r;
            if (m >= p.M) continue;
            if (!kept_col(m, p.k_period, kmax)) continue;
            float* crow = C + (long long)m * p.ldc;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int n = n0 + wn * WCOLS + 16 * j + li;
                if (ncol[j]) atomicAdd(crow + n, acc[i][j][r]);
            }
            if (want_bg && li == 0) atomicAdd(p.bias_grad + m, accb[i][r]);
        }
}

template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 4) void tn8_group_kernel(const TnGroup g) {
    __shared__ __attribute__((aligned(1024))) char smem[(NW == 8 ? 2 : 1) * 2 * Geo<128>::TILE_BYTES];
    for (int bid = (int)blockIdx.x; bid < g.first[MAXG]; bid += (int)gridDim.x) {
        int k = 0;
#pragma unroll
        for (int i = 1; i < MAXG; ++i)
            if (i < g.count && bid >= g.first[i]) k = i;
        tn8_body<NW>(g.a[k], bid - g.first[k], smem);
    }
}

template <int TW, int STAGES> void launch_tw(const vr_gemm_args& a, hipStream_t stream, long long total) {
    const bool mapped = a.a_map.rpi != 0 || a.b_map.rpi != 0;
    if (mapped) {
        if (a.bias_grad) hipLaunchKernelGGL((tn_kernel<true, true, TW, STAGES>), dim3((unsigned)total), dim3(NTHR), 0, stream, a);
        else hipLaunchKernelGGL((tn_kernel<false, true, TW, STAGES>), dim3((unsigned)total), dim3(NTHR), 0, stream, a);
    } else {
        if (a.bias_grad) hipLaunchKernelGGL((tn_kernel<true, false, TW, STAGES>), dim3((unsigned)total), dim3(NTHR), 0, stream, a);
        else hipLaunchKernelGGL((tn_kernel<false, false, TW, STAGES>), dim3((unsigned)total), dim3(NTHR), 0, stream, a);
    }
}

}  // namespace vr_gemm_tn

static bool tn_covers(const vr_gemm_args& a0) {
    if (a0.in_dtype != VR_BF16 || !a0.a_trans || !a0.b_trans || !a0.atomic || a0.out_dtype != VR_F32) return false;
    if (a0.lda % 8 || a0.ldb % 8 || ((uintptr_t)a0.A & 15) || ((uintptr_t)a0.B & 15)) return false;
    if (a0.lda < (a0.M + 7) / 8 * 8 || a0.ldb < (a0.N + 7) / 8 * 8) return false;
    return true;
}

// token split of a multi-architecture batch: the multiple of the group count nearest to the wanted split (gemm_shared.h group_pure)
static bool needs_pure(const vr_gemm_args& a);
static int group_split(long long split, const vr_gemm_args& a) {
    if (split < 1) split = 1;
    if (!vr_gemm_shared::group_pure(a.K, a.m_groups)) return (int)split;
    if (split < a.m_groups && !needs_pure(a)) return (int)split;        // (few tokens: one split stays one split -- and one sum order)
    const long long q = (split + a.m_groups / 2) / a.m_groups;
    return (int)((q < 1 ? 1 : q) * a.m_groups);
}
// sched bit 0x80000: the operands may hold unwritten (fully masked) tiles -- readable only split by split within one group
static bool needs_pure(const vr_gemm_args& a) { return (a.sched & 0x80000) && a.m_groups > 1 && (a.keep_k || a.keep_n); }

// vr_gemm_group: `count` validated weight-gradient problems as one launch.  Returns false when any of them is not a plain
// (un-mapped, automatic split) tn_kernel form -- the caller then issues them one by one.
bool vr_gemm_tn_group_launch(const vr_gemm_args* args, int count, hipStream_t stream, int n_cu) {
    using namespace vr_gemm_tn;
    if (count < 2 || count > MAXG) return false;
    for (int i = 0; i < count; ++i) {
        const vr_gemm_args& a = args[i];
        if (!tn_covers(a) || a.a_map.rpi != 0 || a.b_map.rpi != 0 || a.split_k > 0) return false;
        if (needs_pure(a) && (a.atomic == 2 || !vr_gemm_shared::group_pure(a.K, a.m_groups))) return false;
    }
    bool all_store = true, any_store = false;
    for (int i = 0; i < count; ++i) {
        all_store = all_store && args[i].atomic == 2;
        any_store = any_store || args[i].atomic == 2;
    }
    TnGroup g;
    g.count = count;
    int next = 0;
    const long long slices = (args[0].K + BT - 1) / BT;
    // the lean kernels address a token split through 32-bit offsets of a buffer descriptor: every operand below 2 GB
    bool lean_ok = true;
    for (int i = 0; i < count; ++i)
        lean_ok = lean_ok && (long long)args[i].K * args[i].lda * 2 < 0x7ff00000LL && (long long)args[i].K * args[i].ldb * 2 < 0x7ff00000LL;
    bool same_k = true;
    for (int i = 1; i < count; ++i) same_k = same_k && args[i].K == args[0].K && args[i].m_groups == args[0].m_groups;
    if (!any_store && same_k && lean_ok && (args[0].sched & 0x10000)) {
        // ---- atomic form, OPT-IN (sched 0x10000 on the first problem): tn8_group_kernel<8>, one 8-wave workgroup per CU ----
        // One token split for the whole group.  Cost of s splits in slice-times of one workgroup: rounds of the chip x (slices of an
        // item + its epilogue: 64 KB of fp32 atomics per item, ~40 slice-times while every CU adds at once -- ~1.1 TB/s of payload
        // chip-wide, profiles/r05_wgrad_atomics.txt); a multi-architecture batch splits by whole groups.
        long long tiles = 0;
        for (int i = 0; i < count; ++i) tiles += (long long)((args[i].M + 127) / 128) * ((args[i].N + 127) / 128);
        const int G = vr_gemm_shared::group_pure(args[0].K, args[0].m_groups) ? args[0].m_groups : 1;
        bool pure = false;
        for (int i = 0; i < count; ++i) pure = pure || needs_pure(args[i]);
        long long best = -1, best_s = 1;
        for (long long sq = (pure ? G : 1); sq <= 64; sq += (G > 1 ? (sq < G ? G - sq : G) : 1)) {
            const long long per = (slices + sq - 1) / sq;
            if (per < 4 && sq > (pure ? G : 1)) break;
            const long long rounds = (tiles * sq + n_cu - 1) / n_cu;
            const long long cost = rounds * (per + 40);
            if (best < 0 || cost < best) { best = cost; best_s = sq; }
        }
        for (int i = 0; i < count; ++i) {
            g.a[i] = args[i];
            const long long t_i = (long long)((args[i].M + 127) / 128) * ((args[i].N + 127) / 128);
            g.a[i].split_k = (int)best_s;
            g.first[i] = next;
            next += (int)((t_i * best_s + 7) / 8 * 8);
        }
        for (int i = count; i <= MAXG; ++i) g.first[i] = next;
        const int lim = (n_cu + 7) / 8 * 8;
        hipLaunchKernelGGL(tn8_group_kernel<8>, dim3((unsigned)(next < lim ? next : lim)), dim3(NTHR8), 0, stream, g);
        return true;
    }
    // ---- store-form members (atomic == 2) run one workgroup per tile over all tokens: no token split to fill the chip with, so a group
    // of them uses 64 x 64 tiles -- four times the workgroups at the same output traffic; the 4-wave kernel, capped at two resident
    // workgroups per CU (round 4), ~2 workgroups per CU in total (measured best inside the step), 8 - 32 slices each
    const int TWv = all_store ? 64 : 128;
    long long work = 0;
    for (int i = 0; i < count; ++i)
        work += (long long)((args[i].M + TWv - 1) / TWv) * ((args[i].N + TWv - 1) / TWv) * ((args[i].K + BT - 1) / BT);
    long long spw = work / (2LL * n_cu);
    spw = spw < 8 ? 8 : (spw > 32 ? 32 : spw);
    static const int knob_dbg = std::getenv("VITRES_DBG_TN") ? std::atoi(std::getenv("VITRES_DBG_TN")) : 0;
    for (int i = 0; i < count; ++i) {
        g.a[i] = args[i];
        g.a[i].sched |= (knob_dbg & 3) << 13;
        const long long sl_i = (args[i].K + BT - 1) / BT;
        const long long tiles = (long long)((args[i].M + TWv - 1) / TWv) * ((args[i].N + TWv - 1) / TWv);
        long long split = args[i].atomic == 2 ? 1 : (sl_i + spw - 1) / spw;       // store form: one workgroup per tile
        g.a[i].split_k = args[i].atomic == 2 ? 1 : group_split(split, args[i]);
        g.first[i] = next;
        next += (int)((tiles * g.a[i].split_k + 7) / 8 * 8);
    }
    for (int i = count; i <= MAXG; ++i) g.first[i] = next;
    int grid = next;
    if (!(args[0].sched & 128)) {            // (sched bit 128: nothing runs beside this group -- the step's last)
        const int lim = (TN_CAP10 * n_cu / 10 + 7) / 8 * 8;
        grid = next < lim ? next : lim;
    }
    if (TWv == 64) hipLaunchKernelGGL((tn_group_kernel<false, 64>), dim3((unsigned)grid), dim3(NTHR), 0, stream, g);
    else if (any_store || knob_dbg || !lean_ok || (args[0].sched & 64)) hipLaunchKernelGGL((tn_group_kernel<false, 128>), dim3((unsigned)grid), dim3(NTHR), 0, stream, g);
    else hipLaunchKernelGGL(tn8_group_kernel<4>, dim3((unsigned)grid), dim3(NTHR), 0, stream, g);     // (sched 64: tn_body's instruction stream, tests)
    return true;
}

// Called by vr_gemm after validation.  Returns false when the form is not covered here (odd leading dimensions): the
// general kernel takes those.
bool vr_gemm_tn_launch(const vr_gemm_args& a0, hipStream_t stream, int n_cu) {
    using namespace vr_gemm_tn;
    if (!tn_covers(a0)) return false;
    if (needs_pure(a0) && (a0.atomic == 2 || !vr_gemm_shared::group_pure(a0.K, a0.m_groups))) return false;
    vr_gemm_args a = a0;
    if (a.atomic == 2) a.split_k = 1;
    const long long slices = (a.K + BT - 1) / BT;
    const long long t128 = (long long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    const long long t64 = (long long)((a.M + 63) / 64) * ((a.N + 63) / 64);
    // Every workgroup pays |tile| x 4 B of fp32 atomics whatever its share of the tokens, so the token split is coarse: 32
    // slices (2048 tokens) per workgroup (measured inside the two-stream training step: +3 % over 16/32 adaptive; 24 / 40 /
    // 48 / 64 slower), never more than 4 workgroups per CU.  64 x 64 tiles give 4x the workgroups at the
    // same atomic volume.  (Round 2: splitting the few-tile problems -- patch-embedding and head weights, 64 - 160 workgroups --
    // finer until the chip holds two workgroups per CU: step 7.77 -> 7.86 ms, their atomics and the contention cost more.)
    long long split = (slices + 31) / 32;
    // (64 x 64 tiles for the atomic form: alone on the chip up to 2x faster for stage-1 weights, but inside the training step the extra
    // workgroups take CUs from the data-gradient chain they run beside: measured -3 %; the store form of a single launch keeps 128 too)
    const bool small = false;
    const long long tiles = small ? t64 : t128;
    if (a.split_k <= 0) {
        const long long by_fill = (4LL * n_cu + tiles - 1) / tiles;
        if (split > by_fill) split = by_fill;
        a.split_k = (int)(split < 1 ? 1 : split);
    }
    if (a.atomic != 2) a.split_k = group_split(a.split_k, a);
    const long long total = tiles * a.split_k;
    (void)t64;
    launch_tw<128, 1>(a, stream, total);
    return true;
}
