// bf16 forward / data-gradient GEMM, lean K loop:  C[M,N] = epilogue(A[M,K] * B[N,K]^T)   (or B = W [K][N] k-major, b_trans)
//
// Same tiles, LDS images and epilogue as gemm_nt.hip (reference nets/supernet_blocks.py:37-52,102-119: every nn.Linear of a
// transformer block and its data gradient) -- what changes is the instruction stream around the MFMAs.  Round-5 counters on the
// stage 2 / 3 GEMMs (profiles/r05_gemm_issue_bound.txt): L2 hit rate 73 %, 18 read requests in flight per CU on average, TA busy
// 12 %, yet every 64-wide slice cost 1.4 - 1.9 us whether one, two or three slices were in flight -- and the waves spent 41 % of
// their cycles ISSUING: gemm_nt.hip's loop is ~280 instructions per slice and wave around 32 MFMAs (64-bit address arithmetic
// and zero-page selects per LDS-DMA piece, the live-slice cursor's branches, accumulator copies around conditional blocks).
// With one or two workgroups per CU (the grids of stages 2 and 3) nothing overlaps that serial stream.  Here:
//
//   * operand slices move with `buffer_load_dwordx4 ... lds`: the per-lane byte offset of a piece (row, swizzled k-chunk) is a
//     32-bit VGPR computed ONCE per tile, a slice advances the scalar offset -- one s_mov m0 + one load per piece, no vector
//     arithmetic in the loop.  Rows past the matrix edge are clamped (their products reach no stored output); column chunks of a
//     k-major weight slice past the row's end get an offset beyond the descriptor's range and read as zeros;
//   * the slices with kept k are one 64-bit mask, made once per tile (prefix masks: a count; periodic masks: one scalar pass),
//     walked with s_ff1: no per-slice modulo, no cursor state;
//   * NBUF = 1: one slice buffer, several workgroups per CU overlap each other (the first stage's grids);
//     NBUF = 2 / 3: slice i + 1 (and i + 2) are in flight while slice i is multiplied -- one s_barrier per slice, counted vmcnt.
//
//   * NBUF up to 6 for grids that give a CU one workgroup (the long-K GEMMs of stages 2 / 3): what paces a lone workgroup is the
//     memory latency divided by the slices it keeps in flight;
//   * SPLIT (round 6): the live slices of a tile are dealt round-robin to `ksplit` workgroups (consecutive workgroups of ONE XCD:
//     they run together and share the tile's rows in that XCD's L2).  Each writes its fp32 accumulators to its slab of
//     vr_gemm_args.ws with plain write-through stores, drains and takes the tile's ticket; the holder of the last ticket acquires,
//     sums the slabs IN SHARE ORDER (its own included: the result does not depend on who arrives last) and runs the epilogue.
//     Nobody waits for anybody, no fp32 atomics touch the output, tickets return to zero.  For grids that leave the chip
//     under-filled with a long serial K loop (stage 2 / 3: 136 - 544 tiles walking 16 - 48 slices).
//
// Covered: FAST epilogue forms (N, ldc, ldu, n_period multiples of 8), K a multiple of 64 and at most 4096, FEAT 0 - 3 (no
// positional embedding), operands below 4 GB.  Everything else stays with gemm_nt.hip / gemm.hip (vr_gemm_ntk_launch returns false).
#include <algorithm>
#include <cstdlib>

#include "gemm_nt_parts.h"
#include "lds_dma.h"

namespace vr_gemm_nt {

constexpr int KTHR = 256;
using vr_dma::dma16;
using vr_dma::make_rsrc;
typedef vr_dma::v4i v4i_k;

// largest keep[s], s in [s_lo, s_hi]: the lanes of a wave load in parallel, the scalar unit folds the (few) values
// A negative value -(k + 2) is the host's mark for a sample that is masked ON ITS OWN (DropPath) in an architecture group of width
// k: every kernel reads it as 0; gmax = the largest width with the marks decoded = what the sample's GROUP keeps.
__device__ __forceinline__ int max_keep_wave(const int* keep, int s_lo, int s_hi, int lane, int& gmax) {
    int mk = 0;
    for (int s0 = s_lo; s0 <= s_hi; s0 += 64) {
        const int s = s0 + lane;
        const int v = s <= s_hi ? keep[s] : 0;
        const int d = v < 0 ? -v - 2 : v;
        const int cnt = min(64, s_hi - s0 + 1);
        for (int i = 0; i < cnt; ++i) {
            mk = max(mk, __builtin_amdgcn_readlane(v, i));
            gmax = max(gmax, __builtin_amdgcn_readlane(d, i));
        }
    }
    return mk;
}

__device__ __forceinline__ void store_wt(float* p, const f32x4 v) {        // write-through (sc1) 16-byte store
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
constexpr int SPLIT_TICKET_BYTES = 4096 * 4;      // ticket ints in front of the slabs
constexpr int SPLIT_MAX_TILES = 4096;

// LNF (round 6: the LayerNorm that consumes the Linear's fp32 result -- norm2 of the block, norm1 of the next block, the norm in front
// of a spatial reduction or the heads, reference nets/masked_layer_norm.py:113-125 -- folded into the producer at widths where a
// row spans several tiles, C = 512 / 1024).  Every tile stores its part of the rows write-through, drains and takes the ticket of
// its ROW BLOCK; the holder of the block's last ticket (all N tiles of these rows are in memory) acquires and runs the LayerNorm
// forward's row routine (ln.hip: one wave per row, same arithmetic) on the block's rows -- 128 - 256 KB read back from L2, y / mean /
// rstd written -- while the other workgroups have moved on.  No separate ln_fwd launch (25 per step at stages 2 - 3), no second trip
// of the residual stream through HBM.
// NVJ float4 per lane and row (C <= 256 NVJ); R = 8 / NVJ rows of a wave in flight at once -- the block's 64 rows are 4 - 8 batches of
// loads for the workgroup instead of 16 row round trips per wave (one row at a time: +35 us per launch, 7.69 against 6.85 ms per step)
template <int NVJ>
__device__ __forceinline__ void lnf_rows(const vr_gemm_args& p, const vr_ln_epilogue& ln, int m0, int mend, int wave, int lane) {
    constexpr int R = NVJ >= 8 ? 1 : 8 / NVJ, NWV = KTHR / 64;      // (8 float4 of rows per lane: the kernel keeps its 96 registers)
    const int C = p.N;
    const float* x = reinterpret_cast<const float*>(p.C);
    bf16_t* y = reinterpret_cast<bf16_t*>(ln.y);
    for (int mb = m0 + wave * R; mb < mend; mb += NWV * R) {
        float4 v[R][NVJ];
        int kcs[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int m = min(mb + u, mend - 1);                                    // (rows past the block: loaded again, never stored)
            kcs[u] = ln.keep ? ln.keep[p.rows_in > 0 ? m / p.rows_in : 0] : C;
            const float* xr = x + (long long)m * p.ldc;
#pragma unroll
            for (int j = 0; j < NVJ; ++j) {
                const int c = (lane + 64 * j) * 4;
                v[u][j] = *reinterpret_cast<const float4*>(xr + (c < C ? c : 0));
            }
        }
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int m = mb + u;
            if (m >= mend) break;
            const int kc = kcs[u];
            float s = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < NVJ; ++j) {
                const int c = (lane + 64 * j) * 4;
                if (c >= C) v[u][j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c < C) {
                    if (c + 0 >= kc) v[u][j].x = 0.f;
                    if (c + 1 >= kc) v[u][j].y = 0.f;
                    if (c + 2 >= kc) v[u][j].z = 0.f;
                    if (c + 3 >= kc) v[u][j].w = 0.f;
                    s += v[u][j].x + v[u][j].y + v[u][j].z + v[u][j].w;
                    s2 += v[u][j].x * v[u][j].x + v[u][j].y * v[u][j].y + v[u][j].z * v[u][j].z + v[u][j].w * v[u][j].w;
                }
            }
            s = wave_sum(s);
            const float inv_n = kc > 0 ? 1.0f / (float)kc : 0.f;
            const float mu = s * inv_n;
            float var;
            if (ln.keep) {                                   // masked path: var = E[x^2] / p - mu^2 (masked_layer_norm.py:38-40)
                s2 = wave_sum(s2);
                var = s2 * inv_n - mu * mu;
            } else {                                         // plain F.layer_norm path (:118-122): two-pass variance
                float d2 = 0.f;
#pragma unroll
                for (int j = 0; j < NVJ; ++j) {
                    const int c = (lane + 64 * j) * 4;
                    if (c < C) {
                        const float a = v[u][j].x - mu, b1 = v[u][j].y - mu, c1 = v[u][j].z - mu, d1 = v[u][j].w - mu;
                        d2 += a * a + b1 * b1 + c1 * c1 + d1 * d1;
                    }
                }
                var = wave_sum(d2) * inv_n;
            }
            const float rs = 1.0f / sqrtf(var + ln.eps);
            if (lane == 0) {
                ln.mean[m] = mu;
                ln.rstd[m] = rs;
            }
            bf16_t* yr = y + (long long)m * C;
#pragma unroll
            for (int j = 0; j < NVJ; ++j) {
                const int c = (lane + 64 * j) * 4;
                if (c < C) {
                    const float4 w4 = *reinterpret_cast<const float4*>(ln.w + c), b4 = *reinterpret_cast<const float4*>(ln.b + c);
                    const float o0 = (c + 0 < kc) ? w4.x * ((v[u][j].x - mu) * rs) + b4.x : 0.f;
                    const float o1 = (c + 1 < kc) ? w4.y * ((v[u][j].y - mu) * rs) + b4.y : 0.f;
                    const float o2 = (c + 2 < kc) ? w4.z * ((v[u][j].z - mu) * rs) + b4.z : 0.f;
                    const float o3 = (c + 3 < kc) ? w4.w * ((v[u][j].w - mu) * rs) + b4.w : 0.f;
                    *reinterpret_cast<uint2*>(yr + c) = make_uint2(pack_bf2(o0, o1), pack_bf2(o2, o3));
                }
            }
        }
    }
}
constexpr int LNF_TICKET_OFF = 2048;               // ints: the row blocks' tickets live in the upper half of the ticket area of ws

template <typename TO, int EPI, int MI, int NJ, int NBUF, int FEAT, bool BKM, bool DEEP_EPI = false, bool SPLIT = false, bool LNF = false>
__global__ __launch_bounds__(KTHR, MI == 4 ? 4 : 5) void ntk_kernel(const vr_gemm_args p, const int ksplit, const vr_ln_epilogue ln) {
    constexpr int BM = 32 * MI, WROWS = 16 * MI;
    constexpr int BN = 32 * NJ, WCOLS = 16 * NJ;
    constexpr int A_BYTES = BM * BK * 2, AP = MI;
    constexpr int B_BYTES = BN * BK * 2, BP = NJ;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int META_OFF = NBUF * STAGE_BYTES;
    static_assert(META_OFF >= 4 * 4096, "epilogue park area");
    // ONE shared array (slice ring | row metadata): a second __shared__ object makes hipcc drain the DMA queue before LDS reads
    __shared__ __attribute__((aligned(1024))) char smem[META_OFF + BM * (int)sizeof(RowMeta) + ((SPLIT || LNF) ? 16 : 0)];
    RowMeta* rowmeta = reinterpret_cast<RowMeta*>(smem + META_OFF);
    // forms whose fully masked tiles hold nothing but zeros (bf16 result, no residual): see the write skipping below
    constexpr bool SKIP_FORM = sizeof(TO) == 2 && ((EPI == EPI_STORE && FEAT <= 1) || EPI == EPI_GELU || EPI == EPI_DMUL);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = group_tiles(p.M, BM, p.m_groups);
    const int total = tiles_n * tiles_m;
    int tile = blockIdx.x;
    int zs = 0;                // SPLIT: this workgroup's share of the tile's slices
    if constexpr (SPLIT) {     // grid = 8 x ceil(total / 8) x ksplit: XCD x (= id % 8) owns a contiguous run of tiles, its workgroups
                               // walk (tile, share) share-fastest -- the shares of a tile are neighbours on one XCD
        const int x = tile & 7, i = tile >> 3;
        const int xq = total >> 3, xr = total & 7;
        const int tl = i / ksplit;
        zs = i - tl * ksplit;
        if (tl >= xq + (x < xr ? 1 : 0)) return;
        tile = x * xq + min(x, xr) + tl;
    } else if (total >= 16) {       // workgroup ids go round-robin to the 8 XCDs: an XCD owns one contiguous run of the n-fastest order
        const int xq = total >> 3, xr = total & 7, x = tile & 7;
        tile = x * xq + min(x, xr) + (tile >> 3);
    }
    const int tile_lin = tile;
    // row tiles of a multi-architecture batch never straddle two groups (gemm_shared.h group_tile_rows): rows [m0, mend)
    const int tn = tile % tiles_n, n0 = tn * BN;
    int m0, mend;
    group_tile_rows(tile / tiles_n, p.M, BM, p.m_groups, m0, mend);

    // ---- per-lane byte offsets of this wave's LDS-DMA pieces (constant over the K loop) ----
    unsigned voffA[AP], voffB[BP];
    {
        const RowMap amap = {p.a_map.rpi, p.a_map.rps, p.a_map.off};
        const int ra = wave * (8 * AP) + (lane >> 3);
#pragma unroll
        for (int h = 0; h < AP; ++h) {
            const int r = ra + 8 * h;
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            const int ma = min(m0 + r, mend - 1);
            voffA[h] = (unsigned)((map_row(amap, ma) * (long long)p.lda + c * 8) * 2);
        }
        if constexpr (BKM) {
            typedef KMajor<BN> G;
#pragma unroll
            for (int h = 0; h < BP; ++h) {
                const int tk = (wave * BP + h) * G::TPP + lane / G::SLOTS;
                const int c = (lane % G::SLOTS) ^ G::swz(tk);
                const bool bok = n0 + c * 8 + 8 <= p.ldb;          // chunks past the row's readable width read as zeros
                voffB[h] = bok ? (unsigned)((tk * p.ldb + n0 + c * 8) * 2) : 0xfffffff0u;
            }
        } else {
            const RowMap bmap = {p.b_map.rpi, p.b_map.rps, p.b_map.off};
            const int rb = wave * (8 * BP) + (lane >> 3);
#pragma unroll
            for (int h = 0; h < BP; ++h) {
                const int r = rb + 8 * h;
                const int c = (lane & 7) ^ ((r >> 1) & 7);
                const int nb = min(n0 + r, p.N - 1);
                voffB[h] = (unsigned)((map_row(bmap, nb) * (long long)p.ldb + c * 8) * 2);
            }
        }
    }
    // descriptors: rows are clamped, so every A / row-major B address is inside the operand; the k-major weight's descriptor ends
    // with the matrix (K rows of ldb): offsets >= its size -- the 0xfffffff0 above -- return zeros without an access
    const v4i_k rsA = make_rsrc(p.A, 0xffffff00u);
    const v4i_k rsB = make_rsrc(p.B, BKM ? (unsigned)((long long)p.K * p.ldb * 2) : 0xffffff00u);
    const int stepB = BKM ? BK * p.ldb * 2 : BK * 2;               // bytes a slice advances the weight operand by
    const unsigned lds0 = vr_dma::lds_addr(smem);  // LDS byte address of the ring

    // The loads are inline asm (M0 = LDS destination of the piece, written in the same statement): hipcc waits vmcnt(0) in front
    // of the first LDS read that follows a `raw_ptr_buffer_load_lds` BUILTIN -- the slices in flight would be drained every
    // round.  Hidden from its bookkeeping they are counted by hand below (s_waitcnt vmcnt(N) + s_barrier before any read).
    auto issue = [&](int kt, int buf) {
        const int sa = kt * (BK * 2), sb = kt * stepB;
        const unsigned dst = lds0 + buf * STAGE_BYTES + wave * (AP * 1024);
        const unsigned dstb = lds0 + buf * STAGE_BYTES + A_BYTES + wave * (BP * 1024);
#pragma unroll
        for (int h = 0; h < AP; ++h) dma16(dst + h * 1024, voffA[h], rsA, sa);
#pragma unroll
        for (int h = 0; h < BP; ++h) dma16(dstb + h * 1024, voffB[h], rsB, sb);
    };

    // ---- the first slice leaves before the masks are known (slice 0 is live whenever anything is) ----
    const int ntiles = p.K / BK;
    const int first = SPLIT ? zs : 0;          // (host: ksplit <= ntiles / 4)
    issue(first, 0);

    // ---- masked-work skipping: live slices as a bit mask ----
    unsigned long long live = ntiles >= 64 ? ~0ull : ((1ull << ntiles) - 1ull);
    if (p.keep_k || p.keep_n) {
        int s_lo = 0, s_hi = 0;
        if (p.rows_in > 0) { s_lo = m0 / p.rows_in; s_hi = (min(m0 + BM, mend) - 1) / p.rows_in; }
        bool any = true;
        if (p.keep_n) {
            int gmax = 0;
            any = range_has_kept(n0, BN, p.n_period, max_keep_wave(p.keep_n, s_lo, s_hi, lane, gmax));
            // A tile whose columns are masked for every row would store zeros.  With sched bit 0x40000 the caller vouches that
            // every reader of this output is one of the group-pure bf16 kernels (gemm_ntk / gemm_nt_ln / gemm_tn; the attention
            // cores skip per sample) and reads, for these rows, only the channels their architecture GROUP keeps: a tile beyond
            // the group's width (gmax) is then not written at all.  A sample masked on its own (DropPath) shares its group -- and
            // a reader's tile -- with live rows: below the group's width its zeros ARE written.
            if constexpr (SKIP_FORM) {
                if (!any && !range_has_kept(n0, BN, p.n_period, gmax) && (p.sched & 0x40000) &&
                    (p.m_groups <= 1 || group_pure(p.M, p.m_groups))) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // slice 0 is on its way into this workgroup's LDS
                    return;
                }
            }
        }
        if (p.keep_k) {
            int gk = 0;
            const int kmax = max_keep_wave(p.keep_k, s_lo, s_hi, lane, gk);
            if (kmax <= 0) any = false;
            else if (p.k_period <= 0) {                                  // plain prefix: the slices below kmax
                const int nl = min(ntiles, (kmax + BK - 1) / BK);
                live = nl >= 64 ? ~0ull : ((1ull << nl) - 1ull);
            } else if (p.k_period >= BK) {                              // periodic prefix (per-head widths): one scalar pass
                unsigned long long m = 0;
                int nr = 0;
                for (int kt = 0; kt < ntiles; ++kt) {
                    if (nr < kmax || nr + BK > p.k_period) m |= 1ull << kt;
                    nr += BK;
                    if (nr >= p.k_period) nr -= p.k_period;
                }
                live = m;
            }
        }
        if (!any) live = 0;
    }
    if constexpr (SPLIT) {               // this share's slices: every ksplit-th live one
        unsigned long long m = live, mine = 0;
        int c = 0;
        while (m) {
            const int kt = __builtin_ctzll(m);
            m &= m - 1;
            if (c == zs) mine |= 1ull << kt;
            c = c + 1 == ksplit ? 0 : c + 1;
        }
        live = mine;
    }
    if (t < BM) {                        // per-row epilogue metadata (its loads overlap the first slice)
        const int m = m0 + t;
        RowMeta rm;
        rm.keep = 1 << 30; rm.scale = 1.0f; rm.orow = -1; rm.mloc = 0;
        if (m < mend) {
            const int sample = p.rows_in > 0 ? m / p.rows_in : 0;
            rm.mloc = p.rows_in > 0 ? m - sample * p.rows_in : m;
            rm.orow = (int)map_row({p.c_map.rpi, p.c_map.rps, p.c_map.off}, m);
            if (p.scale) rm.scale = p.scale[sample];
            if (p.keep_n) rm.keep = p.keep_n[sample];
        }
        rowmeta[t] = rm;
    }

    // ---- fragment read offsets: lane -> row (lane & 15) of a 16-row group, k-chunk 4 s + (lane >> 4) ----
    const int frow = lane & 15, fswz = (frow >> 1) & 7;
    const int slot0 = (((lane >> 4)) ^ fswz) << 4, slot1 = ((4 + (lane >> 4)) ^ fswz) << 4;
    const char* As = smem + (wm * WROWS + frow) * 128;
    const char* Bs = smem + A_BYTES + (wn * WCOLS + frow) * 128;
    int offB[NJ];
    if constexpr (BKM) {
        typedef KMajor<BN> G;
        const int li = lane & 15, g4 = lane >> 4;
        const int xr2 = G::swz(8 * g4 + (li >> 2));
        const int rowoff = (8 * g4 + (li >> 2)) * G::ROWB + (li & 1) * 8;
#pragma unroll
        for (int j = 0; j < NJ; ++j) offB[j] = A_BYTES + rowoff + ((((BN / 16) * wn + 2 * j + ((li & 3) >> 1)) ^ xr2) * 16);
    }

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        const char* Ab = As + buf * STAGE_BYTES;
        const char* Bb = Bs + buf * STAGE_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int so = s == 0 ? slot0 : slot1;
            bfv8 a[MI], b[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const bfv8*>(Ab + i * 2048 + so);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if constexpr (BKM) b[j] = tr_frag<KMajor<BN>::ROWB>(smem + buf * STAGE_BYTES + offB[j] + s * 32 * KMajor<BN>::ROWB);
                else b[j] = *reinterpret_cast<const bfv8*>(Bb + j * 2048 + so);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
    };

    // slice 0 went out unconditionally: with nothing live it is drained and dropped
    const bool first_live = ((live >> first) & 1ull) != 0;
    live &= ~(1ull << first);
    if constexpr (NBUF == 1) {
        bool have = first_live;
        if (!first_live && live) {                    // (periodic masks can skip slice 0)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const int kt = __builtin_ctzll(live);
            live &= live - 1;
            issue(kt, 0);
            have = true;
        }
        while (have) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute(0);
            __syncthreads();
            have = live != 0;
            if (have) {
                const int kt = __builtin_ctzll(live);
                live &= live - 1;
                issue(kt, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else {
        // ring of NBUF buffers: the slice multiplied this round was requested NBUF - 1 rounds ago.  in_flight = slices issued and not
        // yet multiplied (the head one is the oldest); a dead slice 0 is multiplied too (zero weight would be wrong: it is real data
        // of masked channels) -- so it is waited for and skipped instead.
        int pending = 1;                              // slices in flight
        int head = 0, tail = 1;                       // ring positions: head = buffer of the oldest slice in flight
        bool skip_head = !first_live;
#pragma unroll 1
        while (pending > 0 || live) {
            // top up the ring
            while (pending < NBUF - 1 && live) {
                const int kt = __builtin_ctzll(live);
                live &= live - 1;
                issue(kt, tail);
                tail = tail + 1 == NBUF ? 0 : tail + 1;
                ++pending;
            }
            // the head slice has landed when at most the younger slices' pieces are outstanding
            if (NBUF >= 6 && pending >= 5) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (AP + BP)) : "memory");
            else if (NBUF >= 5 && pending == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (AP + BP)) : "memory");
            else if (NBUF >= 4 && pending == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (AP + BP)) : "memory");
            else if (pending >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (AP + BP)) : "memory");
            else if (pending == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AP + BP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();             // every wave's pieces of the head slice are in LDS; the buffer multiplied last
                                                      // round is free for the refill below
            if (live) {
                const int kt = __builtin_ctzll(live);
                live &= live - 1;
                issue(kt, tail);
                tail = tail + 1 == NBUF ? 0 : tail + 1;
                ++pending;
            }
            if (!skip_head) compute(head);
            skip_head = false;
            head = head + 1 == NBUF ? 0 : head + 1;
            --pending;
        }
        __syncthreads();
    }

    if constexpr (SPLIT) {
        constexpr int SLAB = BM * BN;
        float* slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(p.ws) + SPLIT_TICKET_BYTES) + (size_t)tile_lin * ksplit * SLAB;
        int* tickets = reinterpret_cast<int*>(p.ws);
        float* mine = slabs + (size_t)zs * SLAB + t * 4;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) store_wt(mine + (i * NJ + j) * (KTHR * 4), acc[i][j]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem + META_OFF + BM * (int)sizeof(RowMeta));
        if (t == 0) *flag = __hip_atomic_fetch_add(tickets + tile_lin, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*flag != ksplit - 1) return;                              // not the last share of this tile to arrive
        if (t == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(tickets + tile_lin, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        // the slabs in share order, this workgroup's own included (read back): a fixed summation order whoever arrives last
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < ksplit; ++z) {
            const float* src = slabs + (size_t)z * SLAB + t * 4;
#pragma unroll
            for (int r0 = 0; r0 < MI * NJ; r0 += 4) {                 // (four 16-byte loads in flight)
                f32x4 part[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) part[r] = *reinterpret_cast<const f32x4*>(src + (r0 + r) * (KTHR * 4));
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[(r0 + r) / NJ][(r0 + r) % NJ] += part[r];
            }
        }
    }

    // side operands of the epilogue (fp32 residual rows, saved gelu'): with two 16-row rounds per wave all of them are requested up
    // front (DEPTH = MI: 32 / 16 registers) -- one exposed HBM latency per tile instead of one per round
    constexpr int EDEPTH = (MI == 2 && DEEP_EPI) ? 2 : 1;
    epilogue<TO, EPI, true, MI, NJ, FEAT, EDEPTH, false, LNF>(p, acc, reinterpret_cast<float*>(smem + wave * 4096), rowmeta + wm * WROWS,
                                                             n0 + wn * WCOLS, lane);
    if constexpr (LNF) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this tile's part of the rows is in memory
        __syncthreads();
        int* tickets = reinterpret_cast<int*>(p.ws) + LNF_TICKET_OFF;
        int* flag = reinterpret_cast<int*>(smem + META_OFF + BM * (int)sizeof(RowMeta));
        const int rb = tile_lin / tiles_n;                           // row block (position in the row-tile order)
        if (t == 0) *flag = __hip_atomic_fetch_add(tickets + rb, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*flag != tiles_n - 1) return;                            // another tile of these rows is still on its way
        if (t == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(tickets + rb, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (p.N <= 512) lnf_rows<2>(p, ln, m0, min(m0 + BM, mend), wave, lane);
        else if (p.N <= 1024) lnf_rows<4>(p, ln, m0, min(m0 + BM, mend), wave, lane);
        else lnf_rows<8>(p, ln, m0, min(m0 + BM, mend), wave, lane);
    }
}

template <typename TO, int EPI, int MI, int NJ, int NBUF, int FEAT, bool BKM> void klaunch(const vr_gemm_args& a, hipStream_t stream, int shares,
                                                                                             const vr_ln_epilogue* ln) {
    const vr_ln_epilogue ln0 = ln ? *ln : vr_ln_epilogue{};
    const long long total = (long long)group_tiles(a.M, 32 * MI, a.m_groups) * ((a.N + 32 * NJ - 1) / (32 * NJ));
    constexpr bool SIDE = (EPI == EPI_STORE && FEAT >= 2) || EPI == EPI_DMUL;
    // the forms of the block Linears have a split kernel (FEAT 2: a block without DropPath -- the first one -- is never a long-K one)
    constexpr bool CAN_SPLIT = NBUF <= 3 && ((EPI == EPI_STORE && (FEAT == 0 || FEAT == 1 || FEAT == 3)) || EPI == EPI_GELU || EPI == EPI_DMUL);
    if constexpr (sizeof(TO) == 4 && EPI == EPI_STORE && FEAT >= 2 && !BKM && NBUF <= 3 && MI == 2 && NJ == 4) {     // (64 x 128 tiles)
        if (ln) {
            hipLaunchKernelGGL((ntk_kernel<TO, EPI, MI, NJ, NBUF, FEAT, BKM, true, false, true>), dim3((unsigned)total), dim3(KTHR), 0, stream, a, 1,
                               ln0);
            return;
        }
    }
    if constexpr (CAN_SPLIT) {
        if (shares > 1) {
            const unsigned grid = (unsigned)(8 * ((total + 7) / 8) * shares);
            hipLaunchKernelGGL((ntk_kernel<TO, EPI, MI, NJ, NBUF, FEAT, BKM, MI == 2 && SIDE, true>), dim3(grid), dim3(KTHR), 0, stream, a, shares, ln0);
            return;
        }
    }
    hipLaunchKernelGGL((ntk_kernel<TO, EPI, MI, NJ, NBUF, FEAT, BKM, MI == 2 && SIDE>), dim3((unsigned)total), dim3(KTHR), 0, stream, a, 1, ln0);
}

template <typename TO, int EPI, int FEAT, bool BKM> void ktile(const vr_gemm_args& a, hipStream_t stream, int tile, int nbuf, int shares,
                                                                const vr_ln_epilogue* ln = nullptr) {
    if (tile == 1) {
        if (nbuf == 1) klaunch<TO, EPI, 4, 4, 1, FEAT, BKM>(a, stream, shares, ln);
        else if (nbuf == 2) klaunch<TO, EPI, 4, 4, 2, FEAT, BKM>(a, stream, shares, ln);
        else if (nbuf == 3) klaunch<TO, EPI, 4, 4, 3, FEAT, BKM>(a, stream, shares, ln);
        else klaunch<TO, EPI, 4, 4, 4, FEAT, BKM>(a, stream, 1, ln);
    } else if (tile == 2) {
        if (nbuf == 1) klaunch<TO, EPI, 2, 4, 1, FEAT, BKM>(a, stream, shares, ln);
        else if (nbuf == 2) klaunch<TO, EPI, 2, 4, 2, FEAT, BKM>(a, stream, shares, ln);
        else if (nbuf == 3) klaunch<TO, EPI, 2, 4, 3, FEAT, BKM>(a, stream, shares, ln);
        else if (nbuf == 4) klaunch<TO, EPI, 2, 4, 4, FEAT, BKM>(a, stream, 1, ln);
        else if (nbuf == 5) klaunch<TO, EPI, 2, 4, 5, FEAT, BKM>(a, stream, 1, ln);
        else klaunch<TO, EPI, 2, 4, 6, FEAT, BKM>(a, stream, 1, ln);
    } else {
        if (nbuf == 1) klaunch<TO, EPI, 2, 2, 1, FEAT, BKM>(a, stream, shares, ln);
        else if (nbuf == 2) klaunch<TO, EPI, 2, 2, 2, FEAT, BKM>(a, stream, shares, ln);
        else if (nbuf == 3) klaunch<TO, EPI, 2, 2, 3, FEAT, BKM>(a, stream, shares, ln);
        else if (nbuf == 4) klaunch<TO, EPI, 2, 2, 4, FEAT, BKM>(a, stream, 1, ln);
        else klaunch<TO, EPI, 2, 2, 6, FEAT, BKM>(a, stream, 1, ln);
    }
}

}  // namespace vr_gemm_nt

bool vr_gemm_panel_launch(const vr_gemm_args& a, hipStream_t stream, int n_cu);      // gemm_panel.hip: panel-resident stage-1 kernels

// Called by vr_gemm_nt_launch in front of gemm_nt.hip's own kernels.  Returns false when the form is not covered here.
// ln != nullptr (vr_gemm_ln_fold): the LNF kernels -- fp32 result with bias + residual on 64 x 128 tiles, whole rows [M][N = ldc],
// tickets in a.ws; anything else returns false and nothing is launched.
bool vr_gemm_ntk_launch(const vr_gemm_args& a, hipStream_t stream, int n_cu, const vr_ln_epilogue* ln) {
    using namespace vr_gemm_nt;
    if (ln) {
        if (a.out_dtype != VR_F32 || !a.bias || !a.resid || a.b_trans || a.dact_u || a.act || a.c_map.rpi || a.ldc != a.N || a.N % 4 ||
            a.N > 2048 || !a.ws || a.ws_bytes < SPLIT_TICKET_BYTES || group_tiles(a.M, 64, a.m_groups) > 2048 || ln->mode != 0 || !ln->w ||
            !ln->b || !ln->y || !ln->mean || !ln->rstd || (a.sched & (8 | 16 | 32 | 0x100)))
            return false;
    } else if (!(a.sched & (8 | 16 | 32 | 0x100)) && vr_gemm_panel_launch(a, stream, n_cu)) return true;
    if (a.sched & (8 | 16 | 32 | 0x100)) return false;      // forms of gemm_nt.hip forced by the caller (tests, measurement aids)
    // an operand with unwritten masked tiles (0x80000) is readable only by the group-pure row tiling: refused where it cannot be had
    if ((a.sched & 0x80000) && a.keep_k && a.m_groups > 1 && !group_pure(a.M, a.m_groups)) return false;
    if (a.in_dtype != VR_BF16 || a.a_trans || a.atomic || a.split_k > 1 || a.bias_grad || a.pos) return false;
    if (a.K % BK || a.K > 64 * BK || a.K < BK) return false;
    const bool fast = a.N % 8 == 0 && a.ldc % 8 == 0 && (!a.dact_u || a.ldu % 8 == 0) && (a.n_period <= 0 || a.n_period % 8 == 0);
    if (!fast || a.lda % 8 || a.ldb % 8 || ((uintptr_t)a.A & 15) || ((uintptr_t)a.B & 15)) return false;
    if (a.k_period > 0 && a.k_period < BK) return false;
    // 32-bit byte offsets: the furthest row an operand reaches (mapped rows included) must stay below 4 GB
    auto reach = [](const vr_rowmap& m, long long rows, long long ld) {
        const long long last = m.rpi == 0 ? rows - 1 : ((rows - 1) / m.rpi) * (long long)m.rps + m.off + (rows - 1) % m.rpi;
        return (last + 1) * ld * 2;
    };
    if (reach(a.a_map, a.M, a.lda) >= 0xfff00000LL) return false;
    if (a.b_trans ? (long long)a.K * a.ldb * 2 >= 0xfff00000LL : reach(a.b_map, a.N, a.ldb) >= 0xfff00000LL) return false;
    const bool of32 = a.out_dtype == VR_F32;
    int feat = -1;
    if (!a.bias && !a.resid && !a.scale) feat = 0;
    else if (a.bias && !a.resid && !a.scale) feat = 1;
    else if (a.bias && a.resid && !a.scale) feat = 2;
    else if (a.bias && a.resid && a.scale) feat = 3;
    if (feat < 0) return false;
    const bool gelu = a.act == 1 || a.act == 3 || (a.act == 2 && !a.dact_u);
    // tile by grid size (gemm_nt.hip's measured crossovers); slices in flight by how many workgroups a CU gets
    const long long tn = (a.N + 127) / 128;
    const long long t128 = (long long)((a.M + 127) / 128) * tn, t64 = (long long)((a.M + 63) / 64) * tn;
    // sched bits 0x1800: tile override (1: 128 x 128, 2: 64 x 128, 3: 64 x 64), 0x600: slice buffers 1 - 3 (tests); vr_gemm_args.ring /
    // k_shares: slice buffers 1 - 6 / shares
    const int s_tile = ln ? 2 : ((a.sched >> 11) & 3), s_buf = (a.sched >> 9) & 3;
    // (in-graph sweep of all nine tile x depth combinations, profiles/r05_ntk_policy_sweep.txt: 64 x 128 tiles for the forms with a
    // bf16 side operand or two outputs once the 128 x 128 grid is below four / six per CU -- they run beside the weight-gradient
    // group's resident workgroups, which leave room for three 25 KB workgroups but only two 34 KB ones --, and 64 x 64 only when
    // even the 64 x 128 grid leaves CUs empty)
    int auto_tile = t128 >= 2LL * n_cu ? 1 : (t64 >= 2LL * n_cu ? 2 : 3);
    if (auto_tile == 3 && t64 >= n_cu) auto_tile = 2;
    if (auto_tile == 1 && a.dact_u && a.b_trans) auto_tile = 2;                       // fc2 data gradient (times the saved gelu')
    if (auto_tile == 1 && gelu && a.C2 && t128 < 4LL * n_cu) auto_tile = 2;           // fc1 forward (training: two outputs) of the second stage
    // K-split (SPLIT kernels): vr_gemm_args.k_shares = 2 - 4 shares wherever the form has a split kernel and the workspace holds the
    // slabs; 0 / 1: never.  No rule turns it on: measured on every stage 2 / 3 shape of the step (tools/ksplit_sweep.py,
    // profiles/r06_ksplit_sweep.txt) every real split loses to the unsplit kernel -- what paces a lone workgroup per CU is not memory
    // latency (rings of 4 - 6 slices: +-0) but its own serial chain of LDS-DMA issue, LDS reads and MFMAs per slice, which a second
    // workgroup on the CU overlaps as well as a share does, without the slab round trip (tiles x shares x 32 - 64 KB written through and
    // read back: 1 - 3x the GEMM's own bytes at these sizes) and without the extra tail round.
    const bool split_form = feat != 2 && a.ws;
    const int ntiles = a.K / BK;
    int shares = 1;
    constexpr int rule_tile = 0, rule_ring = 0;
    if (split_form && a.k_shares > 1 && !ln) shares = std::min(std::min(a.k_shares, 4), std::max(1, ntiles / 2));
    const int tile = s_tile ? s_tile : (rule_tile ? rule_tile : auto_tile);
    const long long wgs = (long long)group_tiles(a.M, tile == 1 ? 128 : 64, a.m_groups) * ((a.N + (tile == 3 ? 63 : 127)) / (tile == 3 ? 64 : 128));
    const int stage_kb = tile == 1 ? 32 : (tile == 2 ? 24 : 16);
    if (shares > 1) {
        const long long slab = (tile == 1 ? 128LL * 128 : (tile == 2 ? 64LL * 128 : 64LL * 64)) * 4;
        if (wgs > SPLIT_MAX_TILES || (long long)SPLIT_TICKET_BYTES + wgs * shares * slab > a.ws_bytes) shares = 1;
    }
    // slices in flight: as many as LDS allows WITHOUT lowering the number of workgroups the grid gives a CU (a 96 KB ring that leaves
    // a CU one workgroup where three single-buffer ones would run loses: candidate scoring, M4352 N2304 K1280 36 -> 44 us)
    int auto_buf = 1;
    if (a.K >= 4 * BK) {
        const long long wg_all = wgs * shares;
        const int per_cu = (int)std::max<long long>(1, (2 * wg_all + n_cu) / (2LL * n_cu));          // workgroups per CU, rounded
        const int want = std::min(per_cu, tile == 1 ? 4 : 5);
        // data gradients run beside the weight-gradient group, whose two resident workgroups hold 64 KB of a CU's LDS
        const int lds_kb = a.b_trans ? 96 : 160;
        const int deepest = ntiles >= 8 ? 3 : 2;
        for (int nb = deepest; nb >= 1; --nb)
            if (lds_kb / (nb * stage_kb + 2) >= want) { auto_buf = nb; break; }
    }
    int nbuf = a.ring > 0 ? a.ring : (s_buf ? s_buf : (rule_ring ? rule_ring : auto_buf));
    if (nbuf > 3) shares = 1;                                   // (the deep rings have no split kernels)
    nbuf = std::min(nbuf, tile == 1 ? 4 : 6);
    if (tile == 3 && nbuf == 5) nbuf = 4;
    if (ln && (a.b_trans || gelu || !of32)) return false;
    if (a.b_trans) {
        if (of32 || feat != 0 || gelu || a.b_map.rpi != 0 || a.ldb < (a.N + 7) / 8 * 8) return false;
        if (a.dact_u) {
            if (a.act != 2) return false;
            ktile<bf16_t, EPI_DMUL, 0, true>(a, stream, tile, nbuf, shares);
        } else {
            ktile<bf16_t, EPI_STORE, 0, true>(a, stream, tile, nbuf, shares);
        }
        return true;
    }
    if (a.dact_u) return false;
    if (gelu) {
        if (of32 || feat != 1 || a.act == 3) return false;
        ktile<bf16_t, EPI_GELU, 1, false>(a, stream, tile, nbuf, shares);
        return true;
    }
    if (of32) {
        if (ln) nbuf = std::min(nbuf, 3);
        if (feat == 3) ktile<float, EPI_STORE, 3, false>(a, stream, tile, nbuf, shares, ln);
        else if (feat == 2) ktile<float, EPI_STORE, 2, false>(a, stream, tile, nbuf, shares, ln);
        else if (feat == 1 && !ln) ktile<float, EPI_STORE, 1, false>(a, stream, tile, nbuf, shares);
        else return false;
        return true;
    }
    if (ln) return false;
    if (feat == 1) ktile<bf16_t, EPI_STORE, 1, false>(a, stream, tile, nbuf, shares);
    else if (feat == 0) ktile<bf16_t, EPI_STORE, 0, false>(a, stream, tile, nbuf, shares);
    else return false;
    return true;
}
