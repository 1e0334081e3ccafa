// bf16 forward / data-gradient GEMM, wide form:  C[M,N] = epilogue(A[M,K] * B[N,K]^T)   (or B = W [K][N] k-major, b_trans)
//
// The Linears of the second and third ViT-Res stages (reference nets/supernet_blocks.py:37-52,102-119 at 65 / 17 tokens per
// sample: M = 8320 / 2176 rows, K and N = 512 ... 3072).  gemm_nt.hip covers them with 128 x 128 (or 64 x 128) tiles, one slice
// in flight per workgroup and several un-synchronised workgroups per CU: at four workgroups on every CU that form reaches
// ~1000 TFLOP/s, but these outputs are 130 - 540 such tiles on 256 CUs -- one or two per CU, each paying the full load latency
// per slice (350 - 430 TFLOP/s dense).  This kernel is built to run ONE workgroup per CU at full speed and to cut the work
// into exactly one share per CU:
//
//   * 256 x 128 tile per 512-thread workgroup, 8 waves as 4 (M) x 2 (N), each 64 x 64 = 4 x 4 v_mfma_f32_16x16x32_bf16;
//   * the upper and lower four waves are the two halves of a ping-pong (waves w and w + 4 share a SIMD): a K slice (64) is two
//     phases per wave, each {8 fragment reads of one k step + 3 LDS-DMA pieces} | s_barrier | {16 MFMAs} | s_barrier, and the
//     second half runs one barrier behind the first -- on every SIMD one wave multiplies while its partner reads and stages;
//   * a ring of three 48 KB slice buffers: the slice two ahead is staged while the current one is multiplied, and ONE counted
//     s_waitcnt vmcnt per slice retires the slice in between (all LDS is one array: a second __shared__ object makes hipcc
//     drain the DMA queue in front of every fragment read);
//   * stream-K (vr_gemm_args.ws): tiles that do not fill a round of the chip are cut into equal contiguous shares of
//     (tile, slice) units, one per workgroup.  A tile cut between workgroups is summed by the LAST of them to arrive: every
//     contributor writes its fp32 accumulators (register layout, 16 B per lane: no transposition) write-through to its own
//     128 KB slab, drains, takes a ticket; the holder of the last ticket acquires, adds the other slabs to its registers and
//     runs the epilogue.  Nobody waits for anybody: no co-residency or dispatch order is assumed; tickets return to zero.
//
// LDS images (XOR-swizzled 128-B rows; k-major weight slice + ds_read_b64_tr_b16 for b_trans) and the epilogue are those of
// gemm_nt.hip (gemm_nt_parts.h).
#include <cstdlib>

#include "../gemm_nt_parts.h"

namespace vr_gemm_nt {

constexpr int WTHR = 512;
constexpr int W_BM = 256, W_BN = 128;
constexpr int SLAB_FLOATS = W_BM * W_BN;            // one workgroup's accumulators: 128 KB
constexpr int TICKET_BYTES = 4096 * 4;

__device__ const uint4 zero_chunk_w[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};

struct NtwPlan {
    int tiles_m, tiles_n, ns;      // output tiles; K slices of a tile
    int sk_tiles;                  // tiles [0, sk_tiles) are shared slice-wise; the others go round-robin, whole
    int grid;                      // workgroups
    float* slabs;                  // [grid][2][SLAB_FLOATS] partial accumulators
    int* tickets;                  // [sk_tiles], zero on entry and on exit
    int dbg;                       // timing experiments (VITRES_NTW_DBG; 1 / 2 / 4 give wrong results): 1 no slab stores, 2 no slab reads,
                                   // 4 plain stores + release fence, 8 record stamps
};

__device__ __forceinline__ void store_wt(float* p, const f32x4 v) {        // write-through (sc1) 16-byte store
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// VITRES_NTW_DBG & 8: workgroups < 64 record wall-clock stamps (100 MHz) of their first segments in the upper half of the ticket
// array (tools/ntw_stamps.py prints them)
#define NTW_STAMP(slot)                                                                                           \
    do {                                                                                                          \
        if ((pl.dbg & 8) && t == 0 && blockIdx.x < 64 && stamp_i < 16)                                            \
            reinterpret_cast<long long*>(pl.tickets + 2048)[blockIdx.x * 16 + stamp_i++] = (long long)wall_clock64() * 16 + (slot); \
    } while (0)

template <typename TO, int EPI, int FEAT, bool BKM>
__global__ __launch_bounds__(WTHR, 2) void ntw_kernel(const vr_gemm_args p, const NtwPlan pl) {
    constexpr int MI = 4, NJ = 4, BM = W_BM, BN = W_BN, WROWS = 64, WCOLS = 64;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, AP = 4, BP = 2;      // LDS-DMA pieces (1 KB) per wave and slice
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES, STAGES = 3;
    constexpr int META_OFF = STAGES * STAGE_BYTES, FLAG_OFF = META_OFF + BM * (int)sizeof(RowMeta);
    __shared__ __attribute__((aligned(1024))) char smem[FLAG_OFF + 16];   // ring of slice buffers | row metadata | ticket broadcast
    RowMeta* rowmeta = reinterpret_cast<RowMeta*>(smem + META_OFF);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 1, wc = wave & 1;            // wave row (rows 64 wr ..), wave column (columns 64 wc ..)
    const int half = wave >> 2;                         // ping-pong half: waves w and w + 4 share a SIMD
    const int G = pl.grid, ns = pl.ns;
    // workgroup ids are dealt round-robin to the 8 XCDs: an XCD owns a contiguous run of shares (n-fastest tile order: its L2
    // fetches an A panel once)
    int w = blockIdx.x;
    if (G >= 16) {
        const int xq = G >> 3, xr = G & 7, x = w & 7;
        w = x * xq + min(x, xr) + (w >> 3);
    }
    const unsigned U = (unsigned)pl.sk_tiles * ns;            // slice units of the shared tiles (host: U * grid < 2^31)
    const int lo = (int)(U * (unsigned)w / (unsigned)G), hi = (int)(U * (unsigned)(w + 1) / (unsigned)G);
    const int tiles = pl.tiles_m * pl.tiles_n;
    const RowMap amap = {p.a_map.rpi, p.a_map.rps, p.a_map.off};
    const RowMap bmap = {p.b_map.rpi, p.b_map.rps, p.b_map.off};
    const char* zero = reinterpret_cast<const char*>(zero_chunk_w);

    // ---- fragment read offsets inside a slice buffer: lane -> row (lane & 15) of a 16-row group, k-chunk 4 s + (lane >> 4) ----
    const int frow = lane & 15, fswz = (frow >> 1) & 7;
    const int slot0 = ((lane >> 4) ^ fswz) << 4, slot1 = ((4 + (lane >> 4)) ^ fswz) << 4;
    const int offA = (wr * WROWS + frow) * 128;                                         // + i * 2048 + slot
    const int offBn = A_BYTES + (wc * WCOLS + frow) * 128;                              // + j * 2048 + slot
    int offB[NJ];                                                                       // k-major weight slice: + s * 32 * ROWB
    if constexpr (BKM) {
        typedef KMajor<BN> KG;
        const int li = lane & 15, g4 = lane >> 4;
        const int xr2 = KG::swz(8 * g4 + (li >> 2));
        const int rowoff = (8 * g4 + (li >> 2)) * KG::ROWB + (li & 1) * 8;
#pragma unroll
        for (int j = 0; j < NJ; ++j) offB[j] = A_BYTES + rowoff + ((((BN / 16) * wc + 2 * j + ((li & 3) >> 1)) ^ xr2) * 16);
    }

    int stamp_i = 0;
    NTW_STAMP(0);
    int u = lo, dp_tile = pl.sk_tiles + w;
    while (true) {
        // ---- next segment: slices [kb, ke) of a tile ----
        int tile, kb, ke;
        bool first_seg = false;
        if (u < hi) {
            tile = u / ns;
            kb = u - tile * ns;
            ke = min(ns, kb + (hi - u));
            first_seg = u == lo;
            u += ke - kb;
        } else if (dp_tile < tiles) {
            tile = dp_tile;
            kb = 0;
            ke = ns;
            dp_tile += G;
        } else {
            break;
        }
        const int tn = tile % pl.tiles_n, tm = interleave_groups(tile / pl.tiles_n, pl.tiles_m, p.m_groups);
        const int m0 = tm * BM, n0 = tn * BN;

        // ---- masked-work skipping (rules of the general kernel; the keep arrays are read with scalar loads) ----
        int kmax = 1 << 30;
        bool n_any = true;
        if (p.keep_k || p.keep_n) {
            int s_lo = 0, s_hi = 0;
            if (p.rows_in > 0) { s_lo = m0 / p.rows_in; s_hi = (min(m0 + BM, p.M) - 1) / p.rows_in; }
            kmax = max_keep(p.keep_k, s_lo, s_hi, 1 << 30);
            const int nmax = max_keep(p.keep_n, s_lo, s_hi, 1 << 30);
            n_any = range_has_kept(n0, BN, p.n_period, nmax);
        }
        LiveSlices live;             // cursor over the slices with kept k (gemm_shared.h)
        live.init(p.keep_k, p.k_period, kb, ke, kmax, n_any);
        auto take = [&]() -> int { return live.take(); };

        // ---- LDS-DMA sources: piece h of this wave = 8 rows of 128 B, lane -> (row, 16-byte slot) ----
        const char* gA[AP];
        const char* gB[BP];
#pragma unroll
        for (int h = 0; h < AP; ++h) {
            const int r = wave * (8 * AP) + 8 * h + (lane >> 3);
            const int ma = min(m0 + r, p.M - 1);
            gA[h] = reinterpret_cast<const char*>(p.A) + map_row(amap, ma) * (long long)p.lda * 2 + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int h = 0; h < BP; ++h) {
            if constexpr (BKM) {
                typedef KMajor<BN> KG;
                const int tk = (wave * BP + h) * KG::TPP + lane / KG::SLOTS;
                const int c = (lane % KG::SLOTS) ^ KG::swz(tk);
                // column chunks past the row's readable width (ldb >= roundup(N, 8)) come from the zero page: their products only
                // reach outputs that are not stored
                const bool bok = n0 + c * 8 + 8 <= p.ldb;
                gB[h] = bok ? reinterpret_cast<const char*>(p.B) + ((long long)tk * p.ldb + n0 + c * 8) * 2 : nullptr;
            } else {
                const int r = wave * (8 * BP) + 8 * h + (lane >> 3);
                const int nb = min(n0 + r, p.N - 1);
                gB[h] = reinterpret_cast<const char*>(p.B) + map_row(bmap, nb) * (long long)p.ldb * 2 + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
            }
        }
        // stage this wave's pieces of slice kt into ring buffer buf: PART 0 = A pieces 0-2, 1 = A piece 3 + both B pieces, 2 = all
        auto stage = [&]<int PART>(int kt, int buf) {
            const long long kbytes = (long long)kt * (BK * 2);
            char* dst = smem + buf * STAGE_BYTES;
#pragma unroll
            for (int h = 0; h < AP; ++h)
                if (PART == 2 || (PART == 0) == (h < 3))
                    __builtin_amdgcn_global_load_lds((glb_void*)(gA[h] + kbytes), (lds_void*)(dst + (wave * AP + h) * 1024), 16, 0, 0);
            if constexpr (PART != 0) {
#pragma unroll
                for (int h = 0; h < BP; ++h) {
                    const char* src;
                    if constexpr (BKM) src = gB[h] ? gB[h] + (long long)kt * BK * p.ldb * 2 : zero;
                    else src = gB[h] + kbytes;
                    __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(dst + A_BYTES + (wave * BP + h) * 1024), 16, 0, 0);
                }
            }
        };

        f32x4 acc[MI][NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // One phase = k step S (32 of the slice's 64) out of ring buffer BUF, staging half of slice c2 into buffer (BUF + 2) % 3
        // (last read a slice ago: every wave retired those reads in front of a barrier this wave has passed).
        auto phase = [&]<int S, int BUF>(const int c2, const bool last_of_slice) {
            const char* sb_ = smem + BUF * STAGE_BYTES;
            const int so = S ? slot1 : slot0;
            bfv8 fa[MI], fb[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const bfv8*>(sb_ + offA + i * 2048 + so);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if constexpr (BKM) fb[j] = tr_frag<KMajor<BN>::ROWB>(sb_ + offB[j] + S * 32 * KMajor<BN>::ROWB);
                else fb[j] = *reinterpret_cast<const bfv8*>(sb_ + offBn + j * 2048 + so);
            }
            if (c2 < ke) stage.template operator()<S>(c2, (BUF + 2) % STAGES);
            if (last_of_slice) {
                // the next slice (staged a slice ago) has landed when at most the six pieces of slice c2 are outstanding
                if (c2 < ke) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // fragment reads retired IN FRONT of the barrier (see above)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };

        NTW_STAMP(1);
        // ---- prologue: slices c0 and c1 whole; per-row epilogue metadata while they fly ----
        int c0 = take();
        int c1 = take();
        if (c0 < ke) stage.template operator()<2>(c0, 0);
        if (c1 < ke) stage.template operator()<2>(c1, 1);
        if (t < BM) {
            const int m = m0 + t;
            RowMeta rm;
            rm.keep = 1 << 30; rm.scale = 1.0f; rm.orow = -1; rm.mloc = 0;
            if (m < p.M) {
                const int sample = p.rows_in > 0 ? m / p.rows_in : 0;
                rm.mloc = p.rows_in > 0 ? m - sample * p.rows_in : m;
                rm.orow = (int)map_row({p.c_map.rpi, p.c_map.rps, p.c_map.off}, m);
                if (p.scale) rm.scale = p.scale[sample];
                if (p.keep_n) rm.keep = p.keep_n[sample];
            }
            rowmeta[t] = rm;
        }
        if (c1 < ke) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        if (half == 1) __builtin_amdgcn_s_barrier();      // the second half runs one barrier behind the first
        __builtin_amdgcn_sched_barrier(0);
        NTW_STAMP(2);
        while (c0 < ke) {
#pragma unroll
            for (int b = 0; b < STAGES; ++b) {
                if (c0 >= ke) break;
                const int c2 = take();
                if (b == 0) { phase.template operator()<0, 0>(c2, false); phase.template operator()<1, 0>(c2, true); }
                else if (b == 1) { phase.template operator()<0, 1>(c2, false); phase.template operator()<1, 1>(c2, true); }
                else { phase.template operator()<0, 2>(c2, false); phase.template operator()<1, 2>(c2, true); }
                c0 = c1;
                c1 = c2;
            }
            // (a segment whose live-slice count is not a multiple of three ends inside the ring: the next segment starts at
            // buffer 0 again behind the workgroup barrier below)
        }
        if (half == 0) __builtin_amdgcn_s_barrier();      // (the barrier the second half is still owed)
        __syncthreads();
        NTW_STAMP(3);

        // ---- a tile cut between workgroups: the last contributor to arrive sums the partial accumulators ----
        bool finish = kb == 0 && ke == ns;
        if (!finish) {
            const unsigned u0 = (unsigned)tile * ns;
            const int w_first = (int)(((u0 + 1) * G - 1) / U), w_last = (int)(((u0 + ns) * G - 1) / U);
            float* mine = pl.slabs + ((size_t)w * 2 + (first_seg ? 0 : 1)) * SLAB_FLOATS + t * 4;
            if (!(pl.dbg & 1)) {
                if (pl.dbg & 4) {
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) *reinterpret_cast<f32x4*>(mine + (i * NJ + j) * (WTHR * 4)) = acc[i][j];
                } else {
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) store_wt(mine + (i * NJ + j) * (WTHR * 4), acc[i][j]);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int* flag = reinterpret_cast<int*>(smem + FLAG_OFF);
            if (t == 0) {
                if (pl.dbg & 4) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                *flag = __hip_atomic_fetch_add(pl.tickets + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            const int ticket = *flag;
            if (ticket == w_last - w_first) {
                if (t == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    __hip_atomic_store(pl.tickets + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
                for (int wc_ = w_first; wc_ <= w_last; ++wc_) {
                    if (wc_ == w || (pl.dbg & 2)) continue;
                    const int lo_c = (int)(U * (unsigned)wc_ / (unsigned)G);
                    const float* src = pl.slabs + ((size_t)wc_ * 2 + (lo_c / ns == tile ? 0 : 1)) * SLAB_FLOATS + t * 4;
                    f32x4 part[MI * NJ];
#pragma unroll
                    for (int r = 0; r < MI * NJ; ++r) part[r] = *reinterpret_cast<const f32x4*>(src + r * (WTHR * 4));
#pragma unroll
                    for (int r = 0; r < MI * NJ; ++r) acc[r / NJ][r % NJ] += part[r];
                }
                finish = true;
            }
        }
        NTW_STAMP(4);
        if (finish)
            epilogue<TO, EPI, true, MI, NJ, FEAT, MI>(p, acc, reinterpret_cast<float*>(smem + wave * 4096), rowmeta + wr * WROWS, n0 + wc * WCOLS, lane);
        __syncthreads();
        NTW_STAMP(5);
    }
}

template <typename TO, int EPI, int FEAT, bool BKM = false> void wlaunch(const vr_gemm_args& a, const NtwPlan& pl, hipStream_t stream) {
    hipLaunchKernelGGL((ntw_kernel<TO, EPI, FEAT, BKM>), dim3((unsigned)pl.grid), dim3(WTHR), 0, stream, a, pl);
}

}  // namespace vr_gemm_nt

// bytes of workspace the wide kernel wants for a chip of n_cu CUs (tickets + two slabs per workgroup)
size_t vr_gemm_ntw_ws_bytes(int n_cu) {
    return (size_t)vr_gemm_nt::TICKET_BYTES + (size_t)n_cu * 2 * vr_gemm_nt::SLAB_FLOATS * sizeof(float);
}

// Called by vr_gemm_nt_launch for forms both kernels cover.  mode: 1 = use the wide kernel where the cost model prefers it,
// 2 = wherever it is admissible.  Returns false when the 4-wave kernels should run.
bool vr_gemm_ntw_launch(const vr_gemm_args& a, hipStream_t stream, int n_cu, int mode) {
    using namespace vr_gemm_nt;
    if (a.in_dtype != VR_BF16 || a.a_trans || a.atomic || a.split_k > 1 || a.bias_grad || a.pos || a.K % BK || a.K < 2 * BK) return false;
    const bool fast = a.N % 8 == 0 && a.ldc % 8 == 0 && (!a.dact_u || a.ldu % 8 == 0) && (a.n_period <= 0 || a.n_period % 8 == 0);
    if (!fast) return false;
    const bool of32 = a.out_dtype == VR_F32;
    int feat = -1;
    if (!a.bias && !a.resid && !a.scale) feat = 0;
    else if (a.bias && !a.resid && !a.scale) feat = 1;
    else if (a.bias && a.resid && !a.scale) feat = 2;
    else if (a.bias && a.resid && a.scale) feat = 3;
    if (feat < 0) return false;
    const bool gelu = a.act == 1 || a.act == 3 || (a.act == 2 && !a.dact_u);
    if (gelu && (of32 || feat > 1)) return false;
    if (a.dact_u && (of32 || feat != 0)) return false;
    if (a.b_trans && (of32 || feat != 0 || gelu)) return false;      // (vr_gemm_nt_launch admitted the k-major form already)

    NtwPlan pl;
    pl.tiles_m = (a.M + W_BM - 1) / W_BM;
    pl.tiles_n = (a.N + W_BN - 1) / W_BN;
    pl.ns = a.K / BK;
    const long long tiles = (long long)pl.tiles_m * pl.tiles_n;
    static const int knob_fix = std::getenv("VITRES_NTW_FIX") ? std::atoi(std::getenv("VITRES_NTW_FIX")) : 8;   // a cut tile ~ this many slices
    static const int knob_sk = std::getenv("VITRES_NTW_SK") ? std::atoi(std::getenv("VITRES_NTW_SK")) : 1;      // 0: never share tiles; 2: always
    const size_t need = vr_gemm_ntw_ws_bytes(n_cu);
    const bool can_sk = knob_sk && a.ws && (size_t)a.ws_bytes >= need && tiles + n_cu <= 2048 && (tiles + n_cu) * pl.ns * (long long)n_cu < (1LL << 31);
    const int G = n_cu;
    const long long rounds = tiles / G, rem = tiles % G;
    const long long dp_cost = (rounds + (rem ? 1 : 0)) * pl.ns;
    long long sk_tiles = 0, cost = dp_cost;
    int grid = (int)(tiles < G ? tiles : G);
    if (can_sk && rem) {
        const long long skt = rounds ? rem + G : rem;               // two-tile form: every share holds at least one whole tile
        const long long units = skt * pl.ns;
        const int g2 = (int)(units / 4 < G ? (units / 4 < 1 ? 1 : units / 4) : G);      // at least four slices per share
        const long long sk_cost = (rounds ? rounds - 1 : 0) * pl.ns + (units + g2 - 1) / g2 + knob_fix;
        if ((sk_cost < dp_cost || knob_sk == 2) && (g2 == G || !rounds)) { sk_tiles = skt; cost = sk_cost; grid = g2; }
    }
    if (mode == 1) {
        // the 4-wave kernels keep the short-K GEMMs of the first stage (HBM-bound: four workgroups per CU hide the latency)
        if (pl.ns < 8) return false;
    }
    (void)cost;
    pl.sk_tiles = (int)sk_tiles;
    pl.grid = grid;
    static const int knob_dbg = std::getenv("VITRES_NTW_DBG") ? std::atoi(std::getenv("VITRES_NTW_DBG")) : 0;
    pl.dbg = knob_dbg;
    pl.tickets = reinterpret_cast<int*>(a.ws);
    pl.slabs = a.ws ? reinterpret_cast<float*>(reinterpret_cast<char*>(a.ws) + TICKET_BYTES) : nullptr;

    if (a.b_trans) {
        if (a.dact_u) {
            if (a.act == 2) wlaunch<bf16_t, EPI_DMUL, 0, true>(a, pl, stream);
            else wlaunch<bf16_t, EPI_DGELU, 0, true>(a, pl, stream);
        } else {
            wlaunch<bf16_t, EPI_STORE, 0, true>(a, pl, stream);
        }
    } else if (gelu) {
        if (feat == 1) wlaunch<bf16_t, EPI_GELU, 1>(a, pl, stream);
        else wlaunch<bf16_t, EPI_GELU, 0>(a, pl, stream);
    } else if (a.dact_u) {
        if (a.act == 2) wlaunch<bf16_t, EPI_DMUL, 0>(a, pl, stream);
        else wlaunch<bf16_t, EPI_DGELU, 0>(a, pl, stream);
    } else if (of32) {
        switch (feat) {
            case 0: wlaunch<float, EPI_STORE, 0>(a, pl, stream); break;
            case 1: wlaunch<float, EPI_STORE, 1>(a, pl, stream); break;
            case 2: wlaunch<float, EPI_STORE, 2>(a, pl, stream); break;
            default: wlaunch<float, EPI_STORE, 3>(a, pl, stream); break;
        }
    } else {
        switch (feat) {
            case 0: wlaunch<bf16_t, EPI_STORE, 0>(a, pl, stream); break;
            case 1: wlaunch<bf16_t, EPI_STORE, 1>(a, pl, stream); break;
            case 2: wlaunch<bf16_t, EPI_STORE, 2>(a, pl, stream); break;
            default: wlaunch<bf16_t, EPI_STORE, 3>(a, pl, stream); break;
        }
    }
    return true;
}
