// bf16 GEMM fast path for K-contiguous operands:  C[M,N] = epilogue(A[M,K] * B[N,K]^T)
//
// Forward of every nn.Linear on the ViT-Res hot path (reference nets/supernet_blocks.py:37-52,102-119) and, with the
// transposed bf16 weight shadow, their data gradients.  vr_gemm (gemm.hip) dispatches here; semantics of every
// vr_gemm_args field are identical to the general kernel.
//
// Structure (gfx950): 128x128 output tile per 256-thread workgroup, 4 waves as 2x2 each owning 64x64 = 4x4
// v_mfma_f32_16x16x32_bf16 tiles (64 accumulator registers -> <= 128 VGPRs -> 4 workgroups per CU).  One K slice =
// 64 bf16 = 128 B per row; both operand slices (32 KB) are moved global -> LDS by LDS-DMA (global_load_lds_dwordx4:
// no staging registers, no ds_write pass).  The slice is single-buffered: the four resident workgroups of a CU are
// what overlaps one workgroup's load latency with another's MFMAs -- at K = 256..1280 a tile is only 4..20 slices
// long, and the general kernel's two workgroups per CU with one slice in flight each were latency bound
// (~1.5 us per slice).
//
// LDS image: row r of a tile = 8 slots of 16 B; slot p holds k-chunk p ^ ((r >> 1) & 7).  LDS-DMA writes lane-linear
// (wave base + lane * 16), so the permutation is applied to each lane's SOURCE address; the fragment reads apply the
// same XOR.  A ds_read_b128 lane group (16 consecutive rows, one k-chunk) then covers all 16 slots of the 256-B bank
// row: conflict free.
//
// The MFMAs compute the transposed tile (weights as the first operand): a lane owns one output row and 4 consecutive
// columns per accumulator.  The epilogue parks 32 rows x 64 columns per wave in that wave's own 8 KB of the (now idle)
// slice buffer -- XOR-swizzled, no workgroup barrier -- and reads it back as whole rows: every store instruction
// writes 8 rows x 128 B (bf16) of full cache lines, side inputs (residual, GELU pre-activation, pos-embed) are loaded
// with the same shape, all loads of a round before its first store (vmcnt counts stores).
#include <cstdlib>

#include "gemm_nt_parts.h"

namespace vr_gemm_nt {

constexpr int NTHR = 256;

__device__ const uint4 zero_chunk[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};

// MI = 16-row fragments per wave along M: 4 -> 128-row tile (4 workgroups / CU), 2 -> 64-row tile (5 / CU; more, smaller
// workgroups for GEMMs that would leave the 128-row grid a partial last round)
// NJ = 16-column fragments per wave along N: 4 -> 128-column tile, 2 -> 64-column tile (long-K GEMMs with few tiles: a lone
// workgroup per CU pays the full ~1.5 us slice latency, four interleaved ones hide it)
// STAGES = 1: one slice buffer, the CU's other workgroups hide the load latency.  STAGES = 3: ring of three buffers with two
// slices in flight (counted s_waitcnt vmcnt, raw s_barrier: __syncthreads would drain the LDS-DMA queue) for grids that leave a
// CU one or two workgroups -- long-K GEMMs of the last stage, where a lone workgroup paid ~1 us per slice.
// sched bit 32 (measurement aid, with vr_gemm_args.ws): workgroups 0..63 record wall-clock stamps (100 MHz) -- slot 0 entry,
// 3 K loop done, 5 epilogue done -- in the upper half of the workspace's ticket array (tools/ntw_stamps.py prints them)
#define NT_STAMP(slot)                                                                                   \
    do {                                                                                                 \
        if ((p.sched & 32) && p.ws && threadIdx.x == 0 && blockIdx.x < 64)                               \
            reinterpret_cast<long long*>(reinterpret_cast<int*>(p.ws) + 2048)[blockIdx.x * 16 + (slot == 0 ? 0 : slot == 3 ? 1 : 2)] = \
                (long long)wall_clock64() * 16 + (slot);                                                 \
    } while (0)

template <typename TO, int EPI, bool FAST, int MI, int NJ, int STAGES, int FEAT, bool BKM = false>
__global__ __launch_bounds__(NTHR, MI == 4 ? 4 : 5) void nt_kernel(const vr_gemm_args p) {
    constexpr int BM = 32 * MI, WROWS = 16 * MI;      // tile rows, rows per wave
    constexpr int BN = 32 * NJ, WCOLS = 16 * NJ;      // tile columns, columns per wave
    constexpr int A_BYTES = BM * BK * 2, AP = MI;     // A slice bytes, LDS-DMA pieces of A per wave
    constexpr int B_BYTES = BN * BK * 2, BP = NJ;     // same for the weight slice
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int NBUF = STAGES > 10 ? STAGES - 10 : STAGES;      // STAGES = 10 + R: ring of R buffers, R - 1 slices in flight
    constexpr bool RING = STAGES == 3 || STAGES > 10;
    __shared__ __attribute__((aligned(1024))) char smem[NBUF * STAGE_BYTES];   // ring of [A slice][B slice]; epilogue: 4 x 4 KB
    __shared__ RowMeta rowmeta[BM];
    const int t = threadIdx.x, lane = t & 63;
    NT_STAMP(0);
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int total = tiles_n * tiles_m;
    // workgroup ids are dealt round-robin to the 8 XCDs: give each XCD one contiguous run of the n-fastest tile order
    int tile = blockIdx.x;
    if (total >= 16) {
        const int xq = total >> 3, xr = total & 7, x = tile & 7;
        tile = x * xq + min(x, xr) + (tile >> 3);
    }
    const int tn = tile % tiles_n, tm = interleave_groups(tile / tiles_n, tiles_m, p.m_groups);
    const int m0 = tm * BM, n0 = tn * BN;
    const RowMap amap = {p.a_map.rpi, p.a_map.rps, p.a_map.off};
    const RowMap bmap = {p.b_map.rpi, p.b_map.rps, p.b_map.off};

    // ---- masked-work skipping (same rules as the general kernel) ----
    int ntiles = (p.K + BK - 1) / BK;
    int kmax = 1 << 30;
    bool n_any = true;
    if (p.keep_k || p.keep_n) {
        int s_lo = 0, s_hi = 0;
        if (p.rows_in > 0) { s_lo = m0 / p.rows_in; s_hi = (min(m0 + BM, p.M) - 1) / p.rows_in; }
        kmax = max_keep(p.keep_k, s_lo, s_hi, 1 << 30);
        const int nmax = max_keep(p.keep_n, s_lo, s_hi, 1 << 30);
        n_any = range_has_kept(n0, BN, p.n_period, nmax);
    }
    LiveSlices live;                 // cursor over the slices with kept k (gemm_shared.h)
    live.init(p.keep_k, p.k_period, 0, ntiles, kmax, n_any);

    // ---- LDS-DMA source addressing: piece h of this wave = tile rows wave*32 + 8h .. +8, lane -> (row, slot) ----
    // Interior tiles of un-mapped operands (every block Linear) take the affine path: one row address per operand, pieces 8 rows
    // apart -- the general path (row clamps at the matrix edge, vr_rowmap divisions) costs ~40 instructions per piece, 8 pieces,
    // on a kernel whose K = 256 loop is 4 slices long (its instruction issue, not HBM or the matrix pipe, set the tile time).
    const char* gA[AP];
    const char* gB[BP];
    int chunkA[AP], chunkB[BP];    // element offset of this lane's k-chunk inside a slice
    int tokB[BP];                  // BKM: k row of this lane inside a slice, per piece
    if constexpr (BKM) {
        typedef KMajor<BN> G;
#pragma unroll
        for (int h = 0; h < BP; ++h) {
            const int tk = (wave * BP + h) * G::TPP + lane / G::SLOTS;
            const int c = (lane % G::SLOTS) ^ G::swz(tk);
            // column chunks past the row's readable width (ldb >= roundup(N, 8)) come from the zero page: their products only
            // reach outputs that are not stored
            const bool bok = n0 + c * 8 + 8 <= p.ldb;
            tokB[h] = tk;
            gB[h] = bok ? reinterpret_cast<const char*>(p.B) + ((long long)tk * p.ldb + n0 + c * 8) * 2 : nullptr;
            chunkB[h] = 0;
        }
    } else {
        const int rb = wave * (8 * BP) + (lane >> 3);
        if (bmap.rpi == 0 && n0 + BN <= p.N) {
            const char* b0 = reinterpret_cast<const char*>(p.B) + (long long)(n0 + rb) * p.ldb * 2;
            const long long step = (long long)p.ldb * 16;
#pragma unroll
            for (int h = 0; h < BP; ++h) {
                const int c = (lane & 7) ^ (((rb + 8 * h) >> 1) & 7);
                gB[h] = b0 + h * step + c * 16;
                chunkB[h] = c * 8;
            }
        } else {
#pragma unroll
            for (int h = 0; h < BP; ++h) {
                const int r = rb + h * 8;
                const int c = (lane & 7) ^ ((r >> 1) & 7);
                const int nb = min(n0 + r, p.N - 1);
                gB[h] = reinterpret_cast<const char*>(p.B) + (map_row(bmap, nb) * (long long)p.ldb + c * 8) * 2;
                chunkB[h] = c * 8;
            }
        }
    }
    {
        const int ra = wave * (8 * AP) + (lane >> 3);
        if (amap.rpi == 0 && m0 + BM <= p.M) {
            const char* a0 = reinterpret_cast<const char*>(p.A) + (long long)(m0 + ra) * p.lda * 2;
            const long long step = (long long)p.lda * 16;
#pragma unroll
            for (int h = 0; h < AP; ++h) {
                const int c = (lane & 7) ^ (((ra + 8 * h) >> 1) & 7);
                gA[h] = a0 + h * step + c * 16;
                chunkA[h] = c * 8;
            }
        } else {
#pragma unroll
            for (int h = 0; h < AP; ++h) {
                const int r = ra + h * 8;
                const int c = (lane & 7) ^ ((r >> 1) & 7);
                const int ma = min(m0 + r, p.M - 1);
                gA[h] = reinterpret_cast<const char*>(p.A) + (map_row(amap, ma) * (long long)p.lda + c * 8) * 2;
                chunkA[h] = c * 8;
            }
        }
    }
    const char* zero = reinterpret_cast<const char*>(zero_chunk);
    const bool ktail = (FEAT == 4 || BKM) && (p.K % BK) != 0;   // FEAT 0..3: K % 64 == 0 (host check; BKM: checked here)

    // ---- fragment read offsets: lane -> row (lane & 15) of a 16-row group, k-chunk 4 s + (lane >> 4) ----
    const int frow = lane & 15, fswz = (frow >> 1) & 7;
    const int slot0 = (((lane >> 4)) ^ fswz) << 4, slot1 = ((4 + (lane >> 4)) ^ fswz) << 4;
    const char* As = smem + (wm * WROWS + frow) * 128;
    const char* Bs = smem + A_BYTES + (wn * WCOLS + frow) * 128;
    int offB[NJ];                  // BKM: byte offset of fragment j's first transposing read inside the k-major weight slice
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

    auto issue = [&](int kt, int buf) {
        const int k0 = kt * BK;
        const long long kb = (long long)k0 * 2;
        char* dst = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int h = 0; h < AP; ++h) {
            const char* sa = (!ktail || (k0 + chunkA[h] < p.K)) ? gA[h] + kb : zero;
            __builtin_amdgcn_global_load_lds((glb_void*)sa, (lds_void*)(dst + (wave * (8 * AP) + h * 8) * 128), 16, 0, 0);
        }
#pragma unroll
        for (int h = 0; h < BP; ++h) {
            const char* sb;
            if constexpr (BKM) sb = (gB[h] && (!ktail || k0 + tokB[h] < p.K)) ? gB[h] + (long long)k0 * p.ldb * 2 : zero;
            else sb = (!ktail || (k0 + chunkB[h] < p.K)) ? gB[h] + kb : zero;
            __builtin_amdgcn_global_load_lds((glb_void*)sb, (lds_void*)(dst + A_BYTES + (wave * (8 * BP) + h * 8) * 128), 16, 0, 0);
        }
    };

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
    auto fill_rowmeta = [&]() {      // per-row epilogue metadata (its loads overlap the first slices)
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
    };

    if constexpr (STAGES == 1) {
        int kt = live.take();
        if (kt < ntiles) issue(kt, 0);
        fill_rowmeta();
        if (kt >= ntiles) __syncthreads();
        while (kt < ntiles) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute(0);
            __syncthreads();
            kt = live.take();
            if (kt < ntiles) issue(kt, 0);
        }
    } else if constexpr (STAGES == 2 || STAGES == 4) {
        // STAGES slices per round: all requested at once, all computed behind one wait -- two (four) times the bytes in flight
        // per workgroup for grids that leave a CU two or three workgroups (one at most: the M = 128 GEMMs of the class-token
        // rows, 8 - 32 workgroups on the whole chip, paid one full load latency per 64-wide slice), in LDS that no other
        // workgroup would have used
        int cs[STAGES];
#pragma unroll
        for (int q = 0; q < STAGES; ++q) {
            cs[q] = live.take();
            if (cs[q] < ntiles) issue(cs[q], q);
        }
        fill_rowmeta();
        if (cs[0] >= ntiles) __syncthreads();
        while (cs[0] < ntiles) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (int q = 0; q < STAGES; ++q)
                if (cs[q] < ntiles) compute(q);
            __syncthreads();
#pragma unroll
            for (int q = 0; q < STAGES; ++q) {
                cs[q] = live.take();
                if (cs[q] < ntiles) issue(cs[q], q);
            }
        }
    } else {
        static_assert(RING && NBUF >= 3 && NBUF <= 5, "ring depth");
        fill_rowmeta();              // its global loads complete (the compiler waits for them) before any LDS-DMA is issued
        // ring of NBUF buffers: slice c[0] is multiplied while c[1] .. c[NBUF - 2] are in flight (counted s_waitcnt vmcnt -- the
        // LDS-DMA pieces retire in issue order -- and a raw s_barrier: __syncthreads would drain the queue)
        int c[NBUF - 1];
#pragma unroll
        for (int q = 0; q < NBUF - 1; ++q) {
            c[q] = live.take();
            if (c[q] < ntiles) issue(c[q], q);
        }
        int buf = 0;
        while (c[0] < ntiles) {
            // slice c[0] has landed when at most the pieces of the younger slices are outstanding
            int younger = 0;
#pragma unroll
            for (int q = 1; q < NBUF - 1; ++q) younger += c[q] < ntiles ? 1 : 0;
            if (younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (AP + BP)) : "memory");
            else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (AP + BP)) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AP + BP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();         // every wave has also finished reading the buffer re-filled next
            const int cn = live.take();
            const int nb = buf == 0 ? NBUF - 1 : buf - 1;      // (buf + NBUF - 1) % NBUF: the buffer multiplied last round
            if (cn < ntiles) issue(cn, nb);
            compute(buf);
#pragma unroll
            for (int q = 0; q < NBUF - 2; ++q) c[q] = c[q + 1];
            c[NBUF - 2] = cn;
            buf = buf == NBUF - 1 ? 0 : buf + 1;
        }
        __syncthreads();
    }

    NT_STAMP(3);
    // (DEPTH = 2 for the fp32-residual forms -- the next round's residual requested once this round's accumulators are parked --
    // measured round 3: +10..17 registers, 4 - 12 spilled at the 128 / 96 caps, residual GEMMs 29.0 -> 30.1 / 19.3 -> 21.0 us: no)
    epilogue<TO, EPI, FAST, MI, NJ, FEAT>(p, acc, reinterpret_cast<float*>(smem + wave * 4096), rowmeta + wm * WROWS, n0 + wn * WCOLS,
                                         lane);
    NT_STAMP(5);
}

template <typename TO, int EPI, int MI, int NJ, int STAGES, int FEAT> void launch3(const vr_gemm_args& a, hipStream_t stream, bool fast) {
    const long long total = (long long)((a.M + 32 * MI - 1) / (32 * MI)) * ((a.N + 32 * NJ - 1) / (32 * NJ));
    if (fast) hipLaunchKernelGGL((nt_kernel<TO, EPI, true, MI, NJ, STAGES, FEAT>), dim3((unsigned)total), dim3(NTHR), 0, stream, a);
    else hipLaunchKernelGGL((nt_kernel<TO, EPI, false, MI, NJ, STAGES, 4>), dim3((unsigned)total), dim3(NTHR), 0, stream, a);
}

template <typename TO, int EPI, int MI, int NJ, int STAGES = 1> void launch2(const vr_gemm_args& a, hipStream_t stream, bool fast) {
    if (a.b_trans) {     // weights as the forward stores them (vr_gemm_nt_launch admitted the form): bf16 data gradients only
        if constexpr (sizeof(TO) == 2 && (EPI == EPI_STORE || EPI == EPI_DMUL || EPI == EPI_DGELU)) {
            const long long total = (long long)((a.M + 32 * MI - 1) / (32 * MI)) * ((a.N + 32 * NJ - 1) / (32 * NJ));
            hipLaunchKernelGGL((nt_kernel<TO, EPI, true, MI, NJ, STAGES, 0, true>), dim3((unsigned)total), dim3(NTHR), 0, stream, a);
        }
        return;
    }
    // epilogue form (see nt_kernel): the forms of the transformer-block Linears get their own kernels
    int feat = 4;
    if (fast && !a.pos && a.K % BK == 0) {
        if (!a.bias && !a.resid && !a.scale) feat = 0;
        else if (a.bias && !a.resid && !a.scale) feat = 1;
        else if (a.bias && a.resid && !a.scale) feat = 2;
        else if (a.bias && a.resid && a.scale) feat = 3;
    }
    if constexpr (EPI == EPI_STORE) {
        switch (feat) {
            case 0: return launch3<TO, EPI, MI, NJ, STAGES, 0>(a, stream, fast);
            case 1: return launch3<TO, EPI, MI, NJ, STAGES, 1>(a, stream, fast);
            case 2: return launch3<TO, EPI, MI, NJ, STAGES, 2>(a, stream, fast);
            case 3: return launch3<TO, EPI, MI, NJ, STAGES, 3>(a, stream, fast);
            default: return launch3<TO, EPI, MI, NJ, STAGES, 4>(a, stream, fast);
        }
    } else if constexpr (EPI == EPI_GELU) {
        if (feat == 1) return launch3<TO, EPI, MI, NJ, STAGES, 1>(a, stream, fast);
        if (feat == 0) return launch3<TO, EPI, MI, NJ, STAGES, 0>(a, stream, fast);
        return launch3<TO, EPI, MI, NJ, STAGES, 4>(a, stream, fast);
    } else {
        if (feat == 0) return launch3<TO, EPI, MI, NJ, STAGES, 0>(a, stream, fast);
        return launch3<TO, EPI, MI, NJ, STAGES, 4>(a, stream, fast);
    }
}

template <typename TO, int EPI> void launch1(const vr_gemm_args& a, hipStream_t stream, int n_cu) {
    const long long tn = (a.N + 127) / 128;
    const long long t128 = (long long)((a.M + 127) / 128) * tn, t64 = (long long)((a.M + 63) / 64) * tn;
    const bool fast = a.N % 8 == 0 && a.ldc % 8 == 0 && (!a.dact_u || a.ldu % 8 == 0) && (a.n_period <= 0 || a.n_period % 8 == 0);
    // tile by grid size (measured crossovers, tools/gemm_bench.py): 128x128 while it gives a CU two workgroups, 64x128
    // below that, 64x64 when even that leaves CUs with a single workgroup (long-K GEMMs of the last stage)
    const int tile = t128 >= 2LL * n_cu ? 1 : (t64 >= 2LL * n_cu ? 2 : 3);
    if (tile == 1) launch2<TO, EPI, 4, 4>(a, stream, fast);
    else if (tile == 2) {
        // every tile resident at three workgroups per CU and >= 8 slices: two slices per round (STAGES = 2)
        if (t64 <= 3LL * n_cu && a.K >= 8 * BK) launch2<TO, EPI, 2, 4, 2>(a, stream, fast);
        else launch2<TO, EPI, 2, 4>(a, stream, fast);
    } else {
        // 64 x 64 tiles: with fewer than ~3 workgroups per CU and a long K the slices are pipelined inside the workgroup
        const long long t3 = (long long)((a.M + 63) / 64) * ((a.N + 63) / 64);
        const bool ring = t3 < 3LL * n_cu && a.K >= 24 * BK;      // measured: +23 % at K = 3072, -3 % at K = 1024
        if (ring) launch2<TO, EPI, 2, 2, 3>(a, stream, fast);
        else if (t3 <= n_cu && a.K >= 4 * BK) launch2<TO, EPI, 2, 2, 4>(a, stream, fast);   // at most one workgroup per CU
        else if (a.K >= 8 * BK) launch2<TO, EPI, 2, 2, 2>(a, stream, fast);
        else launch2<TO, EPI, 2, 2>(a, stream, fast);
    }
}

}  // namespace vr_gemm_nt

bool vr_gemm_ntk_launch(const vr_gemm_args& a, hipStream_t stream, int n_cu, const vr_ln_epilogue* ln);      // gemm_ntk.hip: the lean-loop kernels

// Called by vr_gemm after validation.  Returns false when the form is not covered here.
bool vr_gemm_nt_launch(const vr_gemm_args& a, hipStream_t stream, int n_cu) {
    using namespace vr_gemm_nt;
    if (a.in_dtype != VR_BF16 || a.a_trans || a.atomic || a.split_k > 1 || a.bias_grad) return false;
    if (vr_gemm_ntk_launch(a, stream, n_cu, nullptr)) return true;
    // sched bit 0x80000: the operand may hold unwritten (fully masked) tiles, readable only by the group-pure row tiling of
    // gemm_ntk.hip -- the kernels below tile across architecture groups: refused (vr_gemm then fails loudly)
    if ((a.sched & 0x80000) && a.keep_k && a.m_groups > 1) return false;
    const bool of32 = a.out_dtype == VR_F32;
    if (a.b_trans) {
        // B = W [K][N] row-major (the forward's weight): plain data gradients with a bf16 result (optionally times gelu'), 16-byte
        // rows, the epilogue's vector form; anything else stays with the general kernel
        const bool fast = a.N % 8 == 0 && a.ldc % 8 == 0 && (!a.dact_u || a.ldu % 8 == 0) && (a.n_period <= 0 || a.n_period % 8 == 0);
        if (of32 || a.act == 1 || (a.act == 2 && !a.dact_u) || a.bias || a.resid || a.scale || a.pos || a.C2 ||
            a.b_map.rpi != 0 || !fast || a.ldb % 8 || ((uintptr_t)a.B & 15) || a.ldb < (a.N + 7) / 8 * 8)
            return false;
    }
    if (a.act == 1 || a.act == 3 || (a.act == 2 && !a.dact_u)) {
        if (of32) return false;
        launch1<bf16_t, EPI_GELU>(a, stream, n_cu);
    } else if (a.dact_u) {
        if (of32) return false;
        if (a.act == 2) launch1<bf16_t, EPI_DMUL>(a, stream, n_cu);
        else launch1<bf16_t, EPI_DGELU>(a, stream, n_cu);
    } else if (of32) {
        launch1<float, EPI_STORE>(a, stream, n_cu);
    } else {
        launch1<bf16_t, EPI_STORE>(a, stream, n_cu);
    }
    return true;
}
