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

#include "common.h"
#include "../../include/vitres_hip.h"
#include "gemm_shared.h"

namespace vr_gemm_nt {
using namespace vr_gemm_shared;

typedef __bf16 bfv8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

constexpr int BK = 64, NTHR = 256;

__device__ const uint4 zero_chunk[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};

// per-row epilogue metadata, computed once per tile by threads 0..127 while the first slice is in flight
struct RowMeta {
    int keep;      // kept output-column prefix of the row's sample (1 << 30: dense)
    float scale;   // DropPath scale of the row's sample
    int orow;      // output row after c_map, -1: row >= M
    int mloc;      // row index inside its sample (pos-embed row)
};

// FAST: N % 8 == 0, ldc / ldu % 8 == 0, n_period % 8 == 0 (checked on the host) -- every lane's 8-column group is whole
// or entirely outside the matrix, so the epilogue is branch-free 16-byte accesses.
// MI = 16-row fragments per wave along M: 4 -> 128-row tile (4 workgroups / CU), 2 -> 64-row tile (5 / CU; more, smaller
// workgroups for GEMMs that would leave the 128-row grid a partial last round)
// NJ = 16-column fragments per wave along N: 4 -> 128-column tile, 2 -> 64-column tile (long-K GEMMs with few tiles: a lone
// workgroup per CU pays the full ~1.5 us slice latency, four interleaved ones hide it)
// STAGES = 1: one slice buffer, the CU's other workgroups hide the load latency.  STAGES = 3: ring of three buffers with two
// slices in flight (counted s_waitcnt vmcnt, raw s_barrier: __syncthreads would drain the LDS-DMA queue) for grids that leave a
// CU one or two workgroups -- long-K GEMMs of the last stage, where a lone workgroup paid ~1 us per slice.
// FEAT (FAST kernels): which optional epilogue terms exist is a compile-time fact of the kernel -- 0: none, 1: bias, 2: bias +
// residual, 3: bias + residual + DropPath scale, 4: decided per launch from the arguments (pos-embed, any other mix; the only
// form of the non-FAST kernels).  As run-time uniform conditions the compiler if-converts them into a v_cndmask per element
// and term (measured: 890 VALU instructions per tile and wave against 128 MFMAs at K = 256, VALU pipe busy 2x the matrix pipe).
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

template <typename TO, int EPI, bool FAST, int MI, int NJ, int STAGES, int FEAT, bool BKM = false>
__global__ __launch_bounds__(NTHR, MI == 4 ? 4 : 5) void nt_kernel(const vr_gemm_args p) {
    constexpr int BM = 32 * MI, WROWS = 16 * MI;      // tile rows, rows per wave
    constexpr int BN = 32 * NJ, WCOLS = 16 * NJ;      // tile columns, columns per wave
    constexpr int A_BYTES = BM * BK * 2, AP = MI;     // A slice bytes, LDS-DMA pieces of A per wave
    constexpr int B_BYTES = BN * BK * 2, BP = NJ;     // same for the weight slice
    constexpr int LPR = 2 * NJ, RPP = 64 / LPR, NQ = 16 / RPP;   // epilogue: lanes per row, rows per pass, passes per 16 rows
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    __shared__ __attribute__((aligned(1024))) char smem[STAGES * STAGE_BYTES];   // ring of [A slice][B slice]; epilogue: 4 x 4 KB
    __shared__ RowMeta rowmeta[BM];
    const int t = threadIdx.x, lane = t & 63;
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
    const int tn = tile % tiles_n, tm = tile / tiles_n;
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
    auto slice_live = [&](int kt) -> bool {
        return n_any && (p.keep_k == nullptr || range_has_kept(kt * BK, BK, p.k_period, kmax));
    };
    auto next_live = [&](int kt) -> int {
        while (kt < ntiles && !slice_live(kt)) ++kt;
        return kt;
    };

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
        int kt = next_live(0);
        if (kt < ntiles) issue(kt, 0);
        fill_rowmeta();
        if (kt >= ntiles) __syncthreads();
        while (kt < ntiles) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute(0);
            __syncthreads();
            kt = next_live(kt + 1);
            if (kt < ntiles) issue(kt, 0);
        }
    } else if constexpr (STAGES == 2 || STAGES == 4) {
        // STAGES slices per round: all requested at once, all computed behind one wait -- two (four) times the bytes in flight
        // per workgroup for grids that leave a CU two or three workgroups (one at most: the M = 128 GEMMs of the class-token
        // rows, 8 - 32 workgroups on the whole chip, paid one full load latency per 64-wide slice), in LDS that no other
        // workgroup would have used
        int cs[STAGES];
        int nxt = 0;
#pragma unroll
        for (int q = 0; q < STAGES; ++q) {
            cs[q] = nxt < ntiles ? next_live(nxt) : ntiles;
            nxt = cs[q] < ntiles ? cs[q] + 1 : ntiles;
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
                cs[q] = nxt < ntiles ? next_live(nxt) : ntiles;
                nxt = cs[q] < ntiles ? cs[q] + 1 : ntiles;
                if (cs[q] < ntiles) issue(cs[q], q);
            }
        }
    } else {
        fill_rowmeta();              // its global loads complete (the compiler waits for them) before any LDS-DMA is issued
        int c0 = next_live(0);
        int c1 = c0 < ntiles ? next_live(c0 + 1) : ntiles;
        if (c0 < ntiles) issue(c0, 0);
        if (c1 < ntiles) issue(c1, 1);
        int buf = 0;
        while (c0 < ntiles) {
            // slice c0 has landed when at most the pieces of the younger slice c1 are outstanding
            if (c1 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AP + BP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();         // every wave has also finished reading the buffer re-filled next
            const int c2 = c1 < ntiles ? next_live(c1 + 1) : ntiles;
            const int nb = buf == 0 ? 2 : buf - 1;      // (buf + 2) % 3
            if (c2 < ntiles) issue(c2, nb);
            compute(buf);
            c0 = c1;
            c1 = c2;
            buf = buf == 2 ? 0 : buf + 1;
        }
        __syncthreads();
    }

    // ---- epilogue: lane owns C[m = 16 i + (lane & 15)][n = 16 j + 4 (lane >> 4) + 0..3] of the wave's 64 x 64 ----
    // Optional terms (bias, pos-embed, prefix mask, DropPath scale, residual) are wave-uniform kernel arguments: each is one
    // scalar branch around its code instead of arithmetic on neutral elements -- for the plain Linear forms the epilogue used
    // to issue 5x the instructions of the K = 256 loop.  The prefix mask is applied only by waves that hold a boundary group.
    constexpr int CW = 8;
    float* park = reinterpret_cast<float*>(smem + wave * 4096);      // [16 rows][16 slots of 4 floats], slot ^= row
    const int n = n0 + wn * WCOLS + (lane % LPR) * 8;                 // this lane's 8 columns
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
    const RowMeta* meta = rowmeta + wm * WROWS + (lane / LPR);
    // one round per 16-row fragment: park [16 rows][WCOLS columns] (<= 4 KB per wave), read back as rows.
    // Side operands of the epilogue (saved gelu'(u) / pre-activation of the fc2 data gradient, fp32 residual stream) are
    // requested as raw 16-byte loads ahead of their use -- the bf16 one a whole round ahead, the fp32 one (16 registers per
    // round: a second copy would spill) at the top of its round in front of the park / barrier: issued behind the barrier
    // of their own round, four rounds of exposed HBM latency made the fc2 data gradient 1.4x slower than the fc1 forward
    // of the same shape.
    constexpr bool SIDE_D = EPI == EPI_DGELU || EPI == EPI_DMUL;
    constexpr bool SIDE_R = EPI == EPI_STORE && (FEAT == 2 || FEAT == 3);
    constexpr bool PREF = FAST && (SIDE_D || SIDE_R);
    RowMeta rmn[NQ];
    uint4 dn[NQ];
    float4 rn[NQ][2];
    auto prefetch = [&](int i) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            rmn[q] = meta[i * 16 + q * RPP];
            const long long row = rmn[q].orow < 0 ? 0 : rmn[q].orow;
            if constexpr (SIDE_D)
                dn[q] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.dact_u) + row * p.ldu + nc);
            if constexpr (SIDE_R) {
                const float* r = p.resid + row * p.ldc + nc;
                rn[q][0] = *reinterpret_cast<const float4*>(r);
                rn[q][1] = *reinterpret_cast<const float4*>(r + 4);
            }
        }
    };
    if constexpr (SIDE_D && PREF) prefetch(0);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        if constexpr (SIDE_R && PREF) prefetch(i);
        RowMeta rm[NQ];
        long long oidx[NQ];
        float rv[NQ][CW], pv[NQ][CW];
        uint4 dc[NQ];
        float4 rc[NQ][2];
        if constexpr (PREF) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                rm[q] = rmn[q];
                dc[q] = dn[q];
                rc[q][0] = rn[q][0];
                rc[q][1] = rn[q][1];
            }
            if constexpr (SIDE_D) {
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
                if (any) storew<TO, CW>(p.C, oidx[q], v, vec, mok, nvalid);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
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
    static const int knob = std::getenv("VITRES_NT_TILE") ? std::atoi(std::getenv("VITRES_NT_TILE")) : 0;   // 1/2/3: force
    // tile by grid size (measured crossovers, tools/gemm_bench.py): 128x128 while it gives a CU two workgroups, 64x128
    // below that, 64x64 when even that leaves CUs with a single workgroup (long-K GEMMs of the last stage)
    const int tile = knob ? knob : (t128 >= 2LL * n_cu ? 1 : (t64 >= 2LL * n_cu ? 2 : 3));
    static const int knob_pair = std::getenv("VITRES_NT_PAIR") ? std::atoi(std::getenv("VITRES_NT_PAIR")) : 1;
    if (tile == 1) launch2<TO, EPI, 4, 4>(a, stream, fast);
    else if (tile == 2) {
        // every tile resident at three workgroups per CU and >= 8 slices: two slices per round (STAGES = 2)
        if (knob_pair && t64 <= 3LL * n_cu && a.K >= 8 * BK) launch2<TO, EPI, 2, 4, 2>(a, stream, fast);
        else launch2<TO, EPI, 2, 4>(a, stream, fast);
    } else {
        // 64 x 64 tiles: with fewer than ~3 workgroups per CU and a long K the slices are pipelined inside the workgroup
        static const int knob_st = std::getenv("VITRES_NT_STAGES") ? std::atoi(std::getenv("VITRES_NT_STAGES")) : 0;
        const long long t3 = (long long)((a.M + 63) / 64) * ((a.N + 63) / 64);
        const bool ring = knob_st ? knob_st == 3 : (t3 < 3LL * n_cu && a.K >= 24 * BK);      // measured: +23 % at K = 3072, -3 % at K = 1024
        if (ring) launch2<TO, EPI, 2, 2, 3>(a, stream, fast);
        else if (knob_pair && t3 <= n_cu && a.K >= 4 * BK) launch2<TO, EPI, 2, 2, 4>(a, stream, fast);   // at most one workgroup per CU
        else if (knob_pair && a.K >= 8 * BK) launch2<TO, EPI, 2, 2, 2>(a, stream, fast);
        else launch2<TO, EPI, 2, 2>(a, stream, fast);
    }
}

}  // namespace vr_gemm_nt

// Called by vr_gemm after validation.  Returns false when the form is not covered here.
bool vr_gemm_nt_launch(const vr_gemm_args& a, hipStream_t stream, int n_cu) {
    using namespace vr_gemm_nt;
    if (a.in_dtype != VR_BF16 || a.a_trans || a.atomic || a.split_k > 1 || a.bias_grad) return false;
    const bool of32 = a.out_dtype == VR_F32;
    if (a.b_trans) {
        // B = W [K][N] row-major (the forward's weight): plain data gradients with a bf16 result (optionally times gelu'), 16-byte
        // rows, the epilogue's vector form; anything else stays with the general kernel
        static const bool knob_km = !(std::getenv("VITRES_NT_BKM") && std::getenv("VITRES_NT_BKM")[0] == '0');
        const bool fast = a.N % 8 == 0 && a.ldc % 8 == 0 && (!a.dact_u || a.ldu % 8 == 0) && (a.n_period <= 0 || a.n_period % 8 == 0);
        if (!knob_km || of32 || a.act == 1 || (a.act == 2 && !a.dact_u) || a.bias || a.resid || a.scale || a.pos || a.C2 ||
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
