// vr_gemm_ln for the long, narrow problems of the first stage (N <= 256 channels, tens of thousands of token rows): the same
// two modes as gemm_nt_ln.hip -- Linear + residual + LayerNorm forward, and data gradient + LayerNorm backward -- in a form
// whose K loop streams from HBM instead of waiting for it.
//
// Why a second form.  gemm_nt_ln.hip's 64 x 256 tile is single-buffered (40 KB of LDS, so that two or three workgroups share a
// CU): every K slice is a full memory round trip, ~3.3 us per slice at M = 32896, K = 768 (tools/ntln_bench.py: 58 us, of which
// 40 in the K loop), and a deeper ring there costs the co-residency that hides the row loop.  Here ONE workgroup of eight waves
// owns a CU: a three-stage ring of (A: <= 144 rows, B: 256 weight rows) x 64 k = 3 x 50 KB, two slices always in flight.
//
// Balanced row ownership.  M is cut into 16-row blocks and the blocks are dealt to the workgroups in equal shares (8 or 9
// blocks = 128 or 144 rows at M = 32896 on 256 CUs) -- not into fixed 128-row tiles, whose 257th tile would run alone in a second
// round.  Larger M: every workgroup walks several such shares.
//
// Waves sit side by side along N (32 columns each, all of the tile's rows): acc[9][2] 16 x 16 fragments.  The row loop is
// gemm_nt_ln.hip's: the accumulators are parked in the (drained) ring as fp32 rows -- the whole tile at once, 144 KB -- and
// every wave walks whole rows; the side operands of the next four rows are requested before the current four are worked on.
#include <cstdlib>

#include "../common.h"
#include "../../../include/vitres_hip.h"
#include "../gemm_shared.h"

namespace vr_gemm_ntlnw {
using namespace vr_gemm_shared;

typedef __bf16 bfv8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

constexpr int BK = 64, NTHR = 512, NWAVE = 8;
constexpr int MIX = 9;                                   // 16-row blocks per tile, at most
constexpr int BN = 256, NJ = 2, WCOLS = 32;              // tile width; 16-column fragments / columns per wave
constexpr int A_STAGE = MIX * 16 * BK * 2;               // 18 KB
constexpr int B_STAGE = BN * BK * 2;                     // 32 KB
constexpr int STAGE = A_STAGE + B_STAGE, NST = 3;
constexpr int SLOTS = BN / 4;                            // 16-byte slots per parked row
constexpr int RU = 4;                                    // rows per round of the row loop
static_assert(MIX * 16 * BN * 4 <= NST * STAGE, "the parked tile fits the ring");

__device__ const uint4 zero_chunk[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};

struct RowMeta {
    int keep;      // forward: kept output-column prefix of the GEMM (1 << 30: dense); backward: gt_keep
    float scale;   // forward: DropPath scale of the row's sample; backward: gt_scale
    int orow;      // output row, -1: row >= M
    int lnkeep;    // kept prefix of the LayerNorm (N: dense)
    float mu, rs;  // backward: saved statistics of the row
};
struct Round {     // the same, in scalar registers (a round's rows are wave-uniform)
    int keep, orow, lnkeep;
    float scale, mu, rs;
};
struct Side {
    Round rm[RU];
    float4 rv[RU], xv[RU];
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float sgpr(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

template <int MODE, bool KTAIL>
__global__ __launch_bounds__(NTHR, 1) void ntlnw_kernel(const vr_gemm_args p, const vr_ln_epilogue f, const int nblk, const int ntile) {
    __shared__ __attribute__((aligned(1024))) char smem[NST * STAGE + MIX * 16 * (int)sizeof(RowMeta)];
    RowMeta* rowmeta = reinterpret_cast<RowMeta*>(smem + NST * STAGE);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const RowMap amap = {p.a_map.rpi, p.a_map.rps, p.a_map.off};
    const int ntiles = (p.K + BK - 1) / BK;
    const char* zero = reinterpret_cast<const char*>(zero_chunk);

    // fragment reads (layout of gemm_nt_ln.hip: 128-byte k rows, 16-byte chunk c of row r at slot c ^ ((r >> 1) & 7))
    const int frow = lane & 15, fswz = (frow >> 1) & 7;
    const int slot0 = ((lane >> 4) ^ fswz) << 4, slot1 = ((4 + (lane >> 4)) ^ fswz) << 4;
    const int a_off = frow * 128, b_off = A_STAGE + (wave * WCOLS + frow) * 128;
    const int g4 = lane >> 4, c16 = lane & 15;

    // weight rows of this wave's four LDS-DMA pieces (8 rows x 128 B each): the same for every tile
    const char* gB[4];
    int chunkB[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int r = (wave * 4 + h) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        gB[h] = reinterpret_cast<const char*>(p.B) + ((long long)min(r, p.N - 1) * p.ldb + c * 8) * 2;
        chunkB[h] = c * 8;
    }
    // this lane's four columns in the row loop
    const int col = 4 * lane;
    const bool cin = col < p.N;                                    // N % 8 == 0: a group is whole or outside
    const int cc = cin ? col : 0;
    const float4 lw = ld4(f.w + cc);
    float4 lb = make_float4(0.f, 0.f, 0.f, 0.f), bv = lb;
    if constexpr (MODE == 0) {
        lb = ld4(f.b + cc);
        if (p.bias) bv = ld4(p.bias + cc);
    }
    const bool has_res = p.resid != nullptr;
    float* park = reinterpret_cast<float*>(smem);

    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int blk0 = (int)((long long)tile * nblk / ntile), blk1 = (int)((long long)(tile + 1) * nblk / ntile);
        const int mi = blk1 - blk0;                                // 16-row blocks of this tile (<= MIX)
        const int m0 = blk0 * 16, rows = mi * 16;

        // ---- masked-work skipping (rules of the general kernel) ----
        int kmax = 1 << 30;
        bool n_any = true;
        if (p.keep_k || p.keep_n) {
            int s_lo = 0, s_hi = 0;
            if (p.rows_in > 0) { s_lo = m0 / p.rows_in; s_hi = (min(m0 + rows, p.M) - 1) / p.rows_in; }
            kmax = max_keep(p.keep_k, s_lo, s_hi, 1 << 30);
            n_any = max_keep(p.keep_n, s_lo, s_hi, 1 << 30) > 0;
        }
        LiveSlices live;
        live.init(p.keep_k, p.k_period, 0, ntiles, kmax, n_any);

        // ---- activation rows of this wave's LDS-DMA pieces: piece q = wave + 8 h covers tile rows 8 q .. 8 q + 7 ----
        const char* gA[3];
        int chunkA[3];
        const int nA = (2 * mi - wave + 7) >> 3;                   // pieces q < 2 mi of this wave (0 .. 3)
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            const int r = (wave + 8 * h) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            const int ma = min(m0 + r, p.M - 1);
            gA[h] = reinterpret_cast<const char*>(p.A) + (map_row(amap, ma) * (long long)p.lda + c * 8) * 2;
            chunkA[h] = c * 8;
        }
        auto issue = [&](int kt, int buf) {
            const int k0 = kt * BK;
            const long long kb = (long long)k0 * 2;
            char* st = smem + buf * STAGE;
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                if (h < nA) {
                    const char* sa = gA[h] + kb;
                    if constexpr (KTAIL) sa = (k0 + chunkA[h] < p.K) ? sa : zero;
                    __builtin_amdgcn_global_load_lds((glb_void*)sa, (lds_void*)(st + (wave + 8 * h) * 1024), 16, 0, 0);
                }
            }
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const char* sb = gB[h] + kb;
                if constexpr (KTAIL) sb = (k0 + chunkB[h] < p.K) ? sb : zero;
                __builtin_amdgcn_global_load_lds((glb_void*)sb, (lds_void*)(st + A_STAGE + (wave * 4 + h) * 1024), 16, 0, 0);
            }
        };

        f32x4 acc[MIX][NJ];
#pragma unroll
        for (int i = 0; i < MIX; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        int cur = live.take();
        int nxt = cur < ntiles ? live.take() : ntiles;
        if (cur < ntiles) issue(cur, 0);
        if (nxt < ntiles) issue(nxt, 1);

        if (t < rows) {                        // per-row metadata of the row loop; its loads overlap the first slices
            const int m = m0 + t;
            RowMeta rm;
            rm.keep = 1 << 30; rm.scale = 1.0f; rm.orow = -1; rm.lnkeep = p.N; rm.mu = 0.f; rm.rs = 0.f;
            if (m < p.M) {
                const int sample = p.rows_in > 0 ? m / p.rows_in : 0;
                rm.orow = m;
                if (f.keep) rm.lnkeep = min(f.keep[sample], p.N);
                if constexpr (MODE == 0) {
                    if (p.scale) rm.scale = p.scale[sample];
                    if (p.keep_n) rm.keep = p.keep_n[sample];
                } else {
                    if (f.gt_scale) rm.scale = f.gt_scale[sample];
                    if (f.gt_keep) rm.keep = f.gt_keep[sample];
                    rm.mu = f.mean[m];
                    rm.rs = f.rstd[m];
                }
            }
            rowmeta[t] = rm;
        }

        // ---- K loop: slice `cur` is computed while `nxt` and the one after it are on their way ----
        int buf = 0;
        while (cur < ntiles) {
            const int nn = nxt < ntiles ? live.take() : ntiles;
            if (nxt >= ntiles) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (nA == 3) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else if (nA == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (nA == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __syncthreads();                   // slice `cur` has landed for every wave; the slice before it is no longer read
            const int b2 = buf >= 1 ? buf - 1 : NST - 1;            // == (buf + 2) % 3: the stage that slice just left
            if (nn < ntiles) issue(nn, b2);
            const char* st = smem + buf * STAGE;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int so = s == 0 ? slot0 : slot1;
                bfv8 b[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[j] = *reinterpret_cast<const bfv8*>(st + b_off + j * 2048 + so);
#pragma unroll
                for (int i = 0; i < MIX; ++i) {
                    if (i < mi) {
                        const bfv8 a = *reinterpret_cast<const bfv8*>(st + a_off + i * 2048 + so);
#pragma unroll
                        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a, acc[i][j], 0, 0, 0);
                    }
                }
            }
            cur = nxt;
            nxt = nn;
            buf = buf == NST - 1 ? 0 : buf + 1;
        }
        __syncthreads();                       // the ring is drained and no wave reads it any more (also orders rowmeta)

        // ---- park the tile: lane holds C[16 i + c16][wave * 32 + 16 j + 4 g4 + 0..3]; 16-byte slot s of row r at s ^ (r & 7) ----
#pragma unroll
        for (int i = 0; i < MIX; ++i) {
            if (i < mi) {
                const int r = 16 * i + c16;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int slot = (wave * WCOLS + 16 * j + 4 * g4) >> 2;
                    *reinterpret_cast<f32x4*>(park + ((size_t)r * SLOTS + (slot ^ (r & 7))) * 4) = acc[i][j];
                }
            }
        }

        // ---- row loop: wave w walks tile rows w, w + 8, ... in rounds of RU ----
        const int nrow = 2 * mi;                                   // rows per wave
        const int nround = (nrow + RU - 1) / RU;
        auto request = [&](int it, Side& sd) {
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int k = it * RU + u;
                const RowMeta m = rowmeta[k < nrow ? wave + 8 * k : 0];
                Round& rm = sd.rm[u];
                rm.keep = __builtin_amdgcn_readfirstlane(m.keep);
                rm.orow = k < nrow ? __builtin_amdgcn_readfirstlane(m.orow) : -1;
                rm.lnkeep = __builtin_amdgcn_readfirstlane(m.lnkeep);
                rm.scale = sgpr(m.scale);
                rm.mu = sgpr(m.mu);
                rm.rs = sgpr(m.rs);
                const long long orow = rm.orow < 0 ? 0 : rm.orow;
                sd.rv[u] = has_res ? ld4(p.resid + orow * p.ldc + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (MODE == 1) sd.xv[u] = ld4(f.x + orow * p.ldc + cc);
            }
        };
        float4 gw = make_float4(0.f, 0.f, 0.f, 0.f), gb = gw;
        auto work = [&](int it, const Side& sd) {
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const Round& rm = sd.rm[u];
                if (rm.orow < 0) continue;                             // (wave-uniform) row beyond the tile or beyond M
                const int r = wave + 8 * (it * RU + u);
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(park + ((size_t)r * SLOTS + (lane ^ (r & 7))) * 4);
                const int lk = rm.lnkeep;
                const float inv_n = lk > 0 ? 1.0f / (float)lk : 0.f;
                if constexpr (MODE == 0) {
                    // x1 = resid + scale * mask(acc + bias); LayerNorm over the first lk channels (vr_ln_fwd)
                    const int kn = rm.keep - col;
                    const float sc = rm.scale;
                    float4 a;
                    a.x = (0 < kn) ? (a4[0] + bv.x) * sc : 0.f;
                    a.y = (1 < kn) ? (a4[1] + bv.y) * sc : 0.f;
                    a.z = (2 < kn) ? (a4[2] + bv.z) * sc : 0.f;
                    a.w = (3 < kn) ? (a4[3] + bv.w) * sc : 0.f;
                    a.x += sd.rv[u].x; a.y += sd.rv[u].y; a.z += sd.rv[u].z; a.w += sd.rv[u].w;
                    if (cin) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (long long)rm.orow * p.ldc + col) = a;
                    const int kl = cin ? lk - col : 0;
                    a.x = (0 < kl) ? a.x : 0.f; a.y = (1 < kl) ? a.y : 0.f; a.z = (2 < kl) ? a.z : 0.f; a.w = (3 < kl) ? a.w : 0.f;
                    const float s = wave_sum(a.x + a.y + a.z + a.w);
                    const float mu = s * inv_n;
                    float var;
                    if (f.keep) {                                  // masked path: var = E[x^2] / p - mu^2 (masked_layer_norm.py:38-40)
                        var = wave_sum(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w) * inv_n - mu * mu;
                    } else {                                       // F.layer_norm: two-pass variance
                        float d2 = 0.f;
                        if (cin) {
                            const float d0 = a.x - mu, d1 = a.y - mu, d2_ = a.z - mu, d3 = a.w - mu;
                            d2 = d0 * d0 + d1 * d1 + d2_ * d2_ + d3 * d3;
                        }
                        var = wave_sum(d2) * inv_n;
                    }
                    const float rs = 1.0f / sqrtf(var + f.eps);
                    if (lane == 0) { f.mean[rm.orow] = mu; f.rstd[rm.orow] = rs; }
                    if (cin) {
                        const float o0 = (0 < kl) ? lw.x * ((a.x - mu) * rs) + lb.x : 0.f;
                        const float o1 = (1 < kl) ? lw.y * ((a.y - mu) * rs) + lb.y : 0.f;
                        const float o2 = (2 < kl) ? lw.z * ((a.z - mu) * rs) + lb.z : 0.f;
                        const float o3 = (3 < kl) ? lw.w * ((a.w - mu) * rs) + lb.w : 0.f;
                        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(f.y) + (long long)rm.orow * p.N + col) =
                            make_uint2(pack_bf2(o0, o1), pack_bf2(o2, o3));
                    }
                } else {
                    // dLN/dx of dy (vr_ln_bwd): dz = dy * w; dx = (dz - (mean(dz) + z mean(z dz))) * rstd + resid
                    const float mu = rm.mu, rs = rm.rs;
                    const int kl = cin ? lk - col : 0;
                    float4 a = make_float4(a4[0], a4[1], a4[2], a4[3]), xx = sd.xv[u];
                    if (!(0 < kl)) { a.x = 0.f; xx.x = mu; }
                    if (!(1 < kl)) { a.y = 0.f; xx.y = mu; }
                    if (!(2 < kl)) { a.z = 0.f; xx.z = mu; }
                    if (!(3 < kl)) { a.w = 0.f; xx.w = mu; }
                    const float4 z = make_float4((xx.x - mu) * rs, (xx.y - mu) * rs, (xx.z - mu) * rs, (xx.w - mu) * rs);
                    gw.x += a.x * z.x; gw.y += a.y * z.y; gw.z += a.z * z.z; gw.w += a.w * z.w;
                    gb.x += a.x; gb.y += a.y; gb.z += a.z; gb.w += a.w;
                    const float4 g = make_float4(a.x * lw.x, a.y * lw.y, a.z * lw.z, a.w * lw.w);
                    const float s1 = wave_sum(g.x + g.y + g.z + g.w) * inv_n;
                    const float s2 = wave_sum(g.x * z.x + g.y * z.y + g.z * z.z + g.w * z.w) * inv_n;
                    if (cin) {
                        const int kg = rm.keep - col;
                        const float4 r4 = sd.rv[u];
                        float4 o;
                        o.x = (0 < kl) ? (g.x - (s1 + z.x * s2)) * rs + r4.x : 0.f;
                        o.y = (1 < kl) ? (g.y - (s1 + z.y * s2)) * rs + r4.y : 0.f;
                        o.z = (2 < kl) ? (g.z - (s1 + z.z * s2)) * rs + r4.z : 0.f;
                        o.w = (3 < kl) ? (g.w - (s1 + z.w * s2)) * rs + r4.w : 0.f;
                        const long long oidx = (long long)rm.orow * p.ldc + col;
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + oidx) = o;
                        if (f.gt_out) {
                            const float sc = rm.scale;
                            const float t0 = (0 < kg) ? o.x * sc : 0.f, t1 = (1 < kg) ? o.y * sc : 0.f;
                            const float t2 = (2 < kg) ? o.z * sc : 0.f, t3 = (3 < kg) ? o.w * sc : 0.f;
                            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(f.gt_out) + oidx) =
                                make_uint2(pack_bf2(t0, t1), pack_bf2(t2, t3));
                        }
                    }
                }
            }
        };
        Side sa, sb;
        request(0, sa);                        // (rowmeta is ordered by the barrier above; the loads fly over the park writes)
        __syncthreads();                       // the tile is parked
#pragma unroll 1
        for (int it = 0; it < nround; it += 2) {
            if (it + 1 < nround) request(it + 1, sb);
            work(it, sa);
            if (it + 1 < nround) {
                if (it + 2 < nround) request(it + 2, sa);
                work(it + 1, sb);
            }
        }
        __syncthreads();                       // the park area is reused: column sums below, the next tile's ring

        if constexpr (MODE == 1) {
            // LayerNorm weight / bias gradients: every wave holds partial column sums over its rows -> cross-wave sum through LDS
            float* red = reinterpret_cast<float*>(smem);               // [2][8 waves][BN]
            *reinterpret_cast<float4*>(red + (0 * NWAVE + wave) * BN + col) = gw;
            *reinterpret_cast<float4*>(red + (1 * NWAVE + wave) * BN + col) = gb;
            __syncthreads();
            const long long grow = (long long)(blockIdx.x % (unsigned)(f.grad_copies > 1 ? f.grad_copies : 1)) * p.N;
            if (t < 2 * BN) {
                const int c = t & (BN - 1), which = t >> 8;             // BN == 256
                if (c < p.N) {
                    const float* src = red + which * NWAVE * BN + c;
                    float a = 0.f;
#pragma unroll
                    for (int w8 = 0; w8 < NWAVE; ++w8) a += src[w8 * BN];
                    atomicAdd((which ? f.db : f.dw) + grow + c, a);    // partial row of this workgroup (vr_ln_bwd's grad_copies)
                }
            }
            __syncthreads();
        }
    }
}

}  // namespace vr_gemm_ntlnw

// Launches the wide form when it fits: N <= 256 and enough 16-row blocks that every CU gets a tile of at least `min_blocks`.
// false: not taken (the caller falls back to gemm_nt_ln.hip's kernel).
bool vr_gemm_lnw_launch(const vr_gemm_args& a, const vr_ln_epilogue& f, hipStream_t stream) {
    using namespace vr_gemm_ntlnw;
    static const int n_cu = [] {           // read once per process (one process per GPU)
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            return v;
        return 256;
    }();
    static const int knob = [] { const char* e = getenv("VITRES_NTLN_WIDE"); return e ? atoi(e) : 0; }();   // 0: only when forced (default); 1: by size; n > 1: min blocks
    const bool force = (a.sched & 8) != 0;                         // vr_gemm_args.sched: 8 = this form whenever it applies, 16 = never
    if (a.N > BN || n_cu <= 0 || (a.sched & 16) || (knob <= 0 && !force)) return false;
    const int nblk = (a.M + 15) / 16;
    const int min_blocks = knob > 1 ? knob : 4;                    // below ~4 blocks (64 rows) per CU the 64 x 256 form's co-residency wins
    if (!force && (nblk < (long long)min_blocks * n_cu || a.K < 8 * BK)) return false;   // (K = 256: 39.9 against 33.5 us)
    const int rounds = (nblk + MIX * n_cu - 1) / (MIX * n_cu);
    const int ntile = min(nblk, n_cu * rounds);                    // every tile gets nblk / ntile (+1) blocks <= MIX
    const bool ktail = (a.K % BK) != 0;
    const dim3 grid((unsigned)min(ntile, n_cu)), block(NTHR);
    if (f.mode == 0) {
        if (ktail) hipLaunchKernelGGL((ntlnw_kernel<0, true>), grid, block, 0, stream, a, f, nblk, ntile);
        else hipLaunchKernelGGL((ntlnw_kernel<0, false>), grid, block, 0, stream, a, f, nblk, ntile);
    } else {
        if (ktail) hipLaunchKernelGGL((ntlnw_kernel<1, true>), grid, block, 0, stream, a, f, nblk, ntile);
        else hipLaunchKernelGGL((ntlnw_kernel<1, false>), grid, block, 0, stream, a, f, nblk, ntile);
    }
    return true;
}
