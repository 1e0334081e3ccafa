// Panel-resident bf16 GEMM for the narrow first stage:  C[M,N] = epilogue(A[M,K] * B[N,K]^T),  K = C <= 320
//
// The qkv / fc1 forward Linears of a stage-1 block and the fc2 / proj data gradients (reference nets/supernet_blocks.py:37-52,102:
// F.linear on 32 896 token rows with 256 input channels) move 67 - 117 MB each and multiply for 4 - 5 K slices only: on the tiled
// kernels (gemm_ntk.hip: 1 542 tiles of 128 x 128) a tile is all prologue and epilogue, the A rows are fetched once per N tile
// and every workgroup pays the first-slice HBM latency -- 2.0 - 3.0 TB/s of algorithmic bytes (profiles/r05_launch_table.txt).
// Here a workgroup OWNS a panel of BM = 16 MI token rows:
//
//   * the A panel ([BM][K] bf16, 72 KB at BM = 144, K = 256) is brought into LDS once by LDS-DMA (gemm_ntk.hip's row image:
//     128-byte rows per 64-wide K slice, XOR-swizzled 16-byte slots) -- one exposed HBM latency per panel, one barrier;
//   * after that barrier the waves never synchronise again: wave w walks the 32-column strips w, w + NW, ... of the output.  The
//     weight strip ([32 columns][K] bf16 = 16 KB at K = 256, L2-resident: the whole weight is <= 0.5 MB) lives in REGISTERS --
//     a lane's 16 bytes of W[n][k .. k + 8] are exactly its MFMA operand -- and is refreshed IN PLACE: as soon as the MFMAs of k-step
//     s have read their operand, the same registers are re-loaded with k-step s of the wave's NEXT strip, a whole strip (8 k-steps
//     x MI x 2 MFMAs) ahead of their use.  No weight bytes cross LDS, no s_waitcnt is written by hand: every load is a plain
//     global load whose wait the compiler counts;
//   * a strip's epilogue (gemm_nt_parts.h: bias, GELU pair, saved-gelu' multiply, prefix masks, write skipping) stores while the
//     loads of the next strip are in flight; its stores are never waited for.
//
// Panels: ceil(rows / BM) per architecture group (gemm_shared.h group_tile_rows: a panel never mixes two groups).  BM = 144 with 8
// waves (one workgroup per CU, two waves per SIMD) puts 32 896 rows on 230 of 256 CUs in ONE round (257 = 32 896 / 128 is prime:
// every 64- or 128-row tiling leaves a second round for 2 of 514 / 1 of 257 tiles); BM = 80 with 4 waves (two workgroups per CU)
// serves the half-size batches (16 448 rows: 206 panels).
//
// Covered (vr_gemm_panel_launch returns false otherwise): bf16 operands and result, K-contiguous weight (data gradients read the
// transposed bf16 shadow), K % 32 == 0 <= 320, k_period == 0, no residual / scale / pos / row maps, FAST epilogue alignment.
// PROBE (sched 0x400000, measurement only): the same kernel without weight loads, LDS reads and MFMAs -- what the access pattern
// (A once, kept C strips once) costs by itself; 0x1000000 / 0x2000000: without the weight loads / without LDS reads and MFMAs
// (tools/panel_bench.py).
#include <algorithm>

#include "gemm_nt_parts.h"
#include "lds_dma.h"

namespace vr_gemm_nt {

using vr_dma::dma16;
using vr_dma::make_rsrc;

// PROBE: 0 = the kernel; 1 = bytes only (no weight loads, LDS reads, MFMAs); 2 = without the weight loads; 3 = without LDS reads / MFMAs
template <typename TO, int EPI, int MI, int NW, int KSTEPS, int FEAT, int PROBE>
__global__ __launch_bounds__(NW * 64, MI >= 8 ? 1 : 2) void panel_kernel(const vr_gemm_args p) {
    constexpr int NJ = 2, BM = 16 * MI, NTHR = NW * 64;
    constexpr int KSL = (KSTEPS + 1) / 2;                         // 64-wide K slices of the panel image
    constexpr int A_SLICE = BM * 128, A_BYTES = KSL * A_SLICE;
    constexpr int PARK = 16 * 16 * NJ * 4;                        // per wave: 16 rows x 32 columns of floats
    constexpr int META_OFF = A_BYTES + NW * PARK;
    __shared__ __attribute__((aligned(1024))) char smem[META_OFF + BM * (int)sizeof(RowMeta)];
    RowMeta* rowmeta = reinterpret_cast<RowMeta*>(smem + META_OFF);
    constexpr bool SKIP_FORM = sizeof(TO) == 2 && ((EPI == EPI_STORE && FEAT <= 1) || EPI == EPI_GELU || EPI == EPI_DMUL);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int total = group_tiles(p.M, BM, p.m_groups);
    int panel = blockIdx.x;
    if (total >= 16) {                  // XCD x owns one contiguous run of panels (rows a neighbouring kernel's XCD x wrote / will read)
        const int xq = total >> 3, xr = total & 7, x = panel & 7;
        panel = x * xq + min(x, xr) + (panel >> 3);
    }
    int m0, mend;
    group_tile_rows(panel, p.M, BM, p.m_groups, m0, mend);

    // ---- masks of the panel's samples: kept K prefix (k-steps multiplied), kept / group output widths ----
    int nks = p.K / 32;
    int nmax = 1 << 30, gmax = 1 << 30;
    {
        int s_lo = 0, s_hi = 0;
        if (p.rows_in > 0) { s_lo = m0 / p.rows_in; s_hi = (min(m0 + BM, mend) - 1) / p.rows_in; }
        if (p.keep_k) {
            int kmax = 0;
            for (int s = s_lo; s <= s_hi; ++s) kmax = max(kmax, p.keep_k[s]);      // (negative marks read as 0)
            nks = min(nks, (kmax + 31) / 32);
        }
        if (p.keep_n) {
            nmax = 0; gmax = 0;
            for (int s = s_lo; s <= s_hi; ++s) {
                const int v = p.keep_n[s];
                nmax = max(nmax, v);
                gmax = max(gmax, v < 0 ? -v - 2 : v);
            }
        }
    }
    // ---- the A panel: live 64-wide slices by LDS-DMA, pieces of 8 rows dealt to the waves ----
    const int nsl = (nks + 1) / 2;
    {
        const vr_dma::v4i rsA = make_rsrc(p.A, 0xffffff00u);
        const unsigned lds0 = vr_dma::lds_addr(smem);
        constexpr int RB = BM / 8;
        const int npieces = RB * nsl;
        for (int q = wave; q < npieces; q += NW) {
            const int sl = q / RB, rb = q - sl * RB;
            const int r = rb * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            const int ma = min(m0 + r, mend - 1);
            const unsigned voff = (unsigned)(((long long)ma * p.lda + c * 8) * 2);
            dma16(lds0 + sl * A_SLICE + rb * 1024, voff, rsA, sl * 128);
        }
    }
    for (int r = t; r < BM; r += NTHR) {      // per-row epilogue metadata (its loads overlap the panel's)
        const int m = m0 + r;
        RowMeta rm;
        rm.keep = 1 << 30; rm.scale = 1.0f; rm.orow = -1; rm.mloc = 0;
        if (m < mend) {
            const int sample = p.rows_in > 0 ? m / p.rows_in : 0;
            rm.mloc = p.rows_in > 0 ? m - sample * p.rows_in : m;
            rm.orow = m;
            if (p.keep_n) rm.keep = p.keep_n[sample];
        }
        rowmeta[r] = rm;
    }
    // ---- this wave's strips: w, w + NW, ... (at most 16 of them: N <= 32 x 16 x NW) ----
    const int nstrips = (p.N + 31) / 32;
    const bool skip_ok = SKIP_FORM && (p.sched & 0x40000) && p.keep_n && (p.m_groups <= 1 || group_pure(p.M, p.m_groups)) &&
                         (p.n_period <= 0 || p.n_period % 64 == 0);
    // bit i of `wr`: the wave's i-th strip is written (otherwise every reader stays below the group's 64-wide slices and the strip is
    // left alone); bit i of `ml`: it is multiplied (otherwise: an epilogue on a zero accumulator -- zeros / bias under the masks)
    unsigned wr = 0, ml = 0;
    {
        int i = 0;
        for (int s = wave; s < nstrips; s += NW, ++i) {
            const int n0 = s * 32;
            bool w = true, m = nks > 0;
            if (p.keep_n) {
                if (skip_ok && !range_has_kept(n0 & ~63, 64, p.n_period, gmax)) w = false;
                m = m && range_has_kept(n0, 32, p.n_period, nmax);
            }
            wr |= (w ? 1u : 0u) << i;
            ml |= ((w && m) ? 1u : 0u) << i;
        }
    }

    const int frow = lane & 15, fswz = (frow >> 1) & 7, g4 = lane >> 4;
    const char* As = smem + frow * 128;
    const int slot_e = ((g4) ^ fswz) << 4, slot_o = ((4 + g4) ^ fswz) << 4;
    float* park = reinterpret_cast<float*>(smem + A_BYTES + wave * PARK);

    // The weight strip in registers.  The loads are inline asm, hidden from the compiler's wait bookkeeping: it would put a
    // conservative s_waitcnt vmcnt(0 / 1) in front of a strip's first MFMA -- i.e. wait for the previous strip's STORES.  Instead
    // ONE s_waitcnt vmcnt(0) sits in front of a strip's first store (epilogue WAITV): the next strip's operands have been requested
    // one strip of MFMAs earlier, the previous strip's stores two; the stores issued behind it are waited for by nobody until the
    // next strip's epilogue.  k-steps are requested in PAIRS (2 s, 2 s + 1: the two 64-byte halves of the same 128-byte lines of
    // W[n][.]) -- a half requested 18 MFMAs after the other found its line evicted from the 32 KB L1 by the seven other waves.
    bfv8 bw[KSTEPS][NJ];
    const bf16_t* bptr[NJ];
    auto point = [&](int i) {                                     // operand rows of the wave's i-th strip
        const int s = wave + i * NW;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = min(s * 32 + 16 * j + frow, p.N - 1);
            bptr[j] = reinterpret_cast<const bf16_t*>(p.B) + (long long)n * p.ldb + g4 * 8;
        }
    };
    auto load_step = [&](int ks) {                                // k-step ks of the pointed strip -> this lane's MFMA operands
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bw[ks][j]) : "v"(bptr[j] + ks * 32) : "memory");
    };
    int ci = wr ? __builtin_ctz(wr) : 32;                         // index of the strip being worked on
    if constexpr (PROBE == 0 || PROBE == 3) {
        if (ci < 32 && ((ml >> ci) & 1u)) {                       // the first strip's operands travel with the panel
            point(ci);
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks)
                if (ks < nks) load_step(ks);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    while (ci < 32) {
        const unsigned rest = wr & ~((2u << ci) - 1u);
        const int ni = rest ? __builtin_ctz(rest) : 32;
        const bool mul = (ml >> ci) & 1u;
        const bool pre = ni < 32 && ((ml >> ni) & 1u);
        if (pre) point(ni);
        f32x4 acc[MI][NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (PROBE != 1) {
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                if (ks < nks) {
                    if constexpr (PROBE != 3) {
                        if (mul) {
                            const char* Ab = As + (ks >> 1) * A_SLICE + ((ks & 1) ? slot_o : slot_e);
#pragma unroll
                            for (int i = 0; i < MI; ++i) {
                                const bfv8 a = *reinterpret_cast<const bfv8*>(Ab + i * 2048);
#pragma unroll
                                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[ks][j], a, acc[i][j], 0, 0, 0);
                            }
                        }
                    }
                    if constexpr (PROBE == 3) {                  // (keeps the operand registers allocated until here, like the MFMAs do)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(bw[ks][j]));
                    }
                    if constexpr (PROBE != 2) {
                        // the registers just read take the next strip's k-steps: both halves of a line in one go
                        if (pre && ((ks & 1) || ks + 1 >= nks)) {
                            if (ks & 1) load_step(ks - 1);
                            load_step(ks);
                        }
                    }
                }
            }
        }
        epilogue<TO, EPI, true, MI, NJ, FEAT, 1, true>(p, acc, park, rowmeta, (wave + ci * NW) * 32, lane);
        ci = ni;
    }
}

template <typename TO, int EPI, int FEAT> bool plaunch(const vr_gemm_args& a, hipStream_t stream, int n_cu) {
    const int probe = (a.sched & 0x400000) ? 1 : ((a.sched & 0x1000000) ? 2 : ((a.sched & 0x2000000) ? 3 : 0));   // measurement forms
    // panel height: 144 rows x 8 waves (one workgroup per CU) when that fills >= 3/4 of the chip in one round, else 80 rows x 4
    // waves (two per CU) when those fit in one round; otherwise the tiled kernels are at least as good
    const int p144 = group_tiles(a.M, 144, a.m_groups), p80 = group_tiles(a.M, 80, a.m_groups);
    const bool k8 = a.K <= 256;
    int form = 0;
    if (p144 <= n_cu && 4 * p144 >= 3 * n_cu && k8) form = 9;
    else if (p80 <= 2 * n_cu && 2 * p80 >= n_cu) form = 5;
    if (a.sched & 0x200000) form = (k8 && !(a.sched & 0x800000)) ? 9 : 5;              // forced (tests): 0x800000 picks the 80-row form
    if (!form) return false;
    if (form == 9) {
        switch (probe) {
            case 1: hipLaunchKernelGGL((panel_kernel<TO, EPI, 9, 8, 8, FEAT, 1>), dim3(p144), dim3(512), 0, stream, a); break;
            case 2: hipLaunchKernelGGL((panel_kernel<TO, EPI, 9, 8, 8, FEAT, 2>), dim3(p144), dim3(512), 0, stream, a); break;
            case 3: hipLaunchKernelGGL((panel_kernel<TO, EPI, 9, 8, 8, FEAT, 3>), dim3(p144), dim3(512), 0, stream, a); break;
            default: hipLaunchKernelGGL((panel_kernel<TO, EPI, 9, 8, 8, FEAT, 0>), dim3(p144), dim3(512), 0, stream, a);
        }
    } else {
        switch (probe) {
            case 1: hipLaunchKernelGGL((panel_kernel<TO, EPI, 5, 4, 10, FEAT, 1>), dim3(p80), dim3(256), 0, stream, a); break;
            default: hipLaunchKernelGGL((panel_kernel<TO, EPI, 5, 4, 10, FEAT, 0>), dim3(p80), dim3(256), 0, stream, a);
        }
    }
    return true;
}

}  // namespace vr_gemm_nt

// Called by vr_gemm_ntk_launch in front of the tiled kernels.  OPT-IN (sched 0x200000: wherever the form is covered): measured round 6
// (profiles/r06_panel_resident.txt) the access pattern alone -- A once, C once -- streams at 4.5 - 5.6 TB/s, but the kernel reaches
// 2.4 - 3.6 TB/s against the tiled kernels' 3.0 - 4.5 alone and costs the sr_tiny step +0.25 ms: the weight strips (384 KB per panel
// from L2 as 16-row x 64-byte pieces) arrive at ~40 GB/s per CU and add to, instead of hiding behind, the MFMA phase of the one
// workgroup a CU holds.
bool vr_gemm_panel_launch(const vr_gemm_args& a, hipStream_t stream, int n_cu) {
    using namespace vr_gemm_nt;
    if (!(a.sched & 0x200000)) return false;
    if (a.in_dtype != VR_BF16 || a.out_dtype != VR_BF16 || a.a_trans || a.b_trans || a.atomic || a.split_k > 1 || a.bias_grad || a.pos ||
        a.resid || a.scale)
        return false;
    if (a.K % 32 || a.K > 320 || a.K < 32 || a.k_period > 0) return false;
    if (a.a_map.rpi || a.b_map.rpi || a.c_map.rpi) return false;
    const bool fast = a.N % 8 == 0 && a.ldc % 8 == 0 && (!a.dact_u || a.ldu % 8 == 0) && (a.n_period <= 0 || a.n_period % 8 == 0);
    if (!fast || a.lda % 8 || a.ldb % 8 || ((uintptr_t)a.A & 15) || ((uintptr_t)a.B & 15)) return false;
    if ((long long)a.M * a.lda * 2 >= 0xfff00000LL) return false;
    // an operand with unwritten masked tiles is readable only when no panel mixes two architecture groups
    if ((a.sched & 0x80000) && a.keep_k && a.m_groups > 1 && !group_pure(a.M, a.m_groups)) return false;
    const bool gelu = a.act == 1 || (a.act == 2 && !a.dact_u);
    if (a.act == 3) return false;
    if (a.dact_u) {
        if (a.act != 2 || a.bias) return false;
        return plaunch<bf16_t, EPI_DMUL, 0>(a, stream, n_cu);
    }
    if (gelu) {
        if (!a.bias) return false;
        return plaunch<bf16_t, EPI_GELU, 1>(a, stream, n_cu);
    }
    if (a.bias) return plaunch<bf16_t, EPI_STORE, 1>(a, stream, n_cu);
    return plaunch<bf16_t, EPI_STORE, 0>(a, stream, n_cu);
}
