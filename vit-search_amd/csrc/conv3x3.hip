// Direct 3x3 / stride 1 / pad 1 convolution for the conv patch embedding (reference nets/patch_conv.py:23-73: conv2, conv3 and,
// with flipped weights, their data gradients): NHWC bf16 activations [B,H,W,Cin] (Cin = 24 / 32: a few channels on 112 x 112
// maps), weights [Cout, (kh, kw, ci)] bf16, output [B*H*W, Cout] fp32 or bf16.
//
// The im2col + GEMM form wrote and re-read a 9x larger matrix (694 MB per convolution at B = 128, m = 24) around 8 GFLOP of
// work; here a workgroup stages the 18 x 18 pixel halo patch of its 16 x 16 output tile in LDS once (1.27x the input bytes,
// mostly L2 hits) and the MFMA A fragments are gathered straight from it: contraction index k = (tap, channel chunk), a lane's
// 16-byte chunk = 8 channels of ONE shifted pixel, so "im2col" is just the per-lane LDS address.  HBM traffic = input + output.
//   MFMA 16x16x32: rows = 16 consecutive pixels of a tile row, 4 lane groups = 4 consecutive (tap, chunk) slots,
//   computed transposed (weights first): a lane owns one pixel and 4 consecutive output channels -> 16-byte stores.
// The pixel stride in LDS is Cin*2 bytes, padded by 16 when Cin/8 is even, so the 16 pixels of a fragment read hit 16
// different 16-byte slots.
#include "common.h"
#include "../../include/vitres_hip.h"

namespace {

typedef __bf16 bfv8 __attribute__((ext_vector_type(8)));
constexpr int TH = 16, TW = 16, PH = TH + 2, PW = TW + 2;

template <int NC, typename TO>
__global__ __launch_bounds__(256) void conv3x3_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ w,
                                                      TO* __restrict__ out, int B, int H, int W, int Cout,
                                                      const float* __restrict__ bias, const bf16_t* __restrict__ res, int psz, int rpsz) {
    constexpr int Cin = 8 * NC, K9 = 9 * NC, STEPS = (K9 + 3) / 4;
    constexpr int PS = Cin * 2 + ((NC & 1) ? 0 : 16);                 // pixel stride in LDS, bytes
    __shared__ __attribute__((aligned(16))) char patch[PH * PW * PS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y, b = blockIdx.x / (tiles_x * tiles_y);
    const int y0 = ty * TH, x0 = tx * TW;
    // ---- halo patch -> LDS (all loads first, zero outside the image) ----
    constexpr int NCHUNK = PH * PW * NC, IT = (NCHUNK + 255) / 256;
    uint4 v[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int idx = tid + it * 256;
        const int pix = idx / NC, cc = idx % NC;
        const int iy = y0 + pix / PW - 1, ix = x0 + pix % PW - 1;
        const bool ok = idx < NCHUNK && iy >= 0 && iy < H && ix >= 0 && ix < W;
        v[it] = *reinterpret_cast<const uint4*>(a + (((long long)b * H + (ok ? iy : 0)) * W + (ok ? ix : 0)) * Cin + cc * 8);
        if (!ok) v[it] = make_uint4(0, 0, 0, 0);
    }
    // ---- weights -> registers: fragment (step, n-tile): row co = 16 nt + c, chunk 4 step + g ----
    bfv8 wf[STEPS][2];
#pragma unroll
    for (int ks = 0; ks < STEPS; ++ks)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int chunk = 4 * ks + g, co = 16 * nt + c;
            const bool ok = chunk < K9 && co < Cout;
            uint4 x = *reinterpret_cast<const uint4*>(w + (long long)(ok ? co : 0) * (9 * Cin) + (ok ? chunk : 0) * 8);
            if (!ok) x = make_uint4(0, 0, 0, 0);
            wf[ks][nt] = __builtin_bit_cast(bfv8, x);
        }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int idx = tid + it * 256;
        if (idx < NCHUNK) *reinterpret_cast<uint4*>(patch + (idx / NC) * PS + (idx % NC) * 16) = v[it];
    }
    // per-lane LDS offset of contraction slot (step, g): tap (kh, kw), channel chunk cc; padded slots reuse slot 0 (zero weights)
    int koff[STEPS];
#pragma unroll
    for (int ks = 0; ks < STEPS; ++ks) {
        int chunk = 4 * ks + g;
        chunk = chunk < K9 ? chunk : 0;
        const int tap = chunk / NC, cc = chunk % NC;
        koff[ks] = ((tap / 3) * PW + (tap % 3)) * PS + cc * 16;
    }
    __syncthreads();
    const bool nt1 = Cout > 16;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int r = wave * 4 + mt;                                   // tile row; the fragment's 16 pixels are its 16 columns
        const char* base = patch + (r * PW + c) * PS;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < STEPS; ++ks) {
            const bfv8 af = *reinterpret_cast<const bfv8*>(base + koff[ks]);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][0], af, acc0, 0, 0, 0);
            if (nt1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][1], af, acc1, 0, 0, 0);
        }
        const int oy = y0 + r, ox = x0 + c;
        if (oy < H && ox < W) {
            // psz > 0: the output goes out as the patchify operand of the projection that follows ([B*gh*gw, (i, j, c)], non-
            // overlapping patch x patch windows: vr_patch_unfold's layout) -- a pixel's Cout channels stay contiguous either way
            long long opix = ((long long)b * H + oy) * W + ox;
            if (psz > 0)
                opix = (((long long)b * (H / psz) + oy / psz) * (W / psz) + ox / psz) * (psz * psz) + (oy % psz) * psz + ox % psz;
            TO* dst = out + opix * Cout;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int co = 16 * nt + 4 * g;
                if (co < Cout) {
                    f32x4 r4 = nt == 0 ? acc0 : acc1;
                    if (bias) {      // evaluation: BatchNorm folded into the weights, out = relu(conv + shift) (+ residual)
                        const float4 bb = *reinterpret_cast<const float4*>(bias + co);
                        r4[0] = fmaxf(r4[0] + bb.x, 0.f); r4[1] = fmaxf(r4[1] + bb.y, 0.f);
                        r4[2] = fmaxf(r4[2] + bb.z, 0.f); r4[3] = fmaxf(r4[3] + bb.w, 0.f);
                    }
                    if (res) {       // residual (evaluation: behind the ReLU; data gradients: the gradient of the stem's skip branch)
                        long long rpix = ((long long)b * H + oy) * W + ox;         // rpsz > 0: the residual is stored in patch order
                        if (rpsz > 0)
                            rpix = (((long long)b * (H / rpsz) + oy / rpsz) * (W / rpsz) + ox / rpsz) * (rpsz * rpsz) + (oy % rpsz) * rpsz + ox % rpsz;
                        const uint2 rr = *reinterpret_cast<const uint2*>(res + rpix * Cout + co);
                        r4[0] += __uint_as_float(rr.x << 16); r4[1] += __uint_as_float(rr.x & 0xffff0000u);
                        r4[2] += __uint_as_float(rr.y << 16); r4[3] += __uint_as_float(rr.y & 0xffff0000u);
                    }
                    if constexpr (sizeof(TO) == 4) *reinterpret_cast<float4*>(dst + co) = make_float4(r4[0], r4[1], r4[2], r4[3]);
                    else *reinterpret_cast<uint2*>(dst + co) = make_uint2(pack_bf2(r4[0], r4[1]), pack_bf2(r4[2], r4[3]));
                }
            }
        }
    }
}

// ---- conv1 of the patch embedding: 3x3 / stride 2 / pad 1 from the fp32 NCHW image (3 channels) -> NHWC [B*Ho*Wo, Cout] --------
// (reference nets/patch_conv.py:63: the first convolution).  The gather + GEMM form wrote a [B*Ho*Wo, 32] bf16 im2col matrix
// (205 MB at B = 256) with a one-element-per-thread kernel (0.6 TB/s) and read it back; here a workgroup stages the 17 x 65 x 3
// input patch of its 8 x 32 output tile in LDS as bf16 and gathers the MFMA operand from it: contraction index k = (kh, kw, c),
// 27 of the 32 slots (the weights [Cout, 32] are zero in the rest).  Weights first, as above: a lane owns one pixel and four
// consecutive output channels.  Optional epilogue (evaluation, BatchNorm folded into w): relu(. + bias).
constexpr int C1_TH = 8, C1_TW = 32, C1_IH = 2 * C1_TH + 1, C1_IW = 2 * C1_TW + 1, C1_IWP = C1_IW + 1;

template <typename TO>
__global__ __launch_bounds__(256) void conv1_direct_kernel(const float* __restrict__ img, const bf16_t* __restrict__ w,
                                                           const float* __restrict__ bias, TO* __restrict__ out, int B, int H, int W,
                                                           int Ho, int Wo, int Cout, int relu) {
    __shared__ bf16_t patch[3 * C1_IH * C1_IWP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int tiles_x = (Wo + C1_TW - 1) / C1_TW, tiles_y = (Ho + C1_TH - 1) / C1_TH;
    const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y, b = blockIdx.x / (tiles_x * tiles_y);
    const int oy0 = ty * C1_TH, ox0 = tx * C1_TW;
    const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;
    const float* src = img + (long long)b * 3 * H * W;
    for (int idx = tid; idx < 3 * C1_IH * C1_IW; idx += 256) {
        const int ch = idx / (C1_IH * C1_IW), rem = idx - ch * (C1_IH * C1_IW);
        const int y = rem / C1_IW, x = rem - y * C1_IW;
        const int iy = iy0 + y, ix = ix0 + x;
        const float v = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? src[((long long)ch * H + iy) * W + ix] : 0.f;
        patch[(ch * C1_IH + y) * C1_IWP + x] = f2bf(v);
    }
    // weight fragments: channel 16 nf + c, k = 8 g .. 8 g + 7
    bfv8 wf[2];
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) {
        const int ch = 16 * nf + c;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ch < Cout) v = *reinterpret_cast<const uint4*>(w + (long long)ch * 32 + 8 * g);
        wf[nf] = __builtin_bit_cast(bfv8, v);
    }
    // LDS offsets of this lane's 8 contraction slots relative to the pixel's window origin
    int off[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * g + j;
        const int tap = k / 3, ch = k - 3 * tap, kh = tap / 3, kw = tap - 3 * kh;
        off[j] = k < 27 ? (ch * C1_IH + kh) * C1_IWP + kw : 0;     // slots 27..31 meet zero weights: any finite value will do
    }
    __syncthreads();
    const bool two = Cout > 16;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int oyl = 2 * wave + rr;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int oxl = 16 * f + c;
            const bf16_t* base = patch + (2 * oyl) * C1_IWP + 2 * oxl;
            unsigned short e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = base[off[j]];
            const uint4 av = make_uint4((unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16),
                                        (unsigned)e[4] | ((unsigned)e[5] << 16), (unsigned)e[6] | ((unsigned)e[7] << 16));
            const bfv8 af = __builtin_bit_cast(bfv8, av);
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0], af, acc0, 0, 0, 0);
            if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1], af, acc1, 0, 0, 0);
            const int oy = oy0 + oyl, ox = ox0 + oxl;
            if (oy < Ho && ox < Wo) {
                TO* dst = out + (((long long)b * Ho + oy) * Wo + ox) * Cout;
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) {
                    const int co = 16 * nf + 4 * g;
                    if (co < Cout) {
                        f32x4 r4 = nf == 0 ? acc0 : acc1;
                        if (bias) {
                            const float4 bb = *reinterpret_cast<const float4*>(bias + co);
                            r4[0] += bb.x; r4[1] += bb.y; r4[2] += bb.z; r4[3] += bb.w;
                        }
                        if (relu) { r4[0] = fmaxf(r4[0], 0.f); r4[1] = fmaxf(r4[1], 0.f); r4[2] = fmaxf(r4[2], 0.f); r4[3] = fmaxf(r4[3], 0.f); }
                        if constexpr (sizeof(TO) == 4) *reinterpret_cast<float4*>(dst + co) = make_float4(r4[0], r4[1], r4[2], r4[3]);
                        else *reinterpret_cast<uint2*>(dst + co) = make_uint2(pack_bf2(r4[0], r4[1]), pack_bf2(r4[2], r4[3]));
                    }
                }
            }
        }
    }
}

// ---- weight gradient:  dW[co, (kh, kw, ci)] += sum over pixels of dz[p, co] * a[p + (kh-1, kw-1), ci] --------------------
// Same tiles and halo patch; the contraction now runs over the 256 pixels of a tile, so BOTH MFMA operands are needed
// "pixel-contiguous" while LDS holds [pixel][channel]: both come from transposing reads (ds_read_b64_tr_b16, as gemm_tn.hip), the
// B operand with the tap's shift folded into each lane's address.  Output tile = [Cout <= 32] x [9 taps x ceil(Cin/16) blocks of
// 16 channels]; the 4 waves split the column blocks, accumulate over ALL tiles the (persistent) workgroup visits and add their
// part of dW once at the end (fp32 atomics: 21 MB for 1024 workgroups instead of a 694 MB im2col matrix read by a GEMM).
typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s4v lds_s4v;

__device__ __forceinline__ bfv8 tr8(const char* p0, const char* p1) {
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)p0);
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)p1);
    const s8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bfv8, v);
}

template <int NC, int NCO /* Cout / 8 */>
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ dz,
                                                            float* __restrict__ dw, int B, int H, int W, int ntiles_total) {
    constexpr int Cin = 8 * NC, Cout = 8 * NCO, CB = (Cin + 15) / 16, NT = 9 * CB, MT = (Cout + 15) / 16;
    constexpr int PS = Cin * 2 + ((NC & 1) ? 0 : 16), DS = Cout * 2 + ((NCO & 1) ? 0 : 16);
    constexpr int NTW = (NT + 3) / 4;                                  // column blocks per wave
    __shared__ __attribute__((aligned(16))) char patch[PH * PW * PS + 64];
    __shared__ __attribute__((aligned(16))) char dzt[TH * TW * DS + 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    f32x4 acc[MT][NTW];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    // lane geometry of a transposing read inside a 32-pixel k-step: pixel 8 g + (i >> 2) (+4), i.e. tile row +(g >> 1),
    // column 8 (g & 1) + (i >> 2) (+4); channel quad 4 (i & 3)
    const int prow = g >> 1, pcol = 8 * (g & 1) + (i >> 2);
    const int dz_lane = (prow * TW + pcol) * DS + 4 * (i & 3) * 2;
    const int pa_lane = (prow * PW + pcol) * PS + 4 * (i & 3) * 2;
    for (int tile = blockIdx.x; tile < ntiles_total; tile += gridDim.x) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const int y0 = ty * TH, x0 = tx * TW;
        {   // halo patch of the activation and the dz tile -> LDS (loads first)
            constexpr int NCHUNK = PH * PW * NC, IT = (NCHUNK + 255) / 256;
            constexpr int NDZ = TH * TW * NCO, ITD = (NDZ + 255) / 256;
            uint4 v[IT], d[ITD];
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int idx = tid + it * 256;
                const int pix = idx / NC, cc = idx % NC;
                const int iy = y0 + pix / PW - 1, ix = x0 + pix % PW - 1;
                const bool ok = idx < NCHUNK && iy >= 0 && iy < H && ix >= 0 && ix < W;
                v[it] = *reinterpret_cast<const uint4*>(a + (((long long)b * H + (ok ? iy : 0)) * W + (ok ? ix : 0)) * Cin + cc * 8);
                if (!ok) v[it] = make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int it = 0; it < ITD; ++it) {
                const int idx = tid + it * 256;
                const int pix = idx / NCO, cc = idx % NCO;
                const int iy = y0 + pix / TW, ix = x0 + pix % TW;
                const bool ok = idx < NDZ && iy < H && ix < W;
                d[it] = *reinterpret_cast<const uint4*>(dz + (((long long)b * H + (ok ? iy : 0)) * W + (ok ? ix : 0)) * Cout + cc * 8);
                if (!ok) d[it] = make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int idx = tid + it * 256;
                if (idx < NCHUNK) *reinterpret_cast<uint4*>(patch + (idx / NC) * PS + (idx % NC) * 16) = v[it];
            }
#pragma unroll
            for (int it = 0; it < ITD; ++it) {
                const int idx = tid + it * 256;
                if (idx < NDZ) *reinterpret_cast<uint4*>(dzt + (idx / NCO) * DS + (idx % NCO) * 16) = d[it];
            }
        }
        __syncthreads();
#pragma unroll 2
        for (int s = 0; s < 8; ++s) {                                  // 32 pixels (2 tile rows) per step
            bfv8 af[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const char* q = dzt + dz_lane + (2 * s * TW) * DS + 16 * m * 2;
                af[m] = tr8(q, q + 4 * DS);
            }
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                const int nt = wave + 4 * n;                           // column block (tap, cb)
                if (nt < NT) {
                    const int tap = nt / CB, cb = nt % CB;
                    const char* q = patch + pa_lane + ((2 * s + tap / 3) * PW + tap % 3) * PS + 16 * cb * 2;
                    const bfv8 bf = tr8(q, q + 4 * PS);
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bf, acc[m][n], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // ---- dW += this workgroup's partial: lane holds rows co = 16 m + 4 g + r, column ci = 16 cb + i of block (tap, cb) ----
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        const int nt = wave + 4 * n;
        if (nt >= NT) continue;
        const int tap = nt / CB, ci = 16 * (nt % CB) + i;
        if (ci >= Cin) continue;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 16 * m + 4 * g + r;
                if (co < Cout) atomicAdd(dw + (long long)co * (9 * Cin) + tap * Cin + ci, acc[m][n][r]);
            }
    }
}

template <int NC> int launch(const bf16_t* a, const bf16_t* w, void* out, int B, int H, int W, int Cout, int out_dtype, hipStream_t st,
                             const float* bias = nullptr, const bf16_t* res = nullptr, int patch = 0, int rpatch = 0) {
    const unsigned grid = (unsigned)(B * ((H + TH - 1) / TH) * ((W + TW - 1) / TW));
    if (out_dtype == VR_F32) hipLaunchKernelGGL((conv3x3_kernel<NC, float>), dim3(grid), dim3(256), 0, st, a, w, (float*)out, B, H, W, Cout, bias, res, patch, rpatch);
    else hipLaunchKernelGGL((conv3x3_kernel<NC, bf16_t>), dim3(grid), dim3(256), 0, st, a, w, (bf16_t*)out, B, H, W, Cout, bias, res, patch, rpatch);
    return 0;
}

}  // namespace

extern "C" int vr_conv3x3_wgrad(const void* a, const void* dz, float* dw, int32_t B, int32_t H, int32_t W, int32_t Cin,
                                int32_t Cout, vr_stream_t stream) {
    if (!a || !dz || !dw || B <= 0 || H <= 0 || W <= 0) return VR_EINVAL;
    if (((uintptr_t)a & 15) || ((uintptr_t)dz & 15)) return VR_EALIGN;
    const int tiles = B * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    const unsigned grid = (unsigned)(tiles < 1024 ? tiles : 1024);
    hipStream_t st = (hipStream_t)stream;
    const bf16_t* pa = (const bf16_t*)a;
    const bf16_t* pd = (const bf16_t*)dz;
    if (Cin == 24 && Cout == 24) hipLaunchKernelGGL((conv3x3_wgrad_kernel<3, 3>), dim3(grid), dim3(256), 0, st, pa, pd, dw, B, H, W, tiles);
    else if (Cin == 32 && Cout == 32) hipLaunchKernelGGL((conv3x3_wgrad_kernel<4, 4>), dim3(grid), dim3(256), 0, st, pa, pd, dw, B, H, W, tiles);
    else if (Cin == 16 && Cout == 16) hipLaunchKernelGGL((conv3x3_wgrad_kernel<2, 2>), dim3(grid), dim3(256), 0, st, pa, pd, dw, B, H, W, tiles);
    else return VR_EUNSUPPORTED;
    VR_CHECK_LAUNCH();
    return VR_OK;
}

static int conv3x3_entry(const void* a, const void* w, void* out, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                         int32_t out_dtype, const float* bias, const void* res, vr_stream_t stream, int patch = 0, int rpatch = 0);

extern "C" int vr_conv3x3(const void* a, const void* w, void* out, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                          int32_t out_dtype, vr_stream_t stream) {
    return conv3x3_entry(a, w, out, B, H, W, Cin, Cout, out_dtype, nullptr, nullptr, stream);
}

extern "C" int vr_conv3x3_res(const void* a, const void* w, const void* res, void* out, int32_t B, int32_t H, int32_t W, int32_t Cin,
                              int32_t Cout, int32_t out_dtype, vr_stream_t stream) {
    if (!res || ((uintptr_t)res & 7)) return VR_EINVAL;
    return conv3x3_entry(a, w, out, B, H, W, Cin, Cout, out_dtype, nullptr, res, stream);
}

// the same with `res` stored in patch order (non-overlapping res_patch x res_patch windows, vr_patch_unfold's layout): the gradient
// of the stem's skip connection as the projection's data gradient leaves it
extern "C" int vr_conv3x3_res_patch(const void* a, const void* w, const void* res, void* out, int32_t B, int32_t H, int32_t W, int32_t Cin,
                                    int32_t Cout, int32_t out_dtype, int32_t res_patch, vr_stream_t stream) {
    if (!res || ((uintptr_t)res & 7) || res_patch <= 0 || H % res_patch || W % res_patch) return VR_EINVAL;
    return conv3x3_entry(a, w, out, B, H, W, Cin, Cout, out_dtype, nullptr, res, stream, 0, res_patch);
}

extern "C" int vr_conv3x3_bias_relu(const void* a, const void* w, const float* bias, const void* res, void* out, int32_t B,
                                    int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t out_dtype, vr_stream_t stream) {
    if (!bias || ((uintptr_t)bias & 15) || (res && ((uintptr_t)res & 7))) return VR_EINVAL;
    return conv3x3_entry(a, w, out, B, H, W, Cin, Cout, out_dtype, bias, res, stream);
}

extern "C" int vr_conv3x3_bias_relu_patch(const void* a, const void* w, const float* bias, const void* res, void* out, int32_t B,
                                          int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t patch, int32_t out_dtype,
                                          vr_stream_t stream) {
    if (!bias || ((uintptr_t)bias & 15) || (res && ((uintptr_t)res & 7)) || patch <= 0 || H % patch || W % patch) return VR_EINVAL;
    return conv3x3_entry(a, w, out, B, H, W, Cin, Cout, out_dtype, bias, res, stream, patch);
}

static int conv3x3_entry(const void* a, const void* w, void* out, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                         int32_t out_dtype, const float* bias, const void* res, vr_stream_t stream, int patch, int rpatch) {
    if (!a || !w || !out || B <= 0 || H <= 0 || W <= 0) return VR_EINVAL;
    if (out_dtype != VR_F32 && out_dtype != VR_BF16) return VR_EUNSUPPORTED;
    if (Cout <= 0 || Cout > 32 || Cout % 4) return VR_EUNSUPPORTED;
    if (((uintptr_t)a & 15) || ((uintptr_t)w & 15) || ((uintptr_t)out & 15)) return VR_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    switch (Cin) {
        case 16: launch<2>((const bf16_t*)a, (const bf16_t*)w, out, B, H, W, Cout, out_dtype, st, bias, (const bf16_t*)res, patch, rpatch); break;
        case 24: launch<3>((const bf16_t*)a, (const bf16_t*)w, out, B, H, W, Cout, out_dtype, st, bias, (const bf16_t*)res, patch, rpatch); break;
        case 32: launch<4>((const bf16_t*)a, (const bf16_t*)w, out, B, H, W, Cout, out_dtype, st, bias, (const bf16_t*)res, patch, rpatch); break;
        default: return VR_EUNSUPPORTED;
    }
    VR_CHECK_LAUNCH();
    return VR_OK;
}

extern "C" int vr_conv1_direct(const float* img, const void* w, const float* bias, void* out, int32_t B, int32_t H, int32_t W,
                               int32_t Cout, int32_t relu, int32_t out_dtype, vr_stream_t stream) {
    if (!img || !w || !out || B <= 0 || H <= 0 || W <= 0) return VR_EINVAL;
    if (out_dtype != VR_F32 && out_dtype != VR_BF16) return VR_EUNSUPPORTED;
    if (Cout <= 0 || Cout > 32 || Cout % 4) return VR_EUNSUPPORTED;
    if (((uintptr_t)w & 15) || ((uintptr_t)out & 15) || (bias && ((uintptr_t)bias & 15))) return VR_EALIGN;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const unsigned grid = (unsigned)(B * ((Ho + C1_TH - 1) / C1_TH) * ((Wo + C1_TW - 1) / C1_TW));
    hipStream_t st = (hipStream_t)stream;
    if (out_dtype == VR_F32)
        hipLaunchKernelGGL((conv1_direct_kernel<float>), dim3(grid), dim3(256), 0, st, img, (const bf16_t*)w, bias, (float*)out, B, H, W,
                           Ho, Wo, Cout, relu);
    else
        hipLaunchKernelGGL((conv1_direct_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, img, (const bf16_t*)w, bias, (bf16_t*)out, B, H,
                           W, Ho, Wo, Cout, relu);
    VR_CHECK_LAUNCH();
    return VR_OK;
}
