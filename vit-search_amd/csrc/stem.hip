// Convolutional patch embedding (reference nets/patch_conv.py:23-73: 3 x [Conv3x3 + BatchNorm + ReLU] at 112^2,
// residual, Conv 7x7 / stride 7 -> tokens), expressed as NHWC gathers around the MFMA GEMM:
//   conv3x3      = im2col3x3 (k = (kh, kw, c), channels contiguous)  ->  vr_gemm  ->  z fp32 [pixels, C]
//   BatchNorm    = per-channel sum / sum-of-squares reduction (training: batch statistics) + fused scale/shift/ReLU
//   conv 7x7/s7  = patch unfold (pure permutation, patches do not overlap) -> vr_gemm with the token epilogue
// All kernels are HBM-bound streams: 16-byte accesses along the channel dimension, one pass each.
#include <cstdlib>

#include "common.h"
#include "../../include/vitres_hip.h"

namespace {

template <typename T> struct V8;   // 8 consecutive channels
template <> struct V8<bf16_t> {
    typedef uint4 type;
    static __device__ __forceinline__ void unpack(const uint4& u, float (&f)[8]) {
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(w[i] << 16);
            f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ uint4 pack(const float (&f)[8]) {
        return make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7]));
    }
    static __device__ __forceinline__ void load(const bf16_t* p, float (&f)[8]) { unpack(*reinterpret_cast<const uint4*>(p), f); }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&f)[8]) { *reinterpret_cast<uint4*>(p) = pack(f); }
};
template <> struct V8<float> {
    static __device__ __forceinline__ void load(const float* p, float (&f)[8]) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&f)[8]) {
        *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
    }
};

// ---- conv1: NCHW fp32 image -> col [B*Ho*Wo, ld], k = (kh, kw, c), 3x3 pad 1 -----------------------------
template <typename T>
__global__ __launch_bounds__(256) void im2col3x3_nchw_kernel(const float* __restrict__ img, T* __restrict__ col, int B,
                                                             int C, int H, int W, int stride, int ld, long long total) {
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int k = (int)(i % ld);
        const long long row = i / ld;
        float v = 0.f;
        if (k < 9 * C) {
            const int c = k % C, tap = k / C;
            const int ow = (int)(row % Wo), oh = (int)((row / Wo) % Ho), b = (int)(row / ((long long)Wo * Ho));
            const int ih = oh * stride - 1 + tap / 3, iw = ow * stride - 1 + tap % 3;
            if (ih >= 0 && ih < H && iw >= 0 && iw < W) v = img[(((long long)b * C + c) * H + ih) * W + iw];
        }
        Elem<T>::st(col + i, v);
    }
}

// The same for the shape the patch embedding uses (3 channels, bf16, ld = 32): one thread per output pixel gathers its 27 values and
// writes the whole 64-byte row with four 16-byte stores -- the element-per-thread form above ran at 0.6 TB/s (three integer
// divisions per 2-byte store).  Consecutive threads = consecutive output columns: their image reads share cache lines.
__global__ __launch_bounds__(256) void im2col3x3_nchw3_row_kernel(const float* __restrict__ img, bf16_t* __restrict__ col, int B, int H,
                                                                  int W, int stride, long long rows) {
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    for (long long row = (long long)blockIdx.x * 256 + threadIdx.x; row < rows; row += (long long)gridDim.x * 256) {
        const int ow = (int)(row % Wo), oh = (int)((row / Wo) % Ho), b = (int)(row / ((long long)Wo * Ho));
        const float* src = img + (long long)b * 3 * H * W;
        float v[32];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ih = oh * stride - 1 + tap / 3, iw = ow * stride - 1 + tap % 3;
            const bool in = ih >= 0 && ih < H && iw >= 0 && iw < W;
            const long long o = in ? (long long)ih * W + iw : 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[tap * 3 + c] = in ? src[(long long)c * H * W + o] : 0.f;
        }
#pragma unroll
        for (int k = 27; k < 32; ++k) v[k] = 0.f;
        uint4* dst = reinterpret_cast<uint4*>(col + row * 32);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            dst[q] = make_uint4(pack_bf2(v[8 * q], v[8 * q + 1]), pack_bf2(v[8 * q + 2], v[8 * q + 3]), pack_bf2(v[8 * q + 4], v[8 * q + 5]),
                                pack_bf2(v[8 * q + 6], v[8 * q + 7]));
    }
}

// ---- conv2/3: NHWC activations -> col, stride 1 pad 1, 8 channels per thread ------------------------------
template <typename T>
__global__ __launch_bounds__(256) void im2col3x3_nhwc_kernel(const T* __restrict__ src, T* __restrict__ col, int B, int H,
                                                             int W, int C, long long total) {
    const int c8n = C / 8, per_row = 9 * c8n;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int u = (int)(i % per_row);
        const long long row = i / per_row;
        const int c8 = u % c8n, tap = u / c8n;
        const int ow = (int)(row % W), oh = (int)((row / W) % H), b = (int)(row / ((long long)W * H));
        const int ih = oh - 1 + tap / 3, iw = ow - 1 + tap % 3;
        const bool in = ih >= 0 && ih < H && iw >= 0 && iw < W;
        float f[8];
        V8<T>::load(src + (((long long)b * H + (in ? ih : 0)) * W + (in ? iw : 0)) * C + c8 * 8, f);
        if (!in) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = 0.f;
        }
        V8<T>::store(col + row * (9LL * C) + tap * C + c8 * 8, f);
    }
}

// d src[b,ih,iw,c] = sum over taps of dcol[(b, ih+1-kh, iw+1-kw)][(kh,kw,c)]   (gather form, stride 1)
template <typename T>
__global__ __launch_bounds__(256) void col2im3x3_nhwc_kernel(const T* __restrict__ dcol, T* __restrict__ dsrc, int B, int H,
                                                             int W, int C, long long total) {
    const int c8n = C / 8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c8 = (int)(i % c8n);
        const long long pix = i / c8n;
        const int iw = (int)(pix % W), ih = (int)((pix / W) % H), b = (int)(pix / ((long long)W * H));
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int oh = ih + 1 - kh;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ow = iw + 1 - kw;
                const bool in = oh >= 0 && oh < H && ow >= 0 && ow < W;
                float f[8];
                V8<T>::load(dcol + (((long long)b * H + (in ? oh : 0)) * W + (in ? ow : 0)) * (9LL * C) + (kh * 3 + kw) * C + c8 * 8, f);
                if (in) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += f[e];
                }
            }
        }
        V8<T>::store(dsrc + pix * C + c8 * 8, acc);
    }
}

// ---- BatchNorm pieces --------------------------------------------------------------------------------------
// per-channel sum and sum of squares of z fp32 [R, C]; block = 256 threads = (256/C8N rows) x C8N channel octets
template <typename TZ>
__global__ __launch_bounds__(256) void bn_stats_kernel(const TZ* __restrict__ z, float* __restrict__ sum,
                                                       float* __restrict__ sumsq, long long R, int C) {
    __shared__ float red[2][256 * 8 / 8 * 8];
    const int c8n = C / 8;
    const int rows_par = 256 / c8n;                 // rows handled in parallel by one block
    const int c8 = threadIdx.x % c8n, rl = threadIdx.x / c8n;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, q[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (rl < rows_par) {
        const long long step = (long long)gridDim.x * rows_par;                   // (four rows per trip, as in the backward)
        for (long long r = (long long)blockIdx.x * rows_par + rl; r < R; r += 4 * step) {
            float f[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) V8<TZ>::load(z + (r + u * step < R ? r + u * step : r) * C + c8 * 8, f[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (r + u * step < R) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        s[e] += f[u][e];
                        q[e] += f[u][e] * f[u][e];
                    }
                }
            }
        }
    }
    // reduce over the rows_par row-lanes that share a channel octet
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        red[0][threadIdx.x * 8 + e] = s[e];
        red[1][threadIdx.x * 8 + e] = q[e];
    }
    __syncthreads();
    if (threadIdx.x < C) {
        const int c = threadIdx.x, oc = c / 8, e = c % 8;
        float a = 0.f, b2 = 0.f;
        for (int r = 0; r < rows_par; ++r) {
            a += red[0][(r * c8n + oc) * 8 + e];
            b2 += red[1][(r * c8n + oc) * 8 + e];
        }
        atomicAdd(sum + c, a);
        atomicAdd(sumsq + c, b2);
    }
}

// out = relu(z * scale[c] + shift[c]) (+ res)
// Pixel (b, y, x) of an NHWC tensor -> its row in the patchify operand of the 7 x 7 / stride-7 projection ([B*gh*gw, (i, j, c)],
// vr_patch_unfold's layout): the P x P windows do not overlap, so patch order is a permutation of the pixels and a pixel's C channels
// stay contiguous.  Round 4: the last BatchNorm + ReLU of the stem WRITES patch order, and the backward's readers of d(that tensor)
// (BatchNorm backward, the skip connection's add in the data-gradient convolution) READ it -- the unfold / fold passes (410 MB each
// way at B = 128) are gone from the training step.  P = 0: identity.
__device__ __forceinline__ long long patch_pix(long long pix, int H, int W, int P) {
    if (P <= 0) return pix;
    const int x = (int)(pix % W);
    const long long t = pix / W;
    const int y = (int)(t % H);
    const long long b = t / H;
    return ((b * (H / P) + y / P) * (W / P) + x / P) * (P * P) + (y % P) * P + x % P;
}

template <typename T, typename TZ>
__global__ __launch_bounds__(256) void bn_relu_kernel(const TZ* __restrict__ z, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, const T* __restrict__ res,
                                                      T* __restrict__ out, long long total8, int C, int H, int W, int P) {
    const int c8n = C / 8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total8; i += (long long)gridDim.x * 256) {
        const int c0 = (int)(i % c8n) * 8;
        float f[8], r[8];
        V8<TZ>::load(z + i * 8, f);
        if (res) V8<T>::load(res + i * 8, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = fmaxf(f[e] * scale[c0 + e] + shift[c0 + e], 0.f);
            if (res) v += r[e];
            f[e] = v;
        }
        V8<T>::store(out + patch_pix(i / c8n, H, W, P) * C + c0, f);
    }
}

// backward reduction: g = da * [z*scale+shift > 0];  sg[c] += g ; sgz[c] += g * zhat,  zhat = (z - mean) * rstd
template <typename T, typename TZ>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ da, const TZ* __restrict__ z,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float* __restrict__ sg, float* __restrict__ sgz, long long R,
                                                            int C, int H, int W, int P) {
    __shared__ float red[2][256 * 8];
    const int c8n = C / 8;
    const int rows_par = 256 / c8n;
    const int c8 = threadIdx.x % c8n, rl = threadIdx.x / c8n;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, q[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (rl < rows_par) {
        float sc[8], sh[8], mu[8], rs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sc[e] = scale[c8 * 8 + e]; sh[e] = shift[c8 * 8 + e]; mu[e] = mean[c8 * 8 + e]; rs[e] = rstd[c8 * 8 + e];
        }
        // four rows per trip, all eight loads issued before the first sum (one row per trip: 2.5 TB/s, latency-bound)
        const long long step = (long long)gridDim.x * rows_par;
        for (long long r = (long long)blockIdx.x * rows_par + rl; r < R; r += 4 * step) {
            float f[4][8], g[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long ru = r + u * step < R ? r + u * step : r;         // (rows past the end: row r again, not summed)
                V8<TZ>::load(z + ru * C + c8 * 8, f[u]);
                V8<T>::load(da + patch_pix(ru, H, W, P) * C + c8 * 8, g[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (r + u * step < R) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float gg = (f[u][e] * sc[e] + sh[e] > 0.f) ? g[u][e] : 0.f;
                        s[e] += gg;
                        q[e] += gg * (f[u][e] - mu[e]) * rs[e];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        red[0][threadIdx.x * 8 + e] = s[e];
        red[1][threadIdx.x * 8 + e] = q[e];
    }
    __syncthreads();
    if (threadIdx.x < C) {
        const int c = threadIdx.x, oc = c / 8, e = c % 8;
        float a = 0.f, b2 = 0.f;
        for (int r = 0; r < rows_par; ++r) {
            a += red[0][(r * c8n + oc) * 8 + e];
            b2 += red[1][(r * c8n + oc) * 8 + e];
        }
        atomicAdd(sg + c, a);
        atomicAdd(sgz + c, b2);
    }
}

// dz = gamma*rstd * (g - sg/n - zhat * sgz/n)   [training]   or   gamma*rstd * g   [eval: inv_n = 0]
template <typename T, typename TZ>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ da, const TZ* __restrict__ z,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ sg, const float* __restrict__ sgz,
                                                           float inv_n, T* __restrict__ dz, long long total8, int C, int H, int W,
                                                           int P) {
    const int c8n = C / 8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total8; i += (long long)gridDim.x * 256) {
        const int c0 = (int)(i % c8n) * 8;
        float f[8], g[8];
        V8<TZ>::load(z + i * 8, f);
        V8<T>::load(da + patch_pix(i / c8n, H, W, P) * C + c0, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + e;
            const float gg = (f[e] * scale[c] + shift[c] > 0.f) ? g[e] : 0.f;
            const float zh = (f[e] - mean[c]) * rstd[c];
            g[e] = scale[c] * (gg - sg[c] * inv_n - zh * sgz[c] * inv_n);
        }
        V8<T>::store(dz + i * 8, g);
    }
}

// ---- non-overlapping P x P patches: a NHWC [B, gh*P, gw*P, C] <-> col [B*gh*gw, (i, j, c)] ---------------------
template <typename T, bool FOLD>
__global__ __launch_bounds__(256) void patch_unfold_kernel(T* __restrict__ a, T* __restrict__ col, int B, int gh, int gw,
                                                           int P, int C, long long total) {
    const int c8n = C / 8;
    const int per_row = P * P * c8n;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int u = (int)(i % per_row);
        const long long row = i / per_row;
        const int c8 = u % c8n, j = (u / c8n) % P, ii = u / (c8n * P);
        const int px = (int)(row % gw), py = (int)((row / gw) % gh), b = (int)(row / ((long long)gw * gh));
        T* pa = a + ((((long long)b * gh + py) * P + ii) * (gw * P) + px * P + j) * C + c8 * 8;
        T* pc = col + row * ((long long)P * P * C) + (ii * P + j) * C + c8 * 8;
        float f[8];
        if (FOLD) { V8<T>::load(pc, f); V8<T>::store(pa, f); }
        else { V8<T>::load(pa, f); V8<T>::store(pc, f); }
    }
}

inline unsigned grid_for(long long total) {
    long long b = (total + 255) / 256;
    return (unsigned)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int vr_im2col3x3(const void* src, void* col, int32_t B, int32_t H, int32_t W, int32_t C, int32_t stride,
                            int32_t src_nchw_f32, int32_t ld, int32_t dtype, vr_stream_t stream) {
    if (!src || !col || B <= 0 || H <= 0 || W <= 0 || C <= 0) return VR_EINVAL;
    if (dtype != VR_F32 && dtype != VR_BF16) return VR_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (src_nchw_f32) {
        if (ld < 9 * C) return VR_EINVAL;
        const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
        const long long total = (long long)B * Ho * Wo * ld;
        if (dtype == VR_BF16 && C == 3 && ld == 32 && ((uintptr_t)col & 15) == 0) {
            const long long rows = (long long)B * Ho * Wo;
            hipLaunchKernelGGL(im2col3x3_nchw3_row_kernel, dim3(grid_for(rows)), dim3(256), 0, st, (const float*)src, (bf16_t*)col, B, H, W,
                               stride, rows);
        } else if (dtype == VR_F32)
            hipLaunchKernelGGL((im2col3x3_nchw_kernel<float>), dim3(grid_for(total)), dim3(256), 0, st, (const float*)src, (float*)col, B, C, H, W, stride, ld, total);
        else
            hipLaunchKernelGGL((im2col3x3_nchw_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, st, (const float*)src, (bf16_t*)col, B, C, H, W, stride, ld, total);
    } else {
        if (stride != 1 || C % 8 || ld != 9 * C) return VR_EUNSUPPORTED;
        const long long total = (long long)B * H * W * 9 * (C / 8);
        if (dtype == VR_F32)
            hipLaunchKernelGGL((im2col3x3_nhwc_kernel<float>), dim3(grid_for(total)), dim3(256), 0, st, (const float*)src, (float*)col, B, H, W, C, total);
        else
            hipLaunchKernelGGL((im2col3x3_nhwc_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)col, B, H, W, C, total);
    }
    VR_CHECK_LAUNCH();
    return VR_OK;
}

extern "C" int vr_col2im3x3(const void* dcol, void* dsrc, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype,
                            vr_stream_t stream) {
    if (!dcol || !dsrc || B <= 0 || C <= 0) return VR_EINVAL;
    if (C % 8) return VR_EUNSUPPORTED;
    const long long total = (long long)B * H * W * (C / 8);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VR_F32)
        hipLaunchKernelGGL((col2im3x3_nhwc_kernel<float>), dim3(grid_for(total)), dim3(256), 0, st, (const float*)dcol, (float*)dsrc, B, H, W, C, total);
    else if (dtype == VR_BF16)
        hipLaunchKernelGGL((col2im3x3_nhwc_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, st, (const bf16_t*)dcol, (bf16_t*)dsrc, B, H, W, C, total);
    else
        return VR_EUNSUPPORTED;
    VR_CHECK_LAUNCH();
    return VR_OK;
}

extern "C" int vr_bn_stats(const void* z, float* sum, float* sumsq, int64_t R, int32_t C, int32_t z_dtype, vr_stream_t stream) {
    if (!z || !sum || !sumsq || R <= 0 || C <= 0) return VR_EINVAL;
    if (C % 8 || C > 256) return VR_EUNSUPPORTED;
    if (z_dtype != VR_F32 && z_dtype != VR_BF16) return VR_EUNSUPPORTED;
    const int rows_par = 256 / (C / 8);
    long long blocks = (R + rows_par * 64 - 1) / (rows_par * 64);    // (more workgroups: their atomics on the 2C sums collide, 38 -> 75 us)
    if (blocks > 2048) blocks = 2048;
    if (z_dtype == VR_F32)
        hipLaunchKernelGGL((bn_stats_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)z, sum, sumsq, (long long)R, C);
    else
        hipLaunchKernelGGL((bn_stats_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)z, sum, sumsq, (long long)R, C);
    VR_CHECK_LAUNCH();
    return VR_OK;
}

static int bn_relu_entry(const void* z, const float* scale, const float* shift, const void* res, void* out, int64_t R,
                          int32_t C, int32_t dtype, int32_t z_dtype, vr_stream_t stream, int H, int W, int P) {
    if (!z || !scale || !shift || !out || R <= 0 || C <= 0) return VR_EINVAL;
    if (C % 8) return VR_EUNSUPPORTED;
    if (z_dtype != VR_F32 && !(z_dtype == VR_BF16 && dtype == VR_BF16)) return VR_EUNSUPPORTED;
    const long long total8 = (long long)R * (C / 8);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VR_F32)
        hipLaunchKernelGGL((bn_relu_kernel<float, float>), dim3(grid_for(total8)), dim3(256), 0, st, (const float*)z, scale, shift, (const float*)res, (float*)out, total8, C, H, W, P);
    else if (dtype == VR_BF16 && z_dtype == VR_F32)
        hipLaunchKernelGGL((bn_relu_kernel<bf16_t, float>), dim3(grid_for(total8)), dim3(256), 0, st, (const float*)z, scale, shift, (const bf16_t*)res, (bf16_t*)out, total8, C, H, W, P);
    else if (dtype == VR_BF16)
        hipLaunchKernelGGL((bn_relu_kernel<bf16_t, bf16_t>), dim3(grid_for(total8)), dim3(256), 0, st, (const bf16_t*)z, scale, shift, (const bf16_t*)res, (bf16_t*)out, total8, C, H, W, P);
    else
        return VR_EUNSUPPORTED;
    VR_CHECK_LAUNCH();
    return VR_OK;
}

extern "C" int vr_bn_relu(const void* z, const float* scale, const float* shift, const void* res, void* out, int64_t R,
                          int32_t C, int32_t dtype, int32_t z_dtype, vr_stream_t stream) {
    return bn_relu_entry(z, scale, shift, res, out, R, C, dtype, z_dtype, stream, 0, 0, 0);
}
extern "C" int vr_bn_relu_patch(const void* z, const float* scale, const float* shift, const void* res, void* out, int32_t B, int32_t H,
                                int32_t W, int32_t patch, int32_t C, int32_t dtype, int32_t z_dtype, vr_stream_t stream) {
    if (B <= 0 || H <= 0 || W <= 0 || patch <= 0 || H % patch || W % patch) return VR_EINVAL;
    return bn_relu_entry(z, scale, shift, res, out, (int64_t)B * H * W, C, dtype, z_dtype, stream, H, W, patch);
}

// Train-mode BatchNorm2d between the statistics pass and the normalise pass, one launch instead of a dozen elementwise ones
// (torch.nn.BatchNorm2d semantics, nets/patch_conv.py:23-38): biased variance for the normalisation, unbiased for the running
// estimate, running <- (1 - momentum) * running + momentum * batch, num_batches_tracked += 1.
namespace {
__global__ void bn_finalize_kernel(const float* __restrict__ sum, const float* __restrict__ sumsq, float n, const float* __restrict__ w,
                                   const float* __restrict__ b, float eps, float momentum, float* __restrict__ rmean,
                                   float* __restrict__ rvar, long long* __restrict__ nbt, float* __restrict__ scale,
                                   float* __restrict__ shift, float* __restrict__ mean, float* __restrict__ rstd, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt) *nbt += 1;
    if (c >= C) return;
    const float mu = sum[c] / n;
    const float var = fmaxf(sumsq[c] / n - mu * mu, 0.f);
    if (rmean) {
        rmean[c] = rmean[c] * (1.0f - momentum) + momentum * mu;
        rvar[c] = rvar[c] * (1.0f - momentum) + momentum * (var * (n / fmaxf(n - 1.0f, 1.0f)));
    }
    const float rs = 1.0f / sqrtf(var + eps);
    const float sc = w[c] * rs;
    mean[c] = mu;
    rstd[c] = rs;
    scale[c] = sc;
    shift[c] = b[c] - mu * sc;
}
}  // namespace

extern "C" int vr_bn_finalize(const float* sum, const float* sumsq, int64_t n, const float* weight, const float* bias, float eps,
                              float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* scale,
                              float* shift, float* mean, float* rstd, int32_t C, vr_stream_t stream) {
    if (!sum || !sumsq || !weight || !bias || !scale || !shift || !mean || !rstd || n <= 0 || C <= 0) return VR_EINVAL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return VR_EINVAL;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, (hipStream_t)stream, sum, sumsq, (float)n, weight,
                       bias, eps, momentum, running_mean, running_var, (long long*)num_batches_tracked, scale, shift, mean, rstd, C);
    VR_CHECK_LAUNCH();
    return VR_OK;
}

static int bn_bwd_entry(const void* da, const void* z, const float* scale, const float* shift, const float* mean,
                         const float* rstd, float* sg, float* sgz, void* dz, int64_t R, int32_t C, int32_t training,
                         int32_t dtype, int32_t z_dtype, vr_stream_t stream, int H, int W, int P) {
    if (!da || !z || !scale || !shift || !mean || !rstd || !sg || !sgz || !dz || R <= 0 || C <= 0) return VR_EINVAL;
    if (C % 8 || C > 256) return VR_EUNSUPPORTED;
    if (dtype != VR_F32 && dtype != VR_BF16) return VR_EUNSUPPORTED;
    if (z_dtype != VR_F32 && !(z_dtype == VR_BF16 && dtype == VR_BF16)) return VR_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int rows_par = 256 / (C / 8);
    long long blocks = (R + rows_par * 64 - 1) / (rows_par * 64);    // (8x the workgroups: 93 -> 128 us, colliding atomics)
    if (blocks > 2048) blocks = 2048;
    const long long total8 = (long long)R * (C / 8);
    const float inv_n = training ? 1.0f / (float)R : 0.f;
#define VR_BN_BWD(T, TZ)                                                                                                       \
    do {                                                                                                                       \
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, TZ>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)da, (const TZ*)z,  \
                           scale, shift, mean, rstd, sg, sgz, (long long)R, C, H, W, P);                                                \
        hipLaunchKernelGGL((bn_bwd_apply_kernel<T, TZ>), dim3(grid_for(total8)), dim3(256), 0, st, (const T*)da, (const TZ*)z,   \
                           scale, shift, mean, rstd, sg, sgz, inv_n, (T*)dz, total8, C, H, W, P);                                       \
    } while (0)
    if (dtype == VR_F32) VR_BN_BWD(float, float);
    else if (z_dtype == VR_F32) VR_BN_BWD(bf16_t, float);
    else VR_BN_BWD(bf16_t, bf16_t);
#undef VR_BN_BWD
    VR_CHECK_LAUNCH();
    return VR_OK;
}

extern "C" int vr_bn_bwd(const void* da, const void* z, const float* scale, const float* shift, const float* mean,
                         const float* rstd, float* sg, float* sgz, void* dz, int64_t R, int32_t C, int32_t training,
                         int32_t dtype, int32_t z_dtype, vr_stream_t stream) {
    return bn_bwd_entry(da, z, scale, shift, mean, rstd, sg, sgz, dz, R, C, training, dtype, z_dtype, stream, 0, 0, 0);
}
extern "C" int vr_bn_bwd_patch(const void* da, const void* z, const float* scale, const float* shift, const float* mean,
                               const float* rstd, float* sg, float* sgz, void* dz, int32_t B, int32_t H, int32_t W, int32_t patch,
                               int32_t C, int32_t training, int32_t dtype, int32_t z_dtype, vr_stream_t stream) {
    if (B <= 0 || H <= 0 || W <= 0 || patch <= 0 || H % patch || W % patch) return VR_EINVAL;
    return bn_bwd_entry(da, z, scale, shift, mean, rstd, sg, sgz, dz, (int64_t)B * H * W, C, training, dtype, z_dtype, stream, H, W, patch);
}

extern "C" int vr_patch_unfold(void* a, void* col, int32_t B, int32_t gh, int32_t gw, int32_t P, int32_t C, int32_t fold,
                               int32_t dtype, vr_stream_t stream) {
    if (!a || !col || B <= 0 || gh <= 0 || gw <= 0 || P <= 0 || C <= 0) return VR_EINVAL;
    if (C % 8) return VR_EUNSUPPORTED;
    const long long total = (long long)B * gh * gw * P * P * (C / 8);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VR_F32) {
        if (fold) hipLaunchKernelGGL((patch_unfold_kernel<float, true>), dim3(grid_for(total)), dim3(256), 0, st, (float*)a, (float*)col, B, gh, gw, P, C, total);
        else hipLaunchKernelGGL((patch_unfold_kernel<float, false>), dim3(grid_for(total)), dim3(256), 0, st, (float*)a, (float*)col, B, gh, gw, P, C, total);
    } else if (dtype == VR_BF16) {
        if (fold) hipLaunchKernelGGL((patch_unfold_kernel<bf16_t, true>), dim3(grid_for(total)), dim3(256), 0, st, (bf16_t*)a, (bf16_t*)col, B, gh, gw, P, C, total);
        else hipLaunchKernelGGL((patch_unfold_kernel<bf16_t, false>), dim3(grid_for(total)), dim3(256), 0, st, (bf16_t*)a, (bf16_t*)col, B, gh, gw, P, C, total);
    } else {
        return VR_EUNSUPPORTED;
    }
    VR_CHECK_LAUNCH();
    return VR_OK;
}
