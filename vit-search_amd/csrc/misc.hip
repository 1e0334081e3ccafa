// Small HBM-bound kernels around the GEMMs of the ViT-Res hot path (gfx950).
// Every kernel streams its operands once with 16-byte accesses where the layout allows.
#include "common.h"
#include "../../include/vitres_hip.h"

namespace {

// ---- fp32 -> bf16 ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x * 8;
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
        if (i + 8 <= n) {
            const float4 a = *reinterpret_cast<const float4*>(src + i);
            const float4 b = *reinterpret_cast<const float4*>(src + i + 4);
            *reinterpret_cast<uint4*>(dst + i) =
                make_uint4(pack_bf2(a.x, a.y), pack_bf2(a.z, a.w), pack_bf2(b.x, b.y), pack_bf2(b.z, b.w));
        } else {
            for (long long j = i; j < n; ++j) dst[j] = f2bf(src[j]);
        }
    }
}

// ---- batch of fp32 [rows, cols] -> bf16 [cols, ld] transposes (64 x 64 tiles through LDS) ----
// interior tiles of 16-byte aligned matrices move float4 in and eight bf16 (one 16-byte store) out per thread: the scalar form
// ran at 1 TB/s beside the forward and tripled the duration of the GEMMs it shared the chip with
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                             const vr_tr_desc* __restrict__ descs) {
    __shared__ float tile[64][65];
    const vr_tr_desc d = descs[blockIdx.y];
    const int tc = (d.cols + 63) / 64, tr = (d.rows + 63) / 64;
    if ((int)blockIdx.x >= tc * tr) return;
    const int r0 = ((int)blockIdx.x / tc) * 64, c0 = ((int)blockIdx.x % tc) * 64;
    const float* s = src + d.src_off;
    bf16_t* o = dst + d.dst_off;
    const bool wide = r0 + 64 <= d.rows && c0 + 64 <= d.cols && d.cols % 4 == 0 && d.ld_dst % 8 == 0 &&
                      (reinterpret_cast<uintptr_t>(s) & 15) == 0 && (reinterpret_cast<uintptr_t>(o) & 15) == 0;
    if (wide) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = threadIdx.x + 256 * i, r = idx >> 4, c4 = (idx & 15) * 4;
            const float4 v = *reinterpret_cast<const float4*>(s + (long long)(r0 + r) * d.cols + c0 + c4);
            tile[r][c4 + 0] = v.x; tile[r][c4 + 1] = v.y; tile[r][c4 + 2] = v.z; tile[r][c4 + 3] = v.w;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = threadIdx.x + 256 * i, c = idx >> 3, r8 = (idx & 7) * 8;
            uint4 q;
            q.x = pack_bf2(tile[r8 + 0][c], tile[r8 + 1][c]);
            q.y = pack_bf2(tile[r8 + 2][c], tile[r8 + 3][c]);
            q.z = pack_bf2(tile[r8 + 4][c], tile[r8 + 5][c]);
            q.w = pack_bf2(tile[r8 + 6][c], tile[r8 + 7][c]);
            *reinterpret_cast<uint4*>(o + (long long)(c0 + c) * d.ld_dst + r0 + r8) = q;
        }
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty + 4 * i, c = c0 + tx;
        tile[ty + 4 * i][tx] = (r < d.rows && c < d.cols) ? s[(long long)r * d.cols + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty + 4 * i, r = r0 + tx;
        if (c < d.cols && r < d.rows) o[(long long)c * d.ld_dst + r] = f2bf(tile[tx][ty + 4 * i]);
    }
}

// ---- soft-target cross entropy: wave per row -----------------------------------------------------------
__global__ __launch_bounds__(256) void softce_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                     float* __restrict__ loss, float* __restrict__ dx, int R, int K,
                                                     float gscale) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* xr = x + (long long)r * K;
    const float* tr = t + (long long)r * K;
    float mx = -INFINITY;
    for (int k = lane; k < K; k += 64) mx = fmaxf(mx, xr[k]);
    mx = wave_max(mx);
    float se = 0.f, st = 0.f, stx = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float xv = xr[k], tv = tr[k];
        se += __expf(xv - mx);
        st += tv;
        stx += tv * xv;
    }
    se = wave_sum(se);
    st = wave_sum(st);
    stx = wave_sum(stx);
    const float lse = mx + __logf(se);
    if (lane == 0) loss[r] = lse * st - stx;
    if (dx) {
        float* dr = dx + (long long)r * K;
        const float inv = 1.0f / se;
        for (int k = lane; k < K; k += 64) dr[k] = gscale * (__expf(xr[k] - mx) * inv * st - tr[k]);
    }
}

// Loss step of the training loop in one pass: logits rows are in the model's internal (architecture-grouped) sample order, targets
// in the caller's; the mean loss is accumulated into one scalar and the gradient is written in the dtype / row pitch the head's
// backward GEMMs read (pad columns zeroed) -- no torch glue (gather, mean, add, mul, pad, cast) between the heads and the backward.
template <typename TG>
__global__ __launch_bounds__(256) void softce_train_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                           const long long* __restrict__ sample_map, int rps,
                                                           float* __restrict__ loss_acc, TG* __restrict__ dx, int ld_grad, int R,
                                                           int K, float gscale, float loss_scale) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int s = r / rps;
    const long long trow = (sample_map ? sample_map[s] : (long long)s) * rps + (r - s * rps);
    const float* xr = x + (long long)r * K;
    const float* tr = t + trow * K;
    float mx = -INFINITY;
    for (int k = lane; k < K; k += 64) mx = fmaxf(mx, xr[k]);
    mx = wave_max(mx);
    float se = 0.f, st = 0.f, stx = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float xv = xr[k], tv = tr[k];
        se += __expf(xv - mx);
        st += tv;
        stx += tv * xv;
    }
    se = wave_sum(se);
    st = wave_sum(st);
    stx = wave_sum(stx);
    const float lse = mx + __logf(se);
    if (lane == 0) atomicAdd(loss_acc, (lse * st - stx) * loss_scale);
    TG* dr = dx + (long long)r * ld_grad;
    const float inv = 1.0f / se;
    for (int k = lane; k < ld_grad; k += 64)
        Elem<TG>::st(dr + k, k < K ? gscale * (__expf(xr[k] - mx) * inv * st - tr[k]) : 0.f);
}

// Register-resident form (round 5) for K % 4 == 0, K <= 1024 (the 1000-class heads): a wave owns a row, every lane loads its
// <= 4 float4 of logits and targets ONCE (all in flight together: one memory latency per row instead of one per pass and
// 64-element step), keeps exp(x - max) in registers and writes the gradient from them.  The pass-by-pass kernel above took 41 us
// for the 2048 x 1000 patch logits (20 MB: ~0.5 TB/s).
template <typename TG>
__global__ __launch_bounds__(256) void softce_train_reg_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                               const long long* __restrict__ sample_map, int rps,
                                                               float* __restrict__ loss_acc, TG* __restrict__ dx, int ld_grad, int R,
                                                               int K, float gscale, float loss_scale) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int s = r / rps;
    const long long trow = (sample_map ? sample_map[s] : (long long)s) * rps + (r - s * rps);
    const float* xr = x + (long long)r * K;
    const float* tr = t + trow * K;
    constexpr int NV = 4;
    float4 xv[NV], tv[NV];
    const int K4 = K >> 2;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int q = lane + 64 * v;
        const bool in = q < K4;
        xv[v] = in ? *reinterpret_cast<const float4*>(xr + 4 * q) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        tv[v] = in ? *reinterpret_cast<const float4*>(tr + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int v = 0; v < NV; ++v) mx = fmaxf(fmaxf(mx, fmaxf(xv[v].x, xv[v].y)), fmaxf(xv[v].z, xv[v].w));
    mx = wave_max(mx);
    float se = 0.f, st = 0.f, stx = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const bool in = lane + 64 * v < K4;
        const float4 a = xv[v], b = tv[v];
        st += (b.x + b.y) + (b.z + b.w);
        if (in) stx += (b.x * a.x + b.y * a.y) + (b.z * a.z + b.w * a.w);      // (padding lanes: 0 * -inf)
        xv[v] = make_float4(__expf(a.x - mx), __expf(a.y - mx), __expf(a.z - mx), __expf(a.w - mx));   // exp(-inf) = 0 in the padding
        se += (xv[v].x + xv[v].y) + (xv[v].z + xv[v].w);
    }
    se = wave_sum(se);
    st = wave_sum(st);
    stx = wave_sum(stx);
    const float lse = mx + __logf(se);
    if (lane == 0) atomicAdd(loss_acc, (lse * st - stx) * loss_scale);
    TG* dr = dx + (long long)r * ld_grad;
    const float c = st / se;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int k = 4 * (lane + 64 * v);
        if (k >= ld_grad) continue;                                           // (ld_grad % 4 == 0: a group is whole or outside)
        float4 g;
        g.x = k + 0 < K ? gscale * (xv[v].x * c - tv[v].x) : 0.f;
        g.y = k + 1 < K ? gscale * (xv[v].y * c - tv[v].y) : 0.f;
        g.z = k + 2 < K ? gscale * (xv[v].z * c - tv[v].z) : 0.f;
        g.w = k + 3 < K ? gscale * (xv[v].w * c - tv[v].w) : 0.f;
        if constexpr (sizeof(TG) == 4) *reinterpret_cast<float4*>(dr + k) = g;
        else *reinterpret_cast<uint2*>(dr + k) = make_uint2(pack_bf2(g.x, g.y), pack_bf2(g.z, g.w));
    }
}

// ---- column sums (bias gradients) ----------------------------------------------------------------------
// grid.x over column groups of 256, grid.y over row chunks; thread = one column; atomics at the end.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ in, float* __restrict__ out, int M, int N,
                                                     int ld, int rows_per_block, RowMap rm) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int mb = blockIdx.y * rows_per_block;
    const int me = min(M, mb + rows_per_block);
    float s = 0.f;
    for (int m = mb; m < me; ++m) s += Elem<T>::ld(in + map_row(rm, m) * (long long)ld + n);
    atomicAdd(out + n, s);
}

// out[m,c] = T(in[m,c] * scale[s]) for c < keep[s], else 0   (gradient entering a masked / drop-path'd branch)
template <typename T>
__global__ __launch_bounds__(256) void scale_mask_cast_kernel(const float* __restrict__ in, T* __restrict__ out,
                                                              const float* __restrict__ scale, const int* __restrict__ keep,
                                                              int M, int C, int rps) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int s = m / rps;
    const float sc = scale ? scale[s] : 1.0f;
    const int kc = keep ? keep[s] : C;
    const float* src = in + (long long)m * C;
    T* dst = out + (long long)m * C;
    for (int c = lane * 4; c < C; c += 256) {
        float4 v = *reinterpret_cast<const float4*>(src + c);
        v.x = (c + 0 < kc) ? v.x * sc : 0.f;
        v.y = (c + 1 < kc) ? v.y * sc : 0.f;
        v.z = (c + 2 < kc) ? v.z * sc : 0.f;
        v.w = (c + 3 < kc) ? v.w * sc : 0.f;
        if constexpr (sizeof(T) == 4) *reinterpret_cast<float4*>(dst + c) = v;
        else *reinterpret_cast<uint2*>(dst + c) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
    }
}

// out[i] += sum_b in[b, i]: the batch is split over blockIdx.y (the inner extent alone is only ~N*C/256 = 257 workgroups),
// 8 loads in flight per thread, partial sums combined with fp32 atomics (out is zero-initialised by the caller)
__global__ __launch_bounds__(256) void batchsum_kernel(const float* __restrict__ in, float* __restrict__ out, int B,
                                                       long long inner, int bper) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= inner) return;
    const int b0 = blockIdx.y * bper, b1 = min(B, b0 + bper);
    float s = 0.f;
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = in[(long long)(b + u) * inner + i];
        s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; b < b1; ++b) s += in[(long long)b * inner + i];
    atomicAdd(out + i, s);
}

// ---- patch_output_type == 'avg' (vit_sr_supernet.py:447-449): mean over the patch tokens of a sample, and its backward ----
template <typename T>
__global__ __launch_bounds__(256) void token_mean_kernel(const T* __restrict__ y, T* __restrict__ out, int N, int C, int first) {
    const int b = blockIdx.x;
    const float inv = 1.0f / (float)(N - first);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int n = first; n < N; ++n) s += Elem<T>::ld(y + ((long long)b * N + n) * C + c);
        Elem<T>::st(out + (long long)b * C + c, s * inv);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void token_mean_bwd_kernel(const T* __restrict__ dmean, T* __restrict__ dy, int N, int C,
                                                             int first) {
    const int n = first + blockIdx.x % (N - first), b = blockIdx.x / (N - first);
    const float inv = 1.0f / (float)(N - first);
    for (int c = threadIdx.x; c < C; c += blockDim.x)
        Elem<T>::st(dy + ((long long)b * N + n) * C + c, Elem<T>::ld(dmean + (long long)b * C + c) * inv);
}

// ---- timm PatchEmbed im2col: col[(b,py,px)][(c,i,j)] = img[b,c,py*P+i,px*P+j] ---------------------------
// One workgroup per band of patches (b, py): the band's Cin x P image rows are read once, fully coalesced, into LDS and
// written out as whole patch rows (ldk contiguous elements each) -- a patch row is only P contiguous floats in the image,
// so a gather per patch ran at 0.9 TB/s.
template <typename T>
__global__ __launch_bounds__(256) void im2col_patch_kernel(const float* __restrict__ img, T* __restrict__ col, int B,
                                                           int Cin, int H, int W, int P, int ldk,
                                                           const long long* __restrict__ sample_map) {
    extern __shared__ __attribute__((aligned(16))) float band[];      // [Cin][P][W]
    const int gw = W / P, gh = H / P;
    const int py = blockIdx.x % gh, bo = blockIdx.x / gh;             // output sample bo reads image sample_map[bo]
    const int b = sample_map ? (int)sample_map[bo] : bo;
    const int rows = Cin * P;
    const bool v4 = (W % 4) == 0;
    if (v4) {
        const int w4 = W / 4;
        for (int idx = threadIdx.x; idx < rows * w4; idx += blockDim.x) {
            const int r = idx / w4, x4 = idx % w4;
            const int c = r / P, i = r % P;
            *reinterpret_cast<float4*>(band + r * W + 4 * x4) =
                *reinterpret_cast<const float4*>(img + (((long long)b * Cin + c) * H + py * P + i) * W + 4 * x4);
        }
    } else {
        for (int idx = threadIdx.x; idx < rows * W; idx += blockDim.x) {
            const int r = idx / W, x = idx % W;
            const int c = r / P, i = r % P;
            band[idx] = img[(((long long)b * Cin + c) * H + py * P + i) * W + x];
        }
    }
    __syncthreads();
    const int K = Cin * P * P;
    T* out = col + ((long long)bo * gh + py) * gw * ldk;
    if (sizeof(T) == 2 && ldk % 8 == 0 && (reinterpret_cast<uintptr_t>(col) & 15) == 0) {
        // eight consecutive k per thread, one 16-byte store: (c, i, j) is decoded once per group and stepped with carries -- the
        // element-per-thread form spent ~40 integer instructions (three divisions by run-time values) on every 2-byte store
        const int g8 = ldk / 8;
        for (int q = threadIdx.x; q < gw * g8; q += blockDim.x) {
            const int px = q / g8, k0 = (q - px * g8) * 8;
            int j = k0 % P, r = k0 / P;                  // r = c * P + i: the band row
            const float* src = band + px * P;
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                f[e] = (k0 + e < K) ? src[r * W + j] : 0.f;
                if (++j == P) { j = 0; ++r; }
            }
            *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(out) + (long long)px * ldk + k0) =
                make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7]));
        }
        return;
    }
    for (int idx = threadIdx.x; idx < gw * ldk; idx += blockDim.x) {
        const int px = idx / ldk, k = idx % ldk;
        float v = 0.f;
        if (k < K) {
            const int j = k % P, i = (k / P) % P, c = k / (P * P);
            v = band[(c * P + i) * W + px * P + j];
        }
        Elem<T>::st(out + idx, v);
    }
}

__global__ __launch_bounds__(256) void embed_cls_kernel(const float* __restrict__ tokens, const float* __restrict__ pos,
                                                        float* __restrict__ x, const int* __restrict__ keep, int B, int N,
                                                        int C) {
    const int b = blockIdx.x, t = blockIdx.y;        // token row t (class token, distillation token)
    const int kc = keep ? keep[b] : C;
    for (int c = threadIdx.x; c < C; c += blockDim.x)
        x[((long long)b * N + t) * C + c] = (c < kc) ? tokens[t * C + c] + pos[t * C + c] : 0.f;
}

__global__ __launch_bounds__(256) void mask_rows_kernel(float* __restrict__ x, const int* __restrict__ keep, int M, int C,
                                                        int rps) {
    const int m = blockIdx.x;
    const int kc = keep[m / rps];
    for (int c = kc + threadIdx.x; c < C; c += blockDim.x) x[(long long)m * C + c] = 0.f;
}

// ---- spatial-reduction block pieces (3x3 stride 2 pad 1 conv as GEMM; 2x2 avg-pool residual) ------------
// col[(b,oh,ow)][(kh,kw,c)] = y[b, 1 + (2oh-1+kh)*g + (2ow-1+kw), c]  (0 outside)
template <typename T>
__global__ __launch_bounds__(256) void sr_im2col_kernel(const T* __restrict__ y, T* __restrict__ col, int B, int g, int C, int NT) {
    const int go = g / 2;
    const int row = blockIdx.x;  // (b, oh, ow)
    const int tap = blockIdx.y;  // kh*3+kw
    const int ow = row % go, oh = (row / go) % go, b = row / (go * go);
    const int ih = 2 * oh - 1 + tap / 3, iw = 2 * ow - 1 + tap % 3;
    const bool in = ih >= 0 && ih < g && iw >= 0 && iw < g;
    const T* src = y + ((long long)b * (NT + g * g) + NT + ih * g + iw) * C;
    T* dst = col + (long long)row * 9 * C + tap * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) dst[c] = in ? src[c] : (T)0;
}

// dy[b,1+ih*g+iw,c] = sum over taps (kh,kw) with 2oh-1+kh==ih, 2ow-1+kw==iw of dcol[(b,oh,ow)][(kh,kw,c)]
template <typename T>
__global__ __launch_bounds__(256) void sr_col2im_kernel(const T* __restrict__ dcol, T* __restrict__ dy, int B, int g, int C, int NT) {
    const int go = g / 2;
    const int pix = blockIdx.x;  // (b, ih, iw)
    const int iw = pix % g, ih = (pix / g) % g, b = pix / (g * g);
    T* dst = dy + ((long long)b * (NT + g * g) + NT + ih * g + iw) * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int t = ih + 1 - kh;
            if (t < 0 || (t & 1)) continue;
            const int oh = t >> 1;
            if (oh >= go) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int u = iw + 1 - kw;
                if (u < 0 || (u & 1)) continue;
                const int ow = u >> 1;
                if (ow >= go) continue;
                s += Elem<T>::ld(dcol + ((long long)(b * go + oh) * go + ow) * 9 * C + (kh * 3 + kw) * C + c);
            }
        }
        Elem<T>::st(dst + c, s);
    }
}

// bf16, C % 8 == 0: one 16-byte chunk (8 channels) per thread -- the element-per-thread kernels above launch one 512-byte
// workgroup per (pixel, tap) and were launch-rate bound (0.8 TB/s)
__global__ __launch_bounds__(256) void sr_im2col_v8_kernel(const bf16_t* __restrict__ y, bf16_t* __restrict__ col, int B, int g,
                                                           int C, long long total, int NT) {
    const int go = g / 2, c8n = C / 8;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c8 = (int)(idx % c8n);
        const long long rt = idx / c8n;
        const int tap = (int)(rt % 9);
        const long long row = rt / 9;
        const int ow = (int)(row % go), oh = (int)((row / go) % go), b = (int)(row / (go * go));
        const int ih = 2 * oh - 1 + tap / 3, iw = 2 * ow - 1 + tap % 3;
        const bool in = ih >= 0 && ih < g && iw >= 0 && iw < g;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (in) v = *reinterpret_cast<const uint4*>(y + ((long long)b * (NT + g * g) + NT + ih * g + iw) * C + c8 * 8);
        *reinterpret_cast<uint4*>(col + row * 9 * C + (long long)tap * C + c8 * 8) = v;
    }
}

__global__ __launch_bounds__(256) void sr_col2im_v8_kernel(const bf16_t* __restrict__ dcol, bf16_t* __restrict__ dy, int B, int g,
                                                           int C, long long total, int NT) {
    const int go = g / 2, c8n = C / 8;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c8 = (int)(idx % c8n);
        const long long pix = idx / c8n;
        const int iw = (int)(pix % g), ih = (int)((pix / g) % g), b = (int)(pix / (g * g));
        float s[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int t = ih + 1 - kh;
            const int oh = t >> 1;
            const bool hok = t >= 0 && !(t & 1) && oh < go;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int u = iw + 1 - kw;
                const int ow = u >> 1;
                const bool ok = hok && u >= 0 && !(u & 1) && ow < go;
                if (ok) {       // wave-divergent by pixel parity only; at most 4 of the 9 taps hit
                    const uint4 v = *reinterpret_cast<const uint4*>(dcol + ((long long)(b * go + oh) * go + ow) * 9 * C +
                                                                    (kh * 3 + kw) * C + c8 * 8);
                    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        s[2 * q] += __uint_as_float(w[q] << 16);
                        s[2 * q + 1] += __uint_as_float(w[q] & 0xffff0000u);
                    }
                }
            }
        }
        *reinterpret_cast<uint4*>(dy + ((long long)b * (NT + g * g) + NT + ih * g + iw) * C + c8 * 8) =
            make_uint4(pack_bf2(s[0], s[1]), pack_bf2(s[2], s[3]), pack_bf2(s[4], s[5]), pack_bf2(s[6], s[7]));
    }
}

// out[b,0,:] = pad(x[b,0,:]); out[b,1+(oh,ow),:] = pad(mean of the 2x2 patch rows)
__global__ __launch_bounds__(256) void sr_resid_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int g,
                                                       int Cin, int Cout, int NT) {
    const int go = g / 2, No = NT + go * go, Ni = NT + g * g;
    const int row = blockIdx.x;  // b*No + r
    const int r = row % No, b = row / No;
    float* dst = out + (long long)row * Cout;
    const float* xb = x + (long long)b * Ni * Cin;
    if (r < NT) {                // token rows are copied
        for (int c = threadIdx.x; c < Cout; c += blockDim.x) dst[c] = c < Cin ? xb[(long long)r * Cin + c] : 0.f;
        return;
    }
    const int ow = (r - NT) % go, oh = (r - NT) / go;
    const float* p00 = xb + (long long)(NT + (2 * oh) * g + 2 * ow) * Cin;
    const float* p01 = p00 + Cin;
    const float* p10 = p00 + (long long)g * Cin;
    const float* p11 = p10 + Cin;
    for (int c = threadIdx.x; c < Cout; c += blockDim.x)
        dst[c] = c < Cin ? 0.25f * (p00[c] + p01[c] + p10[c] + p11[c]) : 0.f;
}

__global__ __launch_bounds__(256) void sr_resid_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx, int B,
                                                           int g, int Cin, int Cout, int accumulate, int NT) {
    const int go = g / 2, No = NT + go * go, Ni = NT + g * g;
    const int row = blockIdx.x;  // b*Ni + r  (input rows)
    const int r = row % Ni, b = row / Ni;
    float* dst = dx + (long long)row * Cin;
    const float* src;
    float f;
    if (r < NT) {
        src = dout + ((long long)b * No + r) * Cout;
        f = 1.0f;
    } else {
        const int iw = (r - NT) % g, ih = (r - NT) / g;
        src = dout + ((long long)b * No + NT + (ih / 2) * go + iw / 2) * Cout;
        f = 0.25f;
    }
    for (int c = threadIdx.x; c < Cin; c += blockDim.x) {
        const float v = f * src[c];
        dst[c] = accumulate ? dst[c] + v : v;
    }
}

}  // namespace

extern "C" int vr_version(void) { return 1000; }

extern "C" int vr_cast_f32_bf16(const float* src, void* dst, int64_t n, vr_stream_t stream) {
    if (!src || !dst || n <= 0) return VR_EINVAL;
    if (((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return VR_EALIGN;
    long long blocks = (n / 8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(cast_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, (long long)n);
    VR_CHECK_LAUNCH();
    return VR_OK;
}

extern "C" int vr_cast_transpose_batch(const float* src, void* dst, const vr_tr_desc* descs, int32_t n, int32_t max_tiles,
                                       vr_stream_t stream) {
    if (!src || !dst || !descs || n <= 0 || max_tiles <= 0) return VR_EINVAL;
    hipLaunchKernelGGL(cast_transpose_kernel, dim3((unsigned)max_tiles, (unsigned)n), dim3(256), 0, (hipStream_t)stream, src,
                       (bf16_t*)dst, descs);
    VR_CHECK_LAUNCH();
    return VR_OK;
}

extern "C" int vr_softce(const float* logits, const float* target, float* loss_rows, float* dlogits, int32_t R, int32_t K,
                         float gscale, vr_stream_t stream) {
    if (!logits || !target || !loss_rows || R <= 0 || K <= 0) return VR_EINVAL;
    hipLaunchKernelGGL(softce_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, target, loss_rows,
                       dlogits, R, K, gscale);
    VR_CHECK_LAUNCH();
    return VR_OK;
}

extern "C" int vr_softce_train(const float* logits, const float* target, const int64_t* sample_map, int32_t rows_per_sample,
                               float* loss_acc, void* dlogits, int32_t ld_grad, int32_t grad_dtype, int32_t R, int32_t K,
                               float gscale, float loss_scale, vr_stream_t stream) {
    if (!logits || !target || !loss_acc || !dlogits || R <= 0 || K <= 0 || rows_per_sample <= 0 || ld_grad < K) return VR_EINVAL;
    const dim3 grid((R + 3) / 4);
    const bool reg = K % 4 == 0 && K <= 1024 && ld_grad % 4 == 0 && ld_grad <= 1024 && !((uintptr_t)logits & 15) &&
                     !((uintptr_t)target & 15) && !((uintptr_t)dlogits & 15);
    if (reg && grad_dtype == VR_F32) {
        hipLaunchKernelGGL((softce_train_reg_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, logits, target,
                           (const long long*)sample_map, rows_per_sample, loss_acc, (float*)dlogits, ld_grad, R, K, gscale, loss_scale);
    } else if (reg && grad_dtype == VR_BF16) {
        hipLaunchKernelGGL((softce_train_reg_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, logits, target,
                           (const long long*)sample_map, rows_per_sample, loss_acc, (bf16_t*)dlogits, ld_grad, R, K, gscale, loss_scale);
    } else if (grad_dtype == VR_F32)
        hipLaunchKernelGGL((softce_train_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, logits, target,
                           (const long long*)sample_map, rows_per_sample, loss_acc, (float*)dlogits, ld_grad, R, K, gscale, loss_scale);
    else if (grad_dtype == VR_BF16)
        hipLaunchKernelGGL((softce_train_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, logits, target,
                           (const long long*)sample_map, rows_per_sample, loss_acc, (bf16_t*)dlogits, ld_grad, R, K, gscale, loss_scale);
    else
        return VR_EUNSUPPORTED;
    VR_CHECK_LAUNCH();
    return VR_OK;
}

extern "C" int vr_colsum(const void* in, float* out, int32_t M, int32_t N, int32_t ld, int32_t dtype, vr_rowmap map,
                         vr_stream_t stream) {
    if (!in || !out || M <= 0 || N <= 0) return VR_EINVAL;
    const int rpb = 128;
    dim3 grid((N + 255) / 256, (M + rpb - 1) / rpb);
    const RowMap rm = {map.rpi, map.rps, map.off};
    if (dtype == VR_F32)
        hipLaunchKernelGGL((colsum_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)in, out, M, N, ld, rpb, rm);
    else if (dtype == VR_BF16)
        hipLaunchKernelGGL((colsum_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, out, M, N, ld, rpb, rm);
    else
        return VR_EUNSUPPORTED;
    VR_CHECK_LAUNCH();
    return VR_OK;
}

extern "C" int vr_scale_mask_cast(const float* in, void* out, const float* scale, const int32_t* keep, int32_t M, int32_t C,
                                  int32_t rows_per_sample, int32_t out_dtype, vr_stream_t stream) {
    if (!in || !out || M <= 0 || C <= 0 || C % 4) return VR_EINVAL;
    if (rows_per_sample <= 0) rows_per_sample = M;
    dim3 grid((M + 3) / 4);
    if (out_dtype == VR_F32)
        hipLaunchKernelGGL((scale_mask_cast_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, in, (float*)out, scale, keep, M, C, rows_per_sample);
    else if (out_dtype == VR_BF16)
        hipLaunchKernelGGL((scale_mask_cast_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)out, scale, keep, M, C, rows_per_sample);
    else
        return VR_EUNSUPPORTED;
    VR_CHECK_LAUNCH();
    return VR_OK;
}

extern "C" int vr_token_mean(const void* y, void* out, int32_t B, int32_t N, int32_t C, int32_t first, int32_t dtype,
                             vr_stream_t stream) {
    if (!y || !out || B <= 0 || C <= 0 || first < 0 || first >= N) return VR_EINVAL;
    if (dtype == VR_F32)
        hipLaunchKernelGGL((token_mean_kernel<float>), dim3(B), dim3(256), 0, (hipStream_t)stream, (const float*)y, (float*)out, N, C, first);
    else if (dtype == VR_BF16)
        hipLaunchKernelGGL((token_mean_kernel<bf16_t>), dim3(B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)y, (bf16_t*)out, N, C, first);
    else
        return VR_EUNSUPPORTED;
    VR_CHECK_LAUNCH();
    return VR_OK;
}

extern "C" int vr_token_mean_bwd(const void* dmean, void* dy, int32_t B, int32_t N, int32_t C, int32_t first, int32_t dtype,
                                 vr_stream_t stream) {
    if (!dmean || !dy || B <= 0 || C <= 0 || first < 0 || first >= N) return VR_EINVAL;
    dim3 grid(B * (N - first));
    if (dtype == VR_F32)
        hipLaunchKernelGGL((token_mean_bwd_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)dmean, (float*)dy, N, C, first);
    else if (dtype == VR_BF16)
        hipLaunchKernelGGL((token_mean_bwd_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dmean, (bf16_t*)dy, N, C, first);
    else
        return VR_EUNSUPPORTED;
    VR_CHECK_LAUNCH();
    return VR_OK;
}

extern "C" int vr_batchsum(const float* in, float* out, int32_t B, int64_t inner, vr_stream_t stream) {
    if (!in || !out || B <= 0 || inner <= 0) return VR_EINVAL;
    const int bper = B >= 64 ? 16 : (B >= 16 ? 8 : B);
    hipLaunchKernelGGL(batchsum_kernel, dim3((unsigned)((inner + 255) / 256), (unsigned)((B + bper - 1) / bper)), dim3(256), 0,
                       (hipStream_t)stream, in, out, B, (long long)inner, bper);
    VR_CHECK_LAUNCH();
    return VR_OK;
}

extern "C" int vr_im2col_patch(const float* img, void* col, int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t P,
                               int32_t ldk, int32_t dtype, vr_stream_t stream) {
    return vr_im2col_patch_map(img, col, nullptr, B, Cin, H, W, P, ldk, dtype, stream);
}

extern "C" int vr_im2col_patch_map(const float* img, void* col, const int64_t* sample_map, int32_t B, int32_t Cin, int32_t H,
                                   int32_t W, int32_t P, int32_t ldk, int32_t dtype, vr_stream_t stream) {
    if (!img || !col || B <= 0 || P <= 0 || H % P || W % P || ldk < Cin * P * P) return VR_EINVAL;
    const long long* smap = reinterpret_cast<const long long*>(sample_map);
    dim3 grid(B * (H / P));
    const size_t lds = (size_t)Cin * P * W * sizeof(float);
    if (lds > 64 * 1024) return VR_EUNSUPPORTED;
    if (dtype == VR_F32)
        hipLaunchKernelGGL((im2col_patch_kernel<float>), grid, dim3(256), lds, (hipStream_t)stream, img, (float*)col, B, Cin, H, W, P, ldk, smap);
    else if (dtype == VR_BF16)
        hipLaunchKernelGGL((im2col_patch_kernel<bf16_t>), grid, dim3(256), lds, (hipStream_t)stream, img, (bf16_t*)col, B, Cin, H, W, P, ldk, smap);
    else
        return VR_EUNSUPPORTED;
    VR_CHECK_LAUNCH();
    return VR_OK;
}

extern "C" int vr_embed_cls(const float* tokens, const float* pos, float* x, const int32_t* keep, int32_t B, int32_t N,
                            int32_t C, int32_t num_tokens, vr_stream_t stream) {
    if (!tokens || !pos || !x || B <= 0 || num_tokens < 1 || num_tokens > N) return VR_EINVAL;
    hipLaunchKernelGGL(embed_cls_kernel, dim3(B, num_tokens), dim3(256), 0, (hipStream_t)stream, tokens, pos, x, keep, B, N, C);
    VR_CHECK_LAUNCH();
    return VR_OK;
}

extern "C" int vr_mask_rows(float* x, const int32_t* keep, int32_t M, int32_t C, int32_t rows_per_sample, vr_stream_t stream) {
    if (!x || !keep || M <= 0 || rows_per_sample <= 0) return VR_EINVAL;
    hipLaunchKernelGGL(mask_rows_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, x, keep, M, C, rows_per_sample);
    VR_CHECK_LAUNCH();
    return VR_OK;
}

extern "C" int vr_sr_im2col(const void* y, void* col, int32_t B, int32_t g, int32_t C, int32_t num_tokens, int32_t dtype,
                            vr_stream_t stream) {
    if (!y || !col || B <= 0 || g <= 0 || (g & 1) || num_tokens < 1) return VR_EINVAL;
    const int NT = num_tokens;
    dim3 grid(B * (g / 2) * (g / 2), 9);
    if (dtype == VR_BF16 && C % 8 == 0 && !((uintptr_t)y & 15) && !((uintptr_t)col & 15)) {
        const long long total = (long long)B * (g / 2) * (g / 2) * 9 * (C / 8);
        const long long blocks = (total + 255) / 256;
        hipLaunchKernelGGL(sr_im2col_v8_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)y, (bf16_t*)col, B, g, C, total, NT);
    } else if (dtype == VR_F32)
        hipLaunchKernelGGL((sr_im2col_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)y, (float*)col, B, g, C, NT);
    else if (dtype == VR_BF16)
        hipLaunchKernelGGL((sr_im2col_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)y, (bf16_t*)col, B, g, C, NT);
    else
        return VR_EUNSUPPORTED;
    VR_CHECK_LAUNCH();
    return VR_OK;
}

extern "C" int vr_sr_col2im(const void* dcol, void* dy, int32_t B, int32_t g, int32_t C, int32_t num_tokens, int32_t dtype,
                            vr_stream_t stream) {
    if (!dcol || !dy || B <= 0 || g <= 0 || (g & 1) || num_tokens < 1) return VR_EINVAL;
    const int NT = num_tokens;
    dim3 grid(B * g * g);
    if (dtype == VR_BF16 && C % 8 == 0 && !((uintptr_t)dcol & 15) && !((uintptr_t)dy & 15)) {
        const long long total = (long long)B * g * g * (C / 8);
        const long long blocks = (total + 255) / 256;
        hipLaunchKernelGGL(sr_col2im_v8_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)dcol, (bf16_t*)dy, B, g, C, total, NT);
    } else if (dtype == VR_F32)
        hipLaunchKernelGGL((sr_col2im_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)dcol, (float*)dy, B, g, C, NT);
    else if (dtype == VR_BF16)
        hipLaunchKernelGGL((sr_col2im_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dcol, (bf16_t*)dy, B, g, C, NT);
    else
        return VR_EUNSUPPORTED;
    VR_CHECK_LAUNCH();
    return VR_OK;
}

extern "C" int vr_sr_resid(const float* x, float* out, int32_t B, int32_t g, int32_t Cin, int32_t Cout, int32_t num_tokens,
                           vr_stream_t stream) {
    if (!x || !out || B <= 0 || g <= 0 || (g & 1) || Cout < Cin || num_tokens < 1) return VR_EINVAL;
    hipLaunchKernelGGL(sr_resid_kernel, dim3(B * (num_tokens + (g / 2) * (g / 2))), dim3(256), 0, (hipStream_t)stream, x, out, B, g, Cin,
                       Cout, num_tokens);
    VR_CHECK_LAUNCH();
    return VR_OK;
}

extern "C" int vr_sr_resid_bwd(const float* dout, float* dx, int32_t B, int32_t g, int32_t Cin, int32_t Cout,
                               int32_t accumulate, int32_t num_tokens, vr_stream_t stream) {
    if (!dout || !dx || B <= 0 || g <= 0 || (g & 1) || Cout < Cin || num_tokens < 1) return VR_EINVAL;
    hipLaunchKernelGGL(sr_resid_bwd_kernel, dim3(B * (num_tokens + g * g)), dim3(256), 0, (hipStream_t)stream, dout, dx, B, g, Cin, Cout,
                       accumulate, num_tokens);
    VR_CHECK_LAUNCH();
    return VR_OK;
}

// ---- vr_zero_ranges: zero-fill up to VR_MAX_ZERO_RANGES ranges of one fp32 buffer in ONE launch ----
// The gradient arena is accumulated into by atomics (weight gradients split over tokens, LayerNorm partial rows ...) and must
// start at zero -- except the spans whose weight gradients are written in store form (vr_gemm atomic == 2): the ranges in
// between are what this kernel clears (reference: optimizer.zero_grad(), engine.py:175).
namespace {
__global__ __launch_bounds__(256) void zero_ranges_kernel(float* base, const vr_range_list r) {
    const long long lo = r.lo[blockIdx.y], n = r.count[blockIdx.y];
    float* p = base + lo;
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    // head up to the first 16-byte boundary, float4 body, scalar tail
    const long long head = min(n, (long long)((4 - ((reinterpret_cast<uintptr_t>(p) >> 2) & 3)) & 3));
    if (i < head) p[i] = 0.f;
    float4* p4 = reinterpret_cast<float4*>(p + head);
    const long long n4 = (n - head) >> 2;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    long long k = i;
    for (; k + 3 * stride < n4; k += 4 * stride) {             // four 16-byte stores in flight per thread
        p4[k] = z;
        p4[k + stride] = z;
        p4[k + 2 * stride] = z;
        p4[k + 3 * stride] = z;
    }
    for (; k < n4; k += stride) p4[k] = z;
    const long long tail0 = head + (n4 << 2);
    if (tail0 + i < n && i < 4) p[tail0 + i] = 0.f;
}
}  // namespace

extern "C" int vr_zero_ranges(float* base, const vr_range_list* ranges, vr_stream_t stream) {
    if (!base || !ranges || ranges->n <= 0 || ranges->n > VR_MAX_ZERO_RANGES) return VR_EINVAL;
    long long most = 0;
    for (int i = 0; i < ranges->n; ++i) {
        if (ranges->lo[i] < 0 || ranges->count[i] < 0) return VR_EINVAL;
        most = ranges->count[i] > most ? ranges->count[i] : most;
    }
    if (most == 0) return VR_OK;
    long long bx = (most / 16 + 255) / 256;
    bx = bx < 1 ? 1 : (bx > 4096 ? 4096 : bx);
    hipLaunchKernelGGL(zero_ranges_kernel, dim3((unsigned)bx, (unsigned)ranges->n), dim3(256), 0, (hipStream_t)stream, base, *ranges);
    VR_CHECK_LAUNCH();
    return VR_OK;
}

// ---- vr_relayout: dst[a * dst_ld + c * B + b] = src[a * src_ld + b * C + c]  (fp32 / bf16 in, fp32 / bf16 out) ----
// The small re-layouts around the convolution-shaped weights: [out][in][taps] -> [out][(taps, in)] for the GEMM form of the 3x3 /
// 7x7 convolutions (nets/patch_conv.py:56-58, vit_sr_supernet.py:140) and back for their gradients; with B = 1 a row copy into
// 16-byte-aligned rows (the timm PatchEmbed weight, k = 588 -> ld = 592).  Pad columns of dst are not touched.
namespace {
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void relayout_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int A, int B, int C, long long src_ld,
                                                       long long dst_ld) {
    const long long total = (long long)A * B * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int b = (int)(i % B);                      // destination order: consecutive threads write consecutive elements
        const long long r = i / B;
        const int c = (int)(r % C);
        const int a = (int)(r / C);
        Elem<TD>::st(dst + a * dst_ld + (long long)c * B + b, Elem<TS>::ld(src + a * src_ld + (long long)b * C + c));
    }
}
}  // namespace

extern "C" int vr_relayout(const void* src, void* dst, int32_t A, int32_t B, int32_t C, int64_t src_ld, int64_t dst_ld, int32_t src_dtype,
                           int32_t dst_dtype, vr_stream_t stream) {
    if (!src || !dst || A <= 0 || B <= 0 || C <= 0 || dst_ld < (int64_t)B * C || src_ld < (int64_t)B * C) return VR_EINVAL;
    const long long total = (long long)A * B * C;
    const unsigned grid = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    if (src_dtype == VR_F32 && dst_dtype == VR_F32)
        hipLaunchKernelGGL((relayout_kernel<float, float>), dim3(grid), dim3(256), 0, st, (const float*)src, (float*)dst, A, B, C, src_ld, dst_ld);
    else if (src_dtype == VR_F32 && dst_dtype == VR_BF16)
        hipLaunchKernelGGL((relayout_kernel<float, bf16_t>), dim3(grid), dim3(256), 0, st, (const float*)src, (bf16_t*)dst, A, B, C, src_ld, dst_ld);
    else if (src_dtype == VR_BF16 && dst_dtype == VR_BF16)
        hipLaunchKernelGGL((relayout_kernel<bf16_t, bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, A, B, C, src_ld, dst_ld);
    else if (src_dtype == VR_BF16 && dst_dtype == VR_F32)
        hipLaunchKernelGGL((relayout_kernel<bf16_t, float>), dim3(grid), dim3(256), 0, st, (const bf16_t*)src, (float*)dst, A, B, C, src_ld, dst_ld);
    else
        return VR_EUNSUPPORTED;
    VR_CHECK_LAUNCH();
    return VR_OK;
}

// ---- vr_conv_w_flip: weights of the data-gradient convolution of a 3x3 / stride 1 / pad 1 Conv2d ------------------------------
//   dst[ci, (kh, kw, co)] = src[co, ci, 2 - kh, 2 - kw]   (src fp32 [Co, Ci, 3, 3]; dst fp32 / bf16 [Ci, 9 * Co])
namespace {
template <typename TD>
__global__ __launch_bounds__(256) void conv_w_flip_kernel(const float* __restrict__ src, TD* __restrict__ dst, int Co, int Ci) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Co * Ci * 9) return;
    const int co = i % Co, tap = (i / Co) % 9, ci = i / (9 * Co);
    Elem<TD>::st(dst + i, src[((long long)co * Ci + ci) * 9 + (8 - tap)]);
}
}  // namespace

extern "C" int vr_conv_w_flip(const float* src, void* dst, int32_t Co, int32_t Ci, int32_t dst_dtype, vr_stream_t stream) {
    if (!src || !dst || Co <= 0 || Ci <= 0) return VR_EINVAL;
    const unsigned grid = (unsigned)((Co * Ci * 9 + 255) / 256);
    if (dst_dtype == VR_F32) hipLaunchKernelGGL(conv_w_flip_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (float*)dst, Co, Ci);
    else if (dst_dtype == VR_BF16) hipLaunchKernelGGL(conv_w_flip_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, Co, Ci);
    else return VR_EUNSUPPORTED;
    VR_CHECK_LAUNCH();
    return VR_OK;
}
