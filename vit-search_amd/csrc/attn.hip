// Multi-head self-attention core, exact-fp32 VALU version (reference nets/supernet_blocks.py:105-109).
//
// Used as the parity path (dtype fp32) and as the fallback for shapes the MFMA kernel does not cover.
// One workgroup per (sample, head): the head's K and V (N <= ~290 tokens, D <= 64) live in LDS as fp32
// with rows padded to D+1 words (lane j reads row j -> bank (j + d) mod 32: conflict free); each wave
// owns query rows q = wave, wave+4, ...; scores are lane-parallel over keys, the PV product is
// lane-parallel over the head dimension with p broadcast from LDS.
#include <cstdlib>

#include "common.h"
#include "../../include/vitres_hip.h"

namespace vr_attn_mfma {   // attn_mfma.hip
bool supported(int N, int H, int D);
int fwd(const void* qkv, void* o, float* lse, const int* keep, int B, int N, int H, int D, float scale, hipStream_t st);
int bwd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv, const int* keep, int B,
        int N, int H, int D, float scale, hipStream_t st);
}  // namespace vr_attn_mfma

namespace {

inline bool use_mfma(int dtype, int N, int H, int D) { return dtype == VR_BF16 && vr_attn_mfma::supported(N, H, D); }

constexpr int MAXT = 5;  // key groups of 64 per lane -> N <= 320

template <typename T>
__device__ __forceinline__ void load_head_rows(float* dst, const T* __restrict__ src, int N, int D, int row_stride, int tid,
                                               int nthr) {
    // dst[n][D+1] <- src[n*row_stride + d]
    for (int i = tid; i < N * D; i += nthr) {
        const int n = i / D, d = i - n * D;
        dst[n * (D + 1) + d] = Elem<T>::ld(src + (long long)n * row_stride + d);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ o, float* __restrict__ lse,
                                                       const int* __restrict__ keep_hd, int B, int N, int H, int D,
                                                       float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HD = H * D, RS = 3 * HD;
    const T* base = qkv + (long long)b * N * RS;
    if (keep_hd && h * D >= keep_hd[b]) {   // dropped head: zeros (Attention's ChannelDrop)
        for (int i = tid; i < N * D; i += 256) {
            const int n = i / D, d = i - n * D;
            Elem<T>::st(o + ((long long)b * N + n) * HD + h * D + d, 0.f);
        }
        for (int n = tid; n < N; n += 256) lse[((long long)b * H + h) * N + n] = 0.f;
        return;
    }
    float* Ks = sm;
    float* Vs = Ks + N * (D + 1);
    float* qb = Vs + N * (D + 1) + wave * D;
    float* pb = Vs + N * (D + 1) + 4 * D + wave * (MAXT * 64);
    load_head_rows(Ks, base + HD + h * D, N, D, RS, tid, 256);
    load_head_rows(Vs, base + 2 * HD + h * D, N, D, RS, tid, 256);
    __syncthreads();
    for (int q = wave; q < N; q += 4) {
        if (lane < D) qb[lane] = Elem<T>::ld(base + (long long)q * RS + h * D + lane) * scale;
        __builtin_amdgcn_wave_barrier();
        float s[MAXT];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            const int j = lane + 64 * t;
            s[t] = -INFINITY;
            if (j < N) {
                float a = 0.f;
                const float* kr = Ks + j * (D + 1);
                for (int d = 0; d < D; ++d) a = fmaf(qb[d], kr[d], a);
                s[t] = a;
                mx = fmaxf(mx, a);
            }
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            const int j = lane + 64 * t;
            const float p = (j < N) ? __expf(s[t] - mx) : 0.f;
            sum += p;
            pb[j] = p;
        }
        sum = wave_sum(sum);
        __builtin_amdgcn_wave_barrier();
        if (lane < D) {
            float acc = 0.f;
            for (int j = 0; j < N; ++j) acc = fmaf(pb[j], Vs[j * (D + 1) + lane], acc);
            Elem<T>::st(o + ((long long)b * N + q) * HD + h * D + lane, acc / sum);
        }
        if (lane == 0) lse[((long long)b * H + h) * N + q] = mx + __logf(sum);
        __builtin_amdgcn_wave_barrier();
    }
}

// dQ (and delta = rowsum(dO * O)) : same structure as the forward
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ o,
                                                          const T* __restrict__ d_o, const float* __restrict__ lse,
                                                          float* __restrict__ delta, T* __restrict__ dqkv,
                                                          const int* __restrict__ keep_hd, int B, int N, int H, int D,
                                                          float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HD = H * D, RS = 3 * HD;
    const T* base = qkv + (long long)b * N * RS;
    T* dbase = dqkv + (long long)b * N * RS;
    if (keep_hd && h * D >= keep_hd[b]) {
        for (int i = tid; i < N * D; i += 256) {
            const int n = i / D, d = i - n * D;
            Elem<T>::st(dbase + (long long)n * RS + h * D + d, 0.f);
        }
        return;
    }
    float* Ks = sm;
    float* Vs = Ks + N * (D + 1);
    float* qb = Vs + N * (D + 1) + wave * 2 * D;
    float* gb = qb + D;
    float* pb = Vs + N * (D + 1) + 8 * D + wave * (MAXT * 64);
    load_head_rows(Ks, base + HD + h * D, N, D, RS, tid, 256);
    load_head_rows(Vs, base + 2 * HD + h * D, N, D, RS, tid, 256);
    __syncthreads();
    for (int q = wave; q < N; q += 4) {
        float dl = 0.f;
        if (lane < D) {
            qb[lane] = Elem<T>::ld(base + (long long)q * RS + h * D + lane) * scale;
            const float g = Elem<T>::ld(d_o + ((long long)b * N + q) * HD + h * D + lane);
            gb[lane] = g;
            dl = g * Elem<T>::ld(o + ((long long)b * N + q) * HD + h * D + lane);
        }
        dl = wave_sum(dl);
        const float l = lse[((long long)b * H + h) * N + q];
        if (lane == 0) delta[((long long)b * H + h) * N + q] = dl;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            const int j = lane + 64 * t;
            float ds = 0.f;
            if (j < N) {
                float a = 0.f, dp = 0.f;
                const float* kr = Ks + j * (D + 1);
                const float* vr = Vs + j * (D + 1);
                for (int d = 0; d < D; ++d) {
                    a = fmaf(qb[d], kr[d], a);
                    dp = fmaf(gb[d], vr[d], dp);
                }
                ds = __expf(a - l) * (dp - dl) * scale;
            }
            pb[j] = ds;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < D) {
            float acc = 0.f;
            for (int j = 0; j < N; ++j) acc = fmaf(pb[j], Ks[j * (D + 1) + lane], acc);
            Elem<T>::st(dbase + (long long)q * RS + h * D + lane, acc);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// dK, dV : roles swapped -- Q and dO of the head live in LDS, each wave owns key rows
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const T* __restrict__ qkv, const T* __restrict__ d_o,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           T* __restrict__ dqkv, const int* __restrict__ keep_hd, int B,
                                                           int N, int H, int D, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HD = H * D, RS = 3 * HD;
    const T* base = qkv + (long long)b * N * RS;
    T* dbase = dqkv + (long long)b * N * RS;
    if (keep_hd && h * D >= keep_hd[b]) {
        for (int i = tid; i < N * D; i += 256) {
            const int n = i / D, d = i - n * D;
            Elem<T>::st(dbase + (long long)n * RS + HD + h * D + d, 0.f);
            Elem<T>::st(dbase + (long long)n * RS + 2 * HD + h * D + d, 0.f);
        }
        return;
    }
    float* Qs = sm;
    float* Gs = Qs + N * (D + 1);
    float* Ls = Gs + N * (D + 1);
    float* Ds = Ls + MAXT * 64;
    float* kb = Ds + MAXT * 64 + wave * 2 * D;
    float* vb = kb + D;
    float* pb = Ds + MAXT * 64 + 8 * D + wave * (2 * MAXT * 64);
    float* sb = pb + MAXT * 64;
    load_head_rows(Qs, base + h * D, N, D, RS, tid, 256);
    load_head_rows(Gs, d_o + (long long)b * N * HD + h * D, N, D, HD, tid, 256);
    for (int n = tid; n < MAXT * 64; n += 256) {
        Ls[n] = n < N ? lse[((long long)b * H + h) * N + n] : 0.f;
        Ds[n] = n < N ? delta[((long long)b * H + h) * N + n] : 0.f;
    }
    __syncthreads();
    for (int j = wave; j < N; j += 4) {
        if (lane < D) {
            kb[lane] = Elem<T>::ld(base + (long long)j * RS + HD + h * D + lane) * scale;
            vb[lane] = Elem<T>::ld(base + (long long)j * RS + 2 * HD + h * D + lane);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            const int i = lane + 64 * t;
            float p = 0.f, ds = 0.f;
            if (i < N) {
                float a = 0.f, dp = 0.f;
                const float* qr = Qs + i * (D + 1);
                const float* gr = Gs + i * (D + 1);
                for (int d = 0; d < D; ++d) {
                    a = fmaf(qr[d], kb[d], a);
                    dp = fmaf(gr[d], vb[d], dp);
                }
                p = __expf(a - Ls[i]);
                ds = p * (dp - Ds[i]) * scale;
            }
            pb[i] = p;
            sb[i] = ds;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < D) {
            float dv = 0.f, dk = 0.f;
            for (int i = 0; i < N; ++i) {
                dv = fmaf(pb[i], Gs[i * (D + 1) + lane], dv);
                dk = fmaf(sb[i], Qs[i * (D + 1) + lane], dk);
            }
            Elem<T>::st(dbase + (long long)j * RS + HD + h * D + lane, dk);
            Elem<T>::st(dbase + (long long)j * RS + 2 * HD + h * D + lane, dv);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

inline size_t fwd_lds(int N, int D) { return sizeof(float) * ((size_t)2 * N * (D + 1) + 4 * D + 4 * MAXT * 64); }
inline size_t dq_lds(int N, int D) { return sizeof(float) * ((size_t)2 * N * (D + 1) + 8 * D + 4 * MAXT * 64); }
inline size_t dkv_lds(int N, int D) {
    return sizeof(float) * ((size_t)2 * N * (D + 1) + 2 * MAXT * 64 + 8 * D + 8 * MAXT * 64);
}
constexpr size_t LDS_MAX = 160 * 1024;

template <typename K>
int set_lds(K kernel, size_t bytes) {
    static size_t granted = 0;     // one instance per kernel type K; the driver call is idempotent
    if (bytes > 64 * 1024 && bytes > granted) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)bytes);
        if (e != hipSuccess) return (int)e;
        granted = bytes;
    }
    return 0;
}

}  // namespace

extern "C" int vr_attn_fwd(const void* qkv, void* o, float* lse, const int32_t* keep_hd, int32_t B, int32_t N, int32_t H,
                           int32_t D, float scale, int32_t dtype, vr_stream_t stream) {
    if (!qkv || !o || !lse || B <= 0 || N <= 0 || H <= 0 || D <= 0) return VR_EINVAL;
    if (use_mfma(dtype, N, H, D)) {
        const int rc = vr_attn_mfma::fwd(qkv, o, lse, keep_hd, B, N, H, D, scale, (hipStream_t)stream);
        if (rc) return rc;
        VR_CHECK_LAUNCH();
        return VR_OK;
    }
    if (D > 64 || N > MAXT * 64 || fwd_lds(N, D) > LDS_MAX) return VR_EUNSUPPORTED;
    const size_t lds = fwd_lds(N, D);
    int rc;
    if (dtype == VR_F32) {
        if ((rc = set_lds(attn_fwd_kernel<float>, lds))) return rc;
        hipLaunchKernelGGL((attn_fwd_kernel<float>), dim3(B * H), dim3(256), lds, (hipStream_t)stream, (const float*)qkv,
                           (float*)o, lse, keep_hd, B, N, H, D, scale);
    } else if (dtype == VR_BF16) {
        if ((rc = set_lds(attn_fwd_kernel<bf16_t>, lds))) return rc;
        hipLaunchKernelGGL((attn_fwd_kernel<bf16_t>), dim3(B * H), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)qkv,
                           (bf16_t*)o, lse, keep_hd, B, N, H, D, scale);
    } else {
        return VR_EUNSUPPORTED;
    }
    VR_CHECK_LAUNCH();
    return VR_OK;
}

extern "C" int vr_attn_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv,
                           const int32_t* keep_hd, int32_t B, int32_t N, int32_t H, int32_t D, float scale, int32_t dtype,
                           vr_stream_t stream) {
    if (!qkv || !o || !d_o || !lse || !delta || !dqkv || B <= 0 || N <= 0 || H <= 0 || D <= 0) return VR_EINVAL;
    if (use_mfma(dtype, N, H, D)) {
        const int rc = vr_attn_mfma::bwd(qkv, o, d_o, lse, delta, dqkv, keep_hd, B, N, H, D, scale, (hipStream_t)stream);
        if (rc) return rc;
        VR_CHECK_LAUNCH();
        return VR_OK;
    }
    if (D > 64 || N > MAXT * 64 || dkv_lds(N, D) > LDS_MAX) return VR_EUNSUPPORTED;
    const size_t l1 = dq_lds(N, D), l2 = dkv_lds(N, D);
    int rc;
    if (dtype == VR_F32) {
        if ((rc = set_lds(attn_bwd_dq_kernel<float>, l1))) return rc;
        if ((rc = set_lds(attn_bwd_dkv_kernel<float>, l2))) return rc;
        hipLaunchKernelGGL((attn_bwd_dq_kernel<float>), dim3(B * H), dim3(256), l1, (hipStream_t)stream, (const float*)qkv,
                           (const float*)o, (const float*)d_o, lse, delta, (float*)dqkv, keep_hd, B, N, H, D, scale);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<float>), dim3(B * H), dim3(256), l2, (hipStream_t)stream, (const float*)qkv,
                           (const float*)d_o, lse, delta, (float*)dqkv, keep_hd, B, N, H, D, scale);
    } else if (dtype == VR_BF16) {
        if ((rc = set_lds(attn_bwd_dq_kernel<bf16_t>, l1))) return rc;
        if ((rc = set_lds(attn_bwd_dkv_kernel<bf16_t>, l2))) return rc;
        hipLaunchKernelGGL((attn_bwd_dq_kernel<bf16_t>), dim3(B * H), dim3(256), l1, (hipStream_t)stream, (const bf16_t*)qkv,
                           (const bf16_t*)o, (const bf16_t*)d_o, lse, delta, (bf16_t*)dqkv, keep_hd, B, N, H, D, scale);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<bf16_t>), dim3(B * H), dim3(256), l2, (hipStream_t)stream, (const bf16_t*)qkv,
                           (const bf16_t*)d_o, lse, delta, (bf16_t*)dqkv, keep_hd, B, N, H, D, scale);
    } else {
        return VR_EUNSUPPORTED;
    }
    VR_CHECK_LAUNCH();
    return VR_OK;
}
