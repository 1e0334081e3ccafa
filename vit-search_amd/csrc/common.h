// Common device helpers for the vitres HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VR_F32 0
#define VR_BF16 1

#define VR_OK 0
#define VR_EINVAL (-1)
#define VR_EALIGN (-2)
#define VR_EUNSUPPORTED (-3)

#define VR_CHECK_LAUNCH()                                   \
    do {                                                    \
        hipError_t _e = hipGetLastError();                  \
        if (_e != hipSuccess) return (int)_e;               \
    } while (0)

typedef uint16_t bf16_t;  // raw bf16 bits

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// round-to-nearest-even, NaN-preserving (same rounding as torch's .bfloat16())
__device__ __forceinline__ bf16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)0x7fc0;
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// two floats -> packed bf16 pair with the gfx950 converter (v_cvt_pk_bf16_f32: round-to-nearest-even, one instruction)
typedef __bf16 vr_bf2 __attribute__((ext_vector_type(2)));
typedef float vr_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const vr_f2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, vr_bf2));
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int dtype = VR_F32;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int dtype = VR_BF16;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// XCD-contiguous block order (round 4): workgroups go to the eight XCDs round-robin (block b -> XCD b % 8); xcd_block() gives
// block b the work item x * (n / 8) + ... + b / 8, so XCD x owns ONE contiguous range of the n items -- the rows (or samples) the
// GEMMs' XCD-aware tile order gives the same XCD.  A kernel then finds what the previous kernel wrote to those rows in its own
// XCD's L2 (tools/probes/xcd_reuse_probe.hip: a 32 MB tensor written by one kernel is read back in 6.2 us by the XCD that wrote it,
// 8.9 us by another one -- the L2 keeps the lines across the kernel boundary).
__device__ __forceinline__ int xcd_block(int bid, int n, int on) {
    if (!on || n < 16) return bid;
    const int q = n >> 3, r = n & 7, x = bid & 7;
    return x * q + (x < r ? x : r) + (bid >> 3);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exact (erf) GELU and its derivative, as nn.GELU() default (reference nets/supernet_blocks.py:18)
__device__ __forceinline__ float gelu_f(float u) { return 0.5f * u * (1.0f + erff(u * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float u) {
    const float cdf = 0.5f * (1.0f + erff(u * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * __expf(-0.5f * u * u);
    return cdf + u * pdf;
}

// GELU for bf16 outputs: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below bf16 resolution) -- one rcp, one
// exp and 7 fma instead of libm's erff; the exp is shared with the derivative's pdf term.  Tails are evaluated without
// cancellation (Phi(u) = 0.5 poly e for u < 0).
__device__ __forceinline__ void gelu_terms_fast(float u, float& cdf, float& pdf) {
    const float x = fabsf(u) * 0.70710678118654752f;
    const float tt = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
    const float e = __expf(-x * x);
    float poly = fmaf(1.061405429f, tt, -1.453152027f);
    poly = fmaf(poly, tt, 1.421413741f);
    poly = fmaf(poly, tt, -0.284496736f);
    poly = fmaf(poly, tt, 0.254829592f);
    const float half_tail = 0.5f * poly * tt * e;            // 0.5 * erfc(|x|)
    cdf = u >= 0.f ? 1.0f - half_tail : half_tail;
    pdf = 0.39894228040143268f * e;
}
__device__ __forceinline__ float gelu_fast(float u) {
    float c, d;
    gelu_terms_fast(u, c, d);
    return u * c;
}
__device__ __forceinline__ float dgelu_fast(float u) {
    float c, d;
    gelu_terms_fast(u, c, d);
    return fmaf(u, d, c);
}

// token-row remap:  row(m) = (m / rpi) * rps + off + (m % rpi) ; rpi == 0 -> identity
struct RowMap {
    int rpi, rps, off;
};
__device__ __forceinline__ long long map_row(const RowMap& r, int m) {
    if (r.rpi == 0) return m;
    const int s = m / r.rpi;
    return (long long)s * r.rps + r.off + (m - s * r.rpi);
}
