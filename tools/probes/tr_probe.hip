// dev probe: semantics of ds_read_b64_tr_b16 (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned long long* y) {
    __shared__ unsigned short lds[64 * 128];          // [token 0..63][channel 0..127], value = token * 256 + channel
    for (int i = threadIdx.x; i < 64 * 128; i += 64) lds[i] = (unsigned short)(((i / 128) << 8) | (i % 128));
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, i = l & 15;
    // group g: tokens 8g .. 8g+3, channels 32 .. 47; lane i -> token 8g + (i >> 2), channels 32 + 4 (i & 3)
    const unsigned short* p = lds + (8 * g + (i >> 2)) * 128 + 32 + 4 * (i & 3);
    s4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)p);
    y[l] = __builtin_bit_cast(unsigned long long, r);
}
int main() {
    unsigned long long* d; unsigned long long h[64];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) { unsigned v = (h[l] >> (16 * j)) & 0xffff; printf(" (t%2u,c%3u)", v >> 8, v & 255); }
        printf("\n");
    }
    return 0;
}
