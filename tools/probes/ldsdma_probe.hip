// LDS-DMA throughput of one CU from L2-resident data (dev probe, round 5): a 256-thread workgroup keeps D "slices" of P pieces
// (1 KB each, per wave) in flight with counted vmcnt + s_barrier, like the GEMM K loop without the MFMAs.
// hipcc --offload-arch=gfx950 -O3 -o ldsdma_probe ldsdma_probe.hip ; ./ldsdma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(unsigned lds, unsigned voff, v4i rsrc, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
template <int P, int D>
__global__ __launch_bounds__(256) void probe(const char* src, int region_bytes, int iters, int shared_region, float* sink) {
    __shared__ __attribute__((aligned(1024))) char smem[D * P * 4 * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long a = (unsigned long long)(src + (shared_region ? 0 : (size_t)blockIdx.x * region_bytes));
    v4i rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    rs.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    rs.z = 0x7fffff00; rs.w = 0x00020000;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // GEMM-like pattern: a piece = 8 rows x 128 B, rows 1 KB apart (K = 512 bf16)
    const unsigned voff = (unsigned)((lane >> 3) * 1024 + (lane & 7) * 16);
    const int slice_bytes = P * 4 * 8 * 1024;        // P pieces x 4 waves x 8 rows x 1 KB stride
    auto issue = [&](int it, int buf) {
        const int so = (it * 128) % 1024 + ((it / 8) * slice_bytes) % (region_bytes - slice_bytes);
#pragma unroll
        for (int h = 0; h < P; ++h) dma16(lds0 + (buf * P * 4 + wave * P + h) * 1024, voff + (wave * P + h) * 8192, rs, so);
    };
#pragma unroll
    for (int d = 0; d < D - 1; ++d) issue(d, d);
    int buf = 0;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (D >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * P) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue(it + D - 1, buf == 0 ? D - 1 : buf - 1);
        acc += *reinterpret_cast<float*>(smem + buf * P * 4096 + threadIdx.x * 16);
        buf = buf + 1 == D ? 0 : buf + 1;
        if (D == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 12345.678f) sink[0] = acc;
}
template <int P, int D> void run(const char* src, int region, int wg_per_cu, int shared_region, float* sink) {
    const int iters = 400, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<P, D><<<grid, 256>>>(src, region, 10, shared_region, sink);
    hipEventRecord(e0);
    probe<P, D><<<grid, 256>>>(src, region, iters, shared_region, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * iters * P * 4 * 1024;
    printf("P=%d pieces/wave D=%d bufs  %d wg/cu  %s region: %7.1f us  %6.1f GB/s per CU  %5.2f TB/s chip  (%d KB in flight per CU)\n", P, D, wg_per_cu,
           shared_region ? "shared  " : "per-wg  ", ms * 1e3, bytes / 256 / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12, (D > 1 ? D - 1 : 1) * P * 4 * wg_per_cu);
}
int main() {
    const int region = 2 * 1024 * 1024;                 // shared: every workgroup reads the same 2 MB (L2-hot); per-wg: 0.5 - 2 GB streamed once (HBM)
    char* src; hipMalloc(&src, (size_t)region * 1025); hipMemset(src, 1, (size_t)region * 1025);
    float* sink; hipMalloc(&sink, 64);
    for (int shared = 1; shared >= 0; --shared) {
        for (int g : {1, 2, 4}) {
            run<8, 1>(src, region, g, shared, sink);
            run<8, 2>(src, region, g, shared, sink);
            run<8, 3>(src, region, g, shared, sink);
            if (g <= 2) run<8, 4>(src, region, g, shared, sink);
            run<4, 3>(src, region, g, shared, sink);
            if (g == 1) run<16, 2>(src, region, g, shared, sink);
        }
    }
    return 0;
}
