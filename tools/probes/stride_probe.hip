// How fast does HBM serve the A-operand pattern of a K-contiguous GEMM?  A is [M][K] bf16 row-major; a workgroup owns ROWS
// rows and walks K in slices of SB bytes per row (a GEMM with BK = 64 reads 128 B per row and slice).  Compared with the same
// bytes read as one linear stream.  hipcc --offload-arch=gfx950 -O3 stride_probe.hip -o ../../vit-search_amd/build/stride_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int SB>   // bytes per row and slice
__global__ __launch_bounds__(512) void tile_reader(const char* __restrict__ a, float* sink, int M, int rowbytes, int rows_per_wg) {
    const int t = threadIdx.x;
    constexpr int LPR = SB / 16;                    // lanes per row
    const int r_in = t / LPR, c = (t % LPR) * 16;
    const int rows_pass = 512 / LPR;                // rows covered by one pass of the workgroup
    const long long m0 = (long long)blockIdx.x * rows_per_wg;
    float acc = 0.f;
    for (int k = 0; k < rowbytes; k += SB) {
#pragma unroll 4
        for (int r = r_in; r < rows_per_wg; r += rows_pass) {
            const long long m = m0 + r;
            if (m < M) {
                const float4 v = *reinterpret_cast<const float4*>(a + m * rowbytes + k + c);
                acc += v.x + v.y + v.z + v.w;
            }
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

__global__ __launch_bounds__(512) void linear_reader(const char* __restrict__ a, float* sink, long long bytes) {
    const long long per = (bytes / gridDim.x) & ~15LL;
    const char* p = a + per * blockIdx.x;
    float acc = 0.f;
#pragma unroll 4
    for (long long o = threadIdx.x * 16LL; o < per; o += 512 * 16) {
        const float4 v = *reinterpret_cast<const float4*>(p + o);
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main() {
    const int M = 32896, NB = 8;
    float* sink;
    hipMalloc(&sink, 4);
    for (int K : {256, 768, 1536}) {
        const int rowbytes = K * 2;
        const long long bytes = (long long)M * rowbytes;
        std::vector<char*> bufs(NB);
        for (auto& b : bufs) { hipMalloc(&b, bytes); hipMemset(b, 1, bytes); }
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        auto run = [&](const char* name, auto launch) {
            for (int i = 0; i < 3; ++i) launch(bufs[i % NB]);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            const int n = 40;
            for (int i = 0; i < n; ++i) launch(bufs[i % NB]);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("K %4d  %-28s %7.1f us  %5.2f TB/s\n", K, name, ms / n * 1e3, bytes / (ms / n * 1e-3) / 1e12);
        };
        run("linear, 256 wg", [&](char* b) { hipLaunchKernelGGL(linear_reader, dim3(256), dim3(512), 0, 0, b, sink, bytes); });
        run("linear, 1024 wg", [&](char* b) { hipLaunchKernelGGL(linear_reader, dim3(1024), dim3(512), 0, 0, b, sink, bytes); });
        for (int rows : {129, 64, 32}) {
            const int grid = (M + rows - 1) / rows;
            char nm[64];
            snprintf(nm, 64, "%d rows/wg, 128 B slices", rows);
            run(nm, [&](char* b) { hipLaunchKernelGGL(tile_reader<128>, dim3(grid), dim3(512), 0, 0, b, sink, M, rowbytes, rows); });
            snprintf(nm, 64, "%d rows/wg, 256 B slices", rows);
            run(nm, [&](char* b) { hipLaunchKernelGGL(tile_reader<256>, dim3(grid), dim3(512), 0, 0, b, sink, M, rowbytes, rows); });
            snprintf(nm, 64, "%d rows/wg, 512 B slices", rows);
            run(nm, [&](char* b) { hipLaunchKernelGGL(tile_reader<512>, dim3(grid), dim3(512), 0, 0, b, sink, M, rowbytes, rows); });
        }
        for (auto& b : bufs) hipFree(b);
    }
    return 0;
}
