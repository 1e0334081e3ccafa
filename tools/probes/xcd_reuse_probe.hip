// Does an XCD's L2 keep a producer kernel's output across a kernel boundary?  Kernel A writes `bytes` (workgroup w writes chunk w),
// kernel B reads it back with workgroup w reading chunk (w + shift) % nwg: shift 0 = the XCD that wrote the chunk (workgroups go
// to XCDs round-robin), shift 1 = the neighbouring XCD, shift 8 = the same XCD but another CU.  Prints us per read kernel.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/probes/xcd_reuse_probe.hip -o /tmp/xcd_probe && /tmp/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void writer(float4* p, size_t per_wg, float v) {
    float4* q = p + (size_t)blockIdx.x * per_wg;
    for (size_t i = threadIdx.x; i < per_wg; i += 256) q[i] = make_float4(v, v + 1.f, v + 2.f, (float)i);
}
__global__ __launch_bounds__(256) void reader(const float4* p, size_t per_wg, int shift, float* out) {
    const int w = (blockIdx.x + shift) % gridDim.x;
    const float4* q = p + (size_t)w * per_wg;
    float s = 0.f;
    for (size_t i = threadIdx.x; i < per_wg; i += 256 * 4) {
        float4 a = q[i], b = i + 256 < per_wg ? q[i + 256] : a, c = i + 512 < per_wg ? q[i + 512] : a, d = i + 768 < per_wg ? q[i + 768] : a;
        s += a.x + b.y + c.z + d.w;
    }
    if (s == 12345.678f) out[0] = s;
}

int main() {
    const int nwg = 2048;
    hipStream_t st; CK(hipStreamCreate(&st));
    float* out; CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t mb : {4, 8, 16, 24, 32, 64, 128}) {
        const size_t bytes = mb << 20, per_wg = bytes / nwg / 16;
        float4* buf; CK(hipMalloc(&buf, bytes));
        float4* other; CK(hipMalloc(&other, 512 << 20));
        for (int flush = 0; flush < 2; ++flush)
            for (int shift : {0, 1, 4, 8, 1024}) {
                float tot = 0.f;
                const int reps = 20;
                for (int r = 0; r < reps + 2; ++r) {
                    if (flush) hipLaunchKernelGGL(writer, dim3(nwg), dim3(256), 0, st, other, (size_t)(512 << 20) / nwg / 16, 1.0f);  // evict L2 AND the Infinity Cache
                    hipLaunchKernelGGL(writer, dim3(nwg), dim3(256), 0, st, buf, per_wg, (float)r);
                    CK(hipEventRecord(e0, st));
                    hipLaunchKernelGGL(reader, dim3(nwg), dim3(256), 0, st, buf, per_wg, shift, out);
                    CK(hipEventRecord(e1, st));
                    CK(hipStreamSynchronize(st));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (r >= 2) tot += ms;
                }
                printf("%4zu MB  %s  shift %4d: read %7.2f us  (%.2f TB/s)\n", mb, flush ? "writer first evicted by a 512 MB pass (writes buf AFTER it)" : "plain", shift,
                       tot / reps * 1e3, bytes / (tot / reps * 1e-3) / 1e12);
            }
        CK(hipFree(buf)); CK(hipFree(other));
    }
    return 0;
}
