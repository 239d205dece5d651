// micro-benchmark: how long does a plain float4 copy of the conv's tensors take?  (25.2 MB in, 25.2 MB out = the 3 -> 3 conv at
// 8 x 512 x 512; 75.5 MB in + 25.2 out = its backward) -- the yardstick for the short HBM-bound kernels of the unet.yaml step.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_copy_gs(const float4* __restrict__ a, float4* __restrict__ b, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_read3(const float4* __restrict__ a, const float4* __restrict__ c, const float4* __restrict__ d,
                                               float4* __restrict__ b, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) { float4 x = a[i], y = c[i], z = d[i]; b[i] = make_float4(x.x + y.x + z.x, x.y + y.y + z.y, x.z + y.z + z.z, x.w + y.w + z.w); }
}
int main() {
    const size_t n4 = (size_t)8 * 512 * 512 * 3 / 4;
    float4 *a, *b, *c, *d, *junk;
    hipMalloc(&a, n4 * 16); hipMalloc(&b, n4 * 16); hipMalloc(&c, n4 * 16); hipMalloc(&d, n4 * 16); hipMalloc(&junk, 512u << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, bool cold, auto launch) {
        float best = 1e9;
        for (int r = 0; r < 8; ++r) {
            if (cold) hipMemsetAsync(junk, r, 512u << 20, 0);
            hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-58s %7.2f us\n", name, best * 1e3);
    };
    for (int cold = 0; cold < 2; ++cold) {
        printf(cold ? "-- after a 512 MB memset (cold caches)\n" : "-- back to back (Infinity Cache warm)\n");
        run("copy 25 MB -> 25 MB, one float4 per thread", cold, [&] { hipLaunchKernelGGL(k_copy, dim3((n4 + 255) / 256), dim3(256), 0, 0, a, b, n4); });
        run("copy 25 MB -> 25 MB, 2048 grid-stride blocks", cold, [&] { hipLaunchKernelGGL(k_copy_gs, dim3(2048), dim3(256), 0, 0, a, b, n4); });
        run("read 3 x 25 MB -> write 25 MB", cold, [&] { hipLaunchKernelGGL(k_read3, dim3((n4 + 255) / 256), dim3(256), 0, 0, a, c, d, b, n4); });
    }
    return 0;
}
