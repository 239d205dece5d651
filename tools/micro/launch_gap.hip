// micro-benchmark: what does a kernel boundary cost on one stream?  (a) back-to-back empty kernels, (b) an empty kernel
// after a kernel that leaves N MB of dirty lines in the L2s, (c) the same pair with the writer alone.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty() {}
__global__ void k_one(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ __launch_bounds__(256) void k_write(float4* p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
int main() {
    float* buf; hipMalloc(&buf, 512u << 20);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, int reps, auto body) {
        float best = 1e9;
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(e0, s);
            for (int i = 0; i < reps; ++i) body();
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-60s %8.2f us per iteration\n", name, best * 1e3 / reps);
    };
    run("empty kernel x1000", 1000, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); });
    run("1-thread RMW kernel x1000", 1000, [&] { hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, s, buf); });
    run("empty kernel, 1024 blocks x1000", 1000, [&] { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, s); });
    for (size_t mb : {1, 8, 32, 128}) {
        const size_t n4 = (mb << 20) / 16;
        char nm[96];
        snprintf(nm, 96, "write %zu MB", mb);
        run(nm, 200, [&] { hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, s, (float4*)buf, n4); });
        snprintf(nm, 96, "write %zu MB + empty kernel", mb);
        run(nm, 200, [&] { hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, s, (float4*)buf, n4); hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); });
    }
    return 0;
}
