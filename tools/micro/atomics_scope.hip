// micro-benchmark: float atomic adds of 256 blocks into a 36,864-float slab -- agent scope into one slab (what the split-K weight
// gradient kernels do) against workgroup / agent scope into per-XCD slabs (slab = blockIdx % 8: blocks b, b + 8, ... share an XCD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int SCOPE, bool PERXCD>
__global__ __launch_bounds__(256) void k_atomic(float* dst, int n) {
    float* d = dst + (PERXCD ? (size_t)(blockIdx.x & 7) * n : 0);
    for (int i = threadIdx.x; i < n; i += 256) {
        if (SCOPE == 0) atomicAdd(d + i, 1.0f);
        else __hip_atomic_fetch_add(d + i, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
int main() {
    const int n = 9 * 64 * 64, nb = 256;
    float* dst;
    (void)hipMalloc(&dst, (size_t)n * 8 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    std::vector<float> h((size_t)n * 8);
    auto run = [&](const char* name, auto launch, bool perxcd) {
        float best = 1e9;
        for (int r = 0; r < 6; ++r) {
            (void)hipMemset(dst, 0, (size_t)n * 8 * 4);
            (void)hipEventRecord(e0, 0); launch(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        (void)hipMemcpy(h.data(), dst, (size_t)n * 8 * 4, hipMemcpyDeviceToHost);
        double s = 0; float mn = 1e9, mx = -1e9;
        for (int x = 0; x < (perxcd ? 8 : 1); ++x) for (int i = 0; i < n; ++i) { s += h[(size_t)x * n + i]; }
        for (int i = 0; i < n; ++i) { float v = 0; for (int x = 0; x < (perxcd ? 8 : 1); ++x) v += h[(size_t)x * n + i]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
        printf("%-52s %8.1f us   per-element total min %.0f max %.0f (expect %d)\n", name, best * 1e3, mn, mx, nb);
    };
    run("agent scope, one slab", [&] { hipLaunchKernelGGL((k_atomic<0, false>), dim3(nb), dim3(256), 0, 0, dst, n); }, false);
    run("agent scope, per-XCD slabs", [&] { hipLaunchKernelGGL((k_atomic<0, true>), dim3(nb), dim3(256), 0, 0, dst, n); }, true);
    run("workgroup scope, per-XCD slabs", [&] { hipLaunchKernelGGL((k_atomic<1, true>), dim3(nb), dim3(256), 0, 0, dst, n); }, true);
    run("workgroup scope, one slab (WRONG across XCDs?)", [&] { hipLaunchKernelGGL((k_atomic<1, false>), dim3(nb), dim3(256), 0, 0, dst, n); }, false);
    return 0;
}
