// micro-benchmark: cost of a device-wide barrier inside one persistent kernel (one or two 512-thread blocks per CU) against
// the cost of a kernel boundary -- would fusing the short deep-level kernels of the unet.yaml step into one launch pay?
// Each round every block writes 4 KB, crosses the barrier (release fence, atomic arrive, spin, acquire fence) and reads the
// 4 KB another block wrote (checked), so the fences have real work to order.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned nblocks, unsigned& epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                         // release: this block's writes are visible device-wide
        const unsigned target = (epoch + 1) * nblocks;
        atomicAdd(counter, 1u);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __threadfence();                                         // acquire
    }
    epoch += 1;
    __syncthreads();
}

__global__ __launch_bounds__(512) void k_rounds(float* buf, unsigned* counter, int rounds, int* bad) {
    unsigned epoch = 0;
    const unsigned nb = gridDim.x;
    for (int r = 0; r < rounds; ++r) {
        float* mine = buf + ((size_t)(r & 1) * nb + blockIdx.x) * 1024;
        mine[threadIdx.x] = (float)(r * 1000 + blockIdx.x);
        mine[512 + threadIdx.x] = (float)(r * 1000 + blockIdx.x);
        grid_barrier(counter, nb, epoch);
        const unsigned other = (blockIdx.x * 37 + 11) % nb;      // usually a block on another XCD
        const float* theirs = buf + ((size_t)(r & 1) * nb + other) * 1024;
        if (__builtin_nontemporal_load(theirs + threadIdx.x) != (float)(r * 1000 + other)) atomicAdd(bad, 1);
    }
}
__global__ __launch_bounds__(512) void k_one(float* buf, int r) {
    float* mine = buf + (size_t)blockIdx.x * 1024;
    mine[threadIdx.x] = (float)r;
    mine[512 + threadIdx.x] = (float)r;
}

int main() {
    float* buf; unsigned* counter; int* bad;
    hipMalloc(&buf, 2 * 1024 * 1024 * 4 * 2); hipMalloc(&counter, 4); hipMalloc(&bad, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int rounds = 200;
    for (int nb : {256, 512}) {
        hipMemset(counter, 0, 4); hipMemset(bad, 0, 4);
        hipLaunchKernelGGL(k_rounds, dim3(nb), dim3(512), 0, 0, buf, counter, 4, bad);
        hipDeviceSynchronize();
        hipMemset(counter, 0, 4);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_rounds, dim3(nb), dim3(512), 0, 0, buf, counter, rounds, bad);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        int hb; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
        printf("%d blocks x 512 threads: %.2f us per round with a device-wide barrier (stale reads: %d)\n", nb, ms * 1e3 / rounds, hb);
        hipEventRecord(e0);
        for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(k_one, dim3(nb), dim3(512), 0, 0, buf, r);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("%d blocks x 512 threads: %.2f us per round as separate launches\n", nb, ms * 1e3 / rounds);
    }
    return 0;
}
