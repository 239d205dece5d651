// micro-benchmark: issue rate of v_mfma_f32_16x16x32_bf16 / v_mfma_f32_32x32x16_bf16 with operands in registers,
// NACC independent accumulators per wave, W waves per SIMD (blocks of 256 * W threads, one block per CU).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_bf16_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int NT>
__global__ __launch_bounds__(NT) void k16(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + threadIdx.x * 0.001f); b[i] = (__bf16)0.5f; }
    f32x4 acc[NACC];
    for (int c = 0; c < NACC; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < NACC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < NACC; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * NT + threadIdx.x] = s;
}
template <int NACC, int NT>
__global__ __launch_bounds__(NT) void k32(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + threadIdx.x * 0.001f); b[i] = (__bf16)0.5f; }
    f32x16 acc[NACC];
    for (int c = 0; c < NACC; ++c)
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < NACC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < NACC; ++c)
        for (int i = 0; i < 16; ++i) s += acc[c][i];
    out[blockIdx.x * NT + threadIdx.x] = s;
}

template <typename F>
static void run(const char* name, F launch, double flop_per_block_iter, int iters, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.3f ms  %8.1f TFLOP/s\n", name, ms, flop_per_block_iter * iters * blocks / (ms * 1e-3) / 1e12);
}

int main() {
    float* out; hipMalloc(&out, 2048 * 1024 * 4);
    const int blocks = 256;
    for (int iters : {2000, 40000}) {
        printf("iters %d\n", iters);
        run("16x16x32 1 wave/SIMD 4 acc", [&] { hipLaunchKernelGGL((k16<4, 256>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 4.0 * 4 * 4 * 16384, iters, blocks);
        run("16x16x32 1 wave/SIMD 16 acc", [&] { hipLaunchKernelGGL((k16<16, 256>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 4.0 * 4 * 16 * 16384, iters, blocks);
        run("16x16x32 2 waves/SIMD 16 acc", [&] { hipLaunchKernelGGL((k16<16, 512>), dim3(blocks), dim3(512), 0, 0, out, iters); }, 8.0 * 4 * 16 * 16384, iters, blocks);
        run("16x16x32 4 waves/SIMD 8 acc", [&] { hipLaunchKernelGGL((k16<8, 1024>), dim3(blocks), dim3(1024), 0, 0, out, iters); }, 16.0 * 4 * 8 * 16384, iters, blocks);
        run("32x32x16 1 wave/SIMD 4 acc", [&] { hipLaunchKernelGGL((k32<4, 256>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 4.0 * 4 * 4 * 32768, iters, blocks);
        run("32x32x16 2 waves/SIMD 4 acc", [&] { hipLaunchKernelGGL((k32<4, 512>), dim3(blocks), dim3(512), 0, 0, out, iters); }, 8.0 * 4 * 4 * 32768, iters, blocks);
    }
    return 0;
}
