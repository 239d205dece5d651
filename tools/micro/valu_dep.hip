// micro-benchmark: v_fma_f32 issue rate of ONE wave per SIMD as a function of the distance D between dependent FMAs (D independent
// accumulator chains interleaved), and the cost of DPP wave shifts / v_cndmask / transcendental ops mixed into an FMA stream.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int D, int MIX>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x[D];
    float v[8];
#pragma unroll
    for (int i = 0; i < D; ++i) x[i] = threadIdx.x * 0.001f + i;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = 0.999f + threadIdx.x * 1e-9f + i * 1e-8f; asm volatile("" : "+v"(v[i])); }
    float m = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 96 / D; ++r) {
#pragma unroll
            for (int i = 0; i < D; ++i) x[i] = fmaf(x[i], v[(i + r) % 8], b);
            if (MIX == 1 && r % 4 == 0)       // a DPP wave shift per 4*D FMAs
                m += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x[0]), 0x138, 0xf, 0xf, false));
            if (MIX == 2 && r % 4 == 0) m += __builtin_amdgcn_exp2f(x[0]);
        }
    }
    float s = m;
#pragma unroll
    for (int i = 0; i < D; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int D, int MIX>
void run(float* out, hipEvent_t e0, hipEvent_t e1) {
    const int iters = 2000;
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<D, MIX>), dim3(256), dim3(256), 0, 0, out, iters, 0.999f, 0.001f);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double ninst = (double)iters * (96 / D) * D;
    printf("D=%2d mix=%d: %.2f ns per FMA (one wave per SIMD)\n", D, MIX, best * 1e6 / ninst);
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 1024 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    run<1, 0>(out, e0, e1); run<2, 0>(out, e0, e1); run<3, 0>(out, e0, e1); run<4, 0>(out, e0, e1); run<6, 0>(out, e0, e1); run<8, 0>(out, e0, e1); run<12, 0>(out, e0, e1);
    run<3, 1>(out, e0, e1); run<6, 1>(out, e0, e1); run<3, 2>(out, e0, e1); run<6, 2>(out, e0, e1);
    return 0;
}
