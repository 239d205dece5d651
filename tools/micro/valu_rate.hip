// micro-benchmark: issue rate of v_fma_f32 on gfx950 by operand kind, with 1, 2, 4 waves per SIMD.
//   mode 0: x = fma(x, s, s)       one vector-register operand (the other two scalar)
//   mode 1: x = fma(x, v, s)       two vector-register operands
//   mode 2: acc = fma(v1, v2, acc) three distinct vector-register operands (the weight-gradient outer product of strip_dev.h)
//   mode 3: x = pk_fma(x, s, s)    packed: two FMAs per lane
//   hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate.bin && ./valu_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float a, float b) {
    float x[16], v[8], u[8];
    f2 y[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = f2{x[i], x[i] + 0.5f}; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = 0.999f + threadIdx.x * 1e-9f + i * 1e-8f; u[i] = 1e-6f * (i + 1) + threadIdx.x * 1e-9f; asm volatile("" : "+v"(v[i]), "+v"(u[i])); }
    const f2 a2 = {a, a}, b2 = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) x[i] = fmaf(x[i], a, b);
                if (MODE == 1) x[i] = fmaf(x[i], v[(i + r) % 8], b);
                if (MODE == 2) x[i] = fmaf(v[(i + r) % 8], u[(i * 3 + r) % 8], x[i]);
                if (MODE == 3) y[i] = __builtin_elementwise_fma(y[i], a2, b2);
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += MODE == 3 ? y[i][0] + y[i][1] : x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    const char* names[4] = {"fma(v, s, s)     ", "fma(v, v, s)     ", "fma(v, v, v)     ", "pk_fma(v, s, s)  "};
    for (int mode = 0; mode < 4; ++mode)
        for (int wps : {1, 2, 4}) {          // waves per SIMD: block = 256 * wps threads, one block per CU
            float best = 1e9;
            for (int r = 0; r < 5; ++r) {
                (void)hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256 * wps), 0, 0, out, iters, 0.999f, 0.001f);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256 * wps), 0, 0, out, iters, 0.999f, 0.001f);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256 * wps), 0, 0, out, iters, 0.999f, 0.001f);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256 * wps), 0, 0, out, iters, 0.999f, 0.001f);
                (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double ninst = (double)iters * 128;                      // instructions per wave
            const double flops = ninst * 64 * 2 * (mode == 3 ? 2 : 1) * 256 * 4 * wps;
            printf("%s %d waves/SIMD: %.3f ms  -> %6.1f TFLOP/s, %.2f ns per wave-instruction per SIMD\n", names[mode], wps,
                   best, flops / best * 1e-9, best * 1e6 / (ninst * wps));
        }
    return 0;
}
