// micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 when the A operand of every MFMA comes out of LDS (one ds_read_b32 per MFMA,
// as in the pixel-group convolutions), NCH independent accumulators per wave, W waves per SIMD, lane stride STRIDE floats between
// the 16 rows of an M-tile (12: the 2-way bank conflict of a 12-float pixel group; 17: conflict-free), with s_memtime around it.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f32_lds.hip -o /tmp/mfma_f32_lds && /tmp/mfma_f32_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: operands in registers; 1: A from LDS (double-buffered, next step read before this step's MFMAs); 2: A and B from LDS
template <int NCH, int NT, int MODE, int STRIDE>
__global__ __launch_bounds__(NT) void k(float* out, unsigned long long* ticks, int iters) {
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += NT) lds[i] = 1.0f + 0.001f * (i & 63);
    __syncthreads();
    const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4, wave = threadIdx.x >> 6;
    const float* ap = lds + wave * 1024 + m * STRIDE + q;
    const float* bp = lds + 12288 + lane;
    f32x4 acc[NCH];
    for (int c = 0; c < NCH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float b = 0.5f + lane * 0.01f;
    float av[2][NCH], bv[2];
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
    for (int c = 0; c < NCH; ++c) av[0][c] = MODE ? ap[c * 256] : 1.0f + c;
    bv[0] = MODE == 2 ? bp[0] : b;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (MODE) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) av[(s + 1) & 1][c] = ap[c * 256 + 4 * ((s + 1) & 7)];
                if (MODE == 2) bv[(s + 1) & 1] = bp[64 * ((s + 1) & 7)];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(MODE ? av[s & 1][c] : 1.0f + c, MODE == 2 ? bv[s & 1] : b, acc[c], 0, 0, 0);
            if (MODE) __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
    for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * NT + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <typename F>
static void run(const char* name, F launch, double mfma_per_simd, unsigned long long* ticks_dev) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t[4];
    hipMemcpy(t, ticks_dev, sizeof(t), hipMemcpyDeviceToHost);
    printf("%-56s %7.3f ms  %6.1f TFLOP/s  %6.1f ns/MFMA/SIMD  %6.1f ticks/MFMA/SIMD (tick rate %.2f GHz)\n", name, ms,
           mfma_per_simd * 1024 * 2048 / (ms * 1e-3) / 1e12, ms * 1e6 / mfma_per_simd, (double)t[0] / mfma_per_simd, t[0] / (ms * 1e6));
}

int main() {
    float* out; hipMalloc(&out, 2048 * 1024 * 4);
    unsigned long long* ticks; hipMalloc(&ticks, 4096 * 8);
    const int blocks = 256, iters = 4000;
#define RUN(name, nch, nt, mode, stride) \
    run(name, [&] { hipLaunchKernelGGL((k<nch, nt, mode, stride>), dim3(blocks), dim3(nt), 0, 0, out, ticks, iters); }, (double)(nt / 256) * nch * 8 * iters, ticks)
    RUN("regs        1 wave/SIMD 4 chains", 4, 256, 0, 12);
    RUN("regs        1 wave/SIMD 6 chains", 6, 256, 0, 12);
    RUN("regs        2 waves/SIMD 4 chains", 4, 512, 0, 12);
    RUN("A lds s12   1 wave/SIMD 4 chains", 4, 256, 1, 12);
    RUN("A lds s12   1 wave/SIMD 6 chains", 6, 256, 1, 12);
    RUN("A lds s12   1 wave/SIMD 8 chains", 8, 256, 1, 12);
    RUN("A lds s12   2 waves/SIMD 4 chains", 4, 512, 1, 12);
    RUN("A lds s12   2 waves/SIMD 6 chains", 6, 512, 1, 12);
    RUN("A lds s17   1 wave/SIMD 6 chains", 6, 256, 1, 17);
    RUN("A lds s17   2 waves/SIMD 6 chains", 6, 512, 1, 17);
    RUN("A+B lds s12 1 wave/SIMD 7 chains", 7, 256, 2, 12);
    RUN("A+B lds s12 2 waves/SIMD 7 chains", 7, 512, 2, 12);
    RUN("A+B lds s12 2 waves/SIMD 14 chains", 14, 512, 2, 12);
    RUN("A+B lds s17 2 waves/SIMD 14 chains", 14, 512, 2, 17);
    return 0;
}
