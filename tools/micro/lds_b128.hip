// micro-benchmark: cycles per 64-lane LDS access for the address patterns of the conv kernels (fragment reads with
// row stride S, staging writes of 16 / 8 bytes per lane), to see which ones the 64-bank LDS of gfx950 serialises.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/lds_b128.hip -o /tmp/lds_b128 && /tmp/lds_b128
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(int stride, int iters, unsigned* out, long long* cyc) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<unsigned*>(lds)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int addr;
    if (MODE == 0) addr = (lane & 15) * stride + (lane >> 4) * 16;          // fragment read b128: row m16, K part q
    else if (MODE == 1) addr = (lane >> 2) * stride + (lane & 3) * 16;      // staging write b128: 4 lanes per row
    else if (MODE == 2) addr = (lane >> 3) * stride + (lane & 7) * 8;       // staging write b64: 8 lanes per row
    else addr = (lane >> 4) * stride + (lane & 15) * 8;                     // staging write b64: 16 lanes per row (160-B rows)
    addr += wave * 16384;
    u32x4 acc = {0, 0, 0, 0};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int a = addr + ((it + u) & 3) * 2048;
            if (MODE == 0) {
                u32x4 v = *reinterpret_cast<u32x4*>(lds + a);
                acc += v;
            } else if (MODE == 1) {
                *reinterpret_cast<u32x4*>(lds + a) = acc;
                acc[0] += u;
            } else {
                *reinterpret_cast<u32x2*>(lds + a) = u32x2{acc[0], acc[1]};
                acc[0] += u;
            }
        }
    }
    __syncthreads();
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + lds[threadIdx.x];
}

int main() {
    unsigned* out; long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    const int iters = 2000;
    const char* names[4] = {"read b128 (m16 rows, q*16B)", "write b128 (4 lanes/row)", "write b64 (8 lanes/row)", "write b64 (16 lanes/row)"};
    for (int mode = 0; mode < 4; ++mode)
        for (int stride : {64, 80, 96, 144, 160, 272, 288}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, stride, iters, out, cyc);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, stride, iters, out, cyc);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, stride, iters, out, cyc);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, stride, iters, out, cyc);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // one block per CU, 4 waves: 4 * iters * 8 accesses per CU
            const double per = ms * 1e-3 * 2.4e9 / (4.0 * iters * 8);
            printf("%-30s stride %4d B: %.2f cycles per 64-lane access (per CU, 2.4 GHz assumed)\n", names[mode], stride, per);
        }
    return 0;
}
