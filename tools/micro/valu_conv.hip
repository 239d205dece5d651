// micro-benchmark: 3x3 conv, 3 -> 3 channels, NHWC fp32, 8 x 512 x 512, on the VECTOR ALU (the fp32 VALU peak equals the fp32
// MFMA peak on gfx950, and a 3-channel conv fills a third of a 16x16x4 MFMA).  One thread = one pixel column of a
// 256-wide tile, sliding 3x3 window in registers, 81 weights uniform (SGPR operands), tile rows staged in LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

constexpr int C = 3, TW = 256;
constexpr int LSF = 4 + (TW + 2) * C + 2;          // floats per staged row: 4 lead (1 junk + left halo pixel), tile, right halo, pad -> 780
constexpr int LS4 = LSF / 4;

template <int TH>
__global__ __launch_bounds__(256) void k_conv(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                              float* __restrict__ y, int B, int H, int W) {
    __shared__ float4 lds4[(TH + 2) * LS4];
    float* lds = reinterpret_cast<float*>(lds4);
    const int tid = threadIdx.x;
    const int tiles_x = W / TW, tiles_y = H / TH;
    const int tile = blockIdx.x, bx = tile % tiles_x, by = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int x0 = bx * TW, y0 = by * TH;
    // stage (TH+2) rows: row r = image row y0-1+r, floats [x0*3-4, x0*3-4+LSF)
    const int rowlen = W * C;
    for (int i = tid; i < (TH + 2) * LS4; i += 256) {
        const int r = i / LS4, c4 = i - r * LS4;
        const int iy = y0 - 1 + r, f = x0 * C - 4 + 4 * c4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < H && f >= 0 && f + 3 < rowlen) v = *reinterpret_cast<const float4*>(x + ((size_t)b * H + iy) * rowlen + f);
        else if (iy >= 0 && iy < H) {
            float t[4];
            for (int k = 0; k < 4; ++k) t[k] = (f + k >= 0 && f + k < rowlen) ? x[((size_t)b * H + iy) * rowlen + f + k] : 0.f;
            v = make_float4(t[0], t[1], t[2], t[3]);
        }
        lds4[i] = v;
    }
    float wr[81];
#pragma unroll
    for (int i = 0; i < 81; ++i) wr[i] = w[i];       // uniform -> scalar registers
    const float b0 = bias[0], b1 = bias[1], b2 = bias[2];
    __syncthreads();
    // window rows: win[rr][k], k = dx*3 + ci  (9 floats starting at float 1 + 3*tid of the staged row)
    float win[3][9];
    const float* base = lds + 1 + 3 * tid;
#pragma unroll
    for (int k = 0; k < 9; ++k) { win[0][k] = base[k]; win[1][k] = base[LSF + k]; }
#pragma unroll 1
    for (int r0 = 0; r0 < TH; r0 += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int r = r0 + u;
            if (r < TH) {
                // new bottom row into slot (u + 2) % 3
#pragma unroll
                for (int k = 0; k < 9; ++k) win[(u + 2) % 3][k] = base[(r + 2) * LSF + k];
                float a0 = b0, a1 = b1, a2 = b2;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int k = 0; k < 9; ++k) {          // k = dx*3 + ci ; weight [dy][dx][ci][co]
                        const float xv = win[(u + dy) % 3][k];
                        a0 = fmaf(xv, wr[(dy * 9 + k) * 3 + 0], a0);
                        a1 = fmaf(xv, wr[(dy * 9 + k) * 3 + 1], a1);
                        a2 = fmaf(xv, wr[(dy * 9 + k) * 3 + 2], a2);
                    }
                a0 = fmaxf(a0, 0.f); a1 = fmaxf(a1, 0.f); a2 = fmaxf(a2, 0.f);
                float* o = y + (((size_t)b * H + y0 + r) * W + x0 + tid) * C;
                o[0] = a0; o[1] = a1; o[2] = a2;
            }
        }
    }
}

int main() {
    const int B = 8, H = 512, W = 512;
    const size_t n = (size_t)B * H * W * C;
    std::vector<float> hx(n), hw(81), hb(3), hy(n);
    srand(1);
    for (auto& v : hx) v = rand() / (float)RAND_MAX;
    for (auto& v : hw) v = rand() / (float)RAND_MAX - 0.5f;
    for (auto& v : hb) v = 0.1f;
    float *x, *w, *bias, *y;
    hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMalloc(&w, 81 * 4); hipMalloc(&bias, 12);
    hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), 81 * 4, hipMemcpyHostToDevice);
    hipMemcpy(bias, hb.data(), 12, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    int blocks = 0;
    auto run = [&](auto kern, int th) {
        blocks = B * (H / th) * (W / TW);
        float bst = 1e9;
        for (int r = 0; r < 10; ++r) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, x, w, bias, y, B, H, W);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < bst) bst = ms;
        }
        printf("TH=%d: %.2f us (%d blocks)\n", th, bst * 1e3, blocks);
        best = bst;
    };
    run(k_conv<4>, 4);
    run(k_conv<32>, 32);
    run(k_conv<16>, 16);
    run(k_conv<8>, 8);
    hipMemcpy(hy.data(), y, n * 4, hipMemcpyDeviceToHost);
    // spot check
    double maxerr = 0;
    for (int t = 0; t < 2000; ++t) {
        const int b = rand() % B, yy = rand() % H, xx = rand() % W, co = rand() % 3;
        double a = hb[co];
        for (int dy = 0; dy < 3; ++dy) for (int dx = 0; dx < 3; ++dx) for (int ci = 0; ci < 3; ++ci) {
            const int iy = yy + dy - 1, ix = xx + dx - 1;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            a += (double)hx[(((size_t)b * H + iy) * W + ix) * C + ci] * hw[((dy * 3 + dx) * 3 + ci) * 3 + co];
        }
        a = a > 0 ? a : 0;
        maxerr = fmax(maxerr, fabs(a - hy[(((size_t)b * H + yy) * W + xx) * C + co]));
    }
    printf("VALU 3->3 conv forward 8x512x512: %.2f us (%d blocks), max err %.2e, %.1f GB/s algorithmic\n", best * 1e3, blocks, maxerr,
           2.0 * n * 4 / (best * 1e-3) / 1e9);
    return 0;
}
