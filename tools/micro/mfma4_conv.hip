// micro-benchmark 3: the 3x3 / 3 -> 3 conv forward on v_mfma_f32_4x4x1_16b_f32 -- 16 blocks of (4 pixels x 4 channels) per
// instruction, K = the 27 (tap, input channel) pairs, no structural zeros.  One wave = a strip of 64 pixel columns (lane =
// pixel = row i of block lane / 4), rows walked with a sliding 3-row window in registers, horizontal neighbours by DPP wave
// shifts (valu_conv2.hip), no LDS, no barriers.  Operands: A = the (shifted) pixel value of the lane, B = W[tap][ci][lane % 4]
// (27 registers), D = 4 registers: lane (block b, column j) holds output channel j of pixels 4b .. 4b+3 -- stored straight
// from there (four stores per row, each lane's three channel lanes write one pixel's 12 bytes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef float f3 __attribute__((ext_vector_type(3)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int C = 3;

__device__ __forceinline__ float shr1(float old, float v) {   // lane l <- lane l-1; lane 0 <- old
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float shl1(float old, float v) {   // lane l <- lane l+1; lane 63 <- old
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

template <int RB, int PF>
__global__ __launch_bounds__(256) void k_strip(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                               float* __restrict__ y, int B, int H, int W) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int strips = (W + 63) / 64, bands = H / RB;
    const int wid = blockIdx.x * 4 + wave;
    if (wid >= B * strips * bands) return;
    const int sx = wid % strips, band = (wid / strips) % bands, b = wid / (strips * bands);
    const int x0 = sx * 64, y0 = band * RB, xc = x0 + lane;
    const int j = lane & 3, blk = lane >> 2;
    float wr[27];                      // B operand of K-step k = (dy * 3 + dx) * 3 + ci: W[k][j] (column 3 is padding)
#pragma unroll
    for (int k = 0; k < 27; ++k) wr[k] = j < 3 ? w[k * 3 + j] : 0.f;
    const float bj = j < 3 ? bias[j] : 0.f;
    const int xh = lane == 0 ? x0 - 1 : x0 + 64;
    const bool has_h = (lane == 0 || lane == 63) && xh >= 0 && xh < W;
    const bool in_x = xc < W;
    const float* img = x + (size_t)b * H * W * C;
    auto load_row = [&](int iy, f3& v, f3& h) {
        v = f3{0.f, 0.f, 0.f}; h = f3{0.f, 0.f, 0.f};
        if (iy >= 0 && iy < H) {
            if (in_x) v = *reinterpret_cast<const f3*>(img + ((size_t)iy * W + xc) * C);
            if (has_h) h = *reinterpret_cast<const f3*>(img + ((size_t)iy * W + xh) * C);
        }
    };
    f3 pv[PF], ph[PF];                 // rows y0-1+i in flight
#pragma unroll
    for (int i = 0; i < PF; ++i) load_row(y0 - 1 + i, pv[i], ph[i]);
    float win[3][9];                   // [row slot][dx*3 + ci]
    auto expand = [&](const f3& v, const f3& h, float (&o)[9]) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            o[c] = shr1(h[c], v[c]);       // left neighbour
            o[3 + c] = v[c];
            o[6 + c] = shl1(h[c], v[c]);   // right neighbour
        }
    };
    expand(pv[0], ph[0], win[0]);
    expand(pv[1], ph[1], win[1]);
    load_row(y0 - 1 + PF, pv[0], ph[0]);           // the two consumed slots take the next rows
    load_row(y0 + PF, pv[1], ph[1]);
#pragma unroll 1
    for (int r0 = 0; r0 < RB; r0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int r = r0 + u;
            expand(pv[(u + 2) % PF], ph[(u + 2) % PF], win[(u + 2) % 3]);
            load_row(y0 - 1 + r + 2 + PF, pv[(u + 2) % PF], ph[(u + 2) % PF]);
            f32x4 acc0 = f32x4{bj, bj, bj, bj}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};      // two chains: 4x4x1 results come back after 2 passes
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const float xv = win[(u + dy) % 3][k];
                    if ((dy * 9 + k) & 1) acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(xv, wr[dy * 9 + k], acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(xv, wr[dy * 9 + k], acc0, 0, 0, 0);
                }
            // lane (blk, j): channel j of pixels 4 blk + i
            float* orow = y + (((size_t)b * H + y0 + r) * W + x0 + 4 * blk) * C + j;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (j < 3 && x0 + 4 * blk + i < W) orow[i * C] = fmaxf(acc0[i] + acc1[i], 0.f);
        }
    }
}

int main() {
    const int B = 8, H = 528, W = 512;   // 528 = 11 * 48 rows so that every RB divides it
    const size_t n = (size_t)B * H * W * C;
    std::vector<float> hx(n), hw(81), hb(3), hy(n);
    srand(1);
    for (auto& v : hx) v = rand() / (float)RAND_MAX;
    for (auto& v : hw) v = rand() / (float)RAND_MAX - 0.5f;
    for (auto& v : hb) v = 0.1f;
    float *x, *w, *bias, *y;
    hipMalloc(&x, n * 4 + 64); hipMalloc(&y, n * 4 + 64); hipMalloc(&w, 81 * 4); hipMalloc(&bias, 12);
    hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), 81 * 4, hipMemcpyHostToDevice);
    hipMemcpy(bias, hb.data(), 12, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, int rb, int pf) {
        const int waves = B * ((W + 63) / 64) * (H / rb), blocks = (waves + 3) / 4;
        float bst = 1e9;
        for (int r = 0; r < 10; ++r) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, x, w, bias, y, B, H, W);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < bst) bst = ms;
        }
        hipMemcpy(hy.data(), y, n * 4, hipMemcpyDeviceToHost);
        double maxerr = 0;
        for (int t = 0; t < 3000; ++t) {
            const int b = rand() % B, yy = rand() % H, xx = (t % 3 == 0) ? (rand() % 8) * 64 + (rand() % 2 ? 0 : 63) : rand() % W, co = rand() % 3;
            double a = hb[co];
            for (int dy = 0; dy < 3; ++dy) for (int dx = 0; dx < 3; ++dx) for (int ci = 0; ci < 3; ++ci) {
                const int iy = yy + dy - 1, ix = xx + dx - 1;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                a += (double)hx[(((size_t)b * H + iy) * W + ix) * C + ci] * hw[((dy * 3 + dx) * 3 + ci) * 3 + co];
            }
            a = a > 0 ? a : 0;
            maxerr = fmax(maxerr, fabs(a - hy[(((size_t)b * H + yy) * W + xx) * C + co]));
        }
        printf("RB=%3d PF=%2d: %.2f us (%d waves), max err %.2e\n", rb, pf, bst * 1e3, waves, maxerr);
    };
    run(k_strip<12, 6>, 12, 6);     // PF must be a multiple of 3 (window slots) and divide RB
    run(k_strip<24, 6>, 24, 6);
    run(k_strip<24, 12>, 24, 12);
    run(k_strip<48, 12>, 48, 12);
    run(k_strip<12, 12>, 12, 12);
    run(k_strip<6, 6>, 6, 6);
    return 0;
}
