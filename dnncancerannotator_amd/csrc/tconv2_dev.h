// tconv2_dev.h -- device body of the small-channel 2x2/2 transposed-conv data gradient, shared by its own kernel
// (kernels_misc.hip) and by the one-launch backward of kernels_mfma.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace dnnca {

__device__ __forceinline__ void tc_ld4(float* d, const float* s) {
    const float4 t = *reinterpret_cast<const float4*>(s);
    d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
}
__device__ __forceinline__ void tc_st4(float* d, const float* s) { *reinterpret_cast<float4*>(d) = make_float4(s[0], s[1], s[2], s[3]); }

struct TdArgs {
    const float* dout;
    const float* w;
    const float* in;
    float* din;
    int B, H, W, acc, mask;
    float alpha;
};

// data gradient: one thread = one input pixel; din = ((acc ? din : 0) + sum dout * W) * (mask ? act'(in) : 1)
template <int CIN, int COUT>
__device__ __forceinline__ void tconv2_dgrad_body(const TdArgs& p, int id) {
    const int total = p.B * p.H * p.W;
    if (id >= total) return;
    const int j = id % p.W, bi = id / p.W;
    float d[CIN];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) d[ci] = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        float g[2 * COUT];
        const float* gp = p.dout + (((size_t)bi * 2 + a) * (2 * p.W) + 2 * j) * COUT;
        if constexpr ((2 * COUT) % 4 == 0) {
#pragma unroll
            for (int v = 0; v < 2 * COUT / 4; ++v) tc_ld4(g + 4 * v, gp + 4 * v);
        } else {
#pragma unroll
            for (int v = 0; v < COUT; ++v) {
                float2 t = *reinterpret_cast<const float2*>(gp + 2 * v);
                g[2 * v] = t.x;
                g[2 * v + 1] = t.y;
            }
        }
        const float* wa = p.w + a * 2 * COUT * CIN;
#pragma unroll
        for (int r = 0; r < 2 * COUT; ++r)
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) d[ci] = fmaf(g[r], wa[r * CIN + ci], d[ci]);
    }
    float* dp = p.din + (size_t)id * CIN;
    if (p.acc) {
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) d[ci] += dp[ci];
    }
    if (p.mask) {
        const float* ip = p.in + (size_t)id * CIN;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) d[ci] *= ip[ci] > 0.f ? 1.0f : p.alpha;
    }
    if constexpr (CIN % 4 == 0) {
#pragma unroll
        for (int v = 0; v < CIN / 4; ++v) tc_st4(dp + 4 * v, d + 4 * v);
    } else {
#pragma unroll
        for (int v = 0; v < CIN / 2; ++v) *reinterpret_cast<float2*>(dp + 2 * v) = make_float2(d[2 * v], d[2 * v + 1]);
    }
}

}  // namespace dnnca
