// kernels_misc.hip -- tuned HBM-bound kernels around the convolutions of configs/unet.yaml: 2x2 max-pool (fwd/bwd),
// 2x2 stride-2 transposed conv (fwd / data gradient), and the fused head: 1x1 conv -> logits -> weighted BCE ->
// dlogits -> head gradients -> gradient of the last feature map, all from ONE read of the feature map.
// Every thread moves 16-byte vectors; 12 floats (= 4/2/1 pixels of 3/6/12 channels) are the common work unit so that
// the non-power-of-two pixel sizes stay 16-byte aligned.
#include "fast.h"
#include "kernels.h"

namespace dnnca {

#define DEVINL __device__ __forceinline__

DEVINL void ld4(float* d, const float* p) {
    float4 t = *reinterpret_cast<const float4*>(p);
    d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
}
DEVINL void st4(float* p, const float* s) { *reinterpret_cast<float4*>(p) = make_float4(s[0], s[1], s[2], s[3]); }

static inline bool dense(const View& v) { return v.C == 0 || v.ps == v.C; }

// ------------------------------------------------------------------------------------------------ max pool 2x2 / 2
// one thread = 12 output floats = 24 input floats of each of the two input rows
template <int C>
__global__ __launch_bounds__(256) void k_pool2_fwd(const float* __restrict__ in, float* __restrict__ out, int B, int Ho,
                                                   int Wo) {
    const int cpr = Wo * C / 12;                       // chunks per output row
    const int total = B * Ho * cpr;
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int ch = id % cpr, row = id / cpr;           // row = b*Ho + oy
    const float* r0 = in + ((size_t)row * 2) * (2 * Wo * C) + ch * 24;
    const float* r1 = r0 + 2 * Wo * C;
    float a[24], b[24], o[12];
#pragma unroll
    for (int v = 0; v < 6; ++v) {
        ld4(a + 4 * v, r0 + 4 * v);
        ld4(b + 4 * v, r1 + 4 * v);
    }
#pragma unroll
    for (int f = 0; f < 12; ++f) {
        const int i0 = (2 * (f / C)) * C + f % C, i1 = i0 + C;
        o[f] = fmaxf(fmaxf(a[i0], a[i1]), fmaxf(b[i0], b[i1]));
    }
    float* op = out + (size_t)row * (Wo * C) + ch * 12;
#pragma unroll
    for (int v = 0; v < 3; ++v) st4(op + 4 * v, o + 4 * v);
}

// din = ((acc ? din : 0) + route(dout)) * (mask ? act'(in) : 1); the first maximum in row-major window order gets it
template <int C>
__global__ __launch_bounds__(256) void k_pool2_bwd(const float* __restrict__ in, const float* __restrict__ dout,
                                                   float* __restrict__ din, int B, int Ho, int Wo, int acc, int mask,
                                                   float alpha) {
    const int cpr = Wo * C / 12;
    const int total = B * Ho * cpr;
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int ch = id % cpr, row = id / cpr;
    const size_t o0 = ((size_t)row * 2) * (2 * Wo * C) + ch * 24, o1 = o0 + 2 * Wo * C;
    float a[24], b[24], g[12], da[24], db[24];
#pragma unroll
    for (int v = 0; v < 6; ++v) {
        ld4(a + 4 * v, in + o0 + 4 * v);
        ld4(b + 4 * v, in + o1 + 4 * v);
    }
    const float* gp = dout + (size_t)row * (Wo * C) + ch * 12;
#pragma unroll
    for (int v = 0; v < 3; ++v) ld4(g + 4 * v, gp + 4 * v);
    if (acc) {
#pragma unroll
        for (int v = 0; v < 6; ++v) {
            ld4(da + 4 * v, din + o0 + 4 * v);
            ld4(db + 4 * v, din + o1 + 4 * v);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 24; ++i) da[i] = db[i] = 0.f;
    }
#pragma unroll
    for (int f = 0; f < 12; ++f) {
        const int i0 = (2 * (f / C)) * C + f % C, i1 = i0 + C;
        const float m = fmaxf(fmaxf(a[i0], a[i1]), fmaxf(b[i0], b[i1]));
        const bool h0 = a[i0] == m, h1 = !h0 && a[i1] == m, h2 = !h0 && !h1 && b[i0] == m, h3 = !h0 && !h1 && !h2;
        da[i0] += h0 ? g[f] : 0.f;
        da[i1] += h1 ? g[f] : 0.f;
        db[i0] += h2 ? g[f] : 0.f;
        db[i1] += h3 ? g[f] : 0.f;
    }
    if (mask) {
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            da[i] *= a[i] > 0.f ? 1.0f : alpha;
            db[i] *= b[i] > 0.f ? 1.0f : alpha;
        }
    }
#pragma unroll
    for (int v = 0; v < 6; ++v) {
        st4(din + o0 + 4 * v, da + 4 * v);
        st4(din + o1 + 4 * v, db + 4 * v);
    }
}

template <bool DH>
__global__ void k_pool2_bwd_idx(size_t nwin4, const float* __restrict__ dout, const unsigned* __restrict__ idx,
                                float* __restrict__ din, int C, int H, int W, int acc);        // with the BN kernels below

bool fast_pool_supported(const Model* m, const Op& o) {
    if (o.type != OP_POOL || o.k != 2) return false;
    if (!dense(o.inA.d) || !dense(o.out.d)) return false;
    const int C = o.out.d.C;
    if (!(C == 3 || C == 6 || C == 12)) return false;
    return (o.out.d.W * C) % 12 == 0;
}

bool fast_pool_fwd(Model* m, int B, Op& o, double bytes) {
    if (!fast_pool_supported(m, o)) return false;
    const int C = o.out.d.C, Ho = o.out.d.H, Wo = o.out.d.W;
    const int total = B * Ho * (Wo * C / 12);
    dim3 grid((total + 255) / 256);
#define POOL_CASE(c)                                                                                      \
    if (C == c) {                                                                                         \
        LAUNCH(m, "pool2_fwd_" #c, bytes, 0,                                                              \
               hipLaunchKernelGGL(k_pool2_fwd<c>, grid, dim3(256), 0, m->stream, o.inA.d.p, o.out.d.p, B, Ho, Wo)); \
        return true;                                                                                      \
    }
    POOL_CASE(3) POOL_CASE(6) POOL_CASE(12)
#undef POOL_CASE
    return false;
}

bool fast_pool_bwd(Model* m, int B, Op& o, double bytes) {
    if (o.pool_idx_valid && (o.pool_idx || m->dry) && dense(o.inA.g) && dense(o.out.g) && !o.maskA) {
        // the fused BN-apply + pool pass of this step recorded where every maximum sits
        o.pool_idx_valid = false;
        const int C = o.out.d.C;
        const size_t nwin4 = (size_t)B * o.out.d.H * o.out.d.W * (C / 4);
        if (o.inA.g.h)          // the gradient arriving at a BatchNorm, stored as bf16
            LAUNCH(m, "pool2_bwd_idx", (double)nwin4 * (16 + 4 + (o.accA ? 64 : 32)), 0,
                   hipLaunchKernelGGL(k_pool2_bwd_idx<true>, dim3((unsigned)((nwin4 + 255) / 256)), dim3(256), 0, m->stream, nwin4, o.out.g.p,
                                      reinterpret_cast<const unsigned*>(o.pool_idx), o.inA.g.p, C, o.inA.d.H, o.inA.d.W, (int)o.accA));
        else
            LAUNCH(m, "pool2_bwd_idx", (double)nwin4 * (16 + 4 + (o.accA ? 128 : 64)), 0,
                   hipLaunchKernelGGL(k_pool2_bwd_idx<false>, dim3((unsigned)((nwin4 + 255) / 256)), dim3(256), 0, m->stream, nwin4, o.out.g.p,
                                      reinterpret_cast<const unsigned*>(o.pool_idx), o.inA.g.p, C, o.inA.d.H, o.inA.d.W, (int)o.accA));
        return true;
    }
    if (!fast_pool_supported(m, o)) return false;
    o.pool_idx_valid = false;          // (recorded window positions of this step, if any, are not used by this kernel)
    const int C = o.out.d.C, Ho = o.out.d.H, Wo = o.out.d.W;
    const int total = B * Ho * (Wo * C / 12);
    dim3 grid((total + 255) / 256);
#define POOL_CASE(c)                                                                                      \
    if (C == c) {                                                                                         \
        LAUNCH(m, "pool2_bwd_" #c, bytes, 0,                                                              \
               hipLaunchKernelGGL(k_pool2_bwd<c>, grid, dim3(256), 0, m->stream, o.inA.d.p, o.out.g.p, o.inA.g.p, B, Ho, \
                                  Wo, (int)o.accA, (int)o.maskA, o.mask_alpha));                          \
        return true;                                                                                      \
    }
    POOL_CASE(3) POOL_CASE(6) POOL_CASE(12)
#undef POOL_CASE
    return false;
}

// ------------------------------------------------------------------------------------------------ transposed conv 2x2/2
// forward: one thread = one input pixel and one output row a (uniform per block row) -> 2 output pixels
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void k_tconv2_fwd(const float* __restrict__ in, const float* __restrict__ w,
                                                    const float* __restrict__ bias, float* __restrict__ out, int B, int H,
                                                    int W) {
    const int a = blockIdx.y;
    const int total = B * H * W;
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int j = id % W, bi = id / W;        // bi = b*H + i
    float x[CIN];
    const float* ip = in + (size_t)id * CIN;
    if constexpr (CIN % 4 == 0) {
#pragma unroll
        for (int v = 0; v < CIN / 4; ++v) ld4(x + 4 * v, ip + 4 * v);
    } else {
#pragma unroll
        for (int v = 0; v < CIN / 2; ++v) {
            float2 t = *reinterpret_cast<const float2*>(ip + 2 * v);
            x[2 * v] = t.x;
            x[2 * v + 1] = t.y;
        }
    }
    float o[2 * COUT];
    const float* wa = w + a * 2 * COUT * CIN;     // [e][co][ci]
#pragma unroll
    for (int r = 0; r < 2 * COUT; ++r) {
        float s = bias[r % COUT];
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) s = fmaf(x[ci], wa[r * CIN + ci], s);
        o[r] = s;
    }
    float* op = out + (((size_t)bi * 2 + a) * (2 * W) + 2 * j) * COUT;
    if constexpr ((2 * COUT) % 4 == 0) {
#pragma unroll
        for (int v = 0; v < 2 * COUT / 4; ++v) st4(op + 4 * v, o + 4 * v);
    } else {
#pragma unroll
        for (int v = 0; v < COUT; ++v) *reinterpret_cast<float2*>(op + 2 * v) = make_float2(o[2 * v], o[2 * v + 1]);
    }
}

bool fast_tconv_supported(const Model* m, const Op& o) {
    if (o.type != OP_TCONV || o.k != 2) return false;
    if (!dense(o.inA.d) || !dense(o.out.d)) return false;
    const int CI = o.inA.d.C, CO = o.out.d.C;
    if (!((CI == 12 && CO == 12) || (CI == 12 && CO == 6) || (CI == 6 && CO == 3))) return false;
    return o.inA.d.W % 4 == 0;
}

#define TCONV_CASES(X) X(12, 12) X(12, 6) X(6, 3)

bool fast_tconv_fwd(Model* m, int B, Op& o, double bytes, double flops) {
    if (!fast_tconv_supported(m, o)) return false;
    const int CI = o.inA.d.C, CO = o.out.d.C, H = o.inA.d.H, W = o.inA.d.W;
    dim3 grid((B * H * W + 255) / 256, 2);
#define X(ci, co)                                                                                               \
    if (CI == ci && CO == co) {                                                                                 \
        LAUNCH(m, "tconv2_fwd_" #ci "_" #co, bytes, flops,                                                      \
               hipLaunchKernelGGL((k_tconv2_fwd<ci, co>), grid, dim3(256), 0, m->stream, o.inA.d.p, m->p + o.w_off, \
                                  m->p + o.b_off, o.out.d.p, B, H, W));                                         \
        return true;                                                                                            \
    }
    TCONV_CASES(X)
#undef X
    return false;
}

// ------------------------------------------------------------------------------------------------ fused head (training)
// feat [B,H,W,C] -> logit = b + sum_c w_c feat_c -> weighted BCE (utils/losses.py:17-37) -> dlogit -> dfeat, dW, db.
// One thread = 4 pixels.  scalars[0] (label sum) must be complete (k_label_stats ran earlier on the stream).
struct HeadArgs {
    const float* feat;
    const float* y;
    const float* w;        // C weights + 1 bias (bias at w_bias)
    const float* bias;
    float* dfeat;
    float* dw;             // gradient destinations
    float* dbias;
    double* scalars;
    float* partials;       // [gridDim.x][C + 2] block partial sums (dW..., db, loss); reduced by k_head_reduce
    dnnca_loss_cfg cfg;
    double n_label;
    float gscale;
    int mask;              // multiply dfeat by act'(feat)
    float alpha;
    int n4;                // number of PX-pixel chunks (PX = 4 for C = 3, 1 for wide heads)
};

template <int C, int PX>
__global__ __launch_bounds__(256) void k_head_train(HeadArgs p) {
    __shared__ float red[4][C + 2];
    float wv[C];
#pragma unroll
    for (int c = 0; c < C; ++c) wv[c] = p.w[c];
    const float bias = p.bias[0];
    float wgt;
    if (p.cfg.has_weight) {
        wgt = p.cfg.weight;
    } else {
        float pr = (float)(p.scalars[0] / p.n_label);
        wgt = pr > 0.f ? 1.0f / pr : 1.0f;
    }
    wgt = p.cfg.weight_mul * wgt + p.cfg.weight_add;

    float sdw[C], sdb = 0.f, sloss = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) sdw[c] = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < p.n4; i += gridDim.x * 256) {
        float f[PX * C], z[PX], df[PX * C];
#pragma unroll
        for (int v = 0; v < PX * C / 4; ++v) ld4(f + 4 * v, p.feat + (size_t)i * PX * C + 4 * v);
        if constexpr (PX == 4) ld4(z, p.y + (size_t)i * 4);
        else z[0] = p.y[i];
#pragma unroll
        for (int px = 0; px < PX; ++px) {
            float x = bias;
#pragma unroll
            for (int c = 0; c < C; ++c) x = fmaf(f[px * C + c], wv[c], x);
            const float mk = fmaf(z[px], wgt - 1.0f, 1.0f);
            const float e = expf(-fabsf(x));
            sloss = fmaf(fmaxf(x, 0.f) - x * z[px] + log1pf(e), mk, sloss);
            const float sig = x >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
            const float dl = mk * (sig - z[px]) * p.gscale;
            sdb += dl;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float fv = f[px * C + c];
                sdw[c] = fmaf(fv, dl, sdw[c]);
                float d = dl * wv[c];
                if (p.mask) d *= fv > 0.f ? 1.0f : p.alpha;
                df[px * C + c] = d;
            }
        }
#pragma unroll
        for (int v = 0; v < PX * C / 4; ++v) st4(p.dfeat + (size_t)i * PX * C + 4 * v, df + 4 * v);
    }
    // block reduction: wave shuffles, then 4 partials through LDS, one atomic per value per block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float vals[C + 2];
#pragma unroll
    for (int c = 0; c < C; ++c) vals[c] = sdw[c];
    vals[C] = sdb;
    vals[C + 1] = sloss;
#pragma unroll
    for (int k = 0; k < C + 2; ++k) {
        float v = vals[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < C + 2) {
        const int k = threadIdx.x;
        p.partials[blockIdx.x * (C + 2) + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    }
}

// The same for wide heads (C = 16, 64): C / 4 adjacent lanes share a pixel, each holds four channels -- every load and store
// is one coalesced 16-byte access per lane (the one-thread-per-pixel kernel above walks 256-byte rows with 64 lanes at once
// and keeps 4 C floats per thread); the logit is folded across the lane group with shuffles.  Same partials layout.
template <int C>
__global__ __launch_bounds__(256) void k_head_train_wide(HeadArgs p) {
    constexpr int G = C / 4, PPW = 64 / G, U = 4;        // lanes per pixel, pixels per wave and load, loads in flight
    static_assert(C % 4 == 0 && 64 % G == 0, "lane groups");
    __shared__ float red[4][C + 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cq = lane % G, pl = lane / G;
    const float4 wv = *reinterpret_cast<const float4*>(p.w + 4 * cq);
    const float bias = p.bias[0];
    float wgt;
    if (p.cfg.has_weight) {
        wgt = p.cfg.weight;
    } else {
        float pr = (float)(p.scalars[0] / p.n_label);
        wgt = pr > 0.f ? 1.0f / pr : 1.0f;
    }
    wgt = p.cfg.weight_mul * wgt + p.cfg.weight_add;

    float4 sdw = make_float4(0.f, 0.f, 0.f, 0.f);
    float sdb = 0.f, sloss = 0.f;
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;          // global wave index / count
    for (int p0 = gw * (PPW * U); p0 < p.n4; p0 += nw * (PPW * U)) {
        float4 f[U];
        float z[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int px = p0 + u * PPW + pl;
            f[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            z[u] = 0.f;
            if (px < p.n4) {
                f[u] = *reinterpret_cast<const float4*>(p.feat + (size_t)px * C + 4 * cq);
                z[u] = p.y[px];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int px = p0 + u * PPW + pl;
            float x = fmaf(f[u].x, wv.x, fmaf(f[u].y, wv.y, fmaf(f[u].z, wv.z, f[u].w * wv.w)));
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);     // every lane of the group ends with the sum
            x += bias;
            const float mk = fmaf(z[u], wgt - 1.0f, 1.0f);
            const float e = expf(-fabsf(x));
            const float sig = x >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
            const bool ok = px < p.n4;
            const float dl = ok ? mk * (sig - z[u]) * p.gscale : 0.f;
            if (ok && cq == 0) {
                sloss = fmaf(fmaxf(x, 0.f) - x * z[u] + log1pf(e), mk, sloss);
                sdb += dl;
            }
            sdw.x = fmaf(f[u].x, dl, sdw.x); sdw.y = fmaf(f[u].y, dl, sdw.y);
            sdw.z = fmaf(f[u].z, dl, sdw.z); sdw.w = fmaf(f[u].w, dl, sdw.w);
            float4 d = make_float4(dl * wv.x, dl * wv.y, dl * wv.z, dl * wv.w);
            if (p.mask) {
                d.x *= f[u].x > 0.f ? 1.0f : p.alpha; d.y *= f[u].y > 0.f ? 1.0f : p.alpha;
                d.z *= f[u].z > 0.f ? 1.0f : p.alpha; d.w *= f[u].w > 0.f ? 1.0f : p.alpha;
            }
            if (ok) *reinterpret_cast<float4*>(p.dfeat + (size_t)px * C + 4 * cq) = d;
        }
    }
    // lanes with the same channel quad (lane % G) add up across the wave, then the four waves through LDS
#pragma unroll
    for (int o = G; o < 64; o <<= 1) {
        sdw.x += __shfl_xor(sdw.x, o, 64); sdw.y += __shfl_xor(sdw.y, o, 64);
        sdw.z += __shfl_xor(sdw.z, o, 64); sdw.w += __shfl_xor(sdw.w, o, 64);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sdb += __shfl_xor(sdb, o, 64); sloss += __shfl_xor(sloss, o, 64); }
    if (lane < G) {
        red[wave][4 * cq] = sdw.x; red[wave][4 * cq + 1] = sdw.y; red[wave][4 * cq + 2] = sdw.z; red[wave][4 * cq + 3] = sdw.w;
    }
    if (lane == 0) { red[wave][C] = sdb; red[wave][C + 1] = sloss; }
    __syncthreads();
    if (threadIdx.x < C + 2) {
        const int k = threadIdx.x;
        p.partials[blockIdx.x * (C + 2) + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    }
}

// one block: sums the per-block partials of k_head_train in a fixed order (bit-reproducible) into the gradient vector
template <int C>
__global__ __launch_bounds__(256) void k_head_reduce(const float* __restrict__ partials, int nblocks, float* dw, float* dbias,
                                                     double* scalars) {
    __shared__ double red[256];
    {
        const int k = blockIdx.x;       // one block per reduced value
        double s = 0.0;
        for (int i = threadIdx.x; i < nblocks; i += 256) s += (double)partials[i * (C + 2) + k];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            if (k < C) dw[k] = (float)red[0];
            else if (k == C) dbias[0] = (float)red[0];
            else scalars[3] = red[0];
        }
        __syncthreads();
    }
}

bool fast_head_supported(const Model* m, const Op& o) {
    if (o.type != OP_HEAD || !dense(o.inA.d) || ((size_t)o.inA.d.H * o.inA.d.W) % 4) return false;
    const int C = o.inA.d.C;
    return C == 3 || C == 16 || C == 64;      // n_filters_first of configs/{unet,mulmo_unet,unet_big}.yaml
}

bool fast_head_train(Model* m, int B, Op& o, const float* y, const dnnca_loss_cfg& cfg, float gscale, double bytes) {
    if (!fast_head_supported(m, o)) return false;
    HeadArgs a;
    a.feat = o.inA.d.p;
    a.y = y;
    a.w = m->p + o.w_off;
    a.bias = m->p + o.b_off;
    a.dfeat = o.inA.g.p;
    a.dw = m->g + o.w_off;
    a.dbias = m->g + o.b_off;
    a.scalars = m->scalars;
    a.cfg = cfg;
    size_t npix = (size_t)B * o.inA.d.H * o.inA.d.W;
    a.n_label = (double)npix;
    a.gscale = gscale;
    a.mask = o.maskA;
    a.alpha = o.mask_alpha;
    const int C = o.inA.d.C;
    a.n4 = (int)(C == 3 ? npix / 4 : npix);
    int blocks = (a.n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;          // partials buffer: 2048 x 72 floats
    a.partials = m->head_partials;
#define HEAD_CASE(c, px)                                                                                               \
    if (C == c) {                                                                                                      \
        if constexpr (c >= 16)                                                                                         \
            LAUNCH(m, "head_train_" #c, bytes, 30.0 * npix, hipLaunchKernelGGL((k_head_train_wide<c>), dim3(blocks), dim3(256), 0, m->stream, a)); \
        else                                                                                                           \
            LAUNCH(m, "head_train_" #c, bytes, 30.0 * npix, hipLaunchKernelGGL((k_head_train<c, px>), dim3(blocks), dim3(256), 0, m->stream, a)); \
        if (m->head_defer_ok && m->merged_launches()) {       /* reduced by the launch that ends the backward pass (k_pg_fold) */ \
            m->head_pending.partials = m->head_partials; m->head_pending.nblocks = blocks; m->head_pending.C = c;       \
            m->head_pending.dw = a.dw; m->head_pending.dbias = a.dbias;                                                \
            return true;                                                                                               \
        }                                                                                                              \
        LAUNCH(m, "head_reduce", 0, 0,                                                                                 \
               hipLaunchKernelGGL(k_head_reduce<c>, dim3(c + 2), dim3(256), 0, m->stream, m->head_partials, blocks, a.dw, a.dbias, \
                                  m->scalars));                                                                        \
        return true;                                                                                                   \
    }
    HEAD_CASE(3, 4) HEAD_CASE(16, 1) HEAD_CASE(64, 1)
#undef HEAD_CASE
    return false;
}

// ------------------------------------------------------------------------------------------------ label smoothing
// utils/losses.py:62-67: y_true = tfa.image.gaussian_filter2d(y_true[..., None], filter_shape = k, sigma) -- REFLECT padding of
// (k-1)/2 before / k-1-(k-1)/2 after (tf.pad REFLECT: the edge sample is not repeated), then a VALID depthwise conv with the
// outer product of the 1-D kernel softmax(-u^2 / (2 sigma^2)), u = -k/2+1 .. k/2 (asymmetric for even k).  One thread per pixel.
struct SmoothArgs {
    const float* y;
    float* out;
    int B, H, W, k;
    float g[16];             // normalised 1-D kernel
};

__device__ __forceinline__ int reflect_idx(int i, int n) {      // tf.pad REFLECT (requires pad < n)
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

__global__ __launch_bounds__(256) void k_label_smooth(SmoothArgs p) {
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (size_t)p.B * p.H * p.W) return;
    const int x = (int)(id % p.W), y = (int)((id / p.W) % p.H);
    const float* img = p.y + (id / ((size_t)p.H * p.W)) * ((size_t)p.H * p.W);
    const int before = (p.k - 1) / 2;
    float acc = 0.f;
    for (int i = 0; i < p.k; ++i) {
        const int yy = reflect_idx(y + i - before, p.H);
        float row = 0.f;
        for (int j = 0; j < p.k; ++j) row = fmaf(p.g[j], img[(size_t)yy * p.W + reflect_idx(x + j - before, p.W)], row);
        acc = fmaf(p.g[i], row, acc);
    }
    p.out[id] = acc;
}

bool fast_label_smooth(Model* m, int B, int H, int W, const float* y, float* out, int k, float sigma) {
    if (k < 1 || k > 15 || !(sigma > 0.f) || k - 1 - (k - 1) / 2 >= H || k - 1 - (k - 1) / 2 >= W) return false;
    SmoothArgs a{};
    a.y = y; a.out = out; a.B = B; a.H = H; a.W = W; a.k = k;
    double g[16], sum = 0.0;
    for (int i = 0; i < k; ++i) {
        const double u = (double)(-(k / 2) + 1 + i);          // tf.range(-k // 2 + 1, k // 2 + 1)
        g[i] = exp(-(u * u) / (2.0 * (double)sigma * (double)sigma));
        sum += g[i];
    }
    for (int i = 0; i < k; ++i) a.g[i] = (float)(g[i] / sum);
    const size_t n = (size_t)B * H * W;
    LAUNCH(m, "label_smooth", 8.0 * n, 2.0 * n * k * k,
           hipLaunchKernelGGL(k_label_smooth, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, m->stream, a));
    return true;
}

// ------------------------------------------------------------------------------------------------ label statistics
// Few, fat blocks: every block ends in one atomic on the same address, and same-address atomics execute one after the
// other at the memory side (~56 ns each, tools/micro/bn_reduce.hip) -- 64 of them cost less than the read pass, 256 more.
__global__ __launch_bounds__(1024) void k_label_stats4(int n4, const float* __restrict__ y, double* __restrict__ scalars) {
    __shared__ float red[16][3];
    float s = 0.f, mn = INFINITY, mx = -INFINITY;
    const int T = gridDim.x * 1024;
    for (int i0 = blockIdx.x * 1024 + threadIdx.x; i0 < n4; i0 += 8 * T) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {      // 8 independent 16-byte loads in flight per thread
            const int i = i0 + u * T;
            v[u] = i < n4 ? reinterpret_cast<const float4*>(y)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i0 + u * T >= n4) continue;
            s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
            mn = fminf(fminf(mn, fminf(v[u].x, v[u].y)), fminf(v[u].z, v[u].w));
            mx = fmaxf(fmaxf(mx, fmaxf(v[u].x, v[u].y)), fmaxf(v[u].z, v[u].w));
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_down(s, o, 64);
        mn = fminf(mn, __shfl_down(mn, o, 64));
        mx = fmaxf(mx, __shfl_down(mx, o, 64));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[wave][0] = s;
        red[wave][1] = mn;
        red[wave][2] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ds = 0.0;
        float fmn = INFINITY, fmx = -INFINITY;
        for (int w = 0; w < 16; ++w) {
            ds += (double)red[w][0];
            fmn = fminf(fmn, red[w][1]);
            fmx = fmaxf(fmx, red[w][2]);
        }
        atomicAdd(scalars + 0, ds);
        // min / max: labels outside [0, 1] only matter for the reference's assertions (utils/losses.py:91-99), so the
        // contended compare-and-swap loop is entered only by blocks that would actually change the stored extreme
        if ((double)fmn < __hip_atomic_load(scalars + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            unsigned long long* pmin = (unsigned long long*)(scalars + 1);
            unsigned long long old = *pmin, assumed;
            do {
                assumed = old;
                if (__longlong_as_double(assumed) <= (double)fmn) break;
                old = atomicCAS(pmin, assumed, __double_as_longlong((double)fmn));
            } while (assumed != old);
        }
        if ((double)fmx > __hip_atomic_load(scalars + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            unsigned long long* pmax = (unsigned long long*)(scalars + 2);
            unsigned long long old = *pmax, assumed;
            do {
                assumed = old;
                if (__longlong_as_double(assumed) >= (double)fmx) break;
                old = atomicCAS(pmax, assumed, __double_as_longlong((double)fmx));
            } while (assumed != old);
        }
    }
}

bool fast_label_stats(Model* m, size_t n, const float* y) {
    if (n % 4 || n / 4 > 0x7fffffff) return false;
    int n4 = (int)(n / 4);
    int blocks = (n4 + 8191) / 8192;        // one round of 8 loads per thread
    if (blocks > 64) blocks = 64;
    LAUNCH(m, "label_stats4", 4.0 * n, (double)n,
           hipLaunchKernelGGL(k_label_stats4, dim3(blocks), dim3(1024), 0, m->stream, n4, y, m->scalars));
    return true;
}

}  // namespace dnnca

// ================================================================================================ batch normalisation
// Tuned BatchNormalization for C % 4 == 0 with (C/4) | 256 (16..1024 channels): one thread owns a 4-channel group of a
// pixel (16-byte loads, several in flight), the 256/(C/4) pixel lanes of a block walk chunks of pixels dealt round-robin,
// partial sums meet in LDS and leave the block as atomic adds to one of a few bucket rows (same-address atomics from
// hundreds of blocks cost more than the whole read pass: tools/micro/bn_reduce.hip); the last block to finish folds the rows
// (bn_dev.h): forward -> mean / biased variance / moving statistics / coefficients (raw moments, folded in double),
// backward -> dgamma / dbeta.  No fold launch.
#include "bn_dev.h"

namespace dnnca {

// four consecutive channels of a tensor stored as f32 or (View::h) as bf16; `elem` is the element offset
template <bool H>
__device__ __forceinline__ float4 ld4(const float* base, size_t elem) {
    if constexpr (H) {
        const hbf16x4 h = *reinterpret_cast<const hbf16x4*>(reinterpret_cast<const hbf16*>(base) + elem);
        return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    } else {
        return *reinterpret_cast<const float4*>(base + elem);
    }
}

static inline bool bn_fast_ok(const View& x) { return x.ps == x.C && x.C % 4 == 0 && 256 % (x.C / 4) == 0 && x.C >= 16; }

template <bool XH>      // XH: x is stored as bf16
__global__ __launch_bounds__(256) void k_bn_stats_fast(size_t npix, const float* __restrict__ x, int C, BnSelfFold f) {
    __shared__ double red[256][8];
    const int G = C / 4, cq = threadIdx.x % G, pl = threadIdx.x / G, PL = 256 / G;
    double ds[4] = {0, 0, 0, 0}, dq[4] = {0, 0, 0, 0};
    float s[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0};
    int cnt = 0;
    constexpr int U = 8;            // loads in flight per thread
    const size_t chunk = (size_t)U * PL, nfull = npix / chunk;
    for (size_t k = blockIdx.x; k < nfull; k += gridDim.x) {
        const size_t p = k * chunk + pl;
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld4<XH>(x, (p + (size_t)u * PL) * C + 4 * cq);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            s[0] += v[u].x; s[1] += v[u].y; s[2] += v[u].z; s[3] += v[u].w;
            sq[0] = fmaf(v[u].x, v[u].x, sq[0]); sq[1] = fmaf(v[u].y, v[u].y, sq[1]);
            sq[2] = fmaf(v[u].z, v[u].z, sq[2]); sq[3] = fmaf(v[u].w, v[u].w, sq[3]);
        }
        if (++cnt == 8) {           // bound the float partials' rounding error: 64 terms per flush
            for (int i = 0; i < 4; ++i) { ds[i] += s[i]; dq[i] += sq[i]; s[i] = 0.f; sq[i] = 0.f; }
            cnt = 0;
        }
    }
    if (blockIdx.x == gridDim.x - 1)
        for (size_t p = nfull * chunk + pl; p < npix; p += PL) {
            const float4 v = ld4<XH>(x, p * C + 4 * cq);
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            sq[0] = fmaf(v.x, v.x, sq[0]); sq[1] = fmaf(v.y, v.y, sq[1]); sq[2] = fmaf(v.z, v.z, sq[2]); sq[3] = fmaf(v.w, v.w, sq[3]);
        }
    for (int i = 0; i < 4; ++i) { ds[i] += s[i]; dq[i] += sq[i]; }
    for (int i = 0; i < 4; ++i) { red[threadIdx.x][i] = ds[i]; red[threadIdx.x][4 + i] = dq[i]; }
    __syncthreads();
    // 2C outputs (sum, sumsq per channel), PL terms each
    double* row = bn_bucket(f, (int)blockIdx.x);
    for (int o = threadIdx.x; o < 2 * C; o += 256) {
        const int c = o % C, which = o / C;
        double a = 0.0;
        for (int l = 0; l < PL; ++l) a += red[l * G + (c >> 2)][4 * which + (c & 3)];
        atomicAdd(row + o, a);
    }
    bn_self_fold(f, gridDim.x, blockIdx.x);
}

// One thread = kBnU four-channel groups 256 elements apart (C / 4 divides 256: the same channel group every time, its coefficients are
// loaded once): kBnU loads in flight per thread.  With ONE element per thread a CU had 16 KB of loads in flight (32 waves x 8-byte
// loads of bf16-stored tensors) -- the passes ran at what that buys against the memory latency, 4.5 TB/s, not at what HBM delivers.
// (f32 tensors, 16-byte loads: one element per thread as before -- four were 2 % slower on mulmo_unet.)
constexpr int bn_u(bool any_half) { return any_half ? 4 : 1; }
template <bool YH, bool XH>      // YH / XH: y / x is stored as bf16
__global__ __launch_bounds__(256) void k_bn_apply_fast(size_t n4, const float* __restrict__ x, float* __restrict__ y, int C,
                                                       int yps, const float* __restrict__ coef) {
    constexpr int kBnU = bn_u(YH || XH);
    const size_t i0 = (size_t)blockIdx.x * (256 * kBnU) + threadIdx.x;
    if (i0 >= n4) return;
    const int G = C / 4, cq = (int)(i0 % G);
    float4 v[kBnU];
#pragma unroll
    for (int u = 0; u < kBnU; ++u) {
        const size_t i = i0 + 256 * u;
        v[u] = ld4<XH>(x, 4 * (i < n4 ? i : i0));
    }
    const float4 sc = *reinterpret_cast<const float4*>(coef + 4 * cq), sh = *reinterpret_cast<const float4*>(coef + C + 4 * cq);
#pragma unroll
    for (int u = 0; u < kBnU; ++u) {
        const size_t i = i0 + 256 * u;
        if (i >= n4) break;
        const size_t p = i / G;
        float4 o;
        o.x = fmaf(v[u].x, sc.x, sh.x); o.y = fmaf(v[u].y, sc.y, sh.y); o.z = fmaf(v[u].z, sc.z, sh.z); o.w = fmaf(v[u].w, sc.w, sh.w);
        if (YH) *reinterpret_cast<hbf16x4*>(reinterpret_cast<hbf16*>(y) + p * yps + 4 * cq) = to_bf16x4(o);
        else *reinterpret_cast<float4*>(y + p * yps + 4 * cq) = o;
    }
}

// BatchNorm apply + the MaxPool2D([2,2], 2) that follows it (components.py:54,59): one thread owns a 4-channel group of a 2 x 2
// pixel window, writes the four normalised pixels and their maximum -- the pool pass never re-reads the normalised tensor.
// Grid-stride over the windows (the stride is a multiple of the channel groups: a thread keeps its channel group), so that the
// batch statistics of the POOLED tensor -- the input of the BatchNorm that follows the pool (components.py:59: pool, BatchNorm) --
// can ride along: per-thread sums, one LDS fold and 2C bucket adds per block, self-folding (bn_dev.h; f.tab == nullptr: none).
// WY = false: nobody but the pool reads the normalised tensor (mulmo_unet: the encoders whose skips the decoder does not take,
// unet.py:183-187 reference_index) -- it is not written.
template <bool YH, bool XH, bool WY>
__global__ __launch_bounds__(256) void k_bn_apply_pool_fast(size_t nwin4, const float* __restrict__ x, float* __restrict__ y,
                                                            float* __restrict__ pooled, unsigned* __restrict__ idx, int C, int H,
                                                            int W, const float* __restrict__ coef, BnSelfFold f) {
    __shared__ float red[256][8];
    const int G = C / 4, Wp = W / 2, Hp = H / 2;
    const int cq = (int)(((size_t)blockIdx.x * 256 + threadIdx.x) % G);
    const float4 sc = *reinterpret_cast<const float4*>(coef + 4 * cq), sh = *reinterpret_cast<const float4*>(coef + C + 4 * cq);
    float4 bs = make_float4(0.f, 0.f, 0.f, 0.f), bq = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwin4; i += (size_t)gridDim.x * 256) {
        const size_t wdx = i / G;                          // window index: (b * Hp + yp) * Wp + xp
        const int xp = (int)(wdx % Wp);
        const size_t byp = wdx / Wp;                       // b * Hp + yp
        const size_t b = byp / Hp;
        const int yp = (int)(byp - b * Hp);
        const size_t p00 = ((b * H + 2 * yp) * W + 2 * xp) * C + 4 * cq;
        float4 v4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v4[k] = ld4<XH>(x, p00 + ((size_t)(k >> 1) * W + (k & 1)) * C);
        float4 mx;
        unsigned where = 0;          // byte e: window position of channel e's first maximum (the order g_pool_bwd searches in)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t o = p00 + ((size_t)(k >> 1) * W + (k & 1)) * C;
            const float4 v = v4[k];
            float4 r;
            r.x = fmaf(v.x, sc.x, sh.x); r.y = fmaf(v.y, sc.y, sh.y); r.z = fmaf(v.z, sc.z, sh.z); r.w = fmaf(v.w, sc.w, sh.w);
            if (WY) {
                if (YH) *reinterpret_cast<hbf16x4*>(reinterpret_cast<hbf16*>(y) + o) = to_bf16x4(r);
                else *reinterpret_cast<float4*>(y + o) = r;
            }
            if (k == 0) mx = r;
            else {
                if (r.x > mx.x) { mx.x = r.x; where = (where & 0xffffff00u) | (unsigned)k; }
                if (r.y > mx.y) { mx.y = r.y; where = (where & 0xffff00ffu) | ((unsigned)k << 8); }
                if (r.z > mx.z) { mx.z = r.z; where = (where & 0xff00ffffu) | ((unsigned)k << 16); }
                if (r.w > mx.w) { mx.w = r.w; where = (where & 0x00ffffffu) | ((unsigned)k << 24); }
            }
        }
        *reinterpret_cast<float4*>(pooled + wdx * C + 4 * cq) = mx;
        if (idx) idx[i] = where;
        bs.x += mx.x; bs.y += mx.y; bs.z += mx.z; bs.w += mx.w;
        bq.x = fmaf(mx.x, mx.x, bq.x); bq.y = fmaf(mx.y, mx.y, bq.y); bq.z = fmaf(mx.z, mx.z, bq.z); bq.w = fmaf(mx.w, mx.w, bq.w);
    }
    if (!f.tab) return;
    red[threadIdx.x][0] = bs.x; red[threadIdx.x][1] = bs.y; red[threadIdx.x][2] = bs.z; red[threadIdx.x][3] = bs.w;
    red[threadIdx.x][4] = bq.x; red[threadIdx.x][5] = bq.y; red[threadIdx.x][6] = bq.z; red[threadIdx.x][7] = bq.w;
    __syncthreads();
    // thread t of the block handles channel group (blockIdx * 256 + t) % G: the groups of lanes t, t + G, ... coincide when G | 256
    const int g0 = (int)(((size_t)blockIdx.x * 256) % G), PL = 256 / G;
    double* row = bn_bucket(f, (int)blockIdx.x);
    for (int o = threadIdx.x; o < 2 * C; o += 256) {
        const int c = o % C, which = o / C;
        const int lane0 = ((c >> 2) - g0 + G) % G;          // first thread of the block that owns channel group c / 4
        float a = 0.f;
        for (int l = 0; l < PL; ++l) a += red[lane0 + l * G][4 * which + (c & 3)];
        atomicAdd(row + o, (double)a);
    }
    bn_self_fold(f, gridDim.x, blockIdx.x);
}

// MaxPool2D([2,2], 2) backward by the recorded positions: din = (acc ? din : 0) + route(dout); one thread = a 4-channel group of a window
template <bool DH>      // DH: din is stored as bf16
__global__ __launch_bounds__(256) void k_pool2_bwd_idx(size_t nwin4, const float* __restrict__ dout, const unsigned* __restrict__ idx,
                                                       float* __restrict__ din, int C, int H, int W, int acc) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nwin4) return;
    const int G = C / 4, cq = (int)(i % G), Wp = W / 2, Hp = H / 2;
    const size_t wdx = i / G;
    const int xp = (int)(wdx % Wp);
    const size_t byp = wdx / Wp, b = byp / Hp;
    const int yp = (int)(byp - b * Hp);
    const float4 g = reinterpret_cast<const float4*>(dout)[i];
    const unsigned where = idx[i];
    const size_t p00 = ((b * H + 2 * yp) * W + 2 * xp) * C + 4 * cq;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t e = p00 + ((size_t)(k >> 1) * W + (k & 1)) * C;
        float4 r = acc ? ld4<DH>(din, e) : make_float4(0.f, 0.f, 0.f, 0.f);
        if ((where & 0xffu) == (unsigned)k) r.x += g.x;
        if (((where >> 8) & 0xffu) == (unsigned)k) r.y += g.y;
        if (((where >> 16) & 0xffu) == (unsigned)k) r.z += g.z;
        if ((where >> 24) == (unsigned)k) r.w += g.w;
        if (DH) *reinterpret_cast<hbf16x4*>(reinterpret_cast<hbf16*>(din) + e) = to_bf16x4(r);
        else *reinterpret_cast<float4*>(din + e) = r;
    }
}

// The gradient that reaches a BatchNorm through the 2x2 max-pool behind it (components.py:54,59: ... BatchNorm, pool), routed by the
// recorded window positions while the BatchNorm's two backward passes read their operands: the pool's own backward pass
// (k_pool2_bwd_idx: a read-modify-write of the whole dy, or its first write) is not run, and where the pool is the only reader
// of the BatchNorm's output dy is never materialised.  PG = 1: dy is the routed gradient alone; PG = 2: dy holds the skip
// connection's gradient, the routed one is added.  With bf16-stored gradients the sum is rounded to bf16 as the pool pass stored it.
struct PoolGrad {
    const float* dpool;      // gradient of the pooled tensor, dense NHWC [B][H/2][W/2][C], f32
    const unsigned* idx;     // one byte per pooled element: window position (0..3) of the maximum
    int H, W;                // of the BatchNorm's tensor
    unsigned mW, mH;         // floor(2^32 / W) + 1, floor(2^32 / H) + 1: pixel index -> (b, y, x) without divisions (pixels x W < 2^32)
};

template <int PG, bool GH>
__device__ __forceinline__ float4 with_pool_grad(float4 d, const PoolGrad& g, unsigned pix, int cq, int G) {
    if (PG == 0) return d;
    const unsigned t = __umulhi(pix, g.mW), x = pix - t * (unsigned)g.W;          // t = b * H + y
    const unsigned b = __umulhi(t, g.mH), y = t - b * (unsigned)g.H;
    const size_t w4 = ((size_t)(b * (unsigned)(g.H >> 1) + (y >> 1)) * (unsigned)(g.W >> 1) + (x >> 1)) * G + cq;
    const unsigned k = ((y & 1u) << 1) | (x & 1u), where = g.idx[w4];
    const float4 v = reinterpret_cast<const float4*>(g.dpool)[w4];
    if (PG == 1) d = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((where & 0xffu) == k) d.x += v.x;
    if (((where >> 8) & 0xffu) == k) d.y += v.y;
    if (((where >> 16) & 0xffu) == k) d.z += v.z;
    if ((where >> 24) == k) d.w += v.w;
    if (GH) { d.x = (float)(hbf16)d.x; d.y = (float)(hbf16)d.y; d.z = (float)(hbf16)d.z; d.w = (float)(hbf16)d.w; }
    return d;
}

// Sums of one BatchNorm backward (dgamma = sum dy * xhat, dbeta = sum dy) -- and their fold: every block adds its 2C partial sums
// to one of R bucket rows (double atomics, bn_dev.h; a row meets gridDim / R blocks), takes a ticket, and the block that draws the
// last one folds the R rows, adds the result to dgamma / dbeta and leaves rows and ticket zeroed for the next BatchNorm.
// No fold launch between this pass and the apply pass (it was a 9 us kernel of 8 .. 128 blocks plus a dependent launch, 24 / 48
// times per step of the dense configurations).
template <bool XH, bool GH, int PG>      // XH / GH: x / dy is stored as bf16; PG: the pool behind the BatchNorm sends its gradient along (PoolGrad)
__global__ __launch_bounds__(256) void k_bn_bwd_reduce_fast(size_t npix, const float* __restrict__ x, const float* __restrict__ dy,
                                                            int C, int dps, const float* __restrict__ coef,
                                                            double* __restrict__ tab, int R, unsigned* __restrict__ ticket,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, PoolGrad pg) {
    __shared__ float red[256][8];
    const int G = C / 4, cq = threadIdx.x % G, pl = threadIdx.x / G, PL = 256 / G;
    const float4 mean = *reinterpret_cast<const float4*>(coef + 2 * C + 4 * cq), inv = *reinterpret_cast<const float4*>(coef + 3 * C + 4 * cq);
    float sg[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0};
    constexpr int U = 4;            // 8 loads in flight per thread (16 for the 8-byte loads of bf16-stored tensors: +4 % per launch, measured)
    const size_t chunk = (size_t)U * PL, nfull = npix / chunk;
    // chunks back to front: whichever kernel produced dy wrote it front to back, its tail is what the Infinity Cache still holds;
    // the apply pass that follows walks front to back and finds this pass's last reads there
    for (size_t kk = blockIdx.x; kk < nfull; kk += gridDim.x) {
        const size_t k = nfull - 1 - kk;
        const size_t p = k * chunk + pl;
        float4 v[U], d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[u] = ld4<XH>(x, (p + (size_t)u * PL) * C + 4 * cq);
            d[u] = PG == 1 ? make_float4(0.f, 0.f, 0.f, 0.f) : ld4<GH>(dy, (p + (size_t)u * PL) * dps + 4 * cq);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) d[u] = with_pool_grad<PG, GH>(d[u], pg, (unsigned)(p + (size_t)u * PL), cq, G);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            sg[0] = fmaf(d[u].x, (v[u].x - mean.x) * inv.x, sg[0]); sg[1] = fmaf(d[u].y, (v[u].y - mean.y) * inv.y, sg[1]);
            sg[2] = fmaf(d[u].z, (v[u].z - mean.z) * inv.z, sg[2]); sg[3] = fmaf(d[u].w, (v[u].w - mean.w) * inv.w, sg[3]);
            sb[0] += d[u].x; sb[1] += d[u].y; sb[2] += d[u].z; sb[3] += d[u].w;
        }
    }
    if (blockIdx.x == gridDim.x - 1)
        for (size_t p = nfull * chunk + pl; p < npix; p += PL) {
            const float4 v = ld4<XH>(x, p * C + 4 * cq);
            const float4 d = with_pool_grad<PG, GH>(PG == 1 ? make_float4(0.f, 0.f, 0.f, 0.f) : ld4<GH>(dy, p * dps + 4 * cq), pg, (unsigned)p, cq, G);
            sg[0] = fmaf(d.x, (v.x - mean.x) * inv.x, sg[0]); sg[1] = fmaf(d.y, (v.y - mean.y) * inv.y, sg[1]);
            sg[2] = fmaf(d.z, (v.z - mean.z) * inv.z, sg[2]); sg[3] = fmaf(d.w, (v.w - mean.w) * inv.w, sg[3]);
            sb[0] += d.x; sb[1] += d.y; sb[2] += d.z; sb[3] += d.w;
        }
    for (int i = 0; i < 4; ++i) { red[threadIdx.x][i] = sg[i]; red[threadIdx.x][4 + i] = sb[i]; }
    __syncthreads();
    double* row = tab + (size_t)(blockIdx.x % R) * 2 * C;
    for (int o = threadIdx.x; o < 2 * C; o += 256) {
        const int c = o % C, which = o / C;
        float a = 0.f;
        for (int l = 0; l < PL; ++l) a += red[l * G + (c >> 2)][4 * which + (c & 3)];
        atomicAdd(row + o, (double)a);
    }
    // ticket: a block's adds have been performed (device-scope atomics, each thread waits vmcnt(0) for its own in bn_last_block, then
    // the barrier) before its ticket is drawn; the last block reads the rows with device-scope loads.  No device-scope FENCE anywhere: on gfx950 that is an L2 write-back / invalidate per block,
    // and 512 of them cost more than the reduction (74 us against 25); nothing here is published through plain stores.
    bn_bwd_self_fold(BnBwdFold{tab, ticket, R, C, dgamma, dbeta, nullptr, nullptr}, gridDim.x, blockIdx.x);
}

template <bool DH, bool XH, bool GH, int PG>      // DH: dx is stored as bf16 (never accumulated into); XH / GH: x / dy are; PG: see PoolGrad
__global__ __launch_bounds__(256) void k_bn_bwd_apply_fast(size_t n4, const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ dx, int C, int dps, int acc,
                                                           const float* __restrict__ coef, const float* __restrict__ gamma,
                                                           const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                           float inv_n, int mask, float alpha, PoolGrad pg) {
    // kBnU elements per thread, 256 apart (see k_bn_apply_fast): the same channel group, all loads in flight together
    constexpr int kBnU = bn_u(DH || XH || GH);
    const size_t i0 = (size_t)blockIdx.x * (256 * kBnU) + threadIdx.x;
    if (i0 >= n4) return;
    const int G = C / 4, cq = (int)(i0 % G);
    float4 vv[kBnU], dd[kBnU], tt[kBnU];
#pragma unroll
    for (int u = 0; u < kBnU; ++u) {
        const size_t i = i0 + 256 * u < n4 ? i0 + 256 * u : i0, p = i / G;
        vv[u] = ld4<XH>(x, 4 * i);
        dd[u] = PG == 1 ? make_float4(0.f, 0.f, 0.f, 0.f) : ld4<GH>(dy, p * dps + 4 * cq);
        if (!DH && acc) tt[u] = reinterpret_cast<const float4*>(dx)[i];
    }
    const float4 mean = *reinterpret_cast<const float4*>(coef + 2 * C + 4 * cq), inv = *reinterpret_cast<const float4*>(coef + 3 * C + 4 * cq);
    const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * cq);
    const float4 dg = *reinterpret_cast<const float4*>(dgamma + 4 * cq), db = *reinterpret_cast<const float4*>(dbeta + 4 * cq);
#pragma unroll
    for (int u = 0; u < kBnU; ++u) {
        const size_t i = i0 + 256 * u;
        if (i >= n4) break;
        const size_t p = i / G;
        const float4 v = vv[u];
        const float4 d = with_pool_grad<PG, GH>(dd[u], pg, (unsigned)p, cq, G);
        float4 r;
        r.x = g.x * inv.x * (d.x - inv_n * (db.x + (v.x - mean.x) * inv.x * dg.x));
        r.y = g.y * inv.y * (d.y - inv_n * (db.y + (v.y - mean.y) * inv.y * dg.y));
        r.z = g.z * inv.z * (d.z - inv_n * (db.z + (v.z - mean.z) * inv.z * dg.z));
        r.w = g.w * inv.w * (d.w - inv_n * (db.w + (v.w - mean.w) * inv.w * dg.w));
        if (!DH && acc) { r.x += tt[u].x; r.y += tt[u].y; r.z += tt[u].z; r.w += tt[u].w; }
        if (mask) {      // x is the activated conv output: hand the producing conv its pre-activation gradient directly
            r.x *= v.x > 0.f ? 1.0f : alpha; r.y *= v.y > 0.f ? 1.0f : alpha;
            r.z *= v.z > 0.f ? 1.0f : alpha; r.w *= v.w > 0.f ? 1.0f : alpha;
        }
        if (DH) reinterpret_cast<hbf16x4*>(dx)[i] = to_bf16x4(r);
        else reinterpret_cast<float4*>(dx)[i] = r;
    }
}

static bool bn_table(Model* m) {          // arrives zeroed, stays zeroed between uses
    return m->bn_tab || m->alloc((void**)&m->bn_tab, (size_t)(Model::kBnTab + 32) * 8) == DNNCA_OK;          // + the ticket counters
}

bool bn_self_fold_args(Model* m, Op& bn, int B, BnSelfFold* f) {
    const int C = bn.inA.d.C;
    if (2 * C > Model::kBnTab) return false;
    if (!m->dry) {          // (the dry run lists the launches of the real one)
        if (!bn_table(m)) return false;
        const int R = Model::kBnTab / (2 * C);
        f->tab = m->bn_tab;
        f->ticket = reinterpret_cast<unsigned*>(m->bn_tab + Model::kBnTab);
        f->R = R > kBnRows ? kBnRows : R;
        f->C = C;
        f->momentum = kBnMomentum;
        f->eps = kBnEps;
        f->n = (double)B * bn.inA.d.H * bn.inA.d.W;
        f->gamma = m->p + bn.w_off;
        f->beta = m->p + bn.b_off;
        f->mmean = m->state + bn.mm_off;
        f->mvar = m->state + bn.mv_off;
        f->coef = bn.coef;
    }
    bn.fused_stats_rows = 1;
    return true;
}
#define DN_TRYB(x) do { if (!(x)) return false; } while (0)

static unsigned bn_blocks(size_t npix, int C) {
    const size_t PL = 256 / (C / 4);
    size_t b = (npix + PL * 32 - 1) / (PL * 32);
    static const size_t cap = getenv("DNNCA_BN_BLOCKS") ? (size_t)atoi(getenv("DNNCA_BN_BLOCKS")) : 512;          // tuning aid
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// forward (training statistics, or inference with the moving statistics); returns false when the shape is not covered
bool fast_bn_pool_fusable(const Model* m, const Op& bn, const Op& pool) {
    return fast_bn_supported(m, bn) && pool.type == OP_POOL && pool.k == 2 && pool.inA.d.p == bn.out.d.p && bn.out.d.ps == bn.out.d.C &&
           pool.out.d.ps == pool.out.d.C && bn.out.d.H % 2 == 0 && bn.out.d.W % 2 == 0;
}

bool fast_bn_fwd(Model* m, int B, Op& o, bool training, float momentum, float eps, Op* pool, Op* pool_bn) {
    if (!bn_fast_ok(o.inA.d) || o.out.d.ps % 4) return false;
    const int C = o.inA.d.C;
    const size_t npix = (size_t)B * o.inA.d.H * o.inA.d.W;
    const double tb = 4.0 * npix * C;
    if (training) {
        if (o.fused_stats_rows > 0) {
            // the kernel that produced the input saw every value and has left the coefficients (bn_self_fold_args)
            o.fused_stats_rows = 0;
        } else {
            const unsigned nb = bn_blocks(npix, C);
            BnSelfFold f{};
            DN_TRYB(bn_self_fold_args(m, o, B, &f));
            o.fused_stats_rows = 0;
            if (o.inA.d.h)
                LAUNCH(m, "bn_stats", tb / 2, 3 * tb / 4,
                       hipLaunchKernelGGL(k_bn_stats_fast<true>, dim3(nb), dim3(256), 0, m->stream, npix, o.inA.d.p, C, f));
            else
                LAUNCH(m, "bn_stats", tb, 3 * tb / 4,
                       hipLaunchKernelGGL(k_bn_stats_fast<false>, dim3(nb), dim3(256), 0, m->stream, npix, o.inA.d.p, C, f));
        }
    } else {
        LAUNCH(m, "g_bn_finalize", 0, 0,
               g_bn_finalize(m->stream, C, (double)npix, o.ws, m->p + o.w_off, m->p + o.b_off, m->state + o.mm_off,
                             m->state + o.mv_off, o.coef, 0, momentum, eps));
    }
    if (o.elided) return true;          // the readers apply scale / shift themselves (Op::elided): coefficients are all they need
    const size_t n4 = npix * (C / 4);
    if (pool) {          // the caller checked fast_bn_pool_fusable(o, *pool)
        if (!pool->pool_idx && !m->dry) {
            void* ix = nullptr;
            if (m->alloc(&ix, (size_t)m->desc.max_batch * pool->out.d.H * pool->out.d.W * C) == DNNCA_OK) pool->pool_idx = (unsigned char*)ix;
        }
        pool->pool_idx_valid = m->dry || pool->pool_idx != nullptr;     // the dry run lists the launches of the real one
        // grid-stride: at most 1024 blocks (the statistics of the pooled tensor cost one LDS fold and one ticket per block: measured
        // per launch on unet_big 37.7 / 41.0 / 45.9 / 54.8 us with 1024 / 2048 / 4096 / 8192 blocks); the stride 256 x blocks is a
        // multiple of the channel groups (C / 4 divides 256: bn_fast_ok)
        static const unsigned max_blocks = getenv("DNNCA_POOL_BLOCKS") ? (unsigned)atoi(getenv("DNNCA_POOL_BLOCKS")) : 1024u;      // tuning aid
        const size_t need = (n4 / 4 + 255) / 256;
        const dim3 grid((unsigned)(need < max_blocks ? need : max_blocks));
        unsigned* ix = reinterpret_cast<unsigned*>(pool->pool_idx);
        BnSelfFold pf{};          // batch statistics of the pooled tensor for the BatchNorm behind the pool
        if (pool_bn && training && !dense_switches().no_bn_fusion && !dense_switches().no_pool_stats) (void)bn_self_fold_args(m, *pool_bn, B, &pf);
        // does anybody but the pool read the normalised tensor?  (out_readers: model.hip; -1 = a reader it does not understand)
        bool write_y = o.out_readers.empty();
        for (int r : o.out_readers) write_y = write_y || r < 0 || &m->ops[r] != pool;
        // (the pool's backward must be the one that routes by the recorded positions: any other re-reads the normalised tensor)
        write_y = write_y || !pool->pool_idx_valid || !dense(pool->inA.g) || !dense(pool->out.g) || pool->maskA;
        // bytes: x in (f32 or bf16), y out (f32 or bf16; not when only the pool reads it), pooled out (a quarter, f32)
        const double pb = tb * ((o.inA.d.h ? 0.5 : 1.0) + (write_y ? (o.out.d.h ? 0.5 : 1.0) : 0.0) + 0.25);
#define BNPOOL(YHv, XHv, WYv) LAUNCH(m, "bn_apply_pool", pb, tb / 2,                                                         \
        hipLaunchKernelGGL((k_bn_apply_pool_fast<YHv, XHv, WYv>), grid, dim3(256), 0, m->stream, n4 / 4, o.inA.d.p, o.out.d.p, \
                           pool->out.d.p, ix, C, o.inA.d.H, o.inA.d.W, o.coef, pf))
        m->set_variant("y%d", (int)write_y);
        if (!write_y) { if (o.inA.d.h) BNPOOL(false, true, false); else BNPOOL(false, false, false); }
        else if (o.out.d.h) { if (o.inA.d.h) BNPOOL(true, true, true); else BNPOOL(true, false, true); }
        else { if (o.inA.d.h) BNPOOL(false, true, true); else BNPOOL(false, false, true); }
#undef BNPOOL
        return true;
    }
    const int bu = bn_u(o.inA.d.h || o.out.d.h);
    const dim3 grid((unsigned)((n4 + 256 * bu - 1) / (256 * bu)));
    const double ab = tb * ((o.inA.d.h ? 0.5 : 1.0) + (o.out.d.h ? 0.5 : 1.0));
#define BNAPPLY(YHv, XHv) LAUNCH(m, "bn_apply", ab, tb / 2,                                                                  \
        hipLaunchKernelGGL((k_bn_apply_fast<YHv, XHv>), grid, dim3(256), 0, m->stream, n4, o.inA.d.p, o.out.d.p, C, o.out.d.ps, o.coef))
    if (o.out.d.h) { if (o.inA.d.h) BNAPPLY(true, true); else BNAPPLY(true, false); }
    else { if (o.inA.d.h) BNAPPLY(false, true); else BNAPPLY(false, false); }
#undef BNAPPLY
    return true;
}

bool fast_bn_supported(const Model* m, const Op& o) {
    return o.type == OP_BN && bn_fast_ok(o.inA.d) && o.out.d.ps % 4 == 0 && o.out.g.ps % 4 == 0 && o.inA.g.ps == o.inA.d.C;
}

// the pool's backward rides in the backward passes of the BatchNorm in front of it (PoolGrad): decided when the backward pass reaches
// the pool; fast_bn_bwd finds Op::pool_grad
bool fast_pool_into_bn(Model* m, Op& pool, Op& bn) {
    static const bool off = getenv("DNNCA_NO_POOL_BN_BWD") != nullptr;
    if (off || bn.type != OP_BN || pool.type != OP_POOL || pool.k != 2 || !fast_bn_supported(m, bn)) return false;
    if (bn.out.g.p != pool.inA.g.p || bn.out.g.C != pool.inA.g.C || !dense(pool.inA.g) || !dense(pool.out.g) || pool.out.g.h || pool.maskA) return false;
    if (!(pool.pool_idx_valid && (pool.pool_idx || m->dry))) return false;          // this step's forward recorded the positions
    if (bn.out.g.H % 2 || bn.out.g.W % 2 || (double)m->desc.max_batch * bn.out.g.H * bn.out.g.W * bn.out.g.W >= 4.0e9) return false;
    // f32 tensors only.  (With bf16-stored tensors the routed gradient -- a quarter of f32 values and a byte per pooled element -- is a
    // third more traffic for each of the two passes: measured on unet_big, bn_bwd_reduce + 28 us on the four launches concerned and
    // the step 6.87 -> 6.93 ms, against 11.98 -> 11.82 ms on mulmo_unet.)
    if (bn.inA.d.h || bn.out.g.h || bn.inA.g.h) return false;
    pool.pool_idx_valid = false;
    bn.pool_grad = &pool;
    bn.pool_grad_acc = pool.accA;          // somebody (a skip connection's conv) wrote dy before the pool would have
    return true;
}

// The data-gradient launch that produces ALL of this BatchNorm's dy (a 3x3 conv that is the only reader of the BatchNorm's output)
// takes the backward sums along in its epilogue (ConvArgs::bnb): fills `f` and marks the op so that fast_bn_bwd skips its reduction
// pass.  f32 tensors; the caller has checked that it is the only writer of dy and clears Op::bwd_sums_rode if its kernel declines.
bool bn_bwd_fold_args(Model* m, Op& bn, BnBwdFold* f) {
    if (bn.type != OP_BN || !fast_bn_supported(m, bn) || bn.pool_grad || bn.inA.d.h || bn.out.g.h || bn.inA.g.h) return false;
    if (bn.out.g.ps != bn.inA.d.C || dense_switches().no_bn_fusion) return false;
    const int C = bn.inA.d.C;
    f->C = C;          // (also in the dry run: the launch's variant name says that the sums ride)
    if (!m->dry) {
        if (!bn_table(m)) return false;
        int R = Model::kBnTab / (2 * C);
        f->tab = m->bn_tab;
        f->ticket = reinterpret_cast<unsigned*>(m->bn_tab + Model::kBnTab);
        f->R = R > kBnRows ? kBnRows : R;
        f->dgamma = m->g + bn.w_off;
        f->dbeta = m->g + bn.b_off;
        f->x = bn.inA.d.p;
        f->coef = bn.coef;
    }
    bn.bwd_sums_rode = true;
    return true;
}

bool fast_bn_bwd(Model* m, int B, Op& o) {
    if (!bn_fast_ok(o.inA.d) || o.out.g.ps % 4 || o.inA.g.ps != o.inA.d.C) return false;
    const int C = o.inA.d.C;
    const size_t npix = (size_t)B * o.inA.d.H * o.inA.d.W;
    const double tb = 4.0 * npix * C;
    const unsigned nb = bn_blocks(npix, C);
    if (!m->dry) DN_TRYB(bn_table(m));
    int R = Model::kBnTab / (2 * C);          // C <= 1024 (bn_fast_ok): at least two rows
    if (R > kBnRows) R = kBnRows;
    if (R > (int)nb) R = (int)nb;
    unsigned* ticket = reinterpret_cast<unsigned*>(m->bn_tab + Model::kBnTab);
    const bool xh = o.inA.d.h != 0, gh = o.out.g.h != 0, dh = o.inA.g.h != 0;
    const Op* pool = o.pool_grad;
    o.pool_grad = nullptr;
    const int pgm = pool ? (o.pool_grad_acc ? 2 : 1) : 0;
    PoolGrad pg{};
    if (pool) {
        pg.dpool = pool->out.g.p;
        pg.idx = reinterpret_cast<const unsigned*>(pool->pool_idx);
        pg.H = o.inA.d.H; pg.W = o.inA.d.W;
        pg.mW = (unsigned)(0xffffffffull / (unsigned)pg.W) + 1u;
        pg.mH = (unsigned)(0xffffffffull / (unsigned)pg.H) + 1u;
    }
    // bytes: x, dy (not when the routed gradient is all there is), + a quarter of f32 pooled gradient and a byte per pooled element
    const double pgb = pool ? tb * (0.25 + 0.0625) : 0.0;
    const double rb = tb * ((xh ? 0.5 : 1.0) + (pgm == 1 ? 0.0 : (gh ? 0.5 : 1.0))) + pgb;
    m->set_variant("p%d", pgm);
#define BNRED(XHv, GHv, PGv) LAUNCH(m, "bn_bwd_reduce", rb, tb,                                                              \
        hipLaunchKernelGGL((k_bn_bwd_reduce_fast<XHv, GHv, PGv>), dim3(nb), dim3(256), 0, m->stream, npix, o.inA.d.p, o.out.g.p, C, \
                           o.out.g.ps, o.coef, m->bn_tab, R, ticket, m->g + o.w_off, m->g + o.b_off, pg))
    const bool rode = o.bwd_sums_rode;          // the launch that produced dy has left dgamma / dbeta (bn_bwd_fold_args)
    o.bwd_sums_rode = false;
    if (rode) {
    } else if (pgm) {          // (fast_pool_into_bn: f32 tensors)
        if (pgm == 1) BNRED(false, false, 1); else BNRED(false, false, 2);
    } else if (xh) { if (gh) BNRED(true, true, 0); else BNRED(true, false, 0); }
    else { if (gh) BNRED(false, true, 0); else BNRED(false, false, 0); }
#undef BNRED
    const size_t n4 = npix * (C / 4);
    const int bu = bn_u(!pgm && (dh || xh || gh));          // (the pooled-gradient variants run on f32 tensors)
    const dim3 grid((unsigned)((n4 + 256 * bu - 1) / (256 * bu)));
    m->set_variant("p%d", pgm);
#define BNBWD(DHv, XHv, GHv, PGv) LAUNCH(m, "bn_bwd_apply", rb + tb * (DHv ? 0.5 : 1.0), 2 * tb,                             \
        hipLaunchKernelGGL((k_bn_bwd_apply_fast<DHv, XHv, GHv, PGv>), grid, dim3(256), 0, m->stream, n4, o.inA.d.p, o.out.g.p, \
                           o.inA.g.p, C, o.out.g.ps, DHv ? 0 : (int)o.accA, o.coef, m->p + o.w_off, m->g + o.w_off,          \
                           m->g + o.b_off, (float)(1.0 / (double)npix), (int)o.maskA, o.mask_alpha, pg))
#define BNBWD2(DHv, XHv) do { if (gh) BNBWD(DHv, XHv, true, 0); else BNBWD(DHv, XHv, false, 0); } while (0)
    if (pgm) {
        if (pgm == 1) BNBWD(false, false, false, 1); else BNBWD(false, false, false, 2);
    } else if (dh) { if (xh) BNBWD2(true, true); else BNBWD2(true, false); }
    else { if (xh) BNBWD2(false, true); else BNBWD2(false, false); }
#undef BNBWD2
#undef BNBWD
    return true;
}

}  // namespace dnnca
