// kernels_first.hip -- the 3x3 convolutions that read the network input (one channel in, CO out: the first layer of
// configs/unet_big.yaml and of each configs/mulmo_unet.yaml encoder; components.py:46-52).  K = 9: nothing for the matrix
// cores to do, the layer is a pure HBM pass (write CO channels forward, read CO channels of dY for the weight gradient).
//     thread = (4-channel group cq, pixel lane pl); a block owns an 8 x 32 pixel tile whose 10 x 34 input patch sits in
//     LDS; the 36 weights (+4 biases) of the thread's channel group stay in registers; outputs / dY move as float4,
//     consecutive lanes covering consecutive channels of one pixel (NHWC: 16*G contiguous bytes per pixel).
// The input needs no gradient, so backward is the weight (+bias) gradient only, with the act' mask applied on the fly
// when the producer of dY did not already do so.
#include "bn_dev.h"
#include "fast.h"
#include "kernels.h"

namespace dnnca {
namespace first {

constexpr int TY = 8, TX = 32, PW = TX + 2, PH = TY + 2;

struct Args {
    const float* x;      // network input, pixel stride xps (this encoder's channel already selected)
    int xps;
    const float* w;      // [9][CO]
    const float* bias;   // [CO]
    float* y;            // forward output / backward: activated output (mask source), dense NHWC
    const float* dy;     // backward: gradient of the output
    float* dw;           // [9][CO]   (backward: copy 0 of the bucket copies, see k_first_wgrad)
    float* db;           // [CO]
    int B, H, W, CO;
    int tiles_x, tiles_y, ntiles;
    float alpha;         // forward: activation slope (<0 none).  backward: slope of act' when `mask`
    int mask;
    int nbuckets, bucket_stride;     // backward: block b adds into copy b % nbuckets (copies bucket_stride floats apart)
    BnSelfFold bnf;      // forward: batch statistics of the output for the BatchNorm behind it, self-folding (bn_dev.h; tab == nullptr: none)
};

__device__ __forceinline__ void stage_patch(const Args& p, float* xs, int b, int y0, int x0) {
    for (int i = threadIdx.x; i < PH * PW; i += 256) {
        const int ly = i / PW, lx = i - ly * PW;
        const int iy = y0 - 1 + ly, ix = x0 - 1 + lx;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) v = p.x[(((size_t)b * p.H + iy) * p.W + ix) * p.xps];
        xs[i] = v;
    }
}

__global__ __launch_bounds__(256) void k_first_fwd(Args p) {
    __shared__ float xs[PH * PW];
    __shared__ float red[256][8];
    const int G = p.CO >> 2, cq = threadIdx.x % G, pl = threadIdx.x / G, PL = 256 / G;
    float4 bs = make_float4(0.f, 0.f, 0.f, 0.f), bq = make_float4(0.f, 0.f, 0.f, 0.f);      // batch statistics of this thread's outputs
    float4 w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = *reinterpret_cast<const float4*>(p.w + t * p.CO + 4 * cq);
    const float4 bias = *reinterpret_cast<const float4*>(p.bias + 4 * cq);
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int bx = tile % p.tiles_x, by = (tile / p.tiles_x) % p.tiles_y, b = tile / (p.tiles_x * p.tiles_y);
        const int x0 = bx * TX, y0 = by * TY;
        __syncthreads();
        stage_patch(p, xs, b, y0, x0);
        __syncthreads();
        for (int px = pl; px < TY * TX; px += PL) {
            const int ly = px / TX, lx = px - ly * TX;
            if (y0 + ly >= p.H || x0 + lx >= p.W) continue;
            float4 a = bias;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float v = xs[(ly + t / 3) * PW + lx + t % 3];
                a.x = fmaf(v, w[t].x, a.x); a.y = fmaf(v, w[t].y, a.y); a.z = fmaf(v, w[t].z, a.z); a.w = fmaf(v, w[t].w, a.w);
            }
            if (p.alpha >= 0.f) {
                a.x = a.x > 0.f ? a.x : p.alpha * a.x; a.y = a.y > 0.f ? a.y : p.alpha * a.y;
                a.z = a.z > 0.f ? a.z : p.alpha * a.z; a.w = a.w > 0.f ? a.w : p.alpha * a.w;
            }
            *reinterpret_cast<float4*>(p.y + (((size_t)b * p.H + y0 + ly) * p.W + x0 + lx) * p.CO + 4 * cq) = a;
            bs.x += a.x; bs.y += a.y; bs.z += a.z; bs.w += a.w;
            bq.x = fmaf(a.x, a.x, bq.x); bq.y = fmaf(a.y, a.y, bq.y); bq.z = fmaf(a.z, a.z, bq.z); bq.w = fmaf(a.w, a.w, bq.w);
        }
    }
    if (p.bnf.tab) {        // the block's sums go to one bucket row: [sums (CO), sums of squares (CO)]
        red[threadIdx.x][0] = bs.x; red[threadIdx.x][1] = bs.y; red[threadIdx.x][2] = bs.z; red[threadIdx.x][3] = bs.w;
        red[threadIdx.x][4] = bq.x; red[threadIdx.x][5] = bq.y; red[threadIdx.x][6] = bq.z; red[threadIdx.x][7] = bq.w;
        __syncthreads();
        for (int o = threadIdx.x; o < 2 * p.CO; o += 256) {
            const int c = o % p.CO, which = o / p.CO;
            float acc = 0.f;
            for (int l = 0; l < PL; ++l) acc += red[l * G + (c >> 2)][4 * which + (c & 3)];
            atomicAdd(bn_bucket(p.bnf, (int)blockIdx.x) + o, (double)acc);
        }
        bn_self_fold(p.bnf, gridDim.x, blockIdx.x);
    }
}

// persistent blocks; per-thread partial sums over the block's tiles, one LDS reduction and 10*CO atomics per block into one of
// FIRST_BUCKETS copies of the gradient (a thousand same-address atomics would take longer than the pass itself: they execute
// one after the other, ~56 ns each); k_first_fold sums the copies and leaves them zeroed
__global__ __launch_bounds__(256) void k_first_wgrad(Args p) {
    __shared__ float xs[PH * PW];
    __shared__ float red[256 * 4];
    const int G = p.CO >> 2, cq = threadIdx.x % G, pl = threadIdx.x / G, PL = 256 / G;
    float4 acc[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int bx = tile % p.tiles_x, by = (tile / p.tiles_x) % p.tiles_y, b = tile / (p.tiles_x * p.tiles_y);
        const int x0 = bx * TX, y0 = by * TY;
        __syncthreads();
        stage_patch(p, xs, b, y0, x0);
        __syncthreads();
        constexpr int U = 4;            // dY loads in flight per thread
        for (int px0 = pl; px0 < TY * TX; px0 += U * PL) {
            float4 d[U];
            int ly[U], lx[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int px = px0 + u * PL;
                ly[u] = px / TX; lx[u] = px - ly[u] * TX;
                d[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (px < TY * TX && y0 + ly[u] < p.H && x0 + lx[u] < p.W) {
                    const size_t o = (((size_t)b * p.H + y0 + ly[u]) * p.W + x0 + lx[u]) * p.CO + 4 * cq;
                    d[u] = *reinterpret_cast<const float4*>(p.dy + o);
                    if (p.mask) {
                        const float4 yv = *reinterpret_cast<const float4*>(p.y + o);
                        d[u].x *= yv.x > 0.f ? 1.0f : p.alpha; d[u].y *= yv.y > 0.f ? 1.0f : p.alpha;
                        d[u].z *= yv.z > 0.f ? 1.0f : p.alpha; d[u].w *= yv.w > 0.f ? 1.0f : p.alpha;
                    }
                } else {
                    ly[u] = 0; lx[u] = 0;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float v = xs[(ly[u] + t / 3) * PW + lx[u] + t % 3];
                    acc[t].x = fmaf(v, d[u].x, acc[t].x); acc[t].y = fmaf(v, d[u].y, acc[t].y);
                    acc[t].z = fmaf(v, d[u].z, acc[t].z); acc[t].w = fmaf(v, d[u].w, acc[t].w);
                }
                acc[9].x += d[u].x; acc[9].y += d[u].y; acc[9].z += d[u].z; acc[9].w += d[u].w;
            }
        }
    }
    // sum over the PL pixel lanes of the block, one tap at a time
    for (int t = 0; t < 10; ++t) {
        __syncthreads();
        *reinterpret_cast<float4*>(red + 4 * threadIdx.x) = acc[t];
        __syncthreads();
        if (threadIdx.x < p.CO) {
            const int c = threadIdx.x;          // channel c = 4*cq' + k lives at red[4*(l*G + cq') + k] = red[4*l*G + c]
            float s = 0.f;
            for (int l = 0; l < PL; ++l) s += red[4 * l * G + c];
            atomicAdd((t < 9 ? p.dw + t * p.CO : p.db) + (size_t)(blockIdx.x % p.nbuckets) * p.bucket_stride + c, s);
        }
    }
}

__global__ __launch_bounds__(256) void k_first_fold(float* __restrict__ slabs, int nb, int stride, int n, float* __restrict__ dw,
                                                    float* __restrict__ db, int n_w) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int b = 0; b < nb; ++b) {
        s += slabs[(size_t)b * stride + i];
        slabs[(size_t)b * stride + i] = 0.f;
    }
    if (i < n_w) dw[i] += s;
    else db[i - n_w] += s;
}

}  // namespace first

static bool first_supported(const Op& o) {
    if (o.type != OP_CONV || o.k != 3 || o.need_din) return false;
    if (o.inA.d.C != 1 || o.inB.d.C != 0) return false;
    const int CO = o.out.d.C;
    if (o.out.d.ps != CO || CO % 4 || CO < 4 || CO > 256 || 256 % (CO / 4)) return false;
    return true;
}

static first::Args first_args(Model* m, int B, Op& o) {
    first::Args a{};
    a.x = o.inA.d.p; a.xps = o.inA.d.ps;
    a.w = m->p + o.w_off; a.bias = m->p + o.b_off;
    a.y = o.out.d.p;
    a.dy = o.out.g.p;
    a.dw = m->g + o.w_off; a.db = m->g + o.b_off;
    a.B = B; a.H = o.out.d.H; a.W = o.out.d.W; a.CO = o.out.d.C;
    a.tiles_x = (a.W + first::TX - 1) / first::TX;
    a.tiles_y = (a.H + first::TY - 1) / first::TY;
    a.ntiles = a.tiles_x * a.tiles_y * B;
    return a;
}

bool fast_first_conv_fwd(Model* m, int B, Op& o, double bytes, double flops, Op* bn_next) {
    if (!first_supported(o)) return false;
    first::Args a = first_args(m, B, o);
    a.alpha = o.alpha;
    const int blocks = a.ntiles < 4096 ? a.ntiles : 4096;
    if (bn_next && !dense_switches().no_bn_fusion)       // the BatchNorm behind this conv takes its batch statistics from here
        (void)bn_self_fold_args(m, *bn_next, B, &a.bnf);
    LAUNCH(m, "first_fwd", bytes, flops, hipLaunchKernelGGL(first::k_first_fwd, dim3(blocks), dim3(256), 0, m->stream, a));
    return true;
}

bool fast_first_conv_bwd(Model* m, int B, Op& o, double out_bytes, double in_bytes, double flops) {
    if (!first_supported(o) || o.out.g.ps != o.out.d.C) return false;
    first::Args a = first_args(m, B, o);
    a.mask = (o.alpha >= 0.f && !o.premasked) ? 1 : 0;
    a.alpha = o.alpha;
    const int blocks = a.ntiles < 1024 ? a.ntiles : 1024;
    constexpr int FIRST_BUCKETS = 32, STRIDE = 10 * 256;      // CO <= 256
    if (!m->first_slabs && !m->dry && m->alloc((void**)&m->first_slabs, (size_t)FIRST_BUCKETS * STRIDE * 4) != DNNCA_OK) return false;
    a.dw = m->first_slabs;
    a.db = m->first_slabs + 9 * a.CO;
    a.nbuckets = FIRST_BUCKETS;
    a.bucket_stride = STRIDE;
    LAUNCH(m, "first_wgrad", out_bytes * (a.mask ? 2 : 1) + in_bytes, flops,
           hipLaunchKernelGGL(first::k_first_wgrad, dim3(blocks), dim3(256), 0, m->stream, a));
    LAUNCH(m, "first_fold", 4.0 * FIRST_BUCKETS * 10 * a.CO, 0,
           hipLaunchKernelGGL(first::k_first_fold, dim3((10 * a.CO + 255) / 256), dim3(256), 0, m->stream, m->first_slabs, FIRST_BUCKETS,
                              STRIDE, 10 * a.CO, m->g + o.w_off, m->g + o.b_off, 9 * a.CO));
    return true;
}

}  // namespace dnnca
