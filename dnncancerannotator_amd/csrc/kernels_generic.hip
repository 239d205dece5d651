// kernels_generic.hip -- shape-generic, untuned HIP kernels: any channel count / kernel size / rate, strided views.
// They are the complete functional path (and the in-library cross-check of the tuned kernels); the hot configurations
// are routed to kernels_direct.hip / kernels_mfma.hip by model.hip.
#include "kernels.h"

namespace dnnca {

static constexpr int TB = 256;

static inline unsigned nblk(size_t n, int per = TB) { return (unsigned)((n + per - 1) / per); }

__device__ __forceinline__ float act_apply(float z, float alpha) {
    // alpha < 0: identity
    return alpha < 0.f ? z : (z > 0.f ? z : alpha * z);
}

// ------------------------------------------------------------------------------------------------ conv forward
__global__ void k_conv_fwd(int B, View A, View Bv, const float* __restrict__ w, const float* __restrict__ bias, View out,
                           int K, float alpha) {
    const int H = out.H, W = out.W, Cout = out.C, CA = A.C, CB = Bv.C, Cin = CA + CB;
    const int pad = (K - 1) / 2;
    size_t idx = (size_t)blockIdx.x * TB + threadIdx.x;
    size_t total = (size_t)B * H * W * Cout;
    if (idx >= total) return;
    int co = idx % Cout;
    size_t pix = idx / Cout;
    int x = pix % W;
    int y = (pix / W) % H;
    int b = pix / ((size_t)W * H);
    float acc = bias[co];
    for (int ky = 0; ky < K; ++ky) {
        int iy = y + ky - pad;
        if (iy < 0 || iy >= H) continue;
        for (int kx = 0; kx < K; ++kx) {
            int ix = x + kx - pad;
            if (ix < 0 || ix >= W) continue;
            const float* wp = w + ((size_t)(ky * K + kx) * Cin) * Cout + co;
            const float* ap = A.p + (((size_t)b * H + iy) * W + ix) * A.ps;
            for (int ci = 0; ci < CA; ++ci) acc = fmaf(ap[ci], wp[(size_t)ci * Cout], acc);
            if (CB) {
                const float* bp = Bv.p + (((size_t)b * H + iy) * W + ix) * Bv.ps;
                for (int ci = 0; ci < CB; ++ci) acc = fmaf(bp[ci], wp[(size_t)(CA + ci) * Cout], acc);
            }
        }
    }
    out.p[pix * out.ps + co] = act_apply(acc, alpha);
}

void g_conv_fwd(hipStream_t s, int B, View A, View Bv, const float* w, const float* bias, View out, int K, float alpha) {
    size_t total = (size_t)B * out.H * out.W * out.C;
    hipLaunchKernelGGL(k_conv_fwd, dim3(nblk(total)), dim3(TB), 0, s, B, A, Bv, w, bias, out, K, alpha);
}

__global__ void k_act_bwd(size_t n, float* __restrict__ dy, const float* __restrict__ y, float alpha) {
    size_t i = (size_t)blockIdx.x * TB + threadIdx.x;
    if (i < n) dy[i] = dy[i] * (y[i] > 0.f ? 1.f : alpha);
}

void g_act_bwd(hipStream_t s, size_t n, float* dy, const float* y, float alpha) {
    hipLaunchKernelGGL(k_act_bwd, dim3(nblk(n)), dim3(TB), 0, s, n, dy, y, alpha);
}

// ------------------------------------------------------------------------------------------------ conv dgrad
__global__ void k_conv_dgrad(int B, View dz, const float* __restrict__ w, View dA, int accA, View dB, int accB, int K) {
    const int H = dz.H, W = dz.W, Cout = dz.C, CA = dA.C, CB = dB.C, Cin = CA + CB;
    const int pad = (K - 1) / 2;
    size_t idx = (size_t)blockIdx.x * TB + threadIdx.x;
    size_t total = (size_t)B * H * W * Cin;
    if (idx >= total) return;
    int ci = idx % Cin;
    size_t pix = idx / Cin;
    int x = pix % W;
    int y = (pix / W) % H;
    int b = pix / ((size_t)W * H);
    float acc = 0.f;
    for (int ky = 0; ky < K; ++ky) {
        int oy = y - ky + pad;
        if (oy < 0 || oy >= H) continue;
        for (int kx = 0; kx < K; ++kx) {
            int ox = x - kx + pad;
            if (ox < 0 || ox >= W) continue;
            const float* wp = w + ((size_t)(ky * K + kx) * Cin + ci) * Cout;
            const float* gp = dz.p + (((size_t)b * H + oy) * W + ox) * dz.ps;
            for (int co = 0; co < Cout; ++co) acc = fmaf(gp[co], wp[co], acc);
        }
    }
    if (ci < CA) {
        float* d = dA.p + pix * dA.ps + ci;
        *d = accA ? *d + acc : acc;
    } else {
        float* d = dB.p + pix * dB.ps + (ci - CA);
        *d = accB ? *d + acc : acc;
    }
}

void g_conv_dgrad(hipStream_t s, int B, View dz, const float* w, View dA, int accA, View dB, int accB, int K) {
    size_t total = (size_t)B * dz.H * dz.W * (dA.C + dB.C);
    hipLaunchKernelGGL(k_conv_dgrad, dim3(nblk(total)), dim3(TB), 0, s, B, dz, w, dA, accA, dB, accB, K);
}

// ------------------------------------------------------------------------------------------------ conv wgrad
// block = a chunk of pixels; thread t owns weights t, t+TB, ... (co fastest -> coalesced dz reads)
__global__ void k_conv_wgrad(int B, View A, View Bv, View dz, float* __restrict__ dw, float* __restrict__ dbias, int K,
                             int ppb) {
    const int H = dz.H, W = dz.W, Cout = dz.C, CA = A.C, CB = Bv.C, Cin = CA + CB;
    const int pad = (K - 1) / 2;
    const size_t npix = (size_t)B * H * W;
    const size_t p0 = (size_t)blockIdx.x * ppb;
    const size_t p1 = p0 + ppb < npix ? p0 + ppb : npix;
    const int nW = K * K * Cin * Cout;
    for (int wi = threadIdx.x; wi < nW + Cout; wi += TB) {
        float acc = 0.f;
        if (wi >= nW) {
            int co = wi - nW;
            for (size_t p = p0; p < p1; ++p) acc += dz.p[p * dz.ps + co];
            atomicAdd(dbias + co, acc);
            continue;
        }
        int co = wi % Cout;
        int ci = (wi / Cout) % Cin;
        int tap = wi / (Cout * Cin);
        int ky = tap / K, kx = tap % K;
        for (size_t p = p0; p < p1; ++p) {
            int x = p % W;
            int y = (p / W) % H;
            int iy = y + ky - pad, ix = x + kx - pad;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            size_t q = p + (ptrdiff_t)(ky - pad) * W + (kx - pad);
            float xin = ci < CA ? A.p[q * A.ps + ci] : Bv.p[q * Bv.ps + (ci - CA)];
            acc = fmaf(xin, dz.p[p * dz.ps + co], acc);
        }
        atomicAdd(dw + wi, acc);
    }
}

void g_conv_wgrad(hipStream_t s, int B, View A, View Bv, View dz, float* dw, float* dbias, int K) {
    size_t npix = (size_t)B * dz.H * dz.W;
    int ppb = (int)((npix + 2047) / 2048);
    if (ppb < 16) ppb = 16;
    hipLaunchKernelGGL(k_conv_wgrad, dim3(nblk(npix, ppb)), dim3(TB), 0, s, B, A, Bv, dz, dw, dbias, K, ppb);
}

// ------------------------------------------------------------------------------------------------ max pool
__global__ void k_pool_fwd(int B, View in, View out, int r) {
    const int Ho = out.H, Wo = out.W, C = out.C;
    size_t idx = (size_t)blockIdx.x * TB + threadIdx.x;
    size_t total = (size_t)B * Ho * Wo * C;
    if (idx >= total) return;
    int c = idx % C;
    size_t pix = idx / C;
    int ox = pix % Wo;
    int oy = (pix / Wo) % Ho;
    int b = pix / ((size_t)Wo * Ho);
    float m = -INFINITY;
    for (int a = 0; a < r; ++a)
        for (int e = 0; e < r; ++e) {
            float v = in.p[(((size_t)b * in.H + oy * r + a) * in.W + ox * r + e) * in.ps + c];
            m = v > m ? v : m;
        }
    out.p[pix * out.ps + c] = m;
}

void g_pool_fwd(hipStream_t s, int B, View in, View out, int r) {
    size_t total = (size_t)B * out.H * out.W * out.C;
    hipLaunchKernelGGL(k_pool_fwd, dim3(nblk(total)), dim3(TB), 0, s, B, in, out, r);
}

__global__ void k_pool_bwd(int B, View in, View out, View dout, View din, int acc, int r) {
    const int Ho = out.H, Wo = out.W, C = out.C;
    size_t idx = (size_t)blockIdx.x * TB + threadIdx.x;
    size_t total = (size_t)B * Ho * Wo * C;
    if (idx >= total) return;
    int c = idx % C;
    size_t pix = idx / C;
    int ox = pix % Wo;
    int oy = (pix / Wo) % Ho;
    int b = pix / ((size_t)Wo * Ho);
    float m = out.p[pix * out.ps + c];
    float g = dout.p[pix * dout.ps + c];
    bool found = false;
    for (int a = 0; a < r; ++a)
        for (int e = 0; e < r; ++e) {
            size_t q = ((size_t)b * in.H + oy * r + a) * in.W + ox * r + e;
            float v = in.p[q * in.ps + c];
            float d = 0.f;
            if (!found && v == m) {
                d = g;
                found = true;
            }
            float* dp = din.p + q * din.ps + c;
            *dp = acc ? *dp + d : d;
        }
}

void g_pool_bwd(hipStream_t s, int B, View in, View out, View dout, View din, int acc, int r) {
    size_t total = (size_t)B * out.H * out.W * out.C;
    hipLaunchKernelGGL(k_pool_bwd, dim3(nblk(total)), dim3(TB), 0, s, B, in, out, dout, din, acc, r);
}

// ------------------------------------------------------------------------------------------------ transposed conv
__global__ void k_tconv_fwd(int B, View in, const float* __restrict__ w, const float* __restrict__ bias, View out, int r) {
    const int Ho = out.H, Wo = out.W, Cout = out.C, Cin = in.C;
    size_t idx = (size_t)blockIdx.x * TB + threadIdx.x;
    size_t total = (size_t)B * Ho * Wo * Cout;
    if (idx >= total) return;
    int co = idx % Cout;
    size_t pix = idx / Cout;
    int X = pix % Wo;
    int Y = (pix / Wo) % Ho;
    int b = pix / ((size_t)Wo * Ho);
    int i = Y / r, a = Y % r, j = X / r, e = X % r;
    const float* ip = in.p + (((size_t)b * in.H + i) * in.W + j) * in.ps;
    const float* wp = w + ((size_t)(a * r + e) * Cout + co) * Cin;
    float acc = bias[co];
    for (int ci = 0; ci < Cin; ++ci) acc = fmaf(ip[ci], wp[ci], acc);
    out.p[pix * out.ps + co] = acc;
}

void g_tconv_fwd(hipStream_t s, int B, View in, const float* w, const float* bias, View out, int r) {
    size_t total = (size_t)B * out.H * out.W * out.C;
    hipLaunchKernelGGL(k_tconv_fwd, dim3(nblk(total)), dim3(TB), 0, s, B, in, w, bias, out, r);
}

__global__ void k_tconv_dgrad(int B, View dout, const float* __restrict__ w, View din, int acc, int r) {
    const int H = din.H, W = din.W, Cin = din.C, Cout = dout.C;
    size_t idx = (size_t)blockIdx.x * TB + threadIdx.x;
    size_t total = (size_t)B * H * W * Cin;
    if (idx >= total) return;
    int ci = idx % Cin;
    size_t pix = idx / Cin;
    int j = pix % W;
    int i = (pix / W) % H;
    int b = pix / ((size_t)W * H);
    float s = 0.f;
    for (int a = 0; a < r; ++a)
        for (int e = 0; e < r; ++e) {
            const float* gp = dout.p + (((size_t)b * dout.H + i * r + a) * dout.W + j * r + e) * dout.ps;
            const float* wp = w + ((size_t)(a * r + e) * Cout) * Cin + ci;
            for (int co = 0; co < Cout; ++co) s = fmaf(gp[co], wp[(size_t)co * Cin], s);
        }
    float* d = din.p + pix * din.ps + ci;
    *d = acc ? *d + s : s;
}

void g_tconv_dgrad(hipStream_t s, int B, View dout, const float* w, View din, int acc, int r) {
    size_t total = (size_t)B * din.H * din.W * din.C;
    hipLaunchKernelGGL(k_tconv_dgrad, dim3(nblk(total)), dim3(TB), 0, s, B, dout, w, din, acc, r);
}

__global__ void k_tconv_wgrad(int B, View in, View dout, float* __restrict__ dw, float* __restrict__ dbias, int r, int ppb) {
    const int H = in.H, W = in.W, Cin = in.C, Cout = dout.C;
    const size_t npix = (size_t)B * H * W;
    const size_t p0 = (size_t)blockIdx.x * ppb;
    const size_t p1 = p0 + ppb < npix ? p0 + ppb : npix;
    const int nW = r * r * Cout * Cin;
    for (int wi = threadIdx.x; wi < nW + Cout; wi += TB) {
        float acc = 0.f;
        if (wi >= nW) {
            int co = wi - nW;
            for (size_t p = p0; p < p1; ++p) {
                int j = p % W;
                size_t bi = p / W;   // b*H + i
                for (int a = 0; a < r; ++a)
                    for (int e = 0; e < r; ++e)
                        acc += dout.p[((bi * r + a) * dout.W + j * r + e) * dout.ps + co];
            }
            atomicAdd(dbias + co, acc);
            continue;
        }
        int ci = wi % Cin;
        int co = (wi / Cin) % Cout;
        int tap = wi / (Cin * Cout);
        int a = tap / r, e = tap % r;
        for (size_t p = p0; p < p1; ++p) {
            int j = p % W;
            size_t bi = p / W;
            float g = dout.p[((bi * r + a) * dout.W + j * r + e) * dout.ps + co];
            acc = fmaf(g, in.p[p * in.ps + ci], acc);
        }
        atomicAdd(dw + wi, acc);
    }
}

void g_tconv_wgrad(hipStream_t s, int B, View in, View dout, float* dw, float* dbias, int r) {
    size_t npix = (size_t)B * in.H * in.W;
    int ppb = (int)((npix + 2047) / 2048);
    if (ppb < 16) ppb = 16;
    hipLaunchKernelGGL(k_tconv_wgrad, dim3(nblk(npix, ppb)), dim3(TB), 0, s, B, in, dout, dw, dbias, r, ppb);
}

// ------------------------------------------------------------------------------------------------ batch norm
// Threads stride over the flattened (pixel, channel) index with a stride that is a multiple of C, so one thread
// only ever sees one channel; per-thread float partials are merged with double atomics.
template <int MODE>   // 0: sum x -> ws[c]; 1: sum (x-mean)^2 -> ws[C + c]
__global__ void k_bn_stats(size_t npix, View x, double* __restrict__ ws) {
    const int C = x.C;
    size_t T = (size_t)gridDim.x * TB;
    T = (T / C) * C;
    size_t g = (size_t)blockIdx.x * TB + threadIdx.x;
    if (g >= T) return;
    int c = g % C;
    size_t total = npix * C;
    float mean = 0.f;
    if (MODE == 1) mean = (float)(ws[c] / (double)npix);
    float acc = 0.f;
    double dacc = 0.0;
    int cnt = 0;
    for (size_t i = g; i < total; i += T) {
        float v = x.p[(i / C) * x.ps + c];
        if (MODE == 1) {
            v -= mean;
            v *= v;
        }
        acc += v;
        if (++cnt == 256) {   // bound the float partial's error on very long columns
            dacc += acc;
            acc = 0.f;
            cnt = 0;
        }
    }
    dacc += acc;
    atomicAdd(ws + (MODE == 1 ? C : 0) + c, dacc);
}

static unsigned bn_grid(size_t npix, int C) {
    size_t total = npix * C;
    size_t blocks = (total + TB * 8 - 1) / (TB * 8);
    size_t minb = ((size_t)C + TB - 1) / TB + 1;
    if (blocks < minb) blocks = minb;
    if (blocks > 4096) blocks = 4096;
    return (unsigned)blocks;
}

void g_bn_stats_mean(hipStream_t s, int B, View x, double* ws) {
    size_t npix = (size_t)B * x.H * x.W;
    hipLaunchKernelGGL(k_bn_stats<0>, dim3(bn_grid(npix, x.C)), dim3(TB), 0, s, npix, x, ws);
}
void g_bn_stats_var(hipStream_t s, int B, View x, double* ws) {
    size_t npix = (size_t)B * x.H * x.W;
    hipLaunchKernelGGL(k_bn_stats<1>, dim3(bn_grid(npix, x.C)), dim3(TB), 0, s, npix, x, ws);
}

__global__ void k_bn_finalize(int C, double n, const double* __restrict__ ws, const float* __restrict__ gamma,
                              const float* __restrict__ beta, float* __restrict__ mmean, float* __restrict__ mvar,
                              float* __restrict__ coef, int training, float momentum, float eps) {
    int c = blockIdx.x * TB + threadIdx.x;
    if (c >= C) return;
    float mean, var;
    if (training) {
        mean = (float)(ws[c] / n);
        var = (float)(ws[C + c] / n);
        float unbiased = (float)(ws[C + c] / (n > 1.0 ? n - 1.0 : 1.0));
        mmean[c] = mmean[c] * momentum + mean * (1.f - momentum);
        mvar[c] = mvar[c] * momentum + unbiased * (1.f - momentum);
    } else {
        mean = mmean[c];
        var = mvar[c];
    }
    float inv = 1.0f / sqrtf(var + eps);
    float sc = gamma[c] * inv;
    coef[c] = sc;
    coef[C + c] = beta[c] - mean * sc;
    coef[2 * C + c] = mean;
    coef[3 * C + c] = inv;
}

void g_bn_finalize(hipStream_t s, int C, double n, const double* ws, const float* gamma, const float* beta, float* mmean,
                   float* mvar, float* coef, int training, float momentum, float eps) {
    hipLaunchKernelGGL(k_bn_finalize, dim3(nblk(C)), dim3(TB), 0, s, C, n, ws, gamma, beta, mmean, mvar, coef, training,
                       momentum, eps);
}

__global__ void k_bn_apply(size_t npix, View x, View y, const float* __restrict__ coef) {
    const int C = x.C;
    size_t i = (size_t)blockIdx.x * TB + threadIdx.x;
    if (i >= npix * C) return;
    int c = i % C;
    size_t p = i / C;
    y.p[p * y.ps + c] = fmaf(x.p[p * x.ps + c], coef[c], coef[C + c]);
}

void g_bn_apply(hipStream_t s, int B, View x, View y, const float* coef) {
    size_t npix = (size_t)B * x.H * x.W;
    hipLaunchKernelGGL(k_bn_apply, dim3(nblk(npix * x.C)), dim3(TB), 0, s, npix, x, y, coef);
}

__global__ void k_bn_bwd_reduce(size_t npix, View x, View dy, const float* __restrict__ coef, float* __restrict__ dgamma,
                                float* __restrict__ dbeta) {
    const int C = x.C;
    size_t T = (size_t)gridDim.x * TB;
    T = (T / C) * C;
    size_t g = (size_t)blockIdx.x * TB + threadIdx.x;
    if (g >= T) return;
    int c = g % C;
    size_t total = npix * C;
    float mean = coef[2 * C + c], inv = coef[3 * C + c];
    float sg = 0.f, sb = 0.f;
    for (size_t i = g; i < total; i += T) {
        size_t p = i / C;
        float d = dy.p[p * dy.ps + c];
        float xh = (x.p[p * x.ps + c] - mean) * inv;
        sg = fmaf(d, xh, sg);
        sb += d;
    }
    atomicAdd(dgamma + c, sg);
    atomicAdd(dbeta + c, sb);
}

void g_bn_bwd_reduce(hipStream_t s, int B, View x, View dy, const float* coef, float* dgamma, float* dbeta) {
    size_t npix = (size_t)B * x.H * x.W;
    hipLaunchKernelGGL(k_bn_bwd_reduce, dim3(bn_grid(npix, x.C)), dim3(TB), 0, s, npix, x, dy, coef, dgamma, dbeta);
}

__global__ void k_bn_bwd_apply(size_t npix, View x, View dy, View dx, int acc, const float* __restrict__ coef,
                               const float* __restrict__ gamma, const float* __restrict__ dgamma,
                               const float* __restrict__ dbeta, float inv_n) {
    const int C = x.C;
    size_t i = (size_t)blockIdx.x * TB + threadIdx.x;
    if (i >= npix * C) return;
    int c = i % C;
    size_t p = i / C;
    float mean = coef[2 * C + c], inv = coef[3 * C + c];
    float xh = (x.p[p * x.ps + c] - mean) * inv;
    float d = dy.p[p * dy.ps + c];
    float r = gamma[c] * inv * (d - inv_n * (dbeta[c] + xh * dgamma[c]));
    float* o = dx.p + p * dx.ps + c;
    *o = acc ? *o + r : r;
}

void g_bn_bwd_apply(hipStream_t s, int B, View x, View dy, View dx, int acc, const float* coef, const float* gamma,
                    const float* dgamma, const float* dbeta, double n) {
    size_t npix = (size_t)B * x.H * x.W;
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3(nblk(npix * x.C)), dim3(TB), 0, s, npix, x, dy, dx, acc, coef, gamma, dgamma,
                       dbeta, (float)(1.0 / n));
}

// ------------------------------------------------------------------------------------------------ head
__global__ void k_head_fwd(size_t npix, View feat, const float* __restrict__ w, const float* __restrict__ bias,
                           float* __restrict__ logits) {
    size_t p = (size_t)blockIdx.x * TB + threadIdx.x;
    if (p >= npix) return;
    const float* f = feat.p + p * feat.ps;
    float acc = bias[0];
    for (int c = 0; c < feat.C; ++c) acc = fmaf(f[c], w[c], acc);
    logits[p] = acc;
}

void g_head_fwd(hipStream_t s, int B, View feat, const float* w, const float* bias, float* logits) {
    size_t npix = (size_t)B * feat.H * feat.W;
    hipLaunchKernelGGL(k_head_fwd, dim3(nblk(npix)), dim3(TB), 0, s, npix, feat, w, bias, logits);
}

__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// dfeat = dlogit * w ; dW[c] += sum feat[c]*dlogit ; db += sum dlogit
__global__ void k_head_bwd(size_t npix, View feat, const float* __restrict__ w, const float* __restrict__ dlogits,
                           View dfeat, float* __restrict__ dw, float* __restrict__ dbias) {
    const int C = feat.C;
    size_t T = (size_t)gridDim.x * TB;
    size_t g = (size_t)blockIdx.x * TB + threadIdx.x;
    float sb = 0.f;
    for (size_t p = g; p < npix; p += T) {
        float d = dlogits[p];
        sb += d;
        float* o = dfeat.p + p * dfeat.ps;
        for (int c = 0; c < C; ++c) o[c] = d * w[c];
    }
    sb = wave_sum(sb);
    if ((threadIdx.x & 63) == 0) atomicAdd(dbias, sb);
    for (int c = 0; c < C; ++c) {
        float sw = 0.f;
        for (size_t p = g; p < npix; p += T) sw = fmaf(feat.p[p * feat.ps + c], dlogits[p], sw);
        sw = wave_sum(sw);
        if ((threadIdx.x & 63) == 0) atomicAdd(dw + c, sw);
    }
}

void g_head_bwd(hipStream_t s, int B, View feat, const float* w, const float* dlogits, View dfeat, float* dw, float* dbias) {
    size_t npix = (size_t)B * feat.H * feat.W;
    unsigned blocks = nblk(npix, TB * 8);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_head_bwd, dim3(blocks), dim3(TB), 0, s, npix, feat, w, dlogits, dfeat, dw, dbias);
}

// ------------------------------------------------------------------------------------------------ loss
// scalars: [0] sum(label) [1] min(label) [2] max(label) [3] sum(bce*mask) [4] l2 penalty
__device__ __forceinline__ void atomic_min_d(double* a, double v) {
    unsigned long long* p = (unsigned long long*)a;
    unsigned long long old = *p, assumed;
    do {
        assumed = old;
        if (__longlong_as_double(assumed) <= v) break;
        old = atomicCAS(p, assumed, __double_as_longlong(v));
    } while (assumed != old);
}
__device__ __forceinline__ void atomic_max_d(double* a, double v) {
    unsigned long long* p = (unsigned long long*)a;
    unsigned long long old = *p, assumed;
    do {
        assumed = old;
        if (__longlong_as_double(assumed) >= v) break;
        old = atomicCAS(p, assumed, __double_as_longlong(v));
    } while (assumed != old);
}

__global__ void k_label_stats(size_t n, const float* __restrict__ y, double* __restrict__ scalars) {
    size_t T = (size_t)gridDim.x * TB;
    float s = 0.f, mn = INFINITY, mx = -INFINITY;
    for (size_t i = (size_t)blockIdx.x * TB + threadIdx.x; i < n; i += T) {
        float v = y[i];
        s += v;
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    double ds = wave_sum_d((double)s);
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_down(mn, o, 64));
        mx = fmaxf(mx, __shfl_down(mx, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(scalars + 0, ds);
        atomic_min_d(scalars + 1, (double)mn);
        atomic_max_d(scalars + 2, (double)mx);
    }
}

void g_label_stats(hipStream_t s, size_t n, const float* y, double* scalars) {
    unsigned blocks = nblk(n, TB * 16);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_label_stats, dim3(blocks), dim3(TB), 0, s, n, y, scalars);
}

__device__ __forceinline__ float loss_weight(const dnnca_loss_cfg cfg, double label_sum, double n_label) {
    float w;
    if (cfg.has_weight) {
        w = cfg.weight;
    } else {
        float pr = (float)(label_sum / n_label);           // utils/losses.py:100
        w = pr > 0.f ? 1.0f / pr : 1.0f;                   // utils/losses.py:27
    }
    return cfg.weight_mul * w + cfg.weight_add;            // utils/losses.py:29
}

__global__ void k_loss(size_t n, const float* __restrict__ logits, const float* __restrict__ y, const dnnca_loss_cfg cfg,
                       double n_label, double* __restrict__ scalars, float* __restrict__ dlogits, float* __restrict__ prob,
                       float grad_scale) {
    const float wgt = loss_weight(cfg, scalars[0], n_label);
    size_t T = (size_t)gridDim.x * TB;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * TB + threadIdx.x; i < n; i += T) {
        float x = logits[i], z = y[i];
        float mask = fmaf(z, wgt - 1.0f, 1.0f);            // utils/losses.py:31
        float e = expf(-fabsf(x));
        float bce = fmaxf(x, 0.f) - x * z + log1pf(e);     // BCE-with-logits
        acc = fmaf(bce, mask, acc);
        float sig = x >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
        if (dlogits) dlogits[i] = mask * (sig - z) * grad_scale;
        if (prob) prob[i] = sig;
    }
    // one atomic per block, a few hundred blocks at most: same-address atomics execute one after the other (~56 ns each)
    __shared__ double red[TB / 64];
    const double d = wave_sum_d((double)acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < TB / 64; ++w) t += red[w];
        atomicAdd(scalars + 3, t);
    }
}

void g_loss(hipStream_t s, size_t n, const float* logits, const float* y, const dnnca_loss_cfg cfg, double n_label,
            double* scalars, float* dlogits, float* prob, float grad_scale) {
    unsigned blocks = nblk(n, TB * 8);
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(k_loss, dim3(blocks), dim3(TB), 0, s, n, logits, y, cfg, n_label, scalars, dlogits, prob, grad_scale);
}

__global__ void k_sigmoid(size_t n, const float* __restrict__ logits, float* __restrict__ prob) {
    size_t i = (size_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    float x = logits[i];
    float e = expf(-fabsf(x));
    prob[i] = x >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
}

void g_sigmoid(hipStream_t s, size_t n, const float* logits, float* prob) {
    hipLaunchKernelGGL(k_sigmoid, dim3(nblk(n)), dim3(TB), 0, s, n, logits, prob);
}

__global__ void k_l2(size_t n, const float* __restrict__ w, float* __restrict__ g, float l2, double* __restrict__ scalars) {
    size_t T = (size_t)gridDim.x * TB;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * TB + threadIdx.x; i < n; i += T) {
        float v = w[i];
        acc = fmaf(v, v, acc);
        g[i] = fmaf(2.0f * l2, v, g[i]);
    }
    // one atomic per block, few blocks: same-address atomics execute one after the other (~56 ns each)
    __shared__ double red[TB / 64];
    const double d = wave_sum_d((double)acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < TB / 64; ++w) t += red[w];
        atomicAdd(scalars + 4, t * (double)l2);
    }
}

void g_l2(hipStream_t s, size_t n, const float* w, float* g, float l2, double* scalars) {
    unsigned blocks = nblk(n, TB * 4);
    if (blocks > 128) blocks = 128;
    hipLaunchKernelGGL(k_l2, dim3(blocks), dim3(TB), 0, s, n, w, g, l2, scalars);
}

__global__ void k_finalize_scalars(double* __restrict__ scalars, const dnnca_loss_cfg cfg, double n_label,
                                   double inv_batch_hw, float* __restrict__ out5) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    out5[0] = (float)(scalars[3] * inv_batch_hw + scalars[4]);
    out5[1] = (float)(scalars[0] / n_label);
    out5[2] = loss_weight(cfg, scalars[0], n_label);
    out5[3] = (float)scalars[1];
    out5[4] = (float)scalars[2];
}

void g_finalize_scalars(hipStream_t s, double* scalars, const dnnca_loss_cfg cfg, double n_label, double inv_batch_hw,
                        float* out5) {
    hipLaunchKernelGGL(k_finalize_scalars, dim3(1), dim3(64), 0, s, scalars, cfg, n_label, inv_batch_hw, out5);
}

// ------------------------------------------------------------------------------------------------ Adam (Keras)
// fin != nullptr: block 0 / thread 0 also turns the step's scalar block into the five step outputs (k_finalize_scalars' job;
// single-replica steps only -- under data parallel the loss slot must be final before the all-reduce)
struct AdamFinalize {
    double* scalars;
    dnnca_loss_cfg cfg;
    double n_label, inv_batch_hw;
    float* out5;
};

__global__ void k_adam(size_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, float lr_t, float b1, float b2, float eps, float gscale, AdamFinalize fin) {
    size_t i = (size_t)blockIdx.x * TB + threadIdx.x;
    if (i == 0 && fin.out5) {
        fin.out5[0] = (float)(fin.scalars[3] * fin.inv_batch_hw + fin.scalars[4]);
        fin.out5[1] = (float)(fin.scalars[0] / fin.n_label);
        fin.out5[2] = loss_weight(fin.cfg, fin.scalars[0], fin.n_label);
        fin.out5[3] = (float)fin.scalars[1];
        fin.out5[4] = (float)fin.scalars[2];
    }
    if (i >= n) return;
    float gi = g[i] * gscale;
    float mi = m[i] * b1 + gi * (1.f - b1);
    float vi = v[i] * b2 + (gi * gi) * (1.f - b2);
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
}

void g_adam(hipStream_t s, size_t n, float* p, const float* g, float* m, float* v, float lr_t, float b1, float b2, float eps,
            float gscale) {
    hipLaunchKernelGGL(k_adam, dim3(nblk(n)), dim3(TB), 0, s, n, p, g, m, v, lr_t, b1, b2, eps, gscale, AdamFinalize{});
}

void g_adam_finalize(hipStream_t s, size_t n, float* p, const float* g, float* m, float* v, float lr_t, float b1, float b2, float eps,
                     float gscale, double* scalars, const dnnca_loss_cfg cfg, double n_label, double inv_batch_hw, float* out5) {
    hipLaunchKernelGGL(k_adam, dim3(nblk(n)), dim3(TB), 0, s, n, p, g, m, v, lr_t, b1, b2, eps, gscale,
                       AdamFinalize{scalars, cfg, n_label, inv_batch_hw, out5});
}

// ------------------------------------------------------------------------------------------------ pixel confusion
// One pass over (prob, y) for ANY number of thresholds: with the thresholds sorted ascending, bin(p) = #{t : thr[t] < p}
// and "p > thr[t]" (the Keras Precision/Recall/AUC comparison, metrics.yaml:2-23) holds exactly for t < bin(p), so the
// per-threshold counts are suffix sums of a (positive, negative) histogram over bins 0..nthr -- finished on the host in
// exact 64-bit integers.  Integer work: bit-exact against numpy by construction.  HBM-bound: 8 B per pixel, read once.
// NaN probabilities compare false against everything and land in bin 0, as `nan > t` does.
__global__ __launch_bounds__(256) void k_confusion_hist(size_t n, const float* __restrict__ prob, const float* __restrict__ y,
                                                        const float* __restrict__ thr, int nthr,
                                                        unsigned long long* __restrict__ hist /* [2][nthr + 1] */) {
    __shared__ float sthr[DNNCA_CONF_MAX_THR];
    __shared__ unsigned h[2][DNNCA_CONF_MAX_THR + 1];
    for (int i = threadIdx.x; i < nthr; i += 256) sthr[i] = thr[i];
    for (int i = threadIdx.x; i < 2 * (DNNCA_CONF_MAX_THR + 1); i += 256) (&h[0][0])[i] = 0u;
    __syncthreads();
    // most pixels of a segmentation map sit below every threshold or above all of them: those two bins stay in registers
    unsigned lo[2] = {0u, 0u}, hi[2] = {0u, 0u};
    size_t T = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += T) {
        float p = prob[i];
        int yy = y[i] > 0.5f ? 0 : 1;          // labels are cast to bool [TF-2.6 metrics_utils]; row 0 = positives
        int a = 0, b = nthr;                   // first index with !(thr[idx] < p)
        while (a < b) {
            int mid = (a + b) >> 1;
            if (sthr[mid] < p) a = mid + 1; else b = mid;
        }
        if (a == 0) ++lo[yy];
        else if (a == nthr) ++hi[yy];
        else atomicAdd(&h[yy][a], 1u);
    }
    for (int r = 0; r < 2; ++r) {
        if (lo[r]) atomicAdd(&h[r][0], lo[r]);
        if (hi[r]) atomicAdd(&h[r][nthr], hi[r]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * (nthr + 1); i += 256) {
        int r = i / (nthr + 1), bin = i - r * (nthr + 1);
        unsigned v = h[r][bin];
        if (v) atomicAdd(hist + i, (unsigned long long)v);
    }
}

void g_confusion_hist(hipStream_t s, size_t n, const float* prob, const float* y, const float* thr_sorted, int nthr,
                      unsigned long long* hist) {
    unsigned blocks = nblk(n, 256 * 16);
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(k_confusion_hist, dim3(blocks), dim3(256), 0, s, n, prob, y, thr_sorted, nthr, hist);
}

// one launch at the top of a step: scalar block, flat gradient vector and the weight-gradient slabs
__global__ __launch_bounds__(256) void k_step_init(double* __restrict__ scalars, float4* __restrict__ a, size_t na4,
                                                    float4* __restrict__ b, size_t nb4) {
    const size_t T = (size_t)gridDim.x * TB;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blockIdx.x * TB + threadIdx.x; i < na4; i += T) a[i] = z;
    for (size_t i = (size_t)blockIdx.x * TB + threadIdx.x; i < nb4; i += T) b[i] = z;
    if (blockIdx.x == 0 && threadIdx.x < 8) {
        // label sum 0, min +inf, max -inf, loss 0, l2 0, spare
        scalars[threadIdx.x] = threadIdx.x == 1 ? (double)INFINITY : (threadIdx.x == 2 ? -(double)INFINITY : 0.0);
    }
}

void g_step_init(hipStream_t s, double* scalars, float* a, size_t na, float* b, size_t nb) {
    size_t na4 = (na + 3) / 4, nb4 = (nb + 3) / 4;   // both buffers are allocated with >= 16 bytes of slack
    size_t n = na4 + nb4;
    unsigned blocks = (unsigned)((n + TB * 4 - 1) / (TB * 4));
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_step_init, dim3(blocks), dim3(TB), 0, s, scalars, reinterpret_cast<float4*>(a), na4,
                       reinterpret_cast<float4*>(b), nb4);
}

__global__ void k_scale(size_t n, float* __restrict__ p, float a) {
    size_t i = (size_t)blockIdx.x * TB + threadIdx.x;
    if (i < n) p[i] *= a;
}

void g_scale(hipStream_t s, size_t n, float* p, float a) {
    hipLaunchKernelGGL(k_scale, dim3(nblk(n)), dim3(TB), 0, s, n, p, a);
}

}  // namespace dnnca
