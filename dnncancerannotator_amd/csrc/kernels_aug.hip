// kernels_aug.hip -- train-time augmentation on the device: the reference's per-image tf.data maps
//   base():            centre crop to the stored size, cast to float32, / 255            (annotator/data.py:195-206)
//   random_crop():     crop to output_size at centre + clip(int(N(0, stddev)), min_, max_)  (data.py:677-689)
//   random_flip():     tf.image.random_flip_left_right                                     (data.py:620-625)
//   random_contrast(): (x - mean_hw(x)) * U[lower, upper) + mean_hw(x) on the feature channels (data.py:586-609)
//   to_feature_label() label channel -> y, the others in order -> x                        (data.py:766-788)
// in two launches over a uint8 batch that was uploaded as stored (a quarter of the float bytes over PCIe).  The random draws are
// made by the host (augment.py) and passed per image, so the arithmetic is checkable against the oracle draw for draw.
// random_warp (tfa.image.sparse_image_warp) follows below as dnnca_warp_f32.
#include <string.h>
#include "fast.h"
#include "kernels.h"

namespace dnnca {

struct AugArgs {
    const unsigned char* src;    // [B, Hs, Ws, Cs] uint8
    const dnnca_aug_param* prm;  // [B] (device copy)
    unsigned* sums;              // [B, Cs] integer sums of the crop window (zeroed before the launch)
    float* x;                    // [B, Ho, Wo, Cs - 1]
    float* y;                    // [B, Ho, Wo]
    int B, Hs, Ws, Cs, Ho, Wo, label_index;
    unsigned contrast_mask;      // bit c set: source channel c is contrast-adjusted
};

constexpr int AUG_MAXC = 8;

__device__ __forceinline__ void aug_window(const AugArgs& p, int b, int& top, int& left, int& flip, float& f) {
    const dnnca_aug_param q = p.prm[b];
    top = (p.Hs - p.Ho) / 2 + q.dy;
    left = (p.Ws - p.Wo) / 2 + q.dx;
    flip = q.flip;
    f = q.contrast;
}

// integer sums of every source channel over the crop window of image blockIdx.y (exact: uint8 sums fit 32 bits up to 2^24 pixels)
__global__ __launch_bounds__(256) void k_aug_sums(AugArgs p) {
    __shared__ unsigned red[4][AUG_MAXC];
    const int b = blockIdx.y;
    int top, left, flip;
    float f;
    aug_window(p, b, top, left, flip, f);
    unsigned s[AUG_MAXC];
#pragma unroll
    for (int c = 0; c < AUG_MAXC; ++c) s[c] = 0;
    const int n = p.Ho * p.Wo;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int oy = i / p.Wo, ox = i - oy * p.Wo;
        const unsigned char* px = p.src + (((size_t)b * p.Hs + top + oy) * p.Ws + left + ox) * p.Cs;
#pragma unroll
        for (int c = 0; c < AUG_MAXC; ++c)
            if (c < p.Cs) s[c] += px[c];
    }
#pragma unroll
    for (int c = 0; c < AUG_MAXC; ++c) {
        unsigned v = s[c];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][c] = v;
    }
    __syncthreads();
    if (threadIdx.x < p.Cs) atomicAdd(p.sums + b * p.Cs + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// one thread = one output pixel
__global__ __launch_bounds__(256) void k_aug_apply(AugArgs p) {
    const int n = p.Ho * p.Wo;
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (size_t)p.B * n) return;
    const int b = (int)(id / n), i = (int)(id - (size_t)b * n);
    const int oy = i / p.Wo, ox = i - oy * p.Wo;
    int top, left, flip;
    float f;
    aug_window(p, b, top, left, flip, f);
    const int sx = flip ? p.Wo - 1 - ox : ox;                       // flip of the cropped image
    const unsigned char* px = p.src + (((size_t)b * p.Hs + top + oy) * p.Ws + left + sx) * p.Cs;
    const float inv_n = 1.0f / (float)n;
    float* xo = p.x + id * (p.Cs - 1);
    int k = 0;
#pragma unroll
    for (int c = 0; c < AUG_MAXC; ++c) {
        if (c >= p.Cs) break;
        float v = (float)px[c] / 255.0f;
        if (c == p.label_index) {
            p.y[id] = v;
            continue;
        }
        if (((p.contrast_mask >> c) & 1u) && f != 1.0f) {      // factor 1 is the identity (and the host's "no contrast" value)
            const float mean = ((float)p.sums[b * p.Cs + c] * inv_n) / 255.0f;
            v = (v - mean) * f + mean;                                // tf.image.adjust_contrast (no clipping)
        }
        xo[k++] = v;
    }
}

}  // namespace dnnca

using namespace dnnca;

int dnnca_augment_u8(void* model, const void* src_dev, int batch, int hs, int ws, int cs, int label_index, unsigned contrast_mask,
                     const dnnca_aug_param* params_host, int ho, int wo, float* x_dev, float* y_dev) {
    Model* M = reinterpret_cast<Model*>(model);
    if (!M) { set_error("null model"); return DNNCA_EINVAL; }
    if (!src_dev || !params_host || !x_dev || !y_dev || batch < 1 || cs < 2 || cs > AUG_MAXC || label_index < 0 || label_index >= cs ||
        ho < 1 || wo < 1 || ho > hs || wo > ws) {
        set_error("dnnca_augment_u8: bad arguments (batch %d, %dx%dx%d -> %dx%d, label %d)", batch, hs, ws, cs, ho, wo, label_index);
        return DNNCA_EINVAL;
    }
    for (int b = 0; b < batch; ++b) {      // tf.image.crop_to_bounding_box asserts the window lies inside the image
        const int top = (hs - ho) / 2 + params_host[b].dy, left = (ws - wo) / 2 + params_host[b].dx;
        if (top < 0 || left < 0 || top + ho > hs || left + wo > ws) {
            set_error("dnnca_augment_u8: crop window of image %d leaves the %dx%d source (top %d, left %d, %dx%d)", b, hs, ws, top, left, ho, wo);
            return DNNCA_EINVAL;
        }
    }
    const size_t need = (size_t)batch * sizeof(dnnca_aug_param) + (size_t)batch * AUG_MAXC * 4;
    if (need > M->aug_scratch_bytes) {
        void* p = nullptr;
        DN_TRY(M->alloc(&p, need));
        M->aug_scratch = p;
        M->aug_scratch_bytes = need;
    }
    AugArgs a{};
    a.src = (const unsigned char*)src_dev;
    a.prm = (const dnnca_aug_param*)M->aug_scratch;
    a.sums = (unsigned*)((char*)M->aug_scratch + (size_t)batch * sizeof(dnnca_aug_param));
    a.x = x_dev; a.y = y_dev;
    a.B = batch; a.Hs = hs; a.Ws = ws; a.Cs = cs; a.Ho = ho; a.Wo = wo; a.label_index = label_index;
    a.contrast_mask = contrast_mask & ~(1u << label_index);
    // the draws travel through a pinned row of the model's ring: the caller's buffer is free when this call returns, and the call
    // does not wait for the stream (a row is only waited for when it comes round again, four calls later)
    const size_t prm_bytes = (size_t)batch * sizeof(dnnca_aug_param);
    if (prm_bytes > M->aug_pin_bytes) {
        HIP_TRY(hipStreamSynchronize(M->stream));         // uploads from the old rows are complete
        if (M->aug_pin) (void)hipHostFree(M->aug_pin);
        M->aug_pin = nullptr;
        M->aug_pin_bytes = 0;
        const size_t row = (size_t)M->desc.max_batch * sizeof(dnnca_aug_param) > prm_bytes ? (size_t)M->desc.max_batch * sizeof(dnnca_aug_param) : prm_bytes;
        HIP_TRY(hipHostMalloc(&M->aug_pin, row * Model::kAugRing, hipHostMallocDefault));
        M->aug_pin_bytes = row;
    }
    const int k = M->aug_k;
    M->aug_k = (k + 1) % Model::kAugRing;
    if (!M->aug_ev[k]) HIP_TRY(hipEventCreateWithFlags(&M->aug_ev[k], hipEventDisableTiming));
    else HIP_TRY(hipEventSynchronize(M->aug_ev[k]));
    char* pin = (char*)M->aug_pin + (size_t)k * M->aug_pin_bytes;
    memcpy(pin, params_host, prm_bytes);
    HIP_TRY(hipMemcpyAsync((void*)a.prm, pin, prm_bytes, hipMemcpyHostToDevice, M->stream));
    HIP_TRY(hipEventRecord(M->aug_ev[k], M->stream));
    HIP_TRY(hipMemsetAsync(a.sums, 0, (size_t)batch * cs * 4, M->stream));
    const int n = ho * wo;
    int bx = (n + 256 * 16 - 1) / (256 * 16);
    if (bx > 64) bx = 64;
    if (a.contrast_mask)
        LAUNCH(M, "aug_sums", (double)batch * n * cs, 0, hipLaunchKernelGGL(k_aug_sums, dim3(bx, batch), dim3(256), 0, M->stream, a));
    LAUNCH(M, "aug_apply", (double)batch * n * (cs + 4.0 * cs), 0,
           hipLaunchKernelGGL(k_aug_apply, dim3((unsigned)(((size_t)batch * n + 255) / 256)), dim3(256), 0, M->stream, a));
    return DNNCA_OK;
}

// ------------------------------------------------------------------------------------------------ random_warp (dense part)
// annotator/data.py:725-763 random_warp -> tfa.image.sparse_image_warp(image, source = raw, dest = raw + diff), order 2:
//   flow(q) = sum_i phi(|q - c_i|^2) w_i + [q, 1] v        phi(r) = 0.5 r log(max(r, 1e-10))   (tfa interpolate_spline, order 2)
//   out(q)  = bilinear(image, q - flow(q))                  (tfa dense_image_warp / interpolate_bilinear, "ij" indexing)
// The (n+3) x (n+3) spline system is solved on the host (augment.solve_warp); here every output pixel evaluates its flow from the
// n control points c_i (the *destination* points) and the weights (w, v), and samples features and label with the same flow.
namespace dnnca {

struct WarpArgs {
    const float* x;          // [B, H, W, C] source features
    const float* y;          // [B, H, W] source label
    const double* ctrl;      // [B, n, 2] control points (row, column)
    const double* wv;        // [B, n + 3, 2] spline weights w (n rows) then v (3 rows: row coefficient, column coefficient, constant)
    float* xo;
    float* yo;
    int B, H, W, C, n;
};

__global__ __launch_bounds__(256) void k_warp(WarpArgs p) {
    // The thin-plate weights cancel massively (|phi| ~ 1e4, flows ~ 1): the flow is evaluated in double (the float32 sum is only
    // good to ~1e-2 pixel, which is also all that tfa's own float32 evaluation is good for).
    extern __shared__ double sm[];         // control points + weights of this image: n * 4 + 6 doubles
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < p.n * 2; i += 256) {
        sm[i] = p.ctrl[(size_t)b * p.n * 2 + i];
        sm[p.n * 2 + i] = p.wv[(size_t)b * (p.n + 3) * 2 + i];
    }
    if (threadIdx.x < 6) sm[p.n * 4 + threadIdx.x] = p.wv[((size_t)b * (p.n + 3) + p.n) * 2 + threadIdx.x];
    __syncthreads();
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= p.H * p.W) return;
    const int qy = id / p.W, qx = id - qy * p.W;
    const float fqy = (float)qy, fqx = (float)qx;
    const double* v = sm + p.n * 4;
    double d0 = qy * v[0] + qx * v[2] + v[4];             // linear term [q, 1] v  (v rows: y, x, 1; columns: flow_y, flow_x)
    double d1 = qy * v[1] + qx * v[3] + v[5];
    for (int i = 0; i < p.n; ++i) {
        const double dy = qy - sm[2 * i], dx = qx - sm[2 * i + 1];
        const double r = dy * dy + dx * dx;
        const double ph = 0.5 * r * log(fmax(r, 1e-10));
        d0 = fma(ph, sm[p.n * 2 + 2 * i], d0);
        d1 = fma(ph, sm[p.n * 2 + 2 * i + 1], d1);
    }
    const float f0 = (float)d0, f1 = (float)d1;
    // interpolate_bilinear at (qy - f0, qx - f1): floor clamped to [0, size - 2], alpha clipped to [0, 1]
    const float sy = fqy - f0, sx = fqx - f1;
    const float fy = fminf(fmaxf(floorf(sy), 0.f), (float)(p.H - 2)), fx = fminf(fmaxf(floorf(sx), 0.f), (float)(p.W - 2));
    const float ay = fminf(fmaxf(sy - fy, 0.f), 1.f), ax = fminf(fmaxf(sx - fx, 0.f), 1.f);
    const int iy = (int)fy, ix = (int)fx;
    const size_t o00 = ((size_t)b * p.H + iy) * p.W + ix, o01 = o00 + 1, o10 = o00 + p.W, o11 = o10 + 1;
    const size_t oq = ((size_t)b * p.H + qy) * p.W + qx;
    auto lerp2 = [&](float tl, float tr, float bl, float br) {
        const float top = ax * (tr - tl) + tl, bot = ax * (br - bl) + bl;
        return ay * (bot - top) + top;
    };
    for (int c = 0; c < p.C; ++c)
        p.xo[oq * p.C + c] = lerp2(p.x[o00 * p.C + c], p.x[o01 * p.C + c], p.x[o10 * p.C + c], p.x[o11 * p.C + c]);
    p.yo[oq] = lerp2(p.y[o00], p.y[o01], p.y[o10], p.y[o11]);
}

}  // namespace dnnca

int dnnca_warp_f32(void* model, const float* x_dev, const float* y_dev, int batch, int h, int w, int c, int n_points,
                   const double* ctrl_host, const double* wv_host, float* x_out_dev, float* y_out_dev) {
    Model* M = reinterpret_cast<Model*>(model);
    if (!M) { set_error("null model"); return DNNCA_EINVAL; }
    if (!x_dev || !y_dev || !ctrl_host || !wv_host || !x_out_dev || !y_out_dev || batch < 1 || h < 2 || w < 2 || c < 1 || n_points < 1 ||
        n_points > 2048 || x_out_dev == x_dev || y_out_dev == y_dev) {
        set_error("dnnca_warp_f32: bad arguments");
        return DNNCA_EINVAL;
    }
    const size_t nc = (size_t)batch * n_points * 2 * 8, nw = (size_t)batch * (n_points + 3) * 2 * 8;
    if (nc + nw > M->warp_scratch_bytes) {
        void* p = nullptr;
        DN_TRY(M->alloc(&p, nc + nw));
        M->warp_scratch = p;
        M->warp_scratch_bytes = nc + nw;
    }
    WarpArgs a{};
    a.x = x_dev; a.y = y_dev; a.xo = x_out_dev; a.yo = y_out_dev;
    a.ctrl = (const double*)M->warp_scratch;
    a.wv = (const double*)((char*)M->warp_scratch + nc);
    a.B = batch; a.H = h; a.W = w; a.C = c; a.n = n_points;
    HIP_TRY(hipMemcpyAsync((void*)a.ctrl, ctrl_host, nc, hipMemcpyHostToDevice, M->stream));
    HIP_TRY(hipMemcpyAsync((void*)a.wv, wv_host, nw, hipMemcpyHostToDevice, M->stream));
    LAUNCH(M, "aug_warp", (double)batch * h * w * (c + 1) * 8.0, (double)batch * h * w * n_points * 8.0,
           hipLaunchKernelGGL(k_warp, dim3((h * w + 255) / 256, batch), dim3(256), (size_t)(n_points * 4 + 6) * 8, M->stream, a));
    HIP_TRY(hipStreamSynchronize(M->stream));
    return DNNCA_OK;
}
