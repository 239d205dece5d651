// kernels_aug.hip -- train-time augmentation on the device: the reference's per-image tf.data maps
//   base():            centre crop to the stored size, cast to float32, / 255            (annotator/data.py:195-206)
//   random_crop():     crop to output_size at centre + clip(int(N(0, stddev)), min_, max_)  (data.py:677-689)
//   random_flip():     tf.image.random_flip_left_right                                     (data.py:620-625)
//   random_contrast(): (x - mean_hw(x)) * U[lower, upper) + mean_hw(x) on the feature channels (data.py:586-609)
//   to_feature_label() label channel -> y, the others in order -> x                        (data.py:766-788)
// in two launches over a uint8 batch that was uploaded as stored (a quarter of the float bytes over PCIe).  The random draws are
// made by the host (augment.py) and passed per image, so the arithmetic is checkable against the oracle draw for draw.
// random_warp (tfa.image.sparse_image_warp) is not part of this path.
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
    HIP_TRY(hipMemcpyAsync((void*)a.prm, params_host, (size_t)batch * sizeof(dnnca_aug_param), hipMemcpyHostToDevice, M->stream));
    HIP_TRY(hipMemsetAsync(a.sums, 0, (size_t)batch * cs * 4, M->stream));
    const int n = ho * wo;
    int bx = (n + 256 * 16 - 1) / (256 * 16);
    if (bx > 64) bx = 64;
    if (a.contrast_mask)
        LAUNCH(M, "aug_sums", (double)batch * n * cs, 0, hipLaunchKernelGGL(k_aug_sums, dim3(bx, batch), dim3(256), 0, M->stream, a));
    LAUNCH(M, "aug_apply", (double)batch * n * (cs + 4.0 * cs), 0,
           hipLaunchKernelGGL(k_aug_apply, dim3((unsigned)(((size_t)batch * n + 255) / 256)), dim3(256), 0, M->stream, a));
    HIP_TRY(hipStreamSynchronize(M->stream));        // params_host may be released by the caller
    return DNNCA_OK;
}
