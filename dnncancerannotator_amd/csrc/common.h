// common.h -- shared declarations of libdnnca (MI355X / gfx950 only; no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/dnnca.h"

namespace dnnca {

void set_error(const char* fmt, ...);

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            ::dnnca::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            return DNNCA_EHIP;                                                                  \
        }                                                                                       \
    } while (0)

#define DN_TRY(expr)            \
    do {                        \
        int r_ = (expr);        \
        if (r_ != DNNCA_OK) return r_; \
    } while (0)

// A dense-in-(H,W) NHWC view: element (b,y,x,c) lives at p[((b*H + y)*W + x)*ps + c]; ps >= C lets a tensor be a
// channel slice of a wider buffer (mulmo: input channel slices, bottleneck concat -- unet.py:183,187).
// h = 1: the elements are bf16 (the first half of the same allocation, same geometry in elements).  Only tensors whose every
// reader rounds to bf16 anyway are stored that way (ig_plan_half, kernels_igemm.hip), so results do not depend on it.
struct View {
    float* p = nullptr;
    int H = 0, W = 0, C = 0;
    int ps = 0;
    int h = 0;
};

typedef __bf16 hbf16;
typedef __bf16 hbf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ hbf16x4 to_bf16x4(const float4& v) {
    hbf16x4 r;
    r[0] = (hbf16)v.x; r[1] = (hbf16)v.y; r[2] = (hbf16)v.z; r[3] = (hbf16)v.w;
    return r;
}

inline View slice(const View& v, int c0, int c) {
    View o = v;
    o.p = v.p + c0;
    o.C = c;
    return o;
}

// per-launch accounting used by the profiler and by dnnca_plan_dump
struct LaunchInfo {
    const char* name;
    double bytes;   // algorithmic bytes: every input tensor read once, every output written once
    double flops;
};

}  // namespace dnnca
