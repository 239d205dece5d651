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
struct View {
    float* p = nullptr;
    int H = 0, W = 0, C = 0;
    int ps = 0;
};

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
