// debug_tools.hip -- development aids (NOT part of include/dnnca.h and not used by the product path):
// micro-benchmarks that calibrate what the conv kernels can expect from the f32 matrix pipe, and a read-back of the BatchNorm
// reduction table for the tests.
#include "model.h"
#include <vector>

using namespace dnnca;

// development aid: raw issue rate of v_mfma_f32_16x16x4_f32.  mode 0: operands in registers; mode 1: A operand from LDS
// (one ds_read_b32 per MFMA, stride-12 float pattern of the conv kernels); mode 2: A and B from LDS.
typedef float dbg_f32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int NCH>
__global__ __launch_bounds__(512) void k_mfma_rate(float* out, int iters) {
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    dbg_f32x4 acc[NCH];
    for (int c = 0; c < NCH; ++c) acc[c] = dbg_f32x4{0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + lane * 0.001f, b = 0.5f;
    const int base = wave * 1024 + (lane & 15) * 12 + (lane >> 4);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                float av = MODE >= 1 ? lds[base + c * 200 + 4 * k + (it & 1) * 64] : a;
                float bv = MODE == 2 ? lds[8192 + base + 4 * k + (it & 1) * 64] : b;
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[c], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}


#define MODEL(h)                                             \
    Model* M = reinterpret_cast<Model*>(h);                  \
    if (!M) { set_error("null model handle"); return DNNCA_EINVAL; }

extern "C" {

int dnnca_debug_mfma_rate(void* model, int mode, int nch, int blocks, int iters, float* tflops) {
    MODEL(model);
    float* buf = nullptr;
    HIP_TRY(hipMalloc(&buf, (size_t)blocks * 512 * 4));
    auto launch = [&]() {
        if (mode == 0 && nch == 1) hipLaunchKernelGGL((k_mfma_rate<0, 1>), dim3(blocks), dim3(512), 0, M->stream, buf, iters);
        if (mode == 0 && nch == 2) hipLaunchKernelGGL((k_mfma_rate<0, 2>), dim3(blocks), dim3(512), 0, M->stream, buf, iters);
        if (mode == 0 && nch == 4) hipLaunchKernelGGL((k_mfma_rate<0, 4>), dim3(blocks), dim3(512), 0, M->stream, buf, iters);
        if (mode == 1 && nch == 2) hipLaunchKernelGGL((k_mfma_rate<1, 2>), dim3(blocks), dim3(512), 0, M->stream, buf, iters);
        if (mode == 1 && nch == 4) hipLaunchKernelGGL((k_mfma_rate<1, 4>), dim3(blocks), dim3(512), 0, M->stream, buf, iters);
        if (mode == 2 && nch == 4) hipLaunchKernelGGL((k_mfma_rate<2, 4>), dim3(blocks), dim3(512), 0, M->stream, buf, iters);
    };
    launch();
    HIP_TRY(hipStreamSynchronize(M->stream));
    HIP_TRY(hipEventRecord(M->ev0, M->stream));
    launch();
    HIP_TRY(hipEventRecord(M->ev1, M->stream));
    HIP_TRY(hipEventSynchronize(M->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, M->ev0, M->ev1));
    double flops = (double)blocks * 8 * iters * 16 * nch * 2048.0;
    *tflops = (float)(flops / (ms * 1e-3) / 1e12);
    (void)hipFree(buf);
    return DNNCA_OK;
}

// test aid (tests/test_engine_gpu.py): what the self-folding BatchNorm reductions left in their table (csrc/bn_dev.h) -- the bucket
// rows and the ticket counters must read back as zero after every step.  Synchronises the stream.
int dnnca_debug_bn_table(void* model, double* abs_sum, unsigned* ticket_sum, int* allocated) {
    MODEL(model);
    *abs_sum = 0.0;
    *ticket_sum = 0u;
    *allocated = M->bn_tab != nullptr;
    if (!M->bn_tab) return DNNCA_OK;
    HIP_TRY(hipStreamSynchronize(M->stream));
    std::vector<double> h(Model::kBnTab + 32);
    HIP_TRY(hipMemcpy(h.data(), M->bn_tab, h.size() * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < Model::kBnTab; ++i) *abs_sum += h[i] < 0 ? -h[i] : h[i];
    const unsigned* t = reinterpret_cast<const unsigned*>(h.data() + Model::kBnTab);
    for (int i = 0; i < 64; ++i) *ticket_sum += t[i];
    return DNNCA_OK;
}

}  // extern "C"
