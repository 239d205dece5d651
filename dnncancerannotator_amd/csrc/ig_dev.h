// ig_dev.h -- what the implicit-GEMM conv kernels of kernels_igemm.hip (fp32 / bf16 operands) and kernels_ig3x.hip (fp32 by three
// bf16 planes) share on the device side: the argument block, the register epilogue of the persistent kernels, small helpers.
#pragma once
#include <hip/hip_runtime.h>

#include "bn_dev.h"
#include "common.h"

namespace dnnca {

// Division by a kernel-uniform divisor the persistent kernels repeat per tile: hipcc expands `/` into a ~25-instruction
// dependent chain, which a kernel running one wave per SIMD cannot hide.  q = umulhi(n, ceil(2^32 / d)) is exact while
// n * d < 2^32 (tile / unit / item counts are far below that).
struct FastDiv {
    unsigned d, m;
    __device__ __forceinline__ explicit FastDiv(int dd) : d((unsigned)dd), m(dd > 1 ? (unsigned)(0xffffffffull / (unsigned)dd) + 1u : 0u) {}
    __device__ __forceinline__ int div(int n) const { return d > 1 ? (int)__umulhi((unsigned)n, m) : n; }
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace ig {

constexpr int TY = 8, TX = 16;          // block tile: 8 rows x 16 pixels = 128 GEMM rows, 32 per wave
constexpr int CK = 16;                  // input channels per staged chunk
constexpr int CKP = CK + 4;             // LDS pixel stride of the input patch (floats)
constexpr int PATCH = (TY + 2) * (TX + 2);

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct ConvArgs {
    const float* src[2];     // dense NHWC sources; chunk cc comes from src[cc >= c_src0]
    int c_src0, c_src1;      // channels of the two sources (c_src1 = 0: one source)
    const float* w;          // [9][Cin][Cout] (forward: the Keras kernel; data gradient: flipped/transposed copy)
    const float* bias;       // forward: Cout floats (nullptr: none)
    float* dst[2];           // output tensors; N-tiles below n_dst0 channels go to dst[0], the rest to dst[1]
    int n_dst0, n_dst1;      // channels of the two destinations
    const float* mask[2];    // data gradient: multiply by act'(mask tensor) (nullptr: no mask)
    int acc[2];              // data gradient: accumulate into dst
    int B, H, W;
    int tiles_x, tiles_y;
    float alpha;             // forward: activation slope (<0 none); data gradient: slope of the masked activation
    // forward: batch statistics of the BatchNorm behind this conv ride in the epilogue of the persistent kernels, which fold them
    // themselves (bn_dev.h; bnf.tab == nullptr: none)
    BnSelfFold bnf;
    // forward, k_ig_conv3: source k is the INPUT of a BatchNorm whose apply pass was elided; its scale / shift (norm[k][c],
    // norm[k][c_srck + c]: the BatchNorm's coefficient table) are applied while the patch goes to LDS -- pixels outside the image
    // stay zero, as the padding of the normalised tensor would be (nullptr: the source is used as it is)
    const float* norm[2];
    int src_half;            // the sources are stored as bf16 (View::h; k_igb_conv3 only)
    int dst_half;            // forward, persistent kernels: dst[0] is stored as bf16 (the input of a BatchNorm, ig_plan_half)
    int dsth[2];             // data gradient, persistent kernels: dst[k] is stored as bf16 (the gradient arriving at a BatchNorm)
    // data gradient, k_ig3x_conv3: dst[0] (the only destination, neither accumulated nor masked) is ALL of the gradient arriving at a
    // BatchNorm: that BatchNorm's backward sums (sum dy xhat, sum dy) ride in the epilogue, which reads the BatchNorm's input at the
    // pixels it stores, and fold themselves (bn_dev.h; bnb.tab == nullptr: none) -- the BatchNorm's reduction pass is not launched
    BnBwdFold bnb;
};

// Epilogue of the persistent kernels (k_ig_conv3 / igb::k_igb_conv3), straight from the accumulator registers: lane (m16, q)
// of wave w holds acc[r][j][i] = pixel (row 4w + r, column 4q + i) x channel 16j + m16 of the 16 x 16 tile, so one store
// instruction writes four 64-byte channel runs.  (The earlier version transposed the tile through LDS to store 256-byte rows:
// 64 LDS writes + 16 LDS reads per lane and up to five barriers cost as much as 1.7 K-chunks of MFMAs per tile.)
//   MODE 0: + bias, activation; the BatchNorm behind the conv takes its batch statistics from here (ConvArgs::bnf): per-lane
//           sums over the lane's 16 pixels, the four q groups folded by two wave shuffles, the NW waves through `red`
//           ([NW][2 COT] floats of LDS) -- one barrier;
//   MODE 1: accumulate into dst and multiply by act'(mask tensor) as requested.
// (The backward sums of a BatchNorm were tried in the data-gradient epilogue too, twice: the extra read of the BatchNorm's
// input there costs about what the reduction pass it replaces does -- with the LDS-staged epilogue and one wave per SIMD
// 2.1 -> 3.2 ms of dgrad against 0.7 ms saved; with this epilogue and two waves per SIMD +0.44 ms of dgrad against 0.63 ms on
// unet_big, and a net loss on mulmo_unet, where the fold of the per-tile partials also grows.  Round 3, with the self-folding
// bucket rows of bn_dev.h and the sums taken from the stored bf16 values: 98 -> 192 us per launch -- the 16 two-byte loads per tile
// row cannot move above the previous row's stores, four exposed round trips per unit.  NOTES.md, section 6 of the round-3 document.)
template <int NN, int MODE, int NW = 4>
__device__ __forceinline__ void conv3_epilogue(const ConvArgs& p, const f32x4 (&acc)[4][NN], int b, int y0, int x0, int co0, int tile,
                                               float* red) {
    constexpr int COT = 16 * NN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m16 = lane & 15, q = lane >> 4;
    const int which = co0 >= p.n_dst0;
    const int cw = which ? p.n_dst1 : p.n_dst0, cl = which ? co0 - p.n_dst0 : co0;
    float* dst = p.dst[which];
    const bool bn_on = MODE == 0 && p.bnf.tab != nullptr;
    float bias[NN], bs[NN], bq[NN];
#pragma unroll
    for (int j = 0; j < NN; ++j) {
        bias[j] = (MODE == 0 && p.bias) ? p.bias[co0 + 16 * j + m16] : 0.f;
        bs[j] = 0.f;
        bq[j] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int y = y0 + 4 * wave + r;
        if (y >= p.H) continue;                 // wave-uniform
        float v[4][NN];
        bool ok[4];
        // element offsets in 32 bits (conv3_path checks that every destination has fewer than 2^32 elements): with size_t the address
        // arithmetic of the 16 NN stores was most of the epilogue -- 5.7 k of the 17.8 k ticks a 16-channel unit takes (tools/cv_stamps.py)
        unsigned o[4];
        const unsigned orow = (unsigned)(b * p.H + y) * (unsigned)p.W;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int x = x0 + 4 * q + i;
            ok[i] = x < p.W;
            o[i] = (orow + (unsigned)(ok[i] ? x : 0)) * (unsigned)cw + (unsigned)(cl + m16);
#pragma unroll
            for (int j = 0; j < NN; ++j) v[i][j] = acc[r][j][i];
        }
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NN; ++j) {
                    float t = v[i][j] + bias[j];
                    if (p.alpha >= 0.f) t = t > 0.f ? t : p.alpha * t;
                    if (p.dst_half) t = (float)(hbf16)t;          // the batch statistics are those of the stored values
                    v[i][j] = t;
                    if (bn_on && ok[i]) { bs[j] += t; bq[j] = fmaf(t, t, bq[j]); }
                }
        } else {
            if (p.dsth[which]) {          // bf16 destination (never masked: the BatchNorm backward applies act')
                hbf16* dh = reinterpret_cast<hbf16*>(dst);
                if (p.acc[which]) {
                    float t[4][NN];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < NN; ++j) t[i][j] = ok[i] ? (float)dh[o[i] + 16 * j] : 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < NN; ++j) v[i][j] += t[i][j];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NN; ++j)
                        if (ok[i]) dh[o[i] + 16 * j] = (hbf16)v[i][j];
                continue;
            }
            if (p.acc[which]) {
                float t[4][NN];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NN; ++j) t[i][j] = ok[i] ? dst[o[i] + 16 * j] : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NN; ++j) v[i][j] += t[i][j];
            }
            if (p.mask[which]) {
                float t[4][NN];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NN; ++j) t[i][j] = ok[i] ? p.mask[which][o[i] + 16 * j] : 1.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NN; ++j) v[i][j] *= t[i][j] > 0.f ? 1.0f : p.alpha;
            }
        }
        if (MODE == 0 && p.dst_half) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NN; ++j)
                    if (ok[i]) reinterpret_cast<hbf16*>(dst)[o[i] + 16 * j] = (hbf16)v[i][j];
            continue;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NN; ++j)
                if (ok[i]) dst[o[i] + 16 * j] = v[i][j];
    }
    if (bn_on) {        // this unit's sums go to bucket row tile % R: [2 cw], first half sums, second half sums of squares
#pragma unroll
        for (int j = 0; j < NN; ++j) {
            bs[j] += __shfl_xor(bs[j], 16); bs[j] += __shfl_xor(bs[j], 32);
            bq[j] += __shfl_xor(bq[j], 16); bq[j] += __shfl_xor(bq[j], 32);
            if (q == 0) {
                red[wave * (2 * COT) + 16 * j + m16] = bs[j];
                red[wave * (2 * COT) + COT + 16 * j + m16] = bq[j];
            }
        }
        lds_barrier();
        if (tid < 2 * COT) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) a += red[w * (2 * COT) + tid];
            const int half = tid >= COT, c = half ? tid - COT : tid;
            atomicAdd(bn_bucket(p.bnf, tile) + half * cw + cl + c, (double)a);
        }
    }
}

// sum over the 16 lanes of a DPP row (lanes that share q): quad swaps, then the two mirrors -- four v_add_f32 with DPP operands
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));     // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));     // row_mirror
    return v;
}

// ------------------------------------------------------------------------------------------------ weight gradient
struct WgArgs {
    const float* x;          // one source, dense NHWC, cs channels
    const float* dz;         // dense NHWC, cout channels
    float* dw;               // [9][cin_total][cout] gradient (atomic accumulation; zeroed by the caller)
    float* dbias;            // cout floats or nullptr
    int cs, ci_off, cin_total, cout;
    int B, H, W;
    int tiles_x, tiles_y, psplit;
    // same-address atomics execute one after the other (~56 ns each): when many blocks share a small gradient, block b adds
    // into copy b % nbuckets of it (dw / dbias then point at copy 0, copies bucket_stride floats apart; k_wg_fold sums them)
    int nbuckets, bucket_stride;
    // plain = 1 (k_ig3x_wgrad, igb::k_igb_wgrad64w): no atomics at all -- block (x, y, z) STORES its sums into slab x of a per-launch
    // set of psplit slabs (dw / dbias point at slab 0, slabs bucket_stride floats apart, laid out [9][cs][cout] + bias: cin_total = cs,
    // ci_off = 0) and k_wg_fold_plain sums the slabs into the gradient.  A block's sums are 9 x (channel block) floats whatever the
    // layer, so a 64 x 64-channel launch of 256 blocks drains 37.7 MB: 29 us as float atomics (1.3 TB/s at the memory side), 6 us as
    // plain stores + a 10 us fold pass.
    int plain;
    const float* norm;       // k_ig_wgrad2: x is the input of a BatchNorm whose apply pass was elided (scale norm[c], shift norm[cs + c]); nullptr: none
};

}  // namespace ig

}  // namespace dnnca
