// kernels_ig3x.hip -- fp32 3x3 convolutions (forward / data gradient) of the dense fp32 configurations (configs/mulmo_unet.yaml:
// 16 .. 384 channels; components.py:46-52,122-127) on the BF16 matrix pipe, with fp32 results.
//
// gfx950 has no fast fp32 matrix path: v_mfma_f32_16x16x4_f32 runs at the vector rate (157 TFLOP/s), v_mfma_f32_16x16x32_bf16 at
// sixteen times that.  An fp32 value is the EXACT sum of three bf16 values, f = h0 + h1 + h2 (h0 = bf16(f), h1 = bf16(f - h0),
// h2 = bf16(f - h0 - h1): 8 + 8 + 8 significant bits with signed residuals; both subtractions are exact in fp32), and a product of
// two bf16 values is exact in fp32.  So
//         a b = a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0) + [a1 b2 + a2 b1 + a2 b2],   [...] <= 2^-24 |a b|,
// and the six leading terms, accumulated in fp32 by the matrix pipe, give the convolution to fp32 accuracy (the dropped terms are
// below one fp32 ulp of each product; the summation order differs from an fmaf chain like any other fp32 implementation's).
// Six bf16 products per fp32 product at 16x the rate: 2.67x the fp32 matrix peak -- and the K = 32 of one MFMA holds TWO of them:
// lanes q = 0, 1 (K 0..15) and q = 2, 3 (K 16..31) read different planes of the same 16 channels, so a 16-channel chunk costs three
// MFMAs per tap and 16 x 16 tile:
//         (a0 | a1) . (b0 | b0)  +  (a0 | a1) . (b1 | b1)  +  (a0 | a2) . (b2 | b0).
//
// Structure: the persistent scheme of ig::k_ig_conv3 (units = pixel tile x channel tile, K chunks of 16 input channels, raw buffer
// loads with out-of-range offsets for the zero padding, register epilogue ig::conv3_epilogue with the fused BatchNorm statistics,
// BatchNorm scale / shift applied while staging where the apply pass was elided).  The fp32 patch is split into three bf16 planes
// WHILE IT IS STAGED (once per element, 3 x 8-byte LDS stores per 16-byte load); the weights arrive pre-split (k_ig3x_prep, once per
// step: forward layout [plane][tap][co][ci], data-gradient layout [plane][8 - tap][ci][co]).  Three planes of both operands take 1.5x
// the LDS of the fp32 kernel, so there is ONE LDS buffer: the next item's global loads fly during this item's MFMAs and are committed
// between two barriers.
#include <map>

#include "fast.h"
#include "ig_dev.h"
#include "kernels.h"

namespace dnnca {

namespace ig3x {

using ig::ConvArgs;
using ig::lds_barrier;

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int KC = 16;                  // fp32 input channels per item = one K = 32 MFMA over two planes
constexpr int kMaxBias = 1024;          // output channels at most (forward: the bias vector lives in LDS)
constexpr int T = 16;                   // tile width (pixels); rows: 4 per wave
constexpr unsigned BUF_FLAGS = 0x00020000u, OOB = 0x80000000u;

// f = h0 + h1 + h2 (see the header)
__device__ __forceinline__ void split3(float f, bf16_t& h0, bf16_t& h1, bf16_t& h2) {
    h0 = (bf16_t)f;
    const float r1 = f - (float)h0;
    h1 = (bf16_t)r1;
    const float r2 = r1 - (float)h1;
    h2 = (bf16_t)r2;
}

// phase stamps (tuning builds only: DNNCA_TUNING=1 python -m dnncancerannotator_amd.build; tools/x3_stamps.py): block 0, thread 0
#ifdef DNNCA_TUNING
__device__ unsigned long long g_x3_stamps[64 * 8];
__device__ __forceinline__ unsigned long long x3_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define X3STAMP(item, ph) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (item) < 64) g_x3_stamps[(item) * 8 + (ph)] = x3_now(); } while (0)
#else
#define X3STAMP(item, ph) do { } while (0)
#endif

struct PrepDesc {
    int w_off, cin, cout;
};
// per step: the three bf16 planes of every 3x3 conv kernel [t][ci][co], in the forward layout wf[pl][t][co][ci] (K = ci contiguous)
// and the data-gradient layout wd[pl][8 - t][ci][co] (K = co contiguous); plane pl of a buffer starts at pl * pstride elements
__global__ void k_ig3x_prep(const PrepDesc* __restrict__ descs, const float* __restrict__ params, bf16_t* __restrict__ wf,
                            bf16_t* __restrict__ wd, unsigned pstride) {
    const PrepDesc d = descs[blockIdx.y];
    // 32 x 32 (ci, co) tiles go through LDS so that both layouts are written in 64-byte runs (see igb::k_igb_prep)
    __shared__ float tile[32][33];
    const int ntx = (d.cout + 31) / 32, nty = (d.cin + 31) / 32, ntiles = 9 * nty * ntx;
    const int tx32 = threadIdx.x & 31, ty8 = threadIdx.x >> 5;          // 256 threads: 8 rows of 32
    for (int id = blockIdx.x; id < ntiles; id += gridDim.x) {
        const int t = id / (nty * ntx), r = id - t * (nty * ntx), ty = r / ntx, tx = r - ty * ntx;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ci = ty * 32 + ty8 + 8 * k, co = tx * 32 + tx32;
            float v = 0.f;
            if (ci < d.cin && co < d.cout) {
                v = params[d.w_off + ((size_t)t * d.cin + ci) * d.cout + co];
                bf16_t h0, h1, h2;
                split3(v, h0, h1, h2);
                const size_t o = d.w_off + ((size_t)(8 - t) * d.cin + ci) * d.cout + co;
                wd[o] = h0; wd[pstride + o] = h1; wd[2 * (size_t)pstride + o] = h2;
            }
            tile[ty8 + 8 * k][tx32] = v;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int co = tx * 32 + ty8 + 8 * k, ci = ty * 32 + tx32;
            if (ci < d.cin && co < d.cout) {
                bf16_t h0, h1, h2;
                split3(tile[tx32][ty8 + 8 * k], h0, h1, h2);
                const size_t o = d.w_off + ((size_t)t * d.cout + co) * d.cin + ci;
                wf[o] = h0; wf[pstride + o] = h1; wf[2 * (size_t)pstride + o] = h2;
            }
        }
        __syncthreads();
    }
}

using ig::row16_sum;

// Epilogue straight from the accumulators, CHANNEL-major: the MFMAs run with the operands swapped (rows = channels, columns =
// pixels), so lane (m16, q) of wave w holds acc[r][j][i] = channel 16 j + 4 q + i of pixel (row 4 w + r, column m16): four
// consecutive channels of ONE pixel -- a 16-byte store, and the wave's store instruction covers 16 pixels x 64 bytes (one contiguous
// KB where the tensor has 16 channels).  ig::conv3_epilogue's pixel-major layout needs a 4-byte store per value; at 16 channels per
// tile its 16 stores per lane and unit took longer than the unit's MFMAs.  Same contract otherwise: MODE 0 bias + activation, the
// batch statistics of the BatchNorm behind the conv (per-lane sums, DPP row sums over the 16 pixels, `red`, bucket adds; bn_dev.h);
// MODE 1 accumulate / act' mask / two destinations.  fp32 tensors only (the bf16-stored variants belong to dtype bf16).
// WN > 1: the block's channel tile (16 NN WN channels from co0t) is split over WN groups of NWR row-waves: this wave (row group wave %
// NWR, channel group wave / NWR) holds NN 16-channel tiles from co0t + 16 NN (wave / NWR).
// MODE 2 (ConvArgs::bnb): the values of the BatchNorm's input at the pixels this lane will store, requested BEFORE the unit's last
// MFMA phase -- loaded inside the epilogue, every unit paid one exposed HBM round trip (+13 us on a 512 x 512 x 16-channel launch).
// Rows below the image and columns right of it read element 0 of their row's first pixel / of the tensor: never used.
template <int NN, int NWR, int WN>
__device__ __forceinline__ void bnb_prefetch(const ConvArgs& p, f32x4 (&z)[4][NN], int b, int y0, int x0, int co0t) {
    const int tid = threadIdx.x, lane = tid & 63, wave_all = tid >> 6, wave = wave_all % NWR, wn = wave_all / NWR, m16 = lane & 15, q = lane >> 4;
    const int co0 = co0t + 16 * NN * wn, cw = p.n_dst0;
    const int x = x0 + m16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int y = y0 + 4 * wave + r;
        const bool in = y < p.H && x < p.W;
        const unsigned o = in ? ((unsigned)(b * p.H + y) * (unsigned)p.W + (unsigned)x) * (unsigned)cw + (unsigned)(co0 + 4 * q) : (unsigned)(co0 + 4 * q);
#pragma unroll
        for (int j = 0; j < NN; ++j) z[r][j] = *reinterpret_cast<const f32x4*>(p.bnb.x + o + 16 * j);
    }
}

template <int NN, int MODE, int NWR, int WN = 1>
__device__ __forceinline__ void epilogue_t(const ConvArgs& p, const f32x4 (&acc)[4][NN], int b, int y0, int x0, int co0t, int tile, float* red,
                                           const float* bias_lds, const f32x4 (&zpre)[4][MODE == 2 ? NN : 1]) {
    constexpr int COT = 16 * NN;
    const int tid = threadIdx.x, lane = tid & 63, wave_all = tid >> 6, wave = wave_all % NWR, wn = wave_all / NWR, m16 = lane & 15, q = lane >> 4;
    const int co0 = co0t + COT * wn;
    const int which = co0 >= p.n_dst0;
    const int cw = which ? p.n_dst1 : p.n_dst0, cl = which ? co0 - p.n_dst0 : co0;
    float* dst = p.dst[which];
    const bool bn_on = MODE == 0 && p.bnf.tab != nullptr;
    constexpr bool bb_on = MODE == 2;          // (one destination, neither accumulated nor masked; mean / inv in bias_lds)
    f32x4 bias[NN], bs[NN], bq[NN];
#pragma unroll
    for (int j = 0; j < NN; ++j) {
        bias[j] = MODE == 0 ? *reinterpret_cast<const f32x4*>(bias_lds + co0 + 16 * j + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
        bs[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        bq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int x = x0 + m16;
    const bool okx = x < p.W;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int y = y0 + 4 * wave + r;
        if (y >= p.H) continue;                 // wave-uniform
        // element offset in 32 bits (the launcher checks that every destination has fewer than 2^32 elements)
        const unsigned o = ((unsigned)(b * p.H + y) * (unsigned)p.W + (unsigned)(okx ? x : 0)) * (unsigned)cw + (unsigned)(cl + 4 * q);
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < NN; ++j) {
                f32x4 t = acc[r][j] + bias[j];
                if (p.alpha >= 0.f) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) t[i] = t[i] > 0.f ? t[i] : p.alpha * t[i];
                }
                if (okx) {
                    if (bn_on) {
                        bs[j] += t;
#pragma unroll
                        for (int i = 0; i < 4; ++i) bq[j][i] = fmaf(t[i], t[i], bq[j][i]);
                    }
                    *reinterpret_cast<f32x4*>(dst + o + 16 * j) = t;
                }
            }
        } else if (bb_on) {          // the BatchNorm backward sums of the gradient this launch produces (ConvArgs::bnb)
#pragma unroll
            for (int j = 0; j < NN; ++j) {
                const f32x4 t = acc[r][j];
                const f32x4 (&z)[MODE == 2 ? NN : 1] = zpre[r];
                const f32x4 mean = *reinterpret_cast<const f32x4*>(bias_lds + co0 + 16 * j + 4 * q);
                const f32x4 inv = *reinterpret_cast<const f32x4*>(bias_lds + cw + co0 + 16 * j + 4 * q);
                if (okx) {
                    bs[j] += t;
#pragma unroll
                    for (int i = 0; i < 4; ++i) bq[j][i] = fmaf(t[i], (z[MODE == 2 ? j : 0][i] - mean[i]) * inv[i], bq[j][i]);      // as k_bn_bwd_reduce_fast
                    *reinterpret_cast<f32x4*>(dst + o + 16 * j) = t;
                }
            }
        } else {
            f32x4 t[NN], old[NN], mk[NN];
#pragma unroll
            for (int j = 0; j < NN; ++j) {
                t[j] = acc[r][j];
                if (p.acc[which]) old[j] = okx ? *reinterpret_cast<const f32x4*>(dst + o + 16 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.mask[which]) mk[j] = okx ? *reinterpret_cast<const f32x4*>(p.mask[which] + o + 16 * j) : f32x4{1.f, 1.f, 1.f, 1.f};
            }
#pragma unroll
            for (int j = 0; j < NN; ++j) {
                if (p.acc[which]) t[j] += old[j];
                if (p.mask[which]) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) t[j][i] *= mk[j][i] > 0.f ? 1.0f : p.alpha;
                }
                if (okx) *reinterpret_cast<f32x4*>(dst + o + 16 * j) = t[j];
            }
        }
    }
    if (bn_on || bb_on) {        // this unit's sums go to bucket row tile % R: [2 cw], forward: sums | sums of squares; backward: sums of dy xhat | sums of dy
#pragma unroll
        for (int j = 0; j < NN; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float s1 = row16_sum(MODE == 0 ? bs[j][i] : bq[j][i]), s2 = row16_sum(MODE == 0 ? bq[j][i] : bs[j][i]);
                if (m16 == 0) {
                    red[wave_all * (2 * COT) + 16 * j + 4 * q + i] = s1;
                    red[wave_all * (2 * COT) + COT + 16 * j + 4 * q + i] = s2;
                }
            }
        lds_barrier();
        if (tid < WN * 2 * COT) {          // thread t: channel group t / (2 COT), entry t % (2 COT) of that group's [sums | sums of squares]
            const int g = tid / (2 * COT), e = tid - g * (2 * COT);
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < NWR; ++w) a += red[(g * NWR + w) * (2 * COT) + e];
            const int half = e >= COT, c = half ? e - COT : e;
            // (all channel groups of a block go to the same destination: a channel tile never straddles the two destinations)
            double* row = MODE == 0 ? bn_bucket(p.bnf, tile) : p.bnb.tab + (size_t)(tile % p.bnb.R) * 2 * p.bnb.C;
            atomicAdd(row + half * cw + (cl - COT * wn) + COT * g + c, (double)a);
        }
    }
}

// MODE 0 forward, MODE 1 data gradient (the forward kernel on the flipped / transposed planes), MODE 2 data gradient whose epilogue also
// takes the backward sums of the BatchNorm in front of the conv (ConvArgs::bnb: one destination, neither accumulated nor masked).  w3: the conv's plane 0,
// [9][N channels][K channels] bf16 with K contiguous; planes 1, 2 at + pstride, + 2 pstride elements.
// NW waves: 4 -> 16 x 16-pixel tiles (one wave per SIMD), 8 -> 32 x 16 (two per SIMD).  Channel tile 16 NN.
// WN = 2 (NW = 8 only): 16 x 16-pixel tiles like NW = 4, the eight waves are 4 row groups x 2 channel halves (8 NN channels... 16 NN / 2 each):
// two waves per SIMD on layers that have too few 32 x 16 tiles to fill the chip (the 64^2 / 128^2 levels) -- a single wave per SIMD cannot
// issue v_mfma_f32_16x16x32_bf16 back to back, and its LDS waits and barriers are nobody's cover.
// DB: TWO LDS buffers (where they fit: 16-channel tiles on 32 x 16 pixels, 32-channel tiles on 16 x 16) -- the registers holding item
// i + 1 are split and written into the other buffer, and item i + 2's loads issued, in slices woven between the 27 MFMA steps of item
// i: one barrier per item, and the split's vector instructions run in the shadow of the MFMAs (the scheme of igb::k_igb_conv3).  These
// are the full-resolution layers, bound by HBM and by the commit, not by the matrix pipe.
template <int NN, int MODE, int NW, int WN = 1, bool DB = false>
__global__ __launch_bounds__(64 * NW, (NW == 8 || NN <= 2) ? 2 : 1) void k_ig3x_conv3(ConvArgs p, const bf16_t* __restrict__ w3, unsigned pstride) {
    static_assert(WN == 1 || (WN == 2 && NW == 8 && NN % 2 == 0), "channel split: eight waves, two halves");
    constexpr int NWR = NW / WN, NJ = NN / WN;            // row groups of waves; 16-channel tiles per wave
    constexpr int NT = 64 * NW, TR = 4 * NWR, PATCHX = (TR + 2) * (T + 2);
    constexpr int COT = 16 * NN;
    constexpr int APL = PATCHX * KC;                 // bf16 elements per A plane: [patch pixel][16 channels], 32-byte rows
    constexpr int BPL = 9 * COT * KC;                // per B plane: [tap][channel of the tile][16 K channels]
    constexpr int BOFF = 3 * APL, DUMP = BOFF + 3 * BPL, BUF = DUMP + 64;      // + a dump row for the idle lanes of the last staging element
    static_assert(!DB || 2 * BUF * 2 + 8192 <= 160 * 1024, "two buffers must fit");
    __shared__ __attribute__((aligned(16))) bf16_t lds[DB ? 2 * BUF : BUF];
    __shared__ float bn_red[NW * 2 * 16 * NJ];       // cross-wave fold of the fused BatchNorm statistics (epilogue_t)
    // the bias vector, read by the epilogue through LDS: a global load there queues behind the next item's prefetch (vmcnt retires in
    // order) and cost the epilogue of a 16-channel unit 4 k cycles of waiting (tools/x3_stamps.py)
    // (data gradient with ConvArgs::bnb: the BatchNorm's mean and 1 / sigma instead, [n_dst0] each)
    __shared__ __attribute__((aligned(16))) float bias_lds[MODE == 1 ? 4 : kMaxBias];
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) % NWR, wn = (tid >> 6) / NWR;          // row group, channel half
    const int m16 = lane & 15, q = lane >> 4;
    const int kin = p.c_src0 + p.c_src1, nout = p.n_dst0 + p.n_dst1;
    if (MODE == 0)
        for (int c = tid; c < nout; c += NT) bias_lds[c] = p.bias ? p.bias[c] : 0.f;          // (visible behind the first item's barrier)
    else if (MODE == 2)
        for (int c = tid; c < 2 * nout; c += NT) bias_lds[c] = p.bnb.coef[2 * nout + c];      // [mean | inv] of the [4][C] table, C = nout
    const int nco = nout / COT, ntiles = p.tiles_x * p.tiles_y * p.B, nunits = ntiles * nco;
    const int nchunks = kin / KC;
    const bool xcd_map = (ntiles & 7) == 0 && (gridDim.x & 7) == 0;
    const int my_units = (nunits - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (my_units <= 0) {          // (the launcher sizes the grid to the units: not reached)
        if (MODE == 0 && p.bnf.tab) bn_self_fold(p.bnf, gridDim.x, blockIdx.x);
        if (MODE == 2) bn_bwd_self_fold(p.bnb, gridDim.x, blockIdx.x);
        return;
    }

    const size_t npix = (size_t)p.B * p.H * p.W;
    const unsigned nbytes0 = (unsigned)(npix * p.c_src0 * 4), nbytes1 = (unsigned)(npix * p.c_src1 * 4);
    const __amdgpu_buffer_rsrc_t rsw =
        __builtin_amdgcn_make_buffer_rsrc((void*)w3, 0, (unsigned)((2 * (size_t)pstride + (size_t)9 * nout * kin) * 2), BUF_FLAGS);

    const FastDiv d_nco(nco), d_tx(p.tiles_x), d_ty(p.tiles_y);
    struct Unit { int b, y0, x0, co0, tile; };
    auto unit_of = [&](int k) {
        const int id = blockIdx.x + k * gridDim.x;
        int tile, cot;
        if (xcd_map) {              // ids congruent mod 8 share an XCD: keep a tile's channel blocks there
            const int xcd = id & 7, j = id >> 3, jq = d_nco.div(j);
            cot = j - jq * nco;
            tile = jq * 8 + xcd;
        } else {
            tile = d_nco.div(id);
            cot = id - tile * nco;
        }
        Unit u;
        const int trow = d_tx.div(tile), bx = tile - trow * p.tiles_x;
        u.b = d_ty.div(trow);
        const int by = trow - u.b * p.tiles_y;
        u.x0 = bx * T; u.y0 = by * TR; u.co0 = cot * COT; u.tile = tile;
        return u;
    };

    // ---- staging geometry of this thread.  A element v: patch pixel (tid >> 2) + (NT / 4) v, channels 4 (tid & 3) .. + 3 (one 16-byte
    //      load, three 8-byte LDS stores).  B element v: 16-byte piece i = tid + NT v of the item's 3 x 9 x COT rows of 32 bytes --
    //      LDS is filled linearly, the global offset of a piece is fixed per thread up to the item's (channel tile, K chunk).
    constexpr int AU = (PATCHX * 4 + NT - 1) / NT, NPIECE = 3 * 9 * COT * 2, BU = (NPIECE + NT - 1) / NT;
    const int c4 = tid & 3;
    int a_ly[AU], a_lx[AU];
#pragma unroll
    for (int v = 0; v < AU; ++v) {
        const int px = (tid >> 2) + (NT / 4) * v;
        a_ly[v] = px / (T + 2);
        a_lx[v] = px - a_ly[v] * (T + 2);
        if (px >= PATCHX) a_ly[v] = -4096;          // never inside an image
    }
    unsigned b_off[BU];          // element offset of piece v inside the planes, without the item's part
#pragma unroll
    for (int v = 0; v < BU; ++v) {
        const int i = tid + NT * v, rowi = i >> 1, n = rowi % COT, tp = rowi / COT;
        const int plane = (tp * 57) >> 9, tap = tp - 9 * plane;          // tp < 27
        b_off[v] = i < NPIECE ? (unsigned)plane * pstride + (unsigned)((tap * nout + n) * kin + 8 * (i & 1)) : OOB;
    }
    u32x4 ar[AU], br[BU];
    struct Stage { int b, y0, x0, cc, cs, c0; unsigned oob, wbase; __amdgpu_buffer_rsrc_t rs; };
    int sg_k = 0, sg_cc = 0;
    unsigned sg_oob = 0u;
    Unit sg_u = unit_of(0);
    auto next_stage = [&]() {
        Stage st;
        st.b = sg_u.b; st.y0 = sg_u.y0; st.x0 = sg_u.x0;
        st.cc = sg_cc;
        const bool second = st.cc >= p.c_src0;
        st.cs = second ? p.c_src1 : p.c_src0;
        st.c0 = second ? st.cc - p.c_src0 : st.cc;
        st.oob = sg_oob;                    // past the last item: every offset out of range
        st.wbase = (unsigned)(sg_u.co0 * kin + st.cc);
        st.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(second ? p.src[1] : p.src[0]), 0, second ? nbytes1 : nbytes0, BUF_FLAGS);
        sg_cc += KC;
        if (sg_cc >= kin) {
            sg_cc = 0;
            ++sg_k;
            if (sg_k < my_units) sg_u = unit_of(sg_k);
            else sg_oob = OOB;
        }
        return st;
    };
    // normalise-on-load (MODE 0, ConvArgs::norm): scale / shift quad of the item in the registers + an inside-the-image bit per element
    const bool norm_any = MODE == 0 && (p.norm[0] != nullptr || p.norm[1] != nullptr);          // block-uniform
    float4 n_sc = make_float4(1.f, 1.f, 1.f, 1.f), n_sh = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned a_in = 0u;
    auto issue_item = [&](const Stage& st) {
        a_in = 0u;
#pragma unroll
        for (int v = 0; v < AU; ++v) {
            const int iy = st.y0 - 1 + a_ly[v], ix = st.x0 - 1 + a_lx[v];
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const unsigned off = (ok ? (unsigned)(((((st.b * p.H + iy) * p.W + ix) * st.cs) + st.c0 + 4 * c4) * 4) : OOB) | st.oob;
            ar[v] = __builtin_amdgcn_raw_buffer_load_b128(st.rs, off, 0, 0);
            a_in |= ok ? (1u << v) : 0u;
        }
#pragma unroll
        for (int v = 0; v < BU; ++v)
            br[v] = __builtin_amdgcn_raw_buffer_load_b128(rsw, (b_off[v] == OOB ? OOB : (b_off[v] + st.wbase) * 2u) | st.oob, 0, 0);
        if (norm_any) {
            n_sc = make_float4(1.f, 1.f, 1.f, 1.f);
            n_sh = make_float4(0.f, 0.f, 0.f, 0.f);
            const float* nt = p.norm[st.cc >= p.c_src0 ? 1 : 0];          // uniform
            if (nt) {
                n_sc = *reinterpret_cast<const float4*>(nt + st.c0 + 4 * c4);
                n_sh = *reinterpret_cast<const float4*>(nt + st.cs + st.c0 + 4 * c4);
            }
        }
    };
    // ---- the same per element, for the double-buffered loop: `cur` belongs to the item in the registers, `nxt` to the one being issued
    float4 n_sc_nxt = n_sc, n_sh_nxt = n_sh;
    unsigned a_in_nxt = 0u;
    auto issue_a1 = [&](const Stage& st, int v) {
        const int iy = st.y0 - 1 + a_ly[v], ix = st.x0 - 1 + a_lx[v];
        const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const unsigned off = (ok ? (unsigned)(((((st.b * p.H + iy) * p.W + ix) * st.cs) + st.c0 + 4 * c4) * 4) : OOB) | st.oob;
        ar[v] = __builtin_amdgcn_raw_buffer_load_b128(st.rs, off, 0, 0);
        a_in_nxt |= ok ? (1u << v) : 0u;
    };
    auto issue_b1 = [&](const Stage& st, int v) {
        br[v] = __builtin_amdgcn_raw_buffer_load_b128(rsw, (b_off[v] == OOB ? OOB : (b_off[v] + st.wbase) * 2u) | st.oob, 0, 0);
    };
    auto stage_coef = [&](const Stage& st, float4& sc, float4& sh) {
        sc = make_float4(1.f, 1.f, 1.f, 1.f);
        sh = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* nt = p.norm[st.cc >= p.c_src0 ? 1 : 0];          // uniform
        if (nt) {
            sc = *reinterpret_cast<const float4*>(nt + st.c0 + 4 * c4);
            sh = *reinterpret_cast<const float4*>(nt + st.cs + st.c0 + 4 * c4);
        }
    };
    auto commit_a1 = [&](bf16_t* buf, int v) {
        const int px = (tid >> 2) + (NT / 4) * v;
        f32x4 f = __builtin_bit_cast(f32x4, ar[v]);
        if (norm_any) {
            const bool in = (a_in >> v) & 1u;
            f[0] = in ? fmaf(f[0], n_sc.x, n_sh.x) : 0.f; f[1] = in ? fmaf(f[1], n_sc.y, n_sh.y) : 0.f;
            f[2] = in ? fmaf(f[2], n_sc.z, n_sh.z) : 0.f; f[3] = in ? fmaf(f[3], n_sc.w, n_sh.w) : 0.f;
        }
        bf16x4 h0, h1, h2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bf16_t x0, x1, x2;
            split3(f[e], x0, x1, x2);
            h0[e] = x0; h1[e] = x1; h2[e] = x2;
        }
        bf16_t* dst = buf + (px < PATCHX ? px * KC : DUMP) + 4 * c4;          // idle lanes (last element only): the dump row
        *reinterpret_cast<bf16x4*>(dst) = h0;
        *reinterpret_cast<bf16x4*>(dst + (px < PATCHX ? APL : 16)) = h1;
        *reinterpret_cast<bf16x4*>(dst + (px < PATCHX ? 2 * APL : 32)) = h2;
    };
    auto commit_b1 = [&](bf16_t* buf, int v) {
        const int i = tid + NT * v;
        *reinterpret_cast<u32x4*>(buf + (i < NPIECE ? BOFF + 8 * i : DUMP + 8 * (i & 7))) = br[v];
    };
    auto commit_item = [&]() {
#pragma unroll
        for (int v = 0; v < AU; ++v) {
            const int px = (tid >> 2) + (NT / 4) * v;
            f32x4 f = __builtin_bit_cast(f32x4, ar[v]);
            if (norm_any) {
                const bool in = (a_in >> v) & 1u;
                f[0] = in ? fmaf(f[0], n_sc.x, n_sh.x) : 0.f; f[1] = in ? fmaf(f[1], n_sc.y, n_sh.y) : 0.f;
                f[2] = in ? fmaf(f[2], n_sc.z, n_sh.z) : 0.f; f[3] = in ? fmaf(f[3], n_sc.w, n_sh.w) : 0.f;
            }
            bf16x4 h0, h1, h2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bf16_t x0, x1, x2;
                split3(f[e], x0, x1, x2);
                h0[e] = x0; h1[e] = x1; h2[e] = x2;
            }
            bf16_t* dst = lds + (px < PATCHX ? px * KC : DUMP) + 4 * c4;          // idle lanes (last element only): the dump row
            *reinterpret_cast<bf16x4*>(dst) = h0;
            *reinterpret_cast<bf16x4*>(dst + (px < PATCHX ? APL : 16)) = h1;
            *reinterpret_cast<bf16x4*>(dst + (px < PATCHX ? 2 * APL : 32)) = h2;
        }
#pragma unroll
        for (int v = 0; v < BU; ++v) {
            const int i = tid + NT * v;
            *reinterpret_cast<u32x4*>(lds + (i < NPIECE ? BOFF + 8 * i : DUMP + 8 * (i & 7))) = br[v];
        }
    };

    // ---- MFMA fragment addresses of this lane: rows of 16 channels, K half 8 (q & 1); the plane depends on q >> 1
    const int hA = q >> 1;
    const bf16_t* a_base = lds + ((4 * wave) * (T + 2) + m16) * KC + 8 * (q & 1);
    const bf16_t* aX = a_base + hA * APL;               // (a0 | a1)
    const bf16_t* aY = a_base + 2 * hA * APL;           // (a0 | a2)
    const bf16_t* b_base = lds + BOFF + (16 * NJ * wn + m16) * KC + 8 * (q & 1);
    const bf16_t* bP[3] = {b_base, b_base + BPL, b_base + (hA ? 0 : 2 * BPL)};          // (b0 | b0), (b1 | b1), (b2 | b0)

    f32x4 acc[4][NJ];
    f32x4 zpre[4][MODE == 2 ? NJ : 1];          // MODE 2: bnb_prefetch
    if constexpr (DB) {
        static_assert(AU + BU <= 12, "staging slices of the 27 MFMA steps");
        // prologue: item 0 into buffer 0, item 1 into the registers
        {
            const Stage s0 = next_stage();
            if (norm_any) stage_coef(s0, n_sc, n_sh);
#pragma unroll
            for (int v = 0; v < AU; ++v) issue_a1(s0, v);
#pragma unroll
            for (int v = 0; v < BU; ++v) issue_b1(s0, v);
            a_in = a_in_nxt;
            a_in_nxt = 0u;
#pragma unroll
            for (int v = 0; v < AU; ++v) commit_a1(lds, v);
#pragma unroll
            for (int v = 0; v < BU; ++v) commit_b1(lds, v);
            const Stage s1 = next_stage();
            if (norm_any) stage_coef(s1, n_sc, n_sh);
#pragma unroll
            for (int v = 0; v < AU; ++v) issue_a1(s1, v);
#pragma unroll
            for (int v = 0; v < BU; ++v) issue_b1(s1, v);
            a_in = a_in_nxt;
            a_in_nxt = 0u;
        }
        lds_barrier();
        int it = 0;
#pragma unroll 1
        for (int k = 0; k < my_units; ++k) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int chunk = 0; chunk < nchunks; ++chunk, ++it) {
                const int boff = (it & 1) * BUF;
                bf16_t* other = lds + ((it & 1) ^ 1) * BUF;
                const Stage nx = next_stage();                        // item it + 2
                if (norm_any) stage_coef(nx, n_sc_nxt, n_sh_nxt);     // used by the NEXT item's commits
                if constexpr (MODE == 2) {
                    if (chunk == nchunks - 1) {
                        const Unit uz = unit_of(k);
                        bnb_prefetch<NJ, NWR, WN>(p, zpre, uz.b, uz.y0, uz.x0, uz.co0);
                    }
                }
                bf16x8 fa[2][6], fb[2][NJ];
                auto load_a = [&](bf16x8 (&a)[6], const bf16_t* base, int g) {
#pragma unroll
                    for (int rr = 0; rr < 6; ++rr) a[rr] = *reinterpret_cast<const bf16x8*>(base + boff + (rr * (T + 2) + g) * KC);
                };
                auto load_b = [&](bf16x8 (&b)[NJ], int pr, int tap) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) b[j] = *reinterpret_cast<const bf16x8*>(bP[pr] + boff + (tap * COT + 16 * j) * KC);
                };
                auto mfmas = [&](const bf16x8 (&a)[6], const bf16x8 (&b)[NJ], int dy) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[r + dy], acc[r][j], 0, 0, 0);
                };
                load_a(fa[0], aX, 0);
                load_b(fb[0], 0, 0);
#pragma unroll
                for (int g = 0; g < 3; ++g) {
#pragma unroll
                    for (int s = 0; s < 9; ++s) {
                        const int dy = s < 6 ? s >> 1 : s - 6, cur = (9 * g + s) & 1, step = 9 * g + s;
                        if (s == 0) load_a(fa[1], aY, g);
                        if (s == 6 && g < 2) load_a(fa[0], aX, g + 1);
                        if (s < 8 || g < 2) {
                            const int s1 = s < 8 ? s + 1 : 0, g1 = s < 8 ? g : g + 1;
                            const int dy1 = s1 < 6 ? s1 >> 1 : s1 - 6, pr1 = s1 < 6 ? (s1 & 1) : 2;
                            load_b(fb[cur ^ 1], pr1, dy1 * 3 + g1);
                        }
                        // staging slice: every second step commits one element of item it + 1 into the other buffer and re-uses its
                        // registers for item it + 2 (unconditional: past the last item the loads are out of range and the commit
                        // writes a buffer nobody reads again)
                        if (step >= 1 && (step & 1) && (step >> 1) < AU + BU) {
                            const int v = step >> 1;
                            if (v < AU) { commit_a1(other, v); issue_a1(nx, v); }
                            else { commit_b1(other, v - AU); issue_b1(nx, v - AU); }
                        }
                        mfmas(fa[s < 6 ? 0 : 1], fb[cur], dy);
                    }
                }
                n_sc = n_sc_nxt; n_sh = n_sh_nxt;          // the registers now hold item it + 2
                a_in = a_in_nxt;
                a_in_nxt = 0u;
                lds_barrier();
            }
            const Unit u = unit_of(k);
            epilogue_t<NJ, MODE, NWR, WN>(p, acc, u.b, u.y0, u.x0, u.co0, u.tile, bn_red, bias_lds, zpre);
        }
        if (MODE == 0 && p.bnf.tab) bn_self_fold(p.bnf, gridDim.x, blockIdx.x);
        if (MODE == 2) bn_bwd_self_fold(p.bnb, gridDim.x, blockIdx.x);
        return;
    }
    issue_item(next_stage());
    int it = 0;
#pragma unroll 1
    for (int k = 0; k < my_units; ++k) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int chunk = 0; chunk < nchunks; ++chunk, ++it) {
            X3STAMP(it, 0);
            commit_item();                       // the previous item's MFMAs have been fenced off by the barrier below
            X3STAMP(it, 1);
            lds_barrier();
            X3STAMP(it, 2);
            // the next item's loads: issued BEHIND the barrier -- their address arithmetic (1 - 3 k cycles per item in front of it,
            // tools/x3_stamps.py) then runs beside the other waves' MFMAs instead of holding the whole block up
            issue_item(next_stage());
            if constexpr (MODE == 2) {
                if (chunk == nchunks - 1) {
                    const Unit uz = unit_of(k);
                    bnb_prefetch<NJ, NWR, WN>(p, zpre, uz.b, uz.y0, uz.x0, uz.co0);
                }
            }
            X3STAMP(it, 3);
            // 27 steps of 4 NN MFMAs: dx-major (g), per g first the (a0 | a1) fragments against (b0 | b0) and (b1 | b1) for the three dy taps,
            // then the (a0 | a2) fragments against (b2 | b0).  Two register sets for both operands: the LDS reads of the NEXT step (B) and
            // of the next A set are issued before this step's MFMAs (the compiler's own order put every read right in front of its MFMA
            // behind an lgkmcnt(0): one exposed LDS round trip per step).
            {
                bf16x8 fa[2][6], fb[2][NJ];
                auto load_a = [&](bf16x8 (&a)[6], const bf16_t* base, int g) {
#pragma unroll
                    for (int rr = 0; rr < 6; ++rr) a[rr] = *reinterpret_cast<const bf16x8*>(base + (rr * (T + 2) + g) * KC);
                };
                auto load_b = [&](bf16x8 (&b)[NJ], int pr, int tap) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) b[j] = *reinterpret_cast<const bf16x8*>(bP[pr] + (tap * COT + 16 * j) * KC);
                };
                auto mfmas = [&](const bf16x8 (&a)[6], const bf16x8 (&b)[NJ], int dy) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[r + dy], acc[r][j], 0, 0, 0);      // rows = channels, columns = pixels
                };
                load_a(fa[0], aX, 0);
                load_b(fb[0], 0, 0);
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    // steps 0..5: X set (fa[0]); steps 6..8: Y set (fa[1]).  B sets alternate: step s of this g uses fb[(9 g + s) & 1]
#pragma unroll
                    for (int s = 0; s < 9; ++s) {
                        const int dy = s < 6 ? s >> 1 : s - 6, cur = (9 * g + s) & 1;
                        if (s == 0) load_a(fa[1], aY, g);                         // this g's (a0 | a2) set, needed from step 6
                        if (s == 6 && g < 2) load_a(fa[0], aX, g + 1);            // the next g's (a0 | a1) set
                        if (s < 8 || g < 2) {                                     // B fragments of the next step
                            const int s1 = s < 8 ? s + 1 : 0, g1 = s < 8 ? g : g + 1;
                            const int dy1 = s1 < 6 ? s1 >> 1 : s1 - 6, pr1 = s1 < 6 ? (s1 & 1) : 2;
                            load_b(fb[cur ^ 1], pr1, dy1 * 3 + g1);
                        }
                        mfmas(fa[s < 6 ? 0 : 1], fb[cur], dy);
                    }
                }
            }
            X3STAMP(it, 4);
            lds_barrier();
            X3STAMP(it, 5);
        }
        const Unit u = unit_of(k);
        epilogue_t<NJ, MODE, NWR, WN>(p, acc, u.b, u.y0, u.x0, u.co0, u.tile, bn_red, bias_lds, zpre);
        X3STAMP(it - 1, 6);
    }
    if (MODE == 0 && p.bnf.tab) bn_self_fold(p.bnf, gridDim.x, blockIdx.x);          // block-uniform; every block gets here
    if (MODE == 2) bn_bwd_self_fold(p.bnb, gridDim.x, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[t][ci][co] = sum over pixels of X[pixel + t][ci] dY[pixel][co] (components.py:46-52 under GradientTape): M = input channels,
// N = output channels, K = pixels, all nine taps at once -- the persistent scheme of ig::k_ig_wgrad2 (one block per CU walks its share
// of the pixel tiles with the accumulators in registers; raw buffer loads, the next tile in registers while this one is consumed;
// in-block reduction of the pixel-split waves; bucket copies for small gradients; BatchNorm scale / shift on load; bias gradient from
// the staged dY quads) -- with both operands split into three bf16 planes while they are staged (X: [plane][patch pixel][ci],
// dY: [plane][tile pixel][co]) and the K-contiguous (pixel-major) fragments taken out of the NHWC rows by ds_read_b64_tr_b16 as in
// igb::k_igb_wgrad64.  One K = 32 MFMA step covers 16 pixels (two tile rows x 8 columns) of TWO planes: lane groups q = 0, 1 read
// the first plane of the pair, q = 2, 3 the second, so per tap and 16 x 16 channel tile a step is the three MFMAs
//         (x0 | x1) . (g0 | g0)  +  (x0 | x1) . (g1 | g1)  +  (x0 | x2) . (g2 | g0)
// of the forward kernel.  Eight waves: MW input-channel tiles x WN output-channel halves x WK pixel splits; NJ 16-channel output
// tiles per wave (at most 72 accumulator registers, two waves per SIMD).
typedef bf16x4 __attribute__((address_space(3))) * lds4_t;
__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* row0, const bf16_t* row1) {
    const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)row0);
    const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)row1);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// LDS row strides (bf16 elements): the eight rows a 32-lane half of a transposing read touches must fall into distinct 32-byte bank
// groups -- stride / 16 elements odd
constexpr int wg_stride(int ch) { return ch == 16 ? 16 : ch + 16; }
constexpr int wg_lds_bytes(int cit, int cot, int tyw) { return 3 * 2 * ((tyw + 2) * (T + 2) * wg_stride(cit) + tyw * T * wg_stride(cot)) + 64; }
// tile rows: 8 TM, as many as fit (the smaller the channel block, the taller the tile: the per-tile costs stay amortised)
constexpr int wg_tm(int cit, int cot) {
    return wg_lds_bytes(cit, cot, 32) <= 150 * 1024 ? 4 : (wg_lds_bytes(cit, cot, 16) <= 150 * 1024 ? 2 : 1);
}

template <int MW, int NJ, int WN, int WK>
__global__ __launch_bounds__(512, 2) void k_ig3x_wgrad(ig::WgArgs p) {
    static_assert(MW * WN * WK == 8, "eight waves");
    constexpr int NT = 512;
    constexpr int CIT = 16 * MW, COT = 16 * NJ * WN;
    constexpr int TM = wg_tm(CIT, COT), TYW = 8 * TM, PW = T + 2, PPATCH = (TYW + 2) * PW, NPX = TYW * T;
    constexpr int XS = wg_stride(CIT), GS = wg_stride(COT);
    constexpr int XPL = PPATCH * XS, GPL = NPX * GS;                     // bf16 elements per plane
    constexpr int XQ = CIT / 4, GQ = COT / 4;                            // channel quads per pixel
    constexpr int XU = (PPATCH * XQ + NT - 1) / NT, GU = (NPX * GQ + NT - 1) / NT;
    constexpr int NSTEP = NPX / 16;                                      // K = 32 steps per tile: (row pair, column half)
    static_assert(NSTEP % WK == 0, "K steps per wave");
    constexpr int NKS = NSTEP / WK;
    constexpr int IMG = 3 * XPL + 3 * GPL + 32;                          // bf16 elements (+ a dump row)
    constexpr int RF = MW * WN * 9 * NJ * 256;                           // floats of the in-block reduction (overlays the images)
    constexpr int SMEM = IMG * 2 > RF * 4 ? IMG * 2 : RF * 4;
    static_assert(SMEM <= 160 * 1024 && NT * 16 <= SMEM, "LDS");
    static_assert(NT % XQ == 0 && NT % GQ == 0, "one channel quad per thread");
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SMEM];
    bf16_t* ximg = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* gimg = ximg + 3 * XPL;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m16 = lane & 15, q = lane >> 4;
    const int gq = (lane >> 2) & 3, gp = lane & 3;       // transposing read: lane 4 gq + gp of a group addresses row gq, columns 4 gp ..
    const int wm = wave % MW, wn = (wave / MW) % WN, wk = wave / (MW * WN);
    const int c0 = blockIdx.y * CIT, co0 = blockIdx.z * COT;
    const bool do_bias = p.dbias && blockIdx.y == 0;          // block-uniform
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    const int tiles_y = (p.H + TYW - 1) / TYW;
    const FastDiv d_tx(p.tiles_x), d_ty(tiles_y);
    const int ntiles = p.tiles_x * tiles_y * p.B;
    const size_t npix = (size_t)p.B * p.H * p.W;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (unsigned)(npix * p.cs * 4), BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc((void*)p.dz, 0, (unsigned)(npix * p.cout * 4), BUF_FLAGS);

    f32x4 acc[9][NJ];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging geometry: X element u = patch pixel (tid + NT u) / XQ, channel quad (tid + NT u) % XQ (the same quad for every u);
    // dY likewise with GQ
    u32x4 xr[XU], gr[GU];
    const bool norm_on = p.norm != nullptr;
    float4 n_sc = make_float4(1.f, 1.f, 1.f, 1.f), n_sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (norm_on) {
        n_sc = *reinterpret_cast<const float4*>(p.norm + c0 + 4 * (tid % XQ));
        n_sh = *reinterpret_cast<const float4*>(p.norm + p.cs + c0 + 4 * (tid % XQ));
    }
    unsigned x_in = 0u;                        // bit u: element u of the tile in registers lies inside the image
    auto issue = [&](int tile) {               // tile >= ntiles: stage nothing (every offset out of range)
        x_in = 0u;
        const unsigned oob = tile < ntiles ? 0u : OOB;
        tile = tile < ntiles ? tile : 0;
        const int trow = d_tx.div(tile), bx = tile - trow * p.tiles_x, b = d_ty.div(trow), by = trow - b * tiles_y;
        const int x0 = bx * T, y0 = by * TYW;
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int i = tid + NT * u, px = i / XQ, c4 = i % XQ;
            const int ly = px / PW, lx = px - ly * PW;
            const int iy = y0 - 1 + ly, ix = x0 - 1 + lx;
            const bool ok = px < PPATCH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const unsigned off = (ok ? (unsigned)(((((b * p.H + iy) * p.W + ix) * p.cs) + c0 + 4 * c4) * 4) : OOB) | oob;
            xr[u] = __builtin_amdgcn_raw_buffer_load_b128(rsx, off, 0, 0);
            x_in |= (ok && !oob) ? (1u << u) : 0u;
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int i = tid + NT * u, px = i / GQ, n4 = i % GQ;
            const int ly = px / T, lx = px - ly * T;
            const int iy = y0 + ly, ix = x0 + lx;
            const bool ok = px < NPX && iy < p.H && ix < p.W;
            const unsigned off = (ok ? (unsigned)(((((b * p.H + iy) * p.W + ix) * p.cout) + co0 + 4 * n4) * 4) : OOB) | oob;
            gr[u] = __builtin_amdgcn_raw_buffer_load_b128(rsg, off, 0, 0);
        }
    };
    auto split_store = [&](f32x4 f, bf16_t* dst, int plane_stride) {
        bf16x4 h0, h1, h2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bf16_t a0, a1, a2;
            split3(f[e], a0, a1, a2);
            h0[e] = a0; h1[e] = a1; h2[e] = a2;
        }
        *reinterpret_cast<bf16x4*>(dst) = h0;
        *reinterpret_cast<bf16x4*>(dst + plane_stride) = h1;
        *reinterpret_cast<bf16x4*>(dst + 2 * plane_stride) = h2;
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int i = tid + NT * u, px = i / XQ, c4 = i % XQ;
            f32x4 f = __builtin_bit_cast(f32x4, xr[u]);
            if (norm_on) {          // block-uniform
                const bool in = (x_in >> u) & 1u;
                f[0] = in ? fmaf(f[0], n_sc.x, n_sh.x) : 0.f; f[1] = in ? fmaf(f[1], n_sc.y, n_sh.y) : 0.f;
                f[2] = in ? fmaf(f[2], n_sc.z, n_sh.z) : 0.f; f[3] = in ? fmaf(f[3], n_sc.w, n_sh.w) : 0.f;
            }
            if (px < PPATCH) split_store(f, ximg + px * XS + 4 * c4, XPL);
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int i = tid + NT * u, px = i / GQ, n4 = i % GQ;
            const f32x4 gv = __builtin_bit_cast(f32x4, gr[u]);
            if (px < NPX) {
                split_store(gv, gimg + px * GS + 4 * n4, GPL);
                if (do_bias) { bsum[0] += gv[0]; bsum[1] += gv[1]; bsum[2] += gv[2]; bsum[3] += gv[3]; }      // (pixels outside the image were loaded as zeros)
            }
        }
    };
    // per-lane bases of the transposing reads (bf16 elements): pixel column 4 (q & 1) + gq of the step's 8-column half; plane by q >> 1
    const int hA = q >> 1;
    const int xlane = (4 * (q & 1) + gq) * XS + 16 * wm + 4 * gp;
    const int glane = (4 * (q & 1) + gq) * GS + 16 * NJ * wn + 4 * gp;
    const bf16_t* xX = ximg + hA * XPL + xlane;                     // (x0 | x1)
    const bf16_t* xY = ximg + 2 * hA * XPL + xlane;                 // (x0 | x2)
    const bf16_t* gP[3] = {gimg + glane, gimg + GPL + glane, gimg + (hA ? 0 : 2 * GPL) + glane};      // (g0 | g0), (g1 | g1), (g2 | g0)

    int tile = blockIdx.x;
    issue(tile);
#pragma unroll 1
    for (; tile < ntiles; tile += p.psplit) {
        lds_barrier();              // the previous tile's operand reads are complete
        commit();
        lds_barrier();
        issue(tile + p.psplit);     // in flight during this tile's MFMAs (past the end: nothing)
#pragma unroll 1
        for (int s = 0; s < NKS; ++s) {
            // (unrolling this loop over consecutive row pairs of one column half shares patch rows between steps -- 0.56 instead of 0.89
            // transposing reads per MFMA, the launch alone 60.3 -> 55.7 us -- and made the mulmo_unet STEP 1.5 % slower, beside the data-
            // gradient launches of the main stream: round-4 log in NOTES.md.  Not kept.)
            const int ks = wk + WK * s, rp = ks >> 1, ch = (ks & 1) * 8;          // tile rows 2 rp, 2 rp + 1; columns ch .. ch + 7
            bf16x8 gv[3][NJ];
#pragma unroll
            for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    gv[pr][j] = tr_frag(gP[pr] + ((2 * rp) * T + ch) * GS + 16 * j, gP[pr] + ((2 * rp + 1) * T + ch) * GS + 16 * j);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3, dx = t % 3;
                const int o0 = ((2 * rp + dy) * PW + ch + dx) * XS, o1 = ((2 * rp + 1 + dy) * PW + ch + dx) * XS;
                const bf16x8 ax = tr_frag(xX + o0, xX + o1), ay = tr_frag(xY + o0, xY + o1);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax, gv[0][j], acc[t][j], 0, 0, 0);
                    acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax, gv[1][j], acc[t][j], 0, 0, 0);
                    acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ay, gv[2][j], acc[t][j], 0, 0, 0);
                }
            }
        }
    }
    // the WK pixel-split waves of a (ci tile, co half) first add up inside the block (through the LDS of the images, one wave set at a
    // time), then wave set 0 adds into the gradient (copy blockIdx.x % nbuckets of it)
    float* red = reinterpret_cast<float*>(smem_raw) + (wm + MW * wn) * (9 * NJ * 256);
    if (WK > 1) {
        for (int r = 1; r < WK; ++r) {
            __syncthreads();
            if (wk == r) {
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) red[((t * NJ + j) * 4 + i) * 64 + lane] = acc[t][j][i];
            }
            __syncthreads();
            if (wk == 0) {
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[t][j][i] += red[((t * NJ + j) * 4 + i) * 64 + lane];
            }
        }
    }
    const size_t boff = p.plain ? (size_t)blockIdx.x * p.bucket_stride : (size_t)(p.nbuckets > 1 ? blockIdx.x % p.nbuckets : 0) * p.bucket_stride;
    if (do_bias) {          // fold the NT / GQ threads of every channel quad (the images are free behind a barrier)
        float* fs = reinterpret_cast<float*>(smem_raw);
        __syncthreads();
        *reinterpret_cast<float4*>(fs + 4 * tid) = make_float4(bsum[0], bsum[1], bsum[2], bsum[3]);
        __syncthreads();
        if (tid < COT) {
            const int n4 = tid >> 2, k = tid & 3;
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < NT / GQ; ++r) a += fs[4 * (n4 + GQ * r) + k];
            if (p.plain) p.dbias[boff + co0 + tid] = a;
            else atomicAdd(p.dbias + boff + co0 + tid, a);
        }
    }
    if (wk != 0) return;
    // D[ci = 16 wm + 4q + i][co = 16 (NJ wn + j) + m16]
    float* dwb = p.dw + boff + ((size_t)(p.ci_off + c0 + 16 * wm + 4 * q)) * p.cout + co0 + 16 * NJ * wn + m16;
    if (p.plain) {          // block-uniform
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) dwb[((size_t)t * p.cin_total + i) * p.cout + 16 * j] = acc[t][j][i];
        return;
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) atomicAdd(dwb + ((size_t)t * p.cin_total + i) * p.cout + 16 * j, acc[t][j][i]);
}

}  // namespace ig3x

// ================================================================================================ host side
struct Ig3xPlan {
    bool built = false;
    std::vector<ig3x::PrepDesc> preps;
    ig3x::PrepDesc* preps_dev = nullptr;
    ig3x::bf16_t* wf = nullptr;        // [3][pstride]: forward layout, element offsets as in the parameter vector
    ig3x::bf16_t* wd = nullptr;        // data-gradient layout
    unsigned pstride = 0;
    int max_w = 0;
};
static std::map<Model*, Ig3xPlan> g_ig3x;

void ig3x_release(Model* m) { g_ig3x.erase(m); }

// fp32 models only; DNNCA_NO_X3=1 keeps the exact-fp32 MFMA kernels (read once per process)
bool ig3x_enabled(const Model* m) {
    static const bool off = getenv("DNNCA_NO_X3") != nullptr;
    return !off && m->desc.dtype == DNNCA_F32 && !(m->desc.flags & 1);
}

int ig3x_prepare(Model* m) {
    if (!ig3x_enabled(m)) return DNNCA_OK;
    Ig3xPlan& pl = g_ig3x[m];
    if (!pl.built) {
        pl.built = true;
        for (const Op& o : m->ops) {
            if (!ig_conv_supported(m, o)) continue;
            ig3x::PrepDesc d{(int)o.w_off, o.inA.d.C + o.inB.d.C, o.out.d.C};
            pl.preps.push_back(d);
            const int n = 9 * ((d.cin + 31) / 32) * ((d.cout + 31) / 32);
            if (n > pl.max_w) pl.max_w = n;
        }
        if (!pl.preps.empty()) {
            pl.pstride = (unsigned)((m->nT + 15) / 16 * 16);
            DN_TRY(m->alloc((void**)&pl.preps_dev, pl.preps.size() * sizeof(ig3x::PrepDesc)));
            DN_TRY(m->alloc((void**)&pl.wf, (size_t)pl.pstride * 3 * 2 + 64));
            DN_TRY(m->alloc((void**)&pl.wd, (size_t)pl.pstride * 3 * 2 + 64));
            HIP_TRY(hipMemcpyAsync(pl.preps_dev, pl.preps.data(), pl.preps.size() * sizeof(ig3x::PrepDesc), hipMemcpyHostToDevice, m->stream));
            HIP_TRY(hipStreamSynchronize(m->stream));
        }
    }
    if (!pl.preps.empty()) {
        int bx = pl.max_w;
        if (bx > 512) bx = 512;
        LAUNCH(m, "ig3x_prep", 16.0 * m->nT, 0,
               hipLaunchKernelGGL(ig3x::k_ig3x_prep, dim3(bx, (unsigned)pl.preps.size()), dim3(256), 0, m->stream, pl.preps_dev, m->p,
                                  pl.wf, pl.wd, pl.pstride));
    }
    return DNNCA_OK;
}

// launches the conv (mode 0 forward / 1 data gradient) described by `a` (tiles_x / tiles_y are set here) on the split-bf16 kernel with
// channel tiles of 16 nn; false: not this path (the caller goes on to the exact-fp32 kernels).
// Wave layout: eight waves on 32 x 16-pixel tiles while that gives every CU a unit; else (nn >= 2) eight waves on 16 x 16 tiles, split
// 4 row groups x 2 channel halves; else four waves on 16 x 16 tiles.  DNNCA_IG_NW=4|8 forces the first / last (tuning aid, tests).
// would ig3x_launch take this conv?  (ig_conv_bwd asks before it hands the BatchNorm backward sums to the launch: ConvArgs::bnb)
bool ig3x_accepts(Model* m, const ig::ConvArgs& a, int cout) {
    if (!ig3x_enabled(m)) return false;
    Ig3xPlan& pl = g_ig3x[m];
    if (!m->dry && (!pl.wf || !pl.wd)) return false;
    if (a.src_half || a.dst_half || a.dsth[0] || a.dsth[1] || cout > ig3x::kMaxBias) return false;          // bf16-stored tensors: dtype bf16 only
    if (9.0 * cout * (a.c_src0 + a.c_src1) + 2.0 * pl.pstride > 1.0e9) return false;          // 32-bit byte offsets into the planes
    return true;
}
int ig3x_max_bnb_channels() { return ig3x::kMaxBias / 2; }          // mean and 1 / sigma share the bias table

bool ig3x_launch(Model* m, int mode, const ig::ConvArgs& a, size_t w_off, int cout, int nn, const char* name, double bytes, double flops,
                 bool* bnb_rode) {
    if (bnb_rode) *bnb_rode = false;
    if (!ig3x_accepts(m, a, cout)) return false;
    Ig3xPlan& pl = g_ig3x[m];
    static const int forced = getenv("DNNCA_IG_NW") ? atoi(getenv("DNNCA_IG_NW")) : 0;
    static const bool no_split = getenv("DNNCA_X3_NO_SPLIT") != nullptr;          // tuning aid
    static const bool no_db = getenv("DNNCA_X3_NO_DB") != nullptr;                // tuning aid / A-B arm
    const long units8 = (long)((a.W + 15) / 16) * ((a.H + 31) / 32) * a.B * (cout / (16 * nn));
    int nw = 8, wn = 1;
    if (forced == 4) nw = 4;
    else if (forced != 8 && units8 < 256) {
        if (nn >= 2 && !no_split) wn = 2;
        else nw = 4;
    }
    // two LDS buffers where they fit: 16-channel tiles on 32 x 16 pixels, 32-channel tiles on the split 16 x 16 layout
    if (nn == 2 && nw == 8 && !no_db && !no_split && forced != 8) wn = 2;
    const bool db = !no_db && nw == 8 && ((nn == 1 && wn == 1) || (nn == 2 && wn == 2));
    const int rows = 4 * (nw / wn);
    ig::ConvArgs a2 = a;
    a2.tiles_x = (a.W + ig3x::T - 1) / ig3x::T;
    a2.tiles_y = (a.H + rows - 1) / rows;
    const unsigned units = (unsigned)(a2.tiles_x * a2.tiles_y * a2.B * (cout / (16 * nn)));
    typedef void (*Kern)(ig::ConvArgs, const ig3x::bf16_t*, unsigned);
    // [mode: forward, data gradient, data gradient + BatchNorm backward sums][nn index][layout: 4 waves, 8 waves, 8 waves split]
    // (no <4, 2, 8>: that layout has no registers left for the sums -- such launches leave them to the BatchNorm's reduction pass)
    static const Kern kerns[3][3][3] = {
        {{ig3x::k_ig3x_conv3<1, 0, 4>, ig3x::k_ig3x_conv3<1, 0, 8>, nullptr},
         {ig3x::k_ig3x_conv3<2, 0, 4>, ig3x::k_ig3x_conv3<2, 0, 8>, ig3x::k_ig3x_conv3<2, 0, 8, 2>},
         {ig3x::k_ig3x_conv3<4, 0, 4>, ig3x::k_ig3x_conv3<4, 0, 8>, ig3x::k_ig3x_conv3<4, 0, 8, 2>}},
        {{ig3x::k_ig3x_conv3<1, 1, 4>, ig3x::k_ig3x_conv3<1, 1, 8>, nullptr},
         {ig3x::k_ig3x_conv3<2, 1, 4>, ig3x::k_ig3x_conv3<2, 1, 8>, ig3x::k_ig3x_conv3<2, 1, 8, 2>},
         {ig3x::k_ig3x_conv3<4, 1, 4>, ig3x::k_ig3x_conv3<4, 1, 8>, ig3x::k_ig3x_conv3<4, 1, 8, 2>}},
        {{ig3x::k_ig3x_conv3<1, 2, 4>, ig3x::k_ig3x_conv3<1, 2, 8>, nullptr},
         {ig3x::k_ig3x_conv3<2, 2, 4>, ig3x::k_ig3x_conv3<2, 2, 8>, ig3x::k_ig3x_conv3<2, 2, 8, 2>},
         {ig3x::k_ig3x_conv3<4, 2, 4>, nullptr, ig3x::k_ig3x_conv3<4, 2, 8, 2>}}};
    const int ni = nn == 4 ? 2 : (nn == 2 ? 1 : 0), li = wn == 2 ? 2 : (nw == 8 ? 1 : 0);
    static const Kern kerns_db[3][2] = {{ig3x::k_ig3x_conv3<1, 0, 8, 1, true>, ig3x::k_ig3x_conv3<2, 0, 8, 2, true>},
                                        {ig3x::k_ig3x_conv3<1, 1, 8, 1, true>, ig3x::k_ig3x_conv3<2, 1, 8, 2, true>},
                                        {ig3x::k_ig3x_conv3<1, 2, 8, 1, true>, ig3x::k_ig3x_conv3<2, 2, 8, 2, true>}};
    int mi = mode ? 1 : 0;
    if (mode == 1 && a.bnb.C > 0 && (db ? kerns_db[2][ni] : kerns[2][ni][li]) != nullptr) mi = 2;
    if (bnb_rode) *bnb_rode = mi == 2;
    const Kern kern = db ? kerns_db[mi][ni] : kerns[mi][ni][li];
    // a persistent kernel's grid is the number of blocks that are resident at once: several per CU where LDS and registers allow
    // (a block alternates between committing an item and running its MFMAs; co-resident blocks fill each other's commit phases)
    static int occ[3][3][4] = {};
    int& oc = occ[mi][ni][db ? 3 : li];
    if (oc == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), 64 * nw, 0) != hipSuccess || nb < 1) nb = 1;
        static const int cap = getenv("DNNCA_X3_BLOCKS") ? atoi(getenv("DNNCA_X3_BLOCKS")) : 0;          // tuning aid
        if (cap > 0 && nb > cap) nb = cap;
        oc = nb;
    }
    const unsigned resident = 256u * (unsigned)oc;
    const unsigned g = units < resident ? units : resident;
    const ig3x::bf16_t* w3 = (mode == 0 ? pl.wf : pl.wd) + w_off;
    m->set_variant("x3n%dw%d%s%s%s", nn, nw, wn == 2 ? "s" : "", db ? "d" : "", mi == 2 ? "b" : "");      // b: BatchNorm backward sums ride
    LAUNCH(m, name, bytes, flops, hipLaunchKernelGGL(kern, dim3(g), dim3(64 * nw), 0, m->stream, a2, w3, pl.pstride));
    return true;
}

// the split-bf16 weight-gradient kernel's shape for one source: channel tiles, wave roles, pixel-split blocks; false: not this path
struct Wg3Shape { int mw, nn, wn, nj, wk, tyw, ps; };
static bool wg3_shape(const Model* m, const ig::WgArgs& w, int co, Wg3Shape* sh) {
    if (!ig3x_enabled(m)) return false;
    static const bool off = getenv("DNNCA_NO_X3_WGRAD") != nullptr;
    if (off) return false;
    const int cs = w.cs;
    if (cs % 16 || co % 16) return false;
    if ((double)w.B * w.H * w.W * (cs > co ? cs : co) * 4.0 >= 2.0e9) return false;          // 32-bit byte offsets
    sh->mw = cs % 64 == 0 ? 4 : (cs % 32 == 0 ? 2 : 1);
    sh->nn = co % 64 == 0 ? 4 : (co % 32 == 0 ? 2 : 1);
    const int rest = 8 / sh->mw;
    sh->wn = (sh->nn >= 2 && rest >= 2) ? 2 : 1;
    sh->nj = sh->nn / sh->wn;
    sh->wk = rest / sh->wn;
    const int cit = 16 * sh->mw, cot = 16 * sh->nn;
    sh->tyw = 8 * ig3x::wg_tm(cit, cot);
    const int nt = ((w.W + ig3x::T - 1) / ig3x::T) * ((w.H + sh->tyw - 1) / sh->tyw) * w.B;
    const int combos = (cs / cit) * (co / cot);
    int ps = (256 + combos - 1) / combos;
    if (ps > nt) ps = nt;
    sh->ps = ps < 1 ? 1 : ps;
    return true;
}

// pixel-split blocks (= slabs in plain mode) of the launch ig3x_wgrad_launch would make; 0: not this path
int ig3x_wgrad_psplit(const Model* m, const ig::WgArgs& w, int co) {
    Wg3Shape sh;
    return wg3_shape(m, w, co, &sh) ? sh.ps : 0;
}

// weight gradient of one source on the split-bf16 kernel; w: geometry and pointers filled by the caller (psplit is set here).
// false: not this path.
bool ig3x_wgrad_launch(Model* m, ig::WgArgs w, int co, const char* name, double bytes, double flops) {
    Wg3Shape sh;
    if (!wg3_shape(m, w, co, &sh)) return false;
    const int mw = sh.mw, nn = sh.nn, cit = 16 * mw, cot = 16 * nn;
    w.tiles_x = (w.W + ig3x::T - 1) / ig3x::T;
    w.psplit = sh.ps;
    const dim3 g(w.psplit, w.cs / cit, co / cot);
    m->set_variant("x3m%dj%dn%dk%d%s", mw, sh.nj, sh.wn, sh.wk, w.plain ? "p" : "");
#define WG3(MWv, NJv, WNv, WKv) LAUNCH(m, name, bytes, flops, hipLaunchKernelGGL((ig3x::k_ig3x_wgrad<MWv, NJv, WNv, WKv>), g, dim3(512), 0, m->stream, w))
    if (mw == 4) { if (nn == 4) WG3(4, 2, 2, 1); else if (nn == 2) WG3(4, 1, 2, 1); else WG3(4, 1, 1, 2); }
    else if (mw == 2) { if (nn == 4) WG3(2, 2, 2, 2); else if (nn == 2) WG3(2, 1, 2, 2); else WG3(2, 1, 1, 4); }
    else { if (nn == 4) WG3(1, 2, 2, 4); else if (nn == 2) WG3(1, 1, 2, 4); else WG3(1, 1, 1, 8); }
#undef WG3
    return true;
}

}  // namespace dnnca

#ifdef DNNCA_TUNING
extern "C" int dnnca_debug_x3_stamps(unsigned long long* out, int n) {
    if (n > 64 * 8) n = 64 * 8;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(dnnca::ig3x::g_x3_stamps), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
#endif
