// kernels_igemm.hip -- implicit-GEMM 3x3 convolutions on the fp32 matrix cores for channel counts that are multiples
// of 16 (configs/mulmo_unet.yaml: 16..384, configs/unet_big.yaml: 64..1024).  Here the contraction is dense:
//     forward / data gradient:  M = pixels (block tile 8 rows x 16 px), N = output channels (16*NN per block),
//                               K = 9 taps x input channels, walked in chunks of CK = 16 input channels;
//     weight gradient:          M = 16 input channels, N = 16*NN output channels, K = pixels, all 9 taps at once
//                               (accumulators live in registers while a persistent block walks its pixel tiles).
// Per chunk a block stages the (8+2) x (16+2) x CK input patch and the 9 x CK x 16*NN weight slab in LDS once and issues
// 288 (NN = 4) MFMAs per wave between two barriers; the decoder concat is two sources, never materialised; bias +
// activation (forward) or act' mask + accumulate + destination split (data gradient) are fused into the store.
// The data gradient is the forward kernel on a transposed + flipped copy of the weights (k_ig_flip, once per step).
#include <cstring>
#include <type_traits>

#include "bn_dev.h"
#include "fast.h"
#include "kernels.h"

#include "ig_dev.h"

namespace dnnca {

namespace ig {

// MODE 0 forward, MODE 1 data gradient
template <int NN, int MODE>
__global__ __launch_bounds__(256) void k_ig_conv(ConvArgs p) {
    constexpr int NT = 16 * NN, BSTR = NT + 16;        // weight slab row stride (floats): conflict-free B reads
    __shared__ float a_lds[PATCH * CKP];
    __shared__ float b_lds[9 * CK * BSTR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m16 = lane & 15, q = lane >> 4;
    const int cin = p.c_src0 + p.c_src1, cout = p.n_dst0 + p.n_dst1;
    const int tile = blockIdx.x, bx = tile % p.tiles_x, by = (tile / p.tiles_x) % p.tiles_y, b = tile / (p.tiles_x * p.tiles_y);
    const int x0 = bx * TX, y0 = by * TY;
    const int co0 = blockIdx.y * NT;

    f32x4 acc[2][NN];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < NN; ++j) acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int cc = 0; cc < cin; cc += CK) {
        const bool second = cc >= p.c_src0;
        const float* src = second ? p.src[1] : p.src[0];
        const int cs = second ? p.c_src1 : p.c_src0, c0 = second ? cc - p.c_src0 : cc;
        lds_barrier();        // the previous chunk's reads are complete
        // input patch: PATCH pixels x 4 float4 (16 channels), zero outside the image
        for (int i = tid; i < PATCH * (CK / 4); i += 256) {
            const int px = i >> 2, c4 = i & 3;
            const int ly = px / (TX + 2), lx = px - ly * (TX + 2);
            const int iy = y0 - 1 + ly, ix = x0 - 1 + lx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                v = *reinterpret_cast<const float4*>(src + (((size_t)b * p.H + iy) * p.W + ix) * cs + c0 + 4 * c4);
            *reinterpret_cast<float4*>(a_lds + px * CKP + 4 * c4) = v;
        }
        // weight slab: 9 taps x CK rows x NT columns
        for (int i = tid; i < 9 * CK * (NT / 4); i += 256) {
            const int n4 = i % (NT / 4), rk = i / (NT / 4);      // rk = tap * CK + k
            const int tap = rk / CK, k = rk - tap * CK;
            const float4 v = *reinterpret_cast<const float4*>(p.w + ((size_t)tap * cin + cc + k) * cout + co0 + 4 * n4);
            *reinterpret_cast<float4*>(b_lds + rk * BSTR + 4 * n4) = v;
        }
        lds_barrier();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
#pragma unroll
            for (int k4 = 0; k4 < CK / 4; ++k4) {
                float bv[NN];
#pragma unroll
                for (int j = 0; j < NN; ++j) bv[j] = b_lds[(tap * CK + 4 * k4 + q) * BSTR + 16 * j + m16];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const float av = a_lds[((2 * wave + r + dy) * (TX + 2) + m16 + dx) * CKP + 4 * k4 + q];
#pragma unroll
                    for (int j = 0; j < NN; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[j], acc[r][j], 0, 0, 0);
                }
            }
        }
    }
    // epilogue: D[pixel 4q+i][channel 16j + m16]
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int y = y0 + 2 * wave + r;
        if (y >= p.H) continue;
#pragma unroll
        for (int j = 0; j < NN; ++j) {
            const int co = co0 + 16 * j + m16;
            const int which = co >= p.n_dst0;
            const int cw = which ? p.n_dst1 : p.n_dst0, cl = which ? co - p.n_dst0 : co;
            const float bias = (MODE == 0 && p.bias) ? p.bias[co] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int x = x0 + 4 * q + i;
                if (x >= p.W) continue;
                const size_t o = (((size_t)b * p.H + y) * p.W + x) * cw + cl;
                float v = acc[r][j][i];
                if (MODE == 0) {
                    v += bias;
                    v = p.alpha < 0.f ? v : (v > 0.f ? v : p.alpha * v);
                } else {
                    if (p.acc[which]) v += p.dst[which][o];
                    if (p.mask[which]) v *= p.mask[which][o] > 0.f ? 1.0f : p.alpha;
                }
                p.dst[which][o] = v;
            }
        }
    }
}

// flipped + transposed kernels for the data gradient: wT[t][co][ci] = w[8 - t][ci][co]
struct FlipDesc {
    int w_off, cin, cout;
};
__global__ void k_ig_flip(const FlipDesc* __restrict__ descs, const float* __restrict__ params, float* __restrict__ flipped) {
    const FlipDesc d = descs[blockIdx.y];
    const int n = 9 * d.cin * d.cout;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int ci = i % d.cin, co = (i / d.cin) % d.cout, t = i / (d.cin * d.cout);
        flipped[d.w_off + i] = params[d.w_off + ((8 - t) * d.cin + ci) * d.cout + co];
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// (WgArgs: ig_dev.h)

// grid: x = pixel split, y = input-channel chunk (16), z = output-channel tile (16*NN)
template <int NN>
__global__ __launch_bounds__(256) void k_ig_wgrad(WgArgs p) {
    constexpr int NT = 16 * NN, GSTR = NT + 4;
    __shared__ float x_lds[PATCH * CKP];
    __shared__ float g_lds[TY * TX * GSTR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m16 = lane & 15, q = lane >> 4;
    const int c0 = blockIdx.y * CK, co0 = blockIdx.z * NT;
    const bool do_bias = p.dbias && blockIdx.y == 0;
    const int ntiles = p.tiles_x * p.tiles_y * p.B;

    f32x4 acc[9][NN];
    f32x4 accb[NN];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < NN; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NN; ++j) accb[j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += p.psplit) {
        const int bx = tile % p.tiles_x, by = (tile / p.tiles_x) % p.tiles_y, b = tile / (p.tiles_x * p.tiles_y);
        const int x0 = bx * TX, y0 = by * TY;
        lds_barrier();
        for (int i = tid; i < PATCH * (CK / 4); i += 256) {
            const int px = i >> 2, c4 = i & 3;
            const int ly = px / (TX + 2), lx = px - ly * (TX + 2);
            const int iy = y0 - 1 + ly, ix = x0 - 1 + lx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                v = *reinterpret_cast<const float4*>(p.x + (((size_t)b * p.H + iy) * p.W + ix) * p.cs + c0 + 4 * c4);
            *reinterpret_cast<float4*>(x_lds + px * CKP + 4 * c4) = v;
        }
        for (int i = tid; i < TY * TX * (NT / 4); i += 256) {
            const int n4 = i % (NT / 4), px = i / (NT / 4);
            const int ly = px / TX, lx = px - ly * TX;
            const int iy = y0 + ly, ix = x0 + lx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy < p.H && ix < p.W)
                v = *reinterpret_cast<const float4*>(p.dz + (((size_t)b * p.H + iy) * p.W + ix) * p.cout + co0 + 4 * n4);
            *reinterpret_cast<float4*>(g_lds + px * GSTR + 4 * n4) = v;
        }
        lds_barrier();
        // every wave: 2 rows of the tile = 8 K-steps of 4 pixels
#pragma unroll 1
        for (int ks = 0; ks < 8; ++ks) {
            const int row = 2 * wave + (ks >> 2), px0 = (ks & 3) * 4;
            float bv[NN];
#pragma unroll
            for (int j = 0; j < NN; ++j) bv[j] = g_lds[(row * TX + px0 + q) * GSTR + 16 * j + m16];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float av = x_lds[((row + t / 3) * (TX + 2) + px0 + q + t % 3) * CKP + m16];
#pragma unroll
                for (int j = 0; j < NN; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[j], acc[t][j], 0, 0, 0);
            }
            if (do_bias) {
#pragma unroll
                for (int j = 0; j < NN; ++j) accb[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, bv[j], accb[j], 0, 0, 0);
            }
        }
    }
    // D[ci = 4q + i][co = 16j + m16] -> dW[tap][ci_off + c0 + ci][co0 + co]; four waves add their partial sums
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < NN; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                atomicAdd(p.dw + ((size_t)t * p.cin_total + p.ci_off + c0 + 4 * q + i) * p.cout + co0 + 16 * j + m16, acc[t][j][i]);
    if (do_bias && q == 0) {
#pragma unroll
        for (int j = 0; j < NN; ++j) atomicAdd(p.dbias + co0 + 16 * j + m16, accb[j][0]);     // row 0 of the all-ones A
    }
}


// weight gradient, second generation: block tile = 16*MW input channels x 16*NN output channels x 9 taps, one persistent
// block per CU.  The four waves split the input channels MW ways and the pixels (K) 4/MW ways; the pixel tile grows when
// the channel tile is small so that a tile always carries ~288*NN MFMAs per wave and ~20 KB of staging per thread-round;
// X and dY are read once per (ci block, co block) pair; the next tile's global loads (raw buffer loads: out-of-image
// pixels come back as zeros without branches) are issued into registers before a tile's MFMAs and written to LDS after
// them; the 13 LDS operand words of K-step s+1 are loaded before the 36 MFMAs of step s.
constexpr unsigned WG_FLAGS = 0x00020000u, WG_OOB = 0x80000000u;
// phase stamps of k_ig_wgrad2 / k_ig_conv3 (tuning builds only; tools/wg_stamps.py): block 0, thread 0, the first 64 tiles / items
#ifdef DNNCA_TUNING
__device__ unsigned long long g_wg_stamps[64 * 8];
__device__ __forceinline__ unsigned long long wg_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define WGSTAMP(item, ph) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && (item) < 64) g_wg_stamps[(item) * 8 + (ph)] = wg_now(); } while (0)
#else
#define WGSTAMP(item, ph) do { } while (0)
#endif
typedef unsigned int wg_u32x4 __attribute__((ext_vector_type(4)));

// NWV = 8: eight waves (two per SIMD), the additional four split the tile's pixels (K) further -- for the narrow tiles whose
// accumulators leave room (NN <= 2): the fp32 matrix pipe of a CU then has a second wave to issue from while the first waits.
template <int MW, int NN, int NWV>
__global__ __launch_bounds__(64 * NWV, NWV / 4) void k_ig_wgrad2(WgArgs p) {
    constexpr int NT = 64 * NWV;
    constexpr int CIT = 16 * MW, COT = 16 * NN, WK = NWV / MW;
    constexpr int TM = (4 / MW) < (4 / NN) ? (4 / MW) : (4 / NN);
    constexpr int TYW = 8 * TM, PW = TX + 2, PPATCH = (TYW + 2) * PW, NPX = TYW * TX;
    constexpr int XS = CIT + (CIT == 16 ? 0 : 16), GS = COT + (COT == 16 ? 0 : 16);   // row strides = 16 (mod 32) banks
    constexpr int XQ = CIT / 4, GQ = COT / 4;                                           // float4 per pixel
    constexpr int XU = (PPATCH * XQ + NT - 1) / NT, GU = (NPX * GQ + NT - 1) / NT;
    static_assert(NPX % (4 * WK) == 0 && (NPX / 4 / WK) % 2 == 0, "K-steps per wave");
    constexpr int NKS = NPX / 4 / WK;                                                   // K-steps per wave per tile
    constexpr int XF = PPATCH * XS + 16, GF = NPX * GS;
    constexpr int RF = MW * 9 * NN * 256 > NT * 4 ? MW * 9 * NN * 256 : NT * 4;          // buffer of the in-block reductions (overlays the staged tiles)
    __shared__ __attribute__((aligned(16))) float smem[XF + GF > RF ? XF + GF : RF];
    float* x_lds = smem;
    float* g_lds = smem + XF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m16 = lane & 15, q = lane >> 4;
    const int wm = wave % MW, wk = wave / MW;
    const int c0 = blockIdx.y * CIT, co0 = blockIdx.z * COT;
    // bias gradient = sum of dY over the pixels: every thread sums the channel quad it stages (the same one for every element: NT is a
    // multiple of GQ), folded through LDS at the end.  (As an all-ones MFMA operand it was a branch inside the K-step loop.)
    const bool do_bias = p.dbias && blockIdx.y == 0;          // block-uniform
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    const int tiles_y = (p.H + TYW - 1) / TYW;
    const FastDiv d_tx(p.tiles_x), d_ty(tiles_y);
    const int ntiles = p.tiles_x * tiles_y * p.B;
    const size_t npix = (size_t)p.B * p.H * p.W;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (unsigned)(npix * p.cs * 4), WG_FLAGS);
    const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc((void*)p.dz, 0, (unsigned)(npix * p.cout * 4), WG_FLAGS);

    f32x4 acc[9][NN];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < NN; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging geometry: X element u = patch pixel (tid + 256u) / XQ, float4 (tid + 256u) % XQ; dY likewise with GQ
    wg_u32x4 xr[XU], gr[GU];
    // normalise-on-load (WgArgs::norm): the thread stages the same channel quad of every element (NT is a multiple of XQ)
    static_assert(NT % XQ == 0, "one channel quad per thread");
    const bool norm_on = p.norm != nullptr;
    float4 n_sc = make_float4(1.f, 1.f, 1.f, 1.f), n_sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (norm_on) {
        n_sc = *reinterpret_cast<const float4*>(p.norm + c0 + 4 * (tid % XQ));
        n_sh = *reinterpret_cast<const float4*>(p.norm + p.cs + c0 + 4 * (tid % XQ));
    }
    unsigned x_in = 0u;                        // bit u: element u of the tile in registers lies inside the image
    auto issue = [&](int tile) {               // tile >= ntiles: stage nothing (every offset out of range)
        x_in = 0u;
        const unsigned oob = tile < ntiles ? 0u : WG_OOB;
        tile = tile < ntiles ? tile : 0;
        const int trow = d_tx.div(tile), bx = tile - trow * p.tiles_x, b = d_ty.div(trow), by = trow - b * tiles_y;
        const int x0 = bx * TX, y0 = by * TYW;
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int i = tid + NT * u, px = i / XQ, c4 = i % XQ;
            const int ly = px / PW, lx = px - ly * PW;
            const int iy = y0 - 1 + ly, ix = x0 - 1 + lx;
            const bool ok = px < PPATCH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const unsigned off = (ok ? (unsigned)(((((b * p.H + iy) * p.W + ix) * p.cs) + c0 + 4 * c4) * 4) : WG_OOB) | oob;
            xr[u] = __builtin_amdgcn_raw_buffer_load_b128(rsx, off, 0, 0);
            x_in |= (ok && !oob) ? (1u << u) : 0u;
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int i = tid + NT * u, px = i / GQ, n4 = i % GQ;
            const int ly = px / TX, lx = px - ly * TX;
            const int iy = y0 + ly, ix = x0 + lx;
            const bool ok = px < NPX && iy < p.H && ix < p.W;
            const unsigned off = (ok ? (unsigned)(((((b * p.H + iy) * p.W + ix) * p.cout) + co0 + 4 * n4) * 4) : WG_OOB) | oob;
            gr[u] = __builtin_amdgcn_raw_buffer_load_b128(rsg, off, 0, 0);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int i = tid + NT * u, px = i / XQ, c4 = i % XQ;
            // lanes past the patch (last element only) write into the 16-float dump row behind it
            if (norm_on) {          // block-uniform
                f32x4 f = __builtin_bit_cast(f32x4, xr[u]);
                const bool in = (x_in >> u) & 1u;
                f[0] = in ? fmaf(f[0], n_sc.x, n_sh.x) : 0.f; f[1] = in ? fmaf(f[1], n_sc.y, n_sh.y) : 0.f;
                f[2] = in ? fmaf(f[2], n_sc.z, n_sh.z) : 0.f; f[3] = in ? fmaf(f[3], n_sc.w, n_sh.w) : 0.f;
                xr[u] = __builtin_bit_cast(wg_u32x4, f);
            }
            *reinterpret_cast<wg_u32x4*>(x_lds + (px < PPATCH ? px * XS + 4 * c4 : PPATCH * XS + 4 * (c4 & 3))) = xr[u];
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int i = tid + NT * u, px = i / GQ, n4 = i % GQ;
            if (px < NPX) *reinterpret_cast<wg_u32x4*>(g_lds + px * GS + 4 * n4) = gr[u];
            if (do_bias && px < NPX) {          // (pixels outside the image were loaded as zeros)
                const f32x4 gv = __builtin_bit_cast(f32x4, gr[u]);
                bsum[0] += gv[0]; bsum[1] += gv[1]; bsum[2] += gv[2]; bsum[3] += gv[3];
            }
        }
    };
    // operand words of K-step ks of this wave (4 pixels: tile row ks / 4, pixels 4 (ks % 4) ..)
    auto load_step = [&](int s, float (&av)[9], float (&bv)[NN]) {
        const int ks = wk + WK * s, row = ks >> 2, px0 = (ks & 3) * 4;
#pragma unroll
        for (int j = 0; j < NN; ++j) bv[j] = g_lds[(row * TX + px0 + q) * GS + 16 * j + m16];
#pragma unroll
        for (int t = 0; t < 9; ++t) av[t] = x_lds[((row + t / 3) * PW + px0 + q + t % 3) * XS + 16 * wm + m16];
    };
    auto mfma_step = [&](const float (&av)[9], const float (&bv)[NN]) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < NN; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[j], acc[t][j], 0, 0, 0);
    };

    int tile = blockIdx.x;
    if (tile < ntiles) issue(tile);
    int wg_it = 0;
#pragma unroll 1
    for (; tile < ntiles; tile += p.psplit, ++wg_it) {
        WGSTAMP(wg_it, 0);
        lds_barrier();              // the previous tile's operand reads are complete
        WGSTAMP(wg_it, 1);
        commit();
        WGSTAMP(wg_it, 2);
        issue(tile + p.psplit);
        WGSTAMP(wg_it, 3);
        lds_barrier();
        WGSTAMP(wg_it, 4);
        float a0[9], b0[NN], a1[9], b1[NN];
        load_step(0, a0, b0);
#pragma unroll 1
        for (int s = 0; s < NKS; s += 2) {
            // the scheduling fences keep a step's operand reads in front of the previous step's MFMAs (hipcc sinks them to their
            // first use otherwise: tools/wg_stamps.py read 400 ticks of exposed LDS latency per 1152-tick step on the one wave a SIMD has)
            load_step(s + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            load_step(s + 2 < NKS ? s + 2 : 0, a0, b0);      // past the last step: a harmless reload
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
        WGSTAMP(wg_it, 5);
    }
    WGSTAMP(wg_it < 63 ? wg_it : 63, 6);
    // the WK pixel-split waves of a channel tile first add up inside the block (through the staged-tile LDS, one wave set
    // at a time), then wave set 0 adds into the gradient (copy blockIdx.x % nbuckets of it)
    if (WK > 1) {
        float* red = smem + wm * (9 * NN * 256);
        for (int r = 1; r < WK; ++r) {
            __syncthreads();
            if (wk == r) {
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int j = 0; j < NN; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) red[((t * NN + j) * 4 + i) * 64 + lane] = acc[t][j][i];
            }
            __syncthreads();
            if (wk == 0) {
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int j = 0; j < NN; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[t][j][i] += red[((t * NN + j) * 4 + i) * 64 + lane];
            }
        }
    }
    const size_t boff = (size_t)(p.nbuckets > 1 ? blockIdx.x % p.nbuckets : 0) * p.bucket_stride;
    if (do_bias) {          // fold the NT / GQ threads of every channel quad (the tile buffers are free behind a barrier)
        __syncthreads();
        *reinterpret_cast<float4*>(smem + 4 * tid) = make_float4(bsum[0], bsum[1], bsum[2], bsum[3]);
        __syncthreads();
        if (tid < COT) {
            const int n4 = tid >> 2, k = tid & 3;
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < NT / GQ; ++r) a += smem[4 * (n4 + GQ * r) + k];
            atomicAdd(p.dbias + boff + co0 + tid, a);
        }
    }
    if (wk != 0) return;
    // D[ci = 16 wm + 4q + i][co = 16j + m16]
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < NN; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                atomicAdd(p.dw + boff + ((size_t)t * p.cin_total + p.ci_off + c0 + 16 * wm + 4 * q + i) * p.cout + co0 + 16 * j + m16, acc[t][j][i]);
}

// sum of the bucket copies of a small weight gradient: dst[i] += sum_b slabs[b][i]; the copies are left zeroed for the next use
__global__ __launch_bounds__(256) void k_wg_fold(float* __restrict__ slabs, int nb, int stride, int n_w, float* __restrict__ dw,
                                                 float* __restrict__ dbias, int n_b) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_w + n_b) return;
    float s = 0.f;
    constexpr int U = 8;                 // eight copies in flight per thread (the loop used to wait for every single load)
    for (int b0 = 0; b0 < nb; b0 += U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = slabs[(size_t)min(b0 + u, nb - 1) * stride + i];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (b0 + u < nb) {
                s += v[u];
                slabs[(size_t)(b0 + u) * stride + i] = 0.f;
            }
    }
    if (i < n_w) dw[i] += s;
    else dbias[i - n_w] += s;
}

// One fold (a launch's slabs -> the gradient rows of its source) as the batched kernel sees it.
struct FoldSeg {
    const float* slabs;
    float* dw;               // the conv's gradient [9][cin_total][cout]
    float* dbias;
    int nb, stride, cs, cout, cin_total, ci_off, n_b;
    int g16;                 // 1: the 16-group block shape (small gradients), 0: one group
};

// slab groups per block: 256 / G quads x G groups (small gradients: few quads, many slabs -> G = 16; large: G = 1)
template <int G>
__device__ __forceinline__ void wg_fold_plain_block(const FoldSeg& f, int block, f32x4* red) {
    // thread (quad qi, group g) sums slabs g, g + G, ... (eight loads in flight); the groups meet in LDS
    constexpr int Q = 256 / G;
    const int n_w = 9 * f.cs * f.cout;
    const int qi = threadIdx.x % Q, g = threadIdx.x / Q;
    const int i = 4 * (block * Q + qi);
    const bool live = i < n_w + f.n_b;
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    if (live) {
        constexpr int U = 8;
        for (int b0 = g; b0 < f.nb; b0 += G * U) {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int b = b0 + G * u;
                v[u] = *reinterpret_cast<const f32x4*>(f.slabs + (size_t)(b < f.nb ? b : g) * f.stride + i);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (b0 + G * u < f.nb) s += v[u];
        }
    }
    if (G > 1) {
        red[g * (Q + 1) + qi] = s;
        __syncthreads();
        if (g != 0) return;
#pragma unroll
        for (int k = 1; k < G; ++k) s += red[k * (Q + 1) + qi];
    }
    if (!live) return;
    if (i < n_w) {
        const int t = i / (f.cs * f.cout), r = i - t * (f.cs * f.cout);          // cout % 4 == 0: the quad stays inside one row
        float* d = f.dw + ((size_t)t * f.cin_total + f.ci_off) * f.cout + r;
        *reinterpret_cast<f32x4*>(d) = *reinterpret_cast<const f32x4*>(d) + s;
    } else {
        float* d = f.dbias + (i - n_w);
        *reinterpret_cast<f32x4*>(d) = *reinterpret_cast<const f32x4*>(d) + s;
    }
}
constexpr int kFoldRed = 16 * 17;      // f32x4 entries of the G = 16 shape

// WgArgs::plain: the sum of a launch's psplit slabs goes into the gradient.  A slab is [9][cs][cout] (+ cout bias sums when n_b > 0); the
// gradient rows of this source sit at input channels ci_off .. ci_off + cs of [9][cin_total][cout].  One thread = four consecutive floats
// of a slab; the slabs are not re-zeroed (every launch stores every element of every slab).
template <int G>
__global__ __launch_bounds__(256) void k_wg_fold_plain(FoldSeg f) {
    __shared__ f32x4 red[G > 1 ? kFoldRed : 1];
    wg_fold_plain_block<G>(f, blockIdx.x, red);
}

// Every fold of a backward pass in ONE launch (single-replica steps: nothing reads the gradient before the optimizer): `first[s]` is
// the first block of segment s (first[nseg] = the grid), nseg <= 64 -- lane s of each wave compares its entry, the ballot counts the
// segments that start at or before this block.
constexpr int kFoldBatch = 64;
__global__ __launch_bounds__(256) void k_wg_fold_batch(const FoldSeg* __restrict__ segs, const int* __restrict__ first, int nseg) {
    __shared__ f32x4 red[kFoldRed];
    const int lane = threadIdx.x & 63;
    const int mine = lane < nseg ? first[lane] : 0x7fffffff;
    const int s = __builtin_amdgcn_readfirstlane(__popcll(__ballot(mine <= (int)blockIdx.x)) - 1);
    const int b0 = __builtin_amdgcn_readfirstlane(first[s]);
    const FoldSeg f = segs[s];
    if (f.g16) wg_fold_plain_block<16>(f, blockIdx.x - b0, red);
    else wg_fold_plain_block<1>(f, blockIdx.x - b0, red);
}

// forward / data gradient, fp32, persistent and software-pipelined: the f32 twin of igb::k_igb_conv3 (see there for the
// scheme).  Block tile 16 x 16 pixels x 16*NN channels, K chunks of 16 input channels; per item 576*NN/4... MFMAs per
// wave (4 M-tiles x NN N-tiles x 9 taps x 4 K-steps of v_mfma_f32_16x16x4_f32), walked dx-major: one step = (dx, K-step)
// loads 6 A words that serve the three dy taps, operand words of step s+1 are loaded before the MFMAs of step s.
constexpr int F3T = 16;                                         // tile width (columns); rows: 4 per wave
constexpr int F3AS = CK + 4;                                    // A row stride (floats)

// NW = 8 (NN <= 2: the two buffers of a 32-row tile fit): 32 x 16-pixel tiles, two waves per SIMD -- the second wave covers the
// first one's LDS waits, epilogue and barrier, and the weight slab serves twice the MFMAs (see igb::k_igb_conv3).
template <int NN, int MODE, int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void k_ig_conv3(ConvArgs p) {
    constexpr int NT = 64 * NW, TR = 4 * NW, PATCHX = (TR + 2) * (F3T + 2);
    constexpr int COT = 16 * NN, BS = COT + 16;                 // B row stride (floats): 16 (mod 32) banks
    constexpr int ABUF = PATCHX * F3AS, BBUF = 9 * CK * BS;
    constexpr int BUF = ABUF + BBUF + 16;                       // floats per LDS buffer
    static_assert(NW == 4 || (NW == 8 && NN <= 2), "eight waves: 16- and 32-channel tiles only (LDS)");
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];
    __shared__ float bn_red[NW * 2 * COT];      // cross-wave fold of the fused BatchNorm statistics (conv3_epilogue)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m16 = lane & 15, q = lane >> 4;
    const int kin = p.c_src0 + p.c_src1, nout = p.n_dst0 + p.n_dst1;
    const int nco = nout / COT, ntiles = p.tiles_x * p.tiles_y * p.B, nunits = ntiles * nco;
    const int nchunks = kin / CK;
    const bool xcd_map = (ntiles & 7) == 0 && (gridDim.x & 7) == 0;
    const int my_units = (nunits - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nitems = my_units * nchunks;
    if (nitems <= 0) {          // (the launchers size the grid to the units: not reached)
        if (MODE == 0 && p.bnf.tab) bn_self_fold(p.bnf, gridDim.x, blockIdx.x);
        return;
    }

    const size_t npix = (size_t)p.B * p.H * p.W;
    const unsigned nbytes0 = (unsigned)(npix * p.c_src0 * 4), nbytes1 = (unsigned)(npix * p.c_src1 * 4);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (unsigned)((size_t)9 * nout * kin * 4), WG_FLAGS);

    const FastDiv d_nco(nco), d_tx(p.tiles_x), d_ty(p.tiles_y), d_chunks(nchunks);
    struct Unit { int b, y0, x0, co0, tile; };
    auto unit_of = [&](int k) {
        const int id = blockIdx.x + k * gridDim.x;
        int tile, cot;
        if (xcd_map) {
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
        u.x0 = bx * F3T; u.y0 = by * TR; u.co0 = cot * COT; u.tile = tile;
        return u;
    };

    // staging geometry: A element v = patch pixel (tid >> 2) + (NT / 4) v, float4 tid & 3; B element v = slab row
    // (tid + NT v) / (4 NN) (= tap * 16 + k), float4 (tid + NT v) % (4 NN)
    constexpr int AU = (PATCHX * 4 + NT - 1) / NT, BU = (9 * CK * 4 * NN + NT - 1) / NT;
    static_assert(AU <= 9 && BU <= 9, "nine staging slices per item");
    const int c4 = tid & 3;
    int a_ly[AU], a_lx[AU];
#pragma unroll
    for (int v = 0; v < AU; ++v) {
        const int px = (tid >> 2) + (NT / 4) * v;
        a_ly[v] = px / (F3T + 2);
        a_lx[v] = px - a_ly[v] * (F3T + 2);
        if (px >= PATCHX) a_ly[v] = -4096;
    }
    wg_u32x4 ar[AU], br[BU];
    struct Stage { int b, y0, x0, co0, cc, cs, c0; unsigned oob; __amdgpu_buffer_rsrc_t rs; };
    // the items are staged in order: next_stage() describes the next one (the unit decode runs only when the unit changes;
    // recomputing it per item cost 8 % of an item on the one wave a SIMD has)
    int sg_k = 0, sg_cc = 0;
    unsigned sg_oob = 0u;
    Unit sg_u = unit_of(0);
    auto next_stage = [&]() {
        Stage st;
        st.b = sg_u.b; st.y0 = sg_u.y0; st.x0 = sg_u.x0; st.co0 = sg_u.co0;
        st.cc = sg_cc;
        const bool second = st.cc >= p.c_src0;
        st.cs = second ? p.c_src1 : p.c_src0;
        st.c0 = second ? st.cc - p.c_src0 : st.cc;
        st.oob = sg_oob;                    // past the last item: every offset out of range
        st.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(second ? p.src[1] : p.src[0]), 0, second ? nbytes1 : nbytes0, WG_FLAGS);
        sg_cc += CK;
        if (sg_cc >= kin) {
            sg_cc = 0;
            ++sg_k;
            if (sg_k < my_units) sg_u = unit_of(sg_k);
            else sg_oob = WG_OOB;
        }
        return st;
    };
    // normalise-on-load (MODE 0, ConvArgs::norm): the item in registers carries its thread's scale / shift quad and an
    // inside-the-image bit per element (`cur`); the item being issued builds the next set (`nxt`)
    const bool norm_any = MODE == 0 && (p.norm[0] != nullptr || p.norm[1] != nullptr);          // block-uniform
    float4 n_sc_cur = make_float4(1.f, 1.f, 1.f, 1.f), n_sh_cur = make_float4(0.f, 0.f, 0.f, 0.f), n_sc_nxt = n_sc_cur, n_sh_nxt = n_sh_cur;
    unsigned a_in_cur = 0u, a_in_nxt = 0u;
    auto stage_coef = [&](const Stage& st, float4& sc, float4& sh) {
        sc = make_float4(1.f, 1.f, 1.f, 1.f);
        sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == 0) {
            const float* nt = p.norm[st.cc >= p.c_src0 ? 1 : 0];          // uniform
            if (nt) {
                sc = *reinterpret_cast<const float4*>(nt + st.c0 + 4 * c4);
                sh = *reinterpret_cast<const float4*>(nt + st.cs + st.c0 + 4 * c4);
            }
        }
    };
    auto issue_a = [&](const Stage& st, int v) {
        const int iy = st.y0 - 1 + a_ly[v], ix = st.x0 - 1 + a_lx[v];
        const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const unsigned off = (ok ? (unsigned)(((((st.b * p.H + iy) * p.W + ix) * st.cs) + st.c0 + 4 * c4) * 4) : WG_OOB) | st.oob;
        ar[v] = __builtin_amdgcn_raw_buffer_load_b128(st.rs, off, 0, 0);
        if (MODE == 0) a_in_nxt |= ok ? (1u << v) : 0u;
    };
    auto issue_b = [&](const Stage& st, int v) {
        const int i = tid + NT * v, n4 = i % (4 * NN), r = i / (4 * NN);
        const bool ok = r < 9 * CK;
        const unsigned off = (ok ? (unsigned)(((((r >> 4) * kin + st.cc + (r & 15)) * nout) + st.co0 + 4 * n4) * 4) : WG_OOB) | st.oob;
        br[v] = __builtin_amdgcn_raw_buffer_load_b128(rsw, off, 0, 0);
    };
    auto commit_a = [&](float* buf, int v) {
        const int px = (tid >> 2) + (NT / 4) * v;
        if (norm_any) {
            f32x4 f = __builtin_bit_cast(f32x4, ar[v]);
            const bool in = (a_in_cur >> v) & 1u;
            f[0] = in ? fmaf(f[0], n_sc_cur.x, n_sh_cur.x) : 0.f; f[1] = in ? fmaf(f[1], n_sc_cur.y, n_sh_cur.y) : 0.f;
            f[2] = in ? fmaf(f[2], n_sc_cur.z, n_sh_cur.z) : 0.f; f[3] = in ? fmaf(f[3], n_sc_cur.w, n_sh_cur.w) : 0.f;
            ar[v] = __builtin_bit_cast(wg_u32x4, f);
        }
        *reinterpret_cast<wg_u32x4*>(buf + (px < PATCHX ? px * F3AS : ABUF + BBUF) + 4 * c4) = ar[v];     // idle lanes: dump row
    };
    auto commit_b = [&](float* buf, int v) {
        const int i = tid + NT * v, n4 = i % (4 * NN), r = i / (4 * NN);
        *reinterpret_cast<wg_u32x4*>(buf + (r < 9 * CK ? ABUF + r * BS + 4 * n4 : ABUF + BBUF + 4 * (n4 & 3))) = br[v];
    };

    f32x4 acc[4][NN];
    {
        const Stage s0 = next_stage();
        if (norm_any) stage_coef(s0, n_sc_cur, n_sh_cur);
#pragma unroll
        for (int v = 0; v < AU; ++v) issue_a(s0, v);
#pragma unroll
        for (int v = 0; v < BU; ++v) issue_b(s0, v);
        a_in_cur = a_in_nxt;
        a_in_nxt = 0u;
#pragma unroll
        for (int v = 0; v < AU; ++v) commit_a(lds, v);
#pragma unroll
        for (int v = 0; v < BU; ++v) commit_b(lds, v);
        const Stage s1 = next_stage();
        if (norm_any) stage_coef(s1, n_sc_cur, n_sh_cur);
#pragma unroll
        for (int v = 0; v < AU; ++v) issue_a(s1, v);
#pragma unroll
        for (int v = 0; v < BU; ++v) issue_b(s1, v);
        a_in_cur = a_in_nxt;
        a_in_nxt = 0u;
    }
    lds_barrier();
    int it = 0;          // units outside, K chunks inside (see igb::k_igb_conv3)
#pragma unroll 1
    for (int k = 0; k < my_units; ++k) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < NN; ++j) acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int chunk = 0; chunk < nchunks; ++chunk, ++it) {
        float* buf = lds + (it & 1) * BUF;
        float* other = lds + ((it & 1) ^ 1) * BUF;
        WGSTAMP(it, 0);
        const Stage nx = next_stage();           // item it + 2
        if (norm_any) stage_coef(nx, n_sc_nxt, n_sh_nxt);          // used by the NEXT item's commits: the loads have a whole item to land
        WGSTAMP(it, 1);
        const float* a_lds = buf + ((4 * wave) * (F3T + 2) + m16) * F3AS + q;
        const float* b_lds = buf + ABUF + q * BS + m16;
        // step s = (dx = s / 4, K-step k4 = s % 4): 6 A words (rows 0..5) + 3 x NN B words (dy = 0..2)
        float fa[2][6], fb[2][3][NN];
        auto load_step = [&](int s, float (&a)[6], float (&bw)[3][NN]) {
            const int g = s >> 2, k4 = s & 3;
#pragma unroll
            for (int rr = 0; rr < 6; ++rr) a[rr] = a_lds[(rr * (F3T + 2) + g) * F3AS + 4 * k4];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int j = 0; j < NN; ++j) bw[dy][j] = b_lds[((dy * 3 + g) * CK + 4 * k4) * BS + 16 * j];
        };
        load_step(0, fa[0], fb[0]);
        WGSTAMP(it, 2);
#pragma unroll
        for (int s = 0; s < 12; ++s) {
#ifdef DNNCA_TUNING
            if (s == 6) WGSTAMP(it, 3);
#endif
            if (s + 1 < 12) load_step(s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
            if (s >= 1 && s <= 9) {                     // staging slice s-1 (nine slices: A elements 0..5, B elements 0..8)
                const int v = s - 1;
                if (v < AU) { commit_a(other, v); issue_a(nx, v); }
                if (v < BU) { commit_b(other, v); issue_b(nx, v); }
            }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < NN; ++j)
                        acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s & 1][r + dy], fb[s & 1][dy][j], acc[r][j], 0, 0, 0);
        }
        if (MODE == 0) {          // the registers now hold item it + 2
            n_sc_cur = n_sc_nxt; n_sh_cur = n_sh_nxt;
            a_in_cur = a_in_nxt;
            a_in_nxt = 0u;
        }
        WGSTAMP(it, 4);
        lds_barrier();
        WGSTAMP(it, 5);
    }
        {
            const Unit u = unit_of(k);
            conv3_epilogue<NN, MODE, NW>(p, acc, u.b, u.y0, u.x0, u.co0, u.tile, bn_red);
            WGSTAMP(it - 1, 6);
        }
    }
    if (MODE == 0 && p.bnf.tab) bn_self_fold(p.bnf, gridDim.x, blockIdx.x);          // block-uniform; every block gets here
}

// ------------------------------------------------------------------------------------------------ transposed conv 2x2/2
// Conv2DTranspose(k = s = 2) is four independent 1x1 GEMMs, one per output parity (a, e): out[2i+a][2j+e] = W[a][e] . in[i][j].
// Kernel layout [a][e][Cout][Cin].  M = 128 consecutive input pixels per block (32 per wave).
struct TcArgs {
    const float* in;         // [B,H,W,Cin]
    const float* w;          // [2][2][Cout][Cin]
    const float* bias;
    float* out;              // forward: [B,2H,2W,Cout]
    const float* dout;       // backward: gradient of out
    float* din;              // data gradient destination
    const float* mask;       // multiply din by act'(mask) (nullptr: none)
    float* dw;               // weight gradient (atomic accumulation)
    float* dbias;
    int acc;
    int cin, cout;
    int H, W;                // input height / width
    int npix;                // B*H*W
    int psplit;
    float alpha;
    BnSelfFold bnf;          // forward (bf16 kernel): batch statistics of the output for the BatchNorm behind it, self-folding (bn_dev.h)
    int out_half;            // forward (bf16 kernel): out is stored as bf16 (the input of a BatchNorm, ig_plan_half)
    int din_half;            // data gradient (bf16 kernel): din is stored as bf16 (the gradient arriving at a BatchNorm)
};

__device__ __forceinline__ size_t tc_outpix(int p, int a, int e, int H, int W) {
    const int j = p % W, bi = p / W;          // bi = b*H + i
    return ((size_t)bi * 2 + a) * (2 * W) + 2 * j + e;
}

template <int NN>
__global__ __launch_bounds__(256) void k_ig_tconv_fwd(TcArgs p) {
    constexpr int NT = 16 * NN, BSTR = NT + 16;
    __shared__ float a_lds[128 * CKP];
    __shared__ float b_lds[CK * BSTR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m16 = lane & 15, q = lane >> 4;
    const int p0 = blockIdx.x * 128, co0 = blockIdx.y * NT, ae = blockIdx.z, a = ae >> 1, e = ae & 1;
    f32x4 acc[2][NN];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < NN; ++j) acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int cc = 0; cc < p.cin; cc += CK) {
        lds_barrier();
        for (int i = tid; i < 128 * 4; i += 256) {
            const int px = i >> 2, c4 = i & 3;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p0 + px < p.npix) v = *reinterpret_cast<const float4*>(p.in + (size_t)(p0 + px) * p.cin + cc + 4 * c4);
            *reinterpret_cast<float4*>(a_lds + px * CKP + 4 * c4) = v;
        }
        for (int i = tid; i < NT * 4; i += 256) {       // B[k = ci][n = co] = W[a][e][co][ci]: transposed while staging
            const int n = i >> 2, k4 = i & 3;
            const float4 v = *reinterpret_cast<const float4*>(p.w + ((size_t)ae * p.cout + co0 + n) * p.cin + cc + 4 * k4);
            b_lds[(4 * k4 + 0) * BSTR + n] = v.x;
            b_lds[(4 * k4 + 1) * BSTR + n] = v.y;
            b_lds[(4 * k4 + 2) * BSTR + n] = v.z;
            b_lds[(4 * k4 + 3) * BSTR + n] = v.w;
        }
        lds_barrier();
#pragma unroll
        for (int k4 = 0; k4 < CK / 4; ++k4) {
            float bv[NN];
#pragma unroll
            for (int j = 0; j < NN; ++j) bv[j] = b_lds[(4 * k4 + q) * BSTR + 16 * j + m16];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float av = a_lds[(32 * wave + 16 * r + m16) * CKP + 4 * k4 + q];
#pragma unroll
                for (int j = 0; j < NN; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[j], acc[r][j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = p0 + 32 * wave + 16 * r + 4 * q + i;
            if (px >= p.npix) continue;
            float* op = p.out + tc_outpix(px, a, e, p.H, p.W) * p.cout + co0 + m16;
#pragma unroll
            for (int j = 0; j < NN; ++j) op[16 * j] = acc[r][j][i] + p.bias[co0 + 16 * j + m16];
        }
}

// forward, second generation (round 4): one block = 128 input pixels x 16 NCO output channels x ALL FOUR parities (the first one ran a
// block per parity and re-read the pixels four times), two LDS buffers with register prefetch (chunk c + 2's loads fly during chunk
// c's MFMAs: one barrier per chunk), MFMAs channel-major (rows = output channels: A = kernel rows; columns = pixels: B) so that a
// lane holds four consecutive channels of a pixel -- 16-byte stores -- and the batch statistics of the BatchNorm behind the layer ride
// along (DPP row sums over the 16 pixels of a tile, the self-folding table of bn_dev.h): no bn_stats launch.  Same fp32 MFMA chain per
// output value as before (K ascending).
template <int NCO>
__global__ __launch_bounds__(256, 2) void k_ig_tconv_fwd2(TcArgs p) {
    constexpr int COT = 16 * NCO, XB = 128 * CKP, WB = 4 * COT * CKP, BUF = XB + WB;          // floats per buffer
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];
    __shared__ float red[4 * 2 * COT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m16 = lane & 15, q = lane >> 4;
    const int p0 = blockIdx.x * 128, co0 = blockIdx.y * COT;
    const int nchunks = p.cin / CK;
    f32x4 acc[4][NCO][2];
#pragma unroll
    for (int ae = 0; ae < 4; ++ae)
#pragma unroll
        for (int mt = 0; mt < NCO; ++mt)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) acc[ae][mt][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // staging: X chunk = 128 pixels x 16 channels = 512 float4 (two per thread); W chunk = 4 COT rows (ae, co) x 16 channels = 16 COT
    // float4 (NCO per thread)
    f32x4 xr[2], wr[NCO];
    auto issue = [&](int c) {
        const int cc = c * CK;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + 256 * u, px = i >> 2, c4 = i & 3;
            xr[u] = p0 + px < p.npix ? *reinterpret_cast<const f32x4*>(p.in + (size_t)(p0 + px) * p.cin + cc + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < NCO; ++u) {
            const int i = tid + 256 * u, r = i >> 2, k4 = i & 3, ae = r / COT, n = r - ae * COT;
            wr[u] = *reinterpret_cast<const f32x4*>(p.w + ((size_t)ae * p.cout + co0 + n) * p.cin + cc + 4 * k4);
        }
    };
    auto commit = [&](float* buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + 256 * u, px = i >> 2, c4 = i & 3;
            *reinterpret_cast<f32x4*>(buf + px * CKP + 4 * c4) = xr[u];
        }
#pragma unroll
        for (int u = 0; u < NCO; ++u) {
            const int i = tid + 256 * u, r = i >> 2, k4 = i & 3;
            *reinterpret_cast<f32x4*>(buf + XB + r * CKP + 4 * k4) = wr[u];
        }
    };
    issue(0);
    commit(lds);
    if (nchunks > 1) issue(1);
    lds_barrier();
#pragma unroll 1
    for (int c = 0; c < nchunks; ++c) {
        const float* buf = lds + (c & 1) * BUF;
#pragma unroll
        for (int ks = 0; ks < CK / 4; ++ks) {
            float xf[2];
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) xf[pt] = buf[(32 * wave + 16 * pt + m16) * CKP + 4 * ks + q];
#pragma unroll
            for (int ae = 0; ae < 4; ++ae)
#pragma unroll
                for (int mt = 0; mt < NCO; ++mt) {
                    const float wf = buf[XB + ((ae * NCO + mt) * 16 + m16) * CKP + 4 * ks + q];
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt) acc[ae][mt][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf, xf[pt], acc[ae][mt][pt], 0, 0, 0);
                }
        }
        if (c + 1 < nchunks) {
            commit(lds + ((c + 1) & 1) * BUF);
            if (c + 2 < nchunks) issue(c + 2);
        }
        lds_barrier();
    }
    // epilogue: lane (m16, q) holds channels co0 + 16 mt + 4 q .. + 3 of input pixel p0 + 32 wave + 16 pt + m16, all four parities
    f32x4 bias[NCO], bs[NCO], bq[NCO];
#pragma unroll
    for (int mt = 0; mt < NCO; ++mt) {
        bias[mt] = *reinterpret_cast<const f32x4*>(p.bias + co0 + 16 * mt + 4 * q);
        bs[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        bq[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int px = p0 + 32 * wave + 16 * pt + m16;
        if (px >= p.npix) continue;
#pragma unroll
        for (int ae = 0; ae < 4; ++ae) {
            float* op = p.out + tc_outpix(px, ae >> 1, ae & 1, p.H, p.W) * p.cout + co0 + 4 * q;
#pragma unroll
            for (int mt = 0; mt < NCO; ++mt) {
                const f32x4 v = acc[ae][mt][pt] + bias[mt];
                *reinterpret_cast<f32x4*>(op + 16 * mt) = v;
                bs[mt] += v;
#pragma unroll
                for (int i = 0; i < 4; ++i) bq[mt][i] = fmaf(v[i], v[i], bq[mt][i]);
            }
        }
    }
    if (p.bnf.tab) {        // batch statistics for the BatchNorm behind the transposed conv: a bucket row per pixel block
#pragma unroll
        for (int mt = 0; mt < NCO; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float s1 = row16_sum(bs[mt][i]), s2 = row16_sum(bq[mt][i]);
                if (m16 == 0) { red[wave * (2 * COT) + 16 * mt + 4 * q + i] = s1; red[wave * (2 * COT) + COT + 16 * mt + 4 * q + i] = s2; }
            }
        __syncthreads();
        if (tid < 2 * COT) {
            const float sum = (red[tid] + red[2 * COT + tid]) + (red[4 * COT + tid] + red[6 * COT + tid]);
            const int half = tid >= COT, c = half ? tid - COT : tid;
            atomicAdd(bn_bucket(p.bnf, (int)blockIdx.x) + half * p.cout + co0 + c, (double)sum);
        }
        bn_self_fold(p.bnf, gridDim.x * gridDim.y, blockIdx.y * gridDim.x + blockIdx.x);
    }
}

// data gradient: din[p][ci] = sum_{a,e,co} dout[out(p,a,e)][co] * W[a][e][co][ci]   (N = ci tile, K = 4 x Cout).
// Round 4: two LDS buffers with register prefetch (chunk c + 2's loads fly during chunk c's MFMAs: one barrier per chunk; the first
// version loaded, waited, multiplied, chunk after chunk) and MFMAs channel-major (rows = input channels, columns = pixels): a lane
// holds four consecutive channels of a pixel -- 16-byte loads / stores in the epilogue.  Same fp32 MFMA chain per value (K ascending).
template <int NN>
__global__ __launch_bounds__(256) void k_ig_tconv_dgrad(TcArgs p) {
    constexpr int NT = 16 * NN, BSTR = NT + 16;
    constexpr int AB = 128 * CKP, BB = CK * BSTR, BUF = AB + BB;
    constexpr int BW = CK * (NT / 4) / 256 > 0 ? CK * (NT / 4) / 256 : 1;          // kernel float4 words per thread (NT = 64: one)
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m16 = lane & 15, q = lane >> 4;
    const int p0 = blockIdx.x * 128, n0 = blockIdx.y * NT;
    const int nchunks = 4 * p.cout / CK;
    f32x4 acc[2][NN];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < NN; ++j) acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ar[2], br[BW];
    auto issue = [&](int c) {
        const int kc = c * CK, ae = kc / p.cout, cc = kc - ae * p.cout, a = ae >> 1, e = ae & 1;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + 256 * u, px = i >> 2, c4 = i & 3;
            ar[u] = p0 + px < p.npix ? *reinterpret_cast<const f32x4*>(p.dout + tc_outpix(p0 + px, a, e, p.H, p.W) * p.cout + cc + 4 * c4)
                                     : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < BW; ++u) {
            const int i = tid + 256 * u, n4 = i % (NT / 4), k = i / (NT / 4);
            if (k < CK) br[u] = *reinterpret_cast<const f32x4*>(p.w + ((size_t)ae * p.cout + cc + k) * p.cin + n0 + 4 * n4);
        }
    };
    auto commit = [&](float* buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + 256 * u, px = i >> 2, c4 = i & 3;
            *reinterpret_cast<f32x4*>(buf + px * CKP + 4 * c4) = ar[u];
        }
#pragma unroll
        for (int u = 0; u < BW; ++u) {
            const int i = tid + 256 * u, n4 = i % (NT / 4), k = i / (NT / 4);
            if (k < CK) *reinterpret_cast<f32x4*>(buf + AB + k * BSTR + 4 * n4) = br[u];
        }
    };
    issue(0);
    commit(lds);
    if (nchunks > 1) issue(1);
    lds_barrier();
#pragma unroll 1
    for (int c = 0; c < nchunks; ++c) {
        const float* buf = lds + (c & 1) * BUF;
#pragma unroll
        for (int k4 = 0; k4 < CK / 4; ++k4) {
            float bv[NN];
#pragma unroll
            for (int j = 0; j < NN; ++j) bv[j] = buf[AB + (4 * k4 + q) * BSTR + 16 * j + m16];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float av = buf[(32 * wave + 16 * r + m16) * CKP + 4 * k4 + q];
#pragma unroll
                for (int j = 0; j < NN; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[j], av, acc[r][j], 0, 0, 0);      // rows = channels, columns = pixels
            }
        }
        if (c + 1 < nchunks) {
            commit(lds + ((c + 1) & 1) * BUF);
            if (c + 2 < nchunks) issue(c + 2);
        }
        lds_barrier();
    }
    // lane (m16, q): channels n0 + 16 j + 4 q .. + 3 of pixel p0 + 32 wave + 16 r + m16
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int px = p0 + 32 * wave + 16 * r + m16;
        if (px >= p.npix) continue;
#pragma unroll
        for (int j = 0; j < NN; ++j) {
            const size_t o = (size_t)px * p.cin + n0 + 16 * j + 4 * q;
            f32x4 v = acc[r][j];
            if (p.acc) v += *reinterpret_cast<const f32x4*>(p.din + o);
            if (p.mask) {
                const f32x4 mk = *reinterpret_cast<const f32x4*>(p.mask + o);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] *= mk[i] > 0.f ? 1.0f : p.alpha;
            }
            *reinterpret_cast<f32x4*>(p.din + o) = v;
        }
    }
}

// weight gradient: dW[a][e][co][ci] = sum_p dout[out(p,a,e)][co] * in[p][ci]   (M = 16 co, N = ci tile, K = pixels)
// grid: x = pixel split, y = co chunk (16), z = 4 * (Cin / NT) + ...
template <int NN>
__global__ __launch_bounds__(256) void k_ig_tconv_wgrad(TcArgs p) {
    constexpr int NT = 16 * NN, XSTR = NT + 4;
    __shared__ float g_lds[128 * CKP];
    __shared__ float x_lds[128 * XSTR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m16 = lane & 15, q = lane >> 4;
    const int co0 = blockIdx.y * CK;
    const int ntn = p.cin / NT, ae = blockIdx.z / ntn, n0 = (blockIdx.z % ntn) * NT, a = ae >> 1, e = ae & 1;
    const bool do_bias = p.dbias && n0 == 0;
    f32x4 acc[NN], accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NN; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntiles = (p.npix + 127) / 128;
#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += p.psplit) {
        const int p0 = tile * 128;
        lds_barrier();
        for (int i = tid; i < 128 * 4; i += 256) {
            const int px = i >> 2, c4 = i & 3;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p0 + px < p.npix) v = *reinterpret_cast<const float4*>(p.dout + tc_outpix(p0 + px, a, e, p.H, p.W) * p.cout + co0 + 4 * c4);
            *reinterpret_cast<float4*>(g_lds + px * CKP + 4 * c4) = v;
        }
        for (int i = tid; i < 128 * (NT / 4); i += 256) {
            const int n4 = i % (NT / 4), px = i / (NT / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p0 + px < p.npix) v = *reinterpret_cast<const float4*>(p.in + (size_t)(p0 + px) * p.cin + n0 + 4 * n4);
            *reinterpret_cast<float4*>(x_lds + px * XSTR + 4 * n4) = v;
        }
        lds_barrier();
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int px = 32 * wave + 4 * ks + q;
            const float av = g_lds[px * CKP + m16];
#pragma unroll
            for (int j = 0; j < NN; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, x_lds[px * XSTR + 16 * j + m16], acc[j], 0, 0, 0);
            if (do_bias) accb = __builtin_amdgcn_mfma_f32_16x16x4f32(av, 1.0f, accb, 0, 0, 0);
        }
    }
    // D[co = 4q + i][ci = 16j + m16]
#pragma unroll
    for (int j = 0; j < NN; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            atomicAdd(p.dw + ((size_t)ae * p.cout + co0 + 4 * q + i) * p.cin + n0 + 16 * j + m16, acc[j][i]);
    if (do_bias && m16 == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(p.dbias + co0 + 4 * q + i, accb[i]);
    }
}

// weight gradient of the transposed conv, second generation (fp32): dW[ae][co][ci] = sum_p dout[out(p, ae)][co] * in[p][ci].
// Block tile = 16*MW output channels x 16*NN input channels x 4 parities, one persistent block per CU; the four waves split
// the output channels MW ways and the pixels 4/MW ways; `in` and the four parity slices of `dout` are read once per
// channel-tile pair (the first generation re-read `in` per parity and per 16-channel chunk); next tile prefetched into
// registers by raw buffer loads; in-block reduction over the pixel-split waves before the atomics.
template <int MW, int NN>
__global__ __launch_bounds__(256, 1) void k_ig_tconv_wgrad2(TcArgs p) {
    constexpr int COT = 16 * MW, CIT = 16 * NN, WK = 4 / MW;
    constexpr int TM = (4 / MW) < (4 / NN) ? (4 / MW) : (4 / NN);
    constexpr int NPX = 64 * TM;                                  // input pixels per tile (K)
    constexpr int GS = COT + (COT == 16 ? 0 : 16), XS = CIT + (CIT == 16 ? 0 : 16);
    constexpr int GQ = COT / 4, XQ = CIT / 4;
    constexpr int GU = 4 * NPX * GQ / 256, XU = NPX * XQ / 256;
    constexpr int NKS = NPX / 4 / WK;
    constexpr int GF = 4 * NPX * GS, XF = NPX * XS;
    constexpr int RF = MW * (4 * NN + 1) * 256;
    static_assert(NPX * XQ % 256 == 0 && NKS % 2 == 0, "tile geometry");
    __shared__ __attribute__((aligned(16))) float smem[GF + XF > RF ? GF + XF : RF];
    float* g_lds = smem;              // [parity][pixel][GS]
    float* x_lds = smem + GF;         // [pixel][XS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m16 = lane & 15, q = lane >> 4;
    const int wm = wave % MW, wk = wave / MW;
    const int co0 = blockIdx.y * COT, n0 = blockIdx.z * CIT;
    const bool do_bias = p.dbias && blockIdx.z == 0;
    const int ntiles = (p.npix + NPX - 1) / NPX;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (unsigned)((size_t)p.npix * p.cin * 4), WG_FLAGS);
    const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc((void*)p.dout, 0, (unsigned)((size_t)p.npix * 4 * p.cout * 4), WG_FLAGS);

    f32x4 acc[4][NN], accb[4];
#pragma unroll
    for (int ae = 0; ae < 4; ++ae) {
        accb[ae] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NN; ++j) acc[ae][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    wg_u32x4 xr[XU], gr[GU];
    auto issue = [&](int tile) {
        const unsigned oob = tile < ntiles ? 0u : WG_OOB;
        const int p0 = (tile < ntiles ? tile : 0) * NPX;
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int i = tid + 256 * u, px = i / XQ, c4 = i % XQ;
            const unsigned off = (p0 + px < p.npix ? (unsigned)((((p0 + px) * p.cin) + n0 + 4 * c4) * 4) : WG_OOB) | oob;
            xr[u] = __builtin_amdgcn_raw_buffer_load_b128(rsx, off, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int i = tid + 256 * u, c4 = i % GQ, r = i / GQ, px = r % NPX, ae = r / NPX;
            const int pp = p0 + px, jx = pp % p.W, bi = pp / p.W;
            const unsigned opix = (unsigned)((bi * 2 + (ae >> 1)) * (2 * p.W) + 2 * jx + (ae & 1));
            const unsigned off = (pp < p.npix ? (opix * (unsigned)p.cout + co0 + 4 * c4) * 4u : WG_OOB) | oob;
            gr[u] = __builtin_amdgcn_raw_buffer_load_b128(rsg, off, 0, 0);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int i = tid + 256 * u, px = i / XQ, c4 = i % XQ;
            *reinterpret_cast<wg_u32x4*>(x_lds + px * XS + 4 * c4) = xr[u];
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int i = tid + 256 * u, c4 = i % GQ, r = i / GQ;       // r = parity * NPX + pixel
            *reinterpret_cast<wg_u32x4*>(g_lds + r * GS + 4 * c4) = gr[u];
        }
    };
    auto load_step = [&](int s, float (&av)[4], float (&bv)[NN]) {
        const int px = 4 * (wk + WK * s) + q;
#pragma unroll
        for (int j = 0; j < NN; ++j) bv[j] = x_lds[px * XS + 16 * j + m16];
#pragma unroll
        for (int ae = 0; ae < 4; ++ae) av[ae] = g_lds[(ae * NPX + px) * GS + 16 * wm + m16];
    };
    auto mfma_step = [&](const float (&av)[4], const float (&bv)[NN]) {
#pragma unroll
        for (int ae = 0; ae < 4; ++ae) {
#pragma unroll
            for (int j = 0; j < NN; ++j) acc[ae][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ae], bv[j], acc[ae][j], 0, 0, 0);
            if (do_bias) accb[ae] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ae], 1.0f, accb[ae], 0, 0, 0);
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) issue(tile);
#pragma unroll 1
    for (; tile < ntiles; tile += p.psplit) {
        lds_barrier();
        commit();
        issue(tile + p.psplit);
        lds_barrier();
        float a0[4], b0[NN], a1[4], b1[NN];
        load_step(0, a0, b0);
#pragma unroll 1
        for (int s = 0; s < NKS; s += 2) {
            load_step(s + 1, a1, b1);
            mfma_step(a0, b0);
            load_step(s + 2 < NKS ? s + 2 : 0, a0, b0);
            mfma_step(a1, b1);
        }
    }
    // bias gradient of a row co = sum over the four parities (all 16 columns of the all-ones-B product are equal)
    float bsum[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bsum[i] = (accb[0][i] + accb[1][i]) + (accb[2][i] + accb[3][i]);
    if (WK > 1) {       // the pixel-split wave sets add up inside the block first (one set at a time through LDS)
        float* red = smem + wm * ((4 * NN + 1) * 256);
        for (int r = 1; r < WK; ++r) {
            __syncthreads();
            if (wk == r) {
#pragma unroll
                for (int ae = 0; ae < 4; ++ae)
#pragma unroll
                    for (int j = 0; j < NN; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) red[((ae * NN + j) * 4 + i) * 64 + lane] = acc[ae][j][i];
#pragma unroll
                for (int i = 0; i < 4; ++i) red[4 * NN * 256 + i * 64 + lane] = bsum[i];
            }
            __syncthreads();
            if (wk == 0) {
#pragma unroll
                for (int ae = 0; ae < 4; ++ae)
#pragma unroll
                    for (int j = 0; j < NN; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[ae][j][i] += red[((ae * NN + j) * 4 + i) * 64 + lane];
#pragma unroll
                for (int i = 0; i < 4; ++i) bsum[i] += red[4 * NN * 256 + i * 64 + lane];
            }
        }
    }
    if (wk != 0) return;
    // D[co = 16 wm + 4q + i][ci = 16j + m16]
#pragma unroll
    for (int ae = 0; ae < 4; ++ae)
#pragma unroll
        for (int j = 0; j < NN; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                atomicAdd(p.dw + ((size_t)ae * p.cout + co0 + 16 * wm + 4 * q + i) * p.cin + n0 + 16 * j + m16, acc[ae][j][i]);
    if (do_bias && m16 == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(p.dbias + co0 + 16 * wm + 4 * q + i, bsum[i]);
    }
}

}  // namespace ig

// ================================================================================================ bf16 variants
// dtype = DNNCA_BF16 (configs/unet_big.yaml as benchmarked: bf16 contraction, fp32 master weights / activations /
// accumulation): operands are rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) while they are staged into LDS
// and the contraction runs on v_mfma_f32_16x16x32_bf16 (16x the f32 MFMA rate).  Tensors in HBM stay fp32.
namespace igb {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // register-resident 16-byte staging word (HIP's uint4 struct arrays end up in scratch)

constexpr int TY = ig::TY, TX = ig::TX, PATCH = ig::PATCH;
constexpr int CK = 32;                  // K elements per staged chunk = one 16x16x32 MFMA
constexpr int RS = 40;                  // LDS row stride in bf16 (80 B): conflict-free 16-byte fragment reads

using ig::lds_barrier;
using ig::ConvArgs;

// per-step bf16 copies of the conv kernels: wf[t][co][ci] = w[t][ci][co] (forward: K = ci contiguous),
//                                            wd[t][ci][co] = w[8-t][ci][co] (data gradient: K = co contiguous)
struct PrepDesc {
    int w_off, cin, cout;
    int kind;                // 0: 3x3 conv kernel [9][ci][co]; 1: transposed-conv kernel [4][co][ci]
};
__global__ void k_igb_prep(const PrepDesc* __restrict__ descs, const float* __restrict__ params, bf16_t* __restrict__ wf,
                           bf16_t* __restrict__ wd) {
    const PrepDesc d = descs[blockIdx.y];
    if (d.kind == 1) {          // Conv2DTranspose [ae][co][ci]: forward wants K = ci contiguous (as stored), dgrad K = co
        const int n = 4 * d.cin * d.cout;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
            const int ci = i % d.cin, co = (i / d.cin) % d.cout, ae = i / (d.cin * d.cout);
            const float v = params[d.w_off + i];
            wf[d.w_off + i] = (bf16_t)v;
            wd[d.w_off + ((size_t)ae * d.cin + ci) * d.cout + co] = (bf16_t)v;
        }
        return;
    }
    // 3x3 kernels [t][ci][co] (channel counts are multiples of 32 under bf16): the data-gradient copy keeps co contiguous, the
    // forward copy wants ci contiguous -- 32 x 32 tiles go through LDS so that both copies are written in 64-byte runs
    // (element-wise scattered 2-byte stores made this the slowest bookkeeping kernel of the unet_big step: 109 us)
    __shared__ float tile[32][33];
    const int ntx = d.cout / 32, nty = d.cin / 32, ntiles = 9 * nty * ntx;
    const int tx32 = threadIdx.x & 31, ty8 = threadIdx.x >> 5;          // 256 threads: 8 rows of 32
    for (int id = blockIdx.x; id < ntiles; id += gridDim.x) {
        const int t = id / (nty * ntx), r = id - t * (nty * ntx), ty = r / ntx, tx = r - ty * ntx;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ci = ty * 32 + ty8 + 8 * k, co = tx * 32 + tx32;
            const float v = params[d.w_off + ((size_t)t * d.cin + ci) * d.cout + co];
            tile[ty8 + 8 * k][tx32] = v;
            wd[d.w_off + ((size_t)(8 - t) * d.cin + ci) * d.cout + co] = (bf16_t)v;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int co = tx * 32 + ty8 + 8 * k, ci = ty * 32 + tx32;
            wf[d.w_off + ((size_t)t * d.cout + co) * d.cin + ci] = (bf16_t)tile[tx32][ty8 + 8 * k];
        }
        __syncthreads();
    }
}

// MODE 0 forward, MODE 1 data gradient.  w16: [9][N channels][K channels] bf16 (K contiguous)
template <int NN, int MODE>
__global__ __launch_bounds__(256) void k_igb_conv(ConvArgs p, const bf16_t* __restrict__ w16) {
    constexpr int NT = 16 * NN;
    __shared__ bf16_t a_lds[PATCH * RS];
    __shared__ bf16_t b_lds[9 * NT * RS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m16 = lane & 15, q = lane >> 4;
    const int kin = p.c_src0 + p.c_src1, nout = p.n_dst0 + p.n_dst1;
    const int tile = blockIdx.x, bx = tile % p.tiles_x, by = (tile / p.tiles_x) % p.tiles_y, b = tile / (p.tiles_x * p.tiles_y);
    const int x0 = bx * TX, y0 = by * TY;
    const int co0 = blockIdx.y * NT;

    f32x4 acc[2][NN];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < NN; ++j) acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int cc = 0; cc < kin; cc += CK) {
        const bool second = cc >= p.c_src0;
        const float* src = second ? p.src[1] : p.src[0];
        const int cs = second ? p.c_src1 : p.c_src0, c0 = second ? cc - p.c_src0 : cc;
        lds_barrier();
        for (int i = tid; i < PATCH * (CK / 4); i += 256) {
            const int px = i >> 3, c4 = i & 7;
            const int ly = px / (TX + 2), lx = px - ly * (TX + 2);
            const int iy = y0 - 1 + ly, ix = x0 - 1 + lx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                v = *reinterpret_cast<const float4*>(src + (((size_t)b * p.H + iy) * p.W + ix) * cs + c0 + 4 * c4);
            bf16x4 h;
            h[0] = (bf16_t)v.x; h[1] = (bf16_t)v.y; h[2] = (bf16_t)v.z; h[3] = (bf16_t)v.w;
            *reinterpret_cast<bf16x4*>(a_lds + px * RS + 4 * c4) = h;
        }
        for (int i = tid; i < 9 * NT * 4; i += 256) {
            const int part = i & 3, r = i >> 2;            // r = tap * NT + n
            const int tap = r / NT, n = r - tap * NT;
            const uint4 v = *reinterpret_cast<const uint4*>(w16 + ((size_t)tap * nout + co0 + n) * kin + cc + 8 * part);
            *reinterpret_cast<uint4*>(b_lds + r * RS + 8 * part) = v;
        }
        lds_barrier();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
            bf16x8 bv[NN];
#pragma unroll
            for (int j = 0; j < NN; ++j) bv[j] = *reinterpret_cast<const bf16x8*>(b_lds + (tap * NT + 16 * j + m16) * RS + 8 * q);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(a_lds + ((2 * wave + r + dy) * (TX + 2) + m16 + dx) * RS + 8 * q);
#pragma unroll
                for (int j = 0; j < NN; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv[j], acc[r][j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int y = y0 + 2 * wave + r;
        if (y >= p.H) continue;
#pragma unroll
        for (int j = 0; j < NN; ++j) {
            const int co = co0 + 16 * j + m16;
            const int which = co >= p.n_dst0;
            const int cw = which ? p.n_dst1 : p.n_dst0, cl = which ? co - p.n_dst0 : co;
            const float bias = (MODE == 0 && p.bias) ? p.bias[co] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int x = x0 + 4 * q + i;
                if (x >= p.W) continue;
                const size_t o = (((size_t)b * p.H + y) * p.W + x) * cw + cl;
                float v = acc[r][j][i];
                if (MODE == 0) {
                    v += bias;
                    v = p.alpha < 0.f ? v : (v > 0.f ? v : p.alpha * v);
                } else {
                    if (p.acc[which]) v += p.dst[which][o];
                    if (p.mask[which]) v *= p.mask[which][o] > 0.f ? 1.0f : p.alpha;
                }
                p.dst[which][o] = v;
            }
        }
    }
}

// weight gradient, bf16: M = 16 input channels, N = 16*NN output channels, K = 32 pixels (2 tile rows) per MFMA.
// Both operands need K = pixels contiguous, i.e. the transposes of the NHWC tiles: they are written transposed into LDS
// while staging; the input patch is stored three times, shifted by dx = 0,1,2, so that every fragment read is 16-B aligned.
template <int NN>
__global__ __launch_bounds__(256) void k_igb_wgrad(ig::WgArgs p) {
    constexpr int NT = 16 * NN;
    constexpr int XROW = 16, XCI = (TY + 2) * XROW + 8;      // bf16 strides of xT[dx][ci][row][col]
    constexpr int GCO = TY * 16 + 8;                         // bf16 stride of gT[co][row][col]
    __shared__ bf16_t xT[3 * 16 * XCI];
    __shared__ bf16_t gT[NT * GCO];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m16 = lane & 15, q = lane >> 4;
    const int c0 = blockIdx.y * 16, co0 = blockIdx.z * NT;
    const bool do_bias = p.dbias && blockIdx.y == 0;
    const int ntiles = p.tiles_x * p.tiles_y * p.B;
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (bf16_t)1.0f;

    f32x4 acc[9][NN], accb[NN];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < NN; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NN; ++j) accb[j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += p.psplit) {
        const int bx = tile % p.tiles_x, by = (tile / p.tiles_x) % p.tiles_y, b = tile / (p.tiles_x * p.tiles_y);
        const int x0 = bx * TX, y0 = by * TY;
        lds_barrier();
        for (int i = tid; i < PATCH * 4; i += 256) {
            const int px = i >> 2, c4 = i & 3;
            const int ly = px / (TX + 2), lx = px - ly * (TX + 2);
            const int iy = y0 - 1 + ly, ix = x0 - 1 + lx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                v = *reinterpret_cast<const float4*>(p.x + (((size_t)b * p.H + iy) * p.W + ix) * p.cs + c0 + 4 * c4);
            const bf16_t h[4] = {(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int col = lx - dx;
                if (col < 0 || col >= TX) continue;
#pragma unroll
                for (int k = 0; k < 4; ++k) xT[(dx * 16 + 4 * c4 + k) * XCI + ly * XROW + col] = h[k];
            }
        }
        for (int i = tid; i < TY * TX * (NT / 4); i += 256) {
            const int n4 = i % (NT / 4), px = i / (NT / 4);
            const int ly = px / TX, lx = px - ly * TX;
            const int iy = y0 + ly, ix = x0 + lx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy < p.H && ix < p.W)
                v = *reinterpret_cast<const float4*>(p.dz + (((size_t)b * p.H + iy) * p.W + ix) * p.cout + co0 + 4 * n4);
            gT[(4 * n4 + 0) * GCO + ly * 16 + lx] = (bf16_t)v.x;
            gT[(4 * n4 + 1) * GCO + ly * 16 + lx] = (bf16_t)v.y;
            gT[(4 * n4 + 2) * GCO + ly * 16 + lx] = (bf16_t)v.z;
            gT[(4 * n4 + 3) * GCO + ly * 16 + lx] = (bf16_t)v.w;
        }
        lds_barrier();
        // wave w: K-step = tile rows 2w, 2w+1 (32 pixels); fragment element j of lane quarter q = pixel (row 2w + q/2, col 8(q&1) + j)
        const int row = 2 * wave + (q >> 1), col = 8 * (q & 1);
        bf16x8 bv[NN];
#pragma unroll
        for (int j = 0; j < NN; ++j) bv[j] = *reinterpret_cast<const bf16x8*>(gT + (16 * j + m16) * GCO + row * 16 + col);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const bf16x8 av = *reinterpret_cast<const bf16x8*>(xT + ((t % 3) * 16 + m16) * XCI + (row + t / 3) * XROW + col);
#pragma unroll
            for (int j = 0; j < NN; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv[j], acc[t][j], 0, 0, 0);
        }
        if (do_bias) {
#pragma unroll
            for (int j = 0; j < NN; ++j) accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bv[j], accb[j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < NN; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                atomicAdd(p.dw + ((size_t)t * p.cin_total + p.ci_off + c0 + 4 * q + i) * p.cout + co0 + 16 * j + m16, acc[t][j][i]);
    if (do_bias && q == 0) {
#pragma unroll
        for (int j = 0; j < NN; ++j) atomicAdd(p.dbias + co0 + 16 * j + m16, accb[j][0]);
    }
}

// phase stamps of k_igb_conv3 (tuning builds only: DNNCA_TUNING=1 python -m dnncancerannotator_amd.build; tools/ig_stamps.py)
#ifdef DNNCA_TUNING
__device__ unsigned long long g_ig_stamps[64 * 8];
__device__ __forceinline__ unsigned long long ig_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define IGSTAMP(item, ph) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (item) < 64) g_ig_stamps[(item) * 8 + (ph)] = ig_now(); } while (0)
#else
#define IGSTAMP(item, ph) do { } while (0)
#endif

constexpr int T2 = 16;                              // tile edge of the persistent forward / data-gradient kernel
constexpr int PATCH2 = (T2 + 2) * (T2 + 2);         // 324 staged pixels

// forward / data gradient, bf16, persistent and software-pipelined (one block per CU, the whole register file):
//   work item = (pixel tile, channel tile, K chunk); a block walks its items in order, accumulators restart at chunk 0;
//   two LDS buffers: while the MFMAs of item i read buffer i&1, the registers prefetched for item i+1 are converted and
//   written into the other buffer and the global loads of item i+2 are issued, both in eight slices that sit between the
//   nine tap steps -- one barrier per item, and the prefetch runs across tile boundaries, so the first chunk of the next
//   tile is already in LDS when a tile's epilogue starts;
//   fragments are double-buffered in registers: the LDS reads of tap step s+1 are issued before the 16 MFMAs of step s;
//   taps are walked dx-major so that the six A rows of a dx serve its three dy taps (18 instead of 24 reads per 48 MFMAs);
//   global loads are raw buffer loads: out-of-image patch pixels get an out-of-range offset and come back as zeros, no
//   branches; the epilogue stores straight from the accumulators (ig::conv3_epilogue) and needs no LDS image and no barrier.
constexpr int BUF3 = (PATCH2 + 9 * 64) * RS + 64;   // bf16 elements per LDS buffer: A patch + 9-tap weight slab (72,000 B) + a dump row for the idle lanes of the last A element
constexpr unsigned BUF_FLAGS = 0x00020000u;         // raw buffer descriptor word 3 (gfx9 family)
constexpr unsigned OOB = 0x80000000u;               // beyond every tensor here: the buffer load returns zeros

// Epilogue of k_igb_conv3 (round 4), CHANNEL-major: the MFMAs run with the operands swapped (rows = output channels, columns =
// pixels), so lane (m16, q) of wave w holds acc[r][j][i] = channel 16 j + 4 q + i of pixel (row 4 w + r, column m16).  Same contract
// as ig::conv3_epilogue (bias + activation + batch statistics of the stored values / accumulate, mask, two destinations).
// bf16-STORED destinations (the normal case under dtype bf16) leave through a wave-private LDS tile: per tile row and 8-pixel half the
// lanes of that half write their 64 channels (8-byte writes), then every lane reads 16 bytes and stores them -- eight lanes per
// 128-byte run of a pixel's 64 channels, 1 KB per store instruction.  ig::conv3_epilogue stored one 2-byte value per lane and
// instruction (64 store instructions per lane and unit, 32-byte requests): the store path is bound by requests, not bytes (measured on
// k_igb_tconv_fwd2: the stores alone were half of a full-resolution launch).  otile: NW x [8 pixels][OTS]; wave w's first 512 bytes
// double as its slot of the statistics fold.
constexpr int OTS = 64 + 8;             // bf16 per pixel of the epilogue tile (144 B: the 8-byte writes of a half spread over the banks)
// Where a unit's epilogue time goes (tools/ig_stamps.py, 64 -> 64 channels at 512 x 512: 10.9 k of the unit's 24.7 k ticks): the
// four rows 1.65 k each -- VECTOR-ALU work, ~170 instructions per row for a lane's 16 values (bias, activation, rounding, the two
// statistics) with two waves per SIMD in the epilogue at once --, the DPP sums 1.4 k, then a block barrier (1.5 k) and the bucket adds
// (1 k).  Tried: the bias through LDS (nothing); the block's statistics in one LDS slot by LDS atomics until the kernel ends (the
// barrier and the bucket adds gone, -2 % per launch -- and the float sums then depend on the order the waves arrive in, which under
// dtype bf16 moves gradients by 1e-2 from run to run: not kept).
template <int MODE, int NW>
__device__ __forceinline__ void epilogue_cm(const ConvArgs& p, const f32x4 (&acc)[4][4], int b, int y0, int x0, int co0, int tile, bf16_t* otile) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m16 = lane & 15, q = lane >> 4;
    const int which = co0 >= p.n_dst0;
    const int cw = which ? p.n_dst1 : p.n_dst0, cl = which ? co0 - p.n_dst0 : co0;
    float* dst = p.dst[which];
    const bool bn_on = MODE == 0 && p.bnf.tab != nullptr;
    bf16_t* ot = otile + wave * (8 * OTS);
    f32x4 bias[4], bs[4], bq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bias[j] = (MODE == 0 && p.bias) ? *reinterpret_cast<const f32x4*>(p.bias + co0 + 16 * j + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
        bs[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        bq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int x = x0 + m16;
    const bool okx = x < p.W;
    const bool half_out = MODE == 0 ? p.dst_half != 0 : (p.dsth[which] != 0 && !p.acc[which]);      // block-uniform: through the LDS tile
    const bool act_max = p.alpha >= 0.f && p.alpha <= 1.f;
    IGSTAMP(48, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int y = y0 + 4 * wave + r;
        IGSTAMP(48, 1 + r);
        if (y >= p.H) continue;                 // wave-uniform
        // element offset in 32 bits (the launcher checks that every destination has fewer than 2^32 elements)
        const unsigned orow = (unsigned)(b * p.H + y) * (unsigned)p.W;
        const unsigned o = (orow + (unsigned)(okx ? x : 0)) * (unsigned)cw + (unsigned)(cl + 4 * q);
        if (half_out) {
            hbf16x4 th[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 t = acc[r][j] + bias[j];
                if (MODE == 0 && act_max) {          // 0 <= alpha <= 1: act(t) = max(t, alpha t), two instructions instead of three
#pragma unroll
                    for (int i = 0; i < 4; ++i) t[i] = fmaxf(t[i], p.alpha * t[i]);
                } else if (MODE == 0 && p.alpha >= 0.f) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) t[i] = t[i] > 0.f ? t[i] : p.alpha * t[i];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) { th[j][i] = (hbf16)t[i]; t[i] = (float)th[j][i]; }      // the statistics are those of the stored values
                if (bn_on && okx) {          // (vector forms: packed fp32 instructions)
                    bs[j] += t;
                    bq[j] += t * t;
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if ((m16 >> 3) == h) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) *reinterpret_cast<hbf16x4*>(reinterpret_cast<hbf16*>(ot) + (m16 & 7) * OTS + 16 * j + 4 * q) = th[j];
                }
                const u32x4 v = *reinterpret_cast<const u32x4*>(ot + (lane >> 3) * OTS + 8 * (lane & 7));
                const int xs = x0 + 8 * h + (lane >> 3);
                if (xs < p.W)
                    *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(dst) + (size_t)((orow + (unsigned)xs) * (unsigned)cw + (unsigned)(cl + 8 * (lane & 7)))) = v;
            }
        } else if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
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
        } else if (p.dsth[which]) {          // bf16 destination that accumulates (never masked: the BatchNorm backward applies act')
            hbf16* dh = reinterpret_cast<hbf16*>(dst);
            hbf16x4 oldh[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) oldh[j] = okx ? *reinterpret_cast<const hbf16x4*>(dh + o + 16 * j) : hbf16x4{(hbf16)0.f, (hbf16)0.f, (hbf16)0.f, (hbf16)0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                hbf16x4 th;
#pragma unroll
                for (int i = 0; i < 4; ++i) th[i] = (hbf16)(acc[r][j][i] + (float)oldh[j][i]);
                if (okx) *reinterpret_cast<hbf16x4*>(dh + o + 16 * j) = th;
            }
        } else {
            f32x4 t[4], old[4], mk[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t[j] = acc[r][j];
                if (p.acc[which]) old[j] = okx ? *reinterpret_cast<const f32x4*>(dst + o + 16 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.mask[which]) mk[j] = okx ? *reinterpret_cast<const f32x4*>(p.mask[which] + o + 16 * j) : f32x4{1.f, 1.f, 1.f, 1.f};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (p.acc[which]) t[j] += old[j];
                if (p.mask[which]) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) t[j][i] *= mk[j][i] > 0.f ? 1.0f : p.alpha;
                }
                if (okx) *reinterpret_cast<f32x4*>(dst + o + 16 * j) = t[j];
            }
        }
    }
    IGSTAMP(48, 5);
    if (bn_on) {        // this unit's sums go to bucket row tile % R: [2 cw], first half sums, second half sums of squares
        float* red = reinterpret_cast<float*>(ot);          // the wave's 128-float slot (its tile is free: DS operations execute in order)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float s1 = ig::row16_sum(bs[j][i]), s2 = ig::row16_sum(bq[j][i]);
                if (m16 == 0) { red[16 * j + 4 * q + i] = s1; red[64 + 16 * j + 4 * q + i] = s2; }
            }
        IGSTAMP(48, 6);
        lds_barrier();
        IGSTAMP(48, 7);
        if (tid < 128) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) a += reinterpret_cast<const float*>(otile + w * (8 * OTS))[tid];
            const int half = tid >= 64, c = half ? tid - 64 : tid;
            atomicAdd(bn_bucket(p.bnf, tile) + half * cw + cl + c, (double)a);
        }
    }
}

// A16: the sources are stored as bf16 (View::h): 16-byte loads of 8 channels go to LDS as they are.
// NW: waves per block.  4: 16 x 16-pixel tiles, one wave per SIMD.  8: 32 x 16-pixel tiles, two waves per SIMD (256 registers
// each) -- a single wave cannot issue v_mfma_f32_16x16x32_bf16 back to back (1.7 of 2.46 PFLOP/s, mfma_bf16_rate.hip), the
// second wave fills those slots and the first one's waits, and the weight slab and the barrier serve twice the MFMAs; the
// two 76 KB buffers only fit with unpadded 64-byte rows (a fragment read then covers one contiguous KB: conflict-free too).
template <int MODE, bool A16, int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void k_igb_conv3(ConvArgs p, const bf16_t* __restrict__ w16) {
    constexpr int NT = 64 * NW, TR = 4 * NW;                  // threads, tile rows
    constexpr int PATCHX = (TR + 2) * (T2 + 2);               // staged pixels
    constexpr int RSX = NW == 8 ? 32 : RS;                    // bf16 per LDS row
    constexpr int BOFF = PATCHX * RSX, DUMP = (PATCHX + 9 * 64) * RSX, BUFX = DUMP + 64;
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    static_assert(NW == 8 || BUFX == BUF3, "layout of the 4-wave kernel");
    __shared__ __attribute__((aligned(16))) bf16_t lds[2 * BUFX];
    __shared__ __attribute__((aligned(16))) bf16_t otile[NW * 8 * OTS];      // epilogue_cm: wave-private output tiles + the statistics fold
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m16 = lane & 15, q = lane >> 4;
    const int kin = p.c_src0 + p.c_src1, nout = p.n_dst0 + p.n_dst1;
    const int nco = nout >> 6, ntiles = p.tiles_x * p.tiles_y * p.B, nunits = ntiles * nco;
    const int nchunks = kin / CK;
    const bool xcd_map = (ntiles & 7) == 0 && (gridDim.x & 7) == 0;
    const int my_units = (nunits - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nitems = my_units * nchunks;
    if (nitems <= 0) {          // (the launchers size the grid to the units: not reached)
        if (MODE == 0 && p.bnf.tab) bn_self_fold(p.bnf, gridDim.x, blockIdx.x);
        return;
    }

    const size_t npix = (size_t)p.B * p.H * p.W;
    constexpr unsigned ESZ = A16 ? 2 : 4;
    const unsigned nbytes0 = (unsigned)(npix * p.c_src0 * ESZ), nbytes1 = (unsigned)(npix * p.c_src1 * ESZ);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)w16, 0, (unsigned)((size_t)9 * nout * kin * 2), BUF_FLAGS);

    const FastDiv d_nco(nco), d_tx(p.tiles_x), d_ty(p.tiles_y), d_chunks(nchunks);
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
        u.x0 = bx * T2; u.y0 = by * TR; u.co0 = cot * 64; u.tile = tile;
        return u;
    };

    // ---- staging geometry of this thread: A element v = patch pixel tid / TPP + PPV v, channels CPT (tid % TPP)..
    //      (f32 sources: 8 threads x 4 channels per pixel; bf16 sources: 4 threads x 8 channels);
    //      B element v = tap v, output channel tid >> 2, K part tid & 3
    constexpr int TPP = A16 ? 4 : 8, PPV = NT / TPP, CPT = CK / TPP;
    constexpr int AU = (PATCHX * TPP + NT - 1) / NT, BU = (9 * 256 + NT - 1) / NT;      // B: 9 taps x 64 rows x 4 16-byte parts
    static_assert(AU <= 16 && BU <= 9, "staging slices of the tap loop");
    const int c4 = tid & (TPP - 1);
    int a_ly[AU], a_lx[AU];
#pragma unroll
    for (int v = 0; v < AU; ++v) {
        const int px = tid / TPP + PPV * v;
        a_ly[v] = px / (T2 + 2);
        a_lx[v] = px - a_ly[v] * (T2 + 2);
        if (px >= PATCHX) a_ly[v] = -4096;          // never inside an image
    }
    u32x4 ar[AU];
    u32x4 br[BU];
    // item being staged (uniform per block)
    struct Stage { int b, y0, x0, co0, cc, cs, c0; unsigned oob; __amdgpu_buffer_rsrc_t rs; };
    // the items are staged in order: next_stage() describes the next one (the unit decode runs only when the unit changes;
    // recomputing it per item cost 8 % of an item on the one wave a SIMD has)
    int sg_k = 0, sg_cc = 0;
    unsigned sg_oob = 0u;
    Unit sg_u = unit_of(0);
    auto next_stage = [&]() {
        Stage st;
        st.b = sg_u.b; st.y0 = sg_u.y0; st.x0 = sg_u.x0; st.co0 = sg_u.co0;
        st.cc = sg_cc;
        const bool second = st.cc >= p.c_src0;
        st.cs = second ? p.c_src1 : p.c_src0;
        st.c0 = second ? st.cc - p.c_src0 : st.cc;
        st.oob = sg_oob;                    // past the last item: every offset out of range
        st.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(second ? p.src[1] : p.src[0]), 0, second ? nbytes1 : nbytes0, BUF_FLAGS);
        sg_cc += CK;
        if (sg_cc >= kin) {
            sg_cc = 0;
            ++sg_k;
            if (sg_k < my_units) sg_u = unit_of(sg_k);
            else sg_oob = OOB;
        }
        return st;
    };
    auto issue_a = [&](const Stage& st, int v) {
        const int iy = st.y0 - 1 + a_ly[v], ix = st.x0 - 1 + a_lx[v];
        const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const unsigned off = (ok ? (unsigned)(((((st.b * p.H + iy) * p.W + ix) * st.cs) + st.c0 + CPT * c4) * ESZ) : OOB) | st.oob;
        ar[v] = __builtin_amdgcn_raw_buffer_load_b128(st.rs, off, 0, 0);
    };
    auto issue_b = [&](const Stage& st, int v) {          // piece i = tid + NT v: tap i / 256, output channel (i / 4) % 64, K part i % 4
        const int i = tid + NT * v, tap = i >> 8, bn = (i >> 2) & 63, bpart = i & 3;
        const unsigned off = (i < 9 * 256 ? (unsigned)((((tap * nout + st.co0 + bn) * kin) + st.cc + 8 * bpart) * 2) : OOB) | st.oob;
        br[v] = __builtin_amdgcn_raw_buffer_load_b128(rsw, off, 0, 0);
    };
    auto commit_a = [&](bf16_t* buf, int v) {
        const int px = tid / TPP + PPV * v;
        // branch-free: lanes past the patch (last element only) write into the buffer's dump row
        bf16_t* dst = buf + (px < PATCHX ? px * RSX : DUMP) + CPT * c4;
        if constexpr (A16) {
            *reinterpret_cast<u32x4*>(dst) = ar[v];
        } else {
            const f32x4 f = __builtin_bit_cast(f32x4, ar[v]);
            bf16x4 h;
            h[0] = (bf16_t)f[0]; h[1] = (bf16_t)f[1]; h[2] = (bf16_t)f[2]; h[3] = (bf16_t)f[3];
            *reinterpret_cast<bf16x4*>(dst) = h;
        }
    };
    auto commit_b = [&](bf16_t* buf, int v) {
        const int i = tid + NT * v, bpart = i & 3;
        *reinterpret_cast<u32x4*>(buf + (i < 9 * 256 ? BOFF + (i >> 2) * RSX : DUMP) + 8 * bpart) = br[v];
    };

    f32x4 acc[4][4];
    // ---- prologue: item 0 into buffer 0, item 1 into registers
    {
        const Stage s0 = next_stage();
#pragma unroll
        for (int v = 0; v < AU; ++v) issue_a(s0, v);
#pragma unroll
        for (int v = 0; v < BU; ++v) issue_b(s0, v);
#pragma unroll
        for (int v = 0; v < AU; ++v) commit_a(lds, v);
#pragma unroll
        for (int v = 0; v < BU; ++v) commit_b(lds, v);
        const Stage s1 = next_stage();
#pragma unroll
        for (int v = 0; v < AU; ++v) issue_a(s1, v);
#pragma unroll
        for (int v = 0; v < BU; ++v) issue_b(s1, v);
    }
    lds_barrier();
    // units outside, K chunks inside: the accumulators are plainly zeroed per unit (a conditional reset inside one flat
    // item loop made hipcc shuffle all 64 accumulator registers through copies at the top of every item)
    int it = 0;
#pragma unroll 1
    for (int k = 0; k < my_units; ++k) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int chunk = 0; chunk < nchunks; ++chunk, ++it) {
        bf16_t* buf = lds + (it & 1) * BUFX;
        bf16_t* other = lds + ((it & 1) ^ 1) * BUFX;
        IGSTAMP(it, 0);
        const Stage nx = next_stage();           // item it + 2
        IGSTAMP(it, 1);
        const bf16_t* a_lds = buf + ((4 * wave) * (T2 + 2) + m16) * RSX + 8 * q;
        const bf16_t* b_lds = buf + BOFF + m16 * RSX + 8 * q;
        // B fragments: double-buffered with one wave per SIMD; with two the other wave covers the read latency and the 16
        // registers are what keeps the kernel inside its 256 (the double-buffered version spilled 2-19 of them)
        constexpr int FB = NW == 8 ? 1 : 2;
        bf16x8 fa[2][6], fb[FB][4];
#pragma unroll
        for (int rr = 0; rr < 6; ++rr) fa[0][rr] = *reinterpret_cast<const bf16x8*>(a_lds + (rr * (T2 + 2)) * RSX);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[0][j] = *reinterpret_cast<const bf16x8*>(b_lds + (16 * j) * RSX);
        IGSTAMP(it, 2);
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const int g = s / 3, dy = s % 3;            // tap (dy, dx = g): weight slab index dy * 3 + g
#ifdef DNNCA_TUNING
            if (s == 3) IGSTAMP(it, 3);
            if (s == 6) IGSTAMP(it, 4);
#endif
            if (s + 1 < 9) {                            // fragments of the next step
                const int g1 = (s + 1) / 3, dy1 = (s + 1) % 3;
                if (dy1 == 0) {
#pragma unroll
                    for (int rr = 0; rr < 6; ++rr) fa[g1 & 1][rr] = *reinterpret_cast<const bf16x8*>(a_lds + (rr * (T2 + 2) + g1) * RSX);
                }
                if (FB == 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[(s + 1) & (FB - 1)][j] = *reinterpret_cast<const bf16x8*>(b_lds + ((dy1 * 3 + g1) * 64 + 16 * j) * RSX);
                }
            }
            if (s >= 1) {                               // staging slice s-1: elements {s-1, s-1+8}
                const int v0 = s - 1, v1 = s + 7;
                // unconditional (straight-line code the scheduler can weave between the MFMAs): past the last item the
                // commit writes stale registers into the buffer nobody reads again and the loads are out of range
                if (v0 < AU) { commit_a(other, v0); issue_a(nx, v0); }
                if (v1 < AU) { commit_a(other, v1); issue_a(nx, v1); }
                if (v0 < BU) { commit_b(other, v0); issue_b(nx, v0); }
                if (v1 < BU) { commit_b(other, v1); issue_b(nx, v1); }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[s & (FB - 1)][j], fa[g & 1][r + dy], acc[r][j], 0, 0, 0);      // rows = channels, columns = pixels
            if (FB == 1 && s + 1 < 9) {
                const int g1 = (s + 1) / 3, dy1 = (s + 1) % 3;
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[0][j] = *reinterpret_cast<const bf16x8*>(b_lds + ((dy1 * 3 + g1) * 64 + 16 * j) * RSX);
            }
        }
        IGSTAMP(it, 5);
        lds_barrier();
        IGSTAMP(it, 6);
    }
        {
            const Unit u = unit_of(k);
            epilogue_cm<MODE, NW>(p, acc, u.b, u.y0, u.x0, u.co0, u.tile, otile);
            IGSTAMP(it - 1, 7);
        }
    }
    if (MODE == 0 && p.bnf.tab) bn_self_fold(p.bnf, gridDim.x, blockIdx.x);          // block-uniform; every block gets here
}

// weight gradient, bf16, 64 x 64 channel blocks (all unet_big layers): M = 64 input channels (16 per wave), N = 64
// output channels, 9 taps, K = pixels.  One persistent block per CU (one wave per SIMD, the whole register file):
//   * the NHWC tiles are staged as they are -- [pixel][64 ch] bf16 rows of 160 B -- with 8-byte stores, and the
//     K-contiguous (pixel-major) MFMA fragments come out of ds_read_b64_tr_b16, the hardware transposing read: group kg of a
//     wave reads pixels 4kg..4kg+3 of tile row 2s (first read) and of row 2s+1 (second read); the contraction index may be
//     permuted freely as long as both operands agree, and this order makes every read bank-conflict free (160-B rows);
//   * the next tile's global loads are issued into registers before the current tile's 144 MFMAs per wave and written to
//     LDS after them;
//   * X and dY are read once per (ci block, co block) pair instead of once per 16 input channels.
constexpr int WRS = 80;                 // bf16 per staged pixel row: 64 channels + 16 pad

__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* row0, const bf16_t* row1) {
    typedef bf16x4 __attribute__((address_space(3))) * lds4;
    const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)row0);
    const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)row1);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// X16 / G16: x / dz are stored as bf16 (View::h) and go to LDS as they are
template <bool X16, bool G16>
__global__ __launch_bounds__(256, 1) void k_igb_wgrad64(ig::WgArgs p) {
    __shared__ __attribute__((aligned(16))) bf16_t ximg[PATCH * WRS];
    __shared__ __attribute__((aligned(16))) bf16_t gimg[TY * TX * WRS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m16 = lane & 15, q = lane >> 4;
    const int gq = (lane >> 2) & 3, gp = lane & 3;       // transposing read: lane 4*gq + gp of a group addresses row gq, columns 4gp..
    const int c0 = blockIdx.y * 64, co0 = blockIdx.z * 64;
    const bool do_bias = p.dbias && blockIdx.y == 0 && wave == 0;
    const int ntiles = p.tiles_x * p.tiles_y * p.B;
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (bf16_t)1.0f;

    f32x4 acc[9][4], accb[4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) accb[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int XU = (PATCH * 16 + 255) / 256, GU = TY * TX * 16 / 256;      // 12 and 8 float4 per thread
    const FastDiv d_tx(p.tiles_x), d_ty(p.tiles_y);
    using XR = std::conditional_t<X16, bf16x4, float4>;
    using GR = std::conditional_t<G16, bf16x4, float4>;
    XR xr[XU];
    GR gr[GU];
    const bf16_t* x16 = reinterpret_cast<const bf16_t*>(p.x);
    const bf16_t* g16 = reinterpret_cast<const bf16_t*>(p.dz);
    auto issue = [&](int tile) {
        const int trow = d_tx.div(tile), bx = tile - trow * p.tiles_x, b = d_ty.div(trow), by = trow - b * p.tiles_y;
        const int x0 = bx * TX, y0 = by * TY;
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int i = tid + 256 * u, px = i >> 4, c4 = i & 15;
            const int ly = px / (TX + 2), lx = px - ly * (TX + 2);
            const int iy = y0 - 1 + ly, ix = x0 - 1 + lx;
            const bool ok = px < PATCH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const size_t o = (((size_t)b * p.H + iy) * p.W + ix) * p.cs + c0 + 4 * c4;
            if constexpr (X16) {
                xr[u] = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
                if (ok) xr[u] = *reinterpret_cast<const bf16x4*>(x16 + o);
            } else {
                xr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) xr[u] = *reinterpret_cast<const float4*>(p.x + o);
            }
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int i = tid + 256 * u, px = i >> 4, n4 = i & 15;
            const int ly = px / TX, lx = px - ly * TX;
            const int iy = y0 + ly, ix = x0 + lx;
            const bool ok = iy < p.H && ix < p.W;
            const size_t o = (((size_t)b * p.H + iy) * p.W + ix) * p.cout + co0 + 4 * n4;
            if constexpr (G16) {
                gr[u] = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
                if (ok) gr[u] = *reinterpret_cast<const bf16x4*>(g16 + o);
            } else {
                gr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) gr[u] = *reinterpret_cast<const float4*>(p.dz + o);
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int i = tid + 256 * u, px = i >> 4, c4 = i & 15;
            if (px >= PATCH) continue;
            bf16x4 h;
            if constexpr (X16) h = xr[u];
            else { h[0] = (bf16_t)xr[u].x; h[1] = (bf16_t)xr[u].y; h[2] = (bf16_t)xr[u].z; h[3] = (bf16_t)xr[u].w; }
            *reinterpret_cast<bf16x4*>(ximg + px * WRS + 4 * c4) = h;
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int i = tid + 256 * u, px = i >> 4, n4 = i & 15;
            bf16x4 h;
            if constexpr (G16) h = gr[u];
            else { h[0] = (bf16_t)gr[u].x; h[1] = (bf16_t)gr[u].y; h[2] = (bf16_t)gr[u].z; h[3] = (bf16_t)gr[u].w; }
            *reinterpret_cast<bf16x4*>(gimg + px * WRS + 4 * n4) = h;
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) issue(tile);
    // per-lane bases of the transposing reads (bf16 elements)
    const int xbase = (4 * q + gq) * WRS + 16 * wave + 4 * gp;
    const int gbase = (4 * q + gq) * WRS + 4 * gp;
#pragma unroll 1
    for (; tile < ntiles; tile += p.psplit) {
        lds_barrier();              // the previous tile's fragment reads are complete
        commit();
        if (tile + p.psplit < ntiles) issue(tile + p.psplit);
        lds_barrier();
#pragma unroll 1
        for (int s = 0; s < TY / 2; ++s) {
            bf16x8 bv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                bv[j] = tr_frag(gimg + gbase + (2 * s * TX) * WRS + 16 * j, gimg + gbase + ((2 * s + 1) * TX) * WRS + 16 * j);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3, dx = t % 3;
                const bf16x8 av = tr_frag(ximg + xbase + ((2 * s + dy) * (TX + 2) + dx) * WRS,
                                          ximg + xbase + ((2 * s + 1 + dy) * (TX + 2) + dx) * WRS);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv[j], acc[t][j], 0, 0, 0);
            }
            if (do_bias) {
#pragma unroll
                for (int j = 0; j < 4; ++j) accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bv[j], accb[j], 0, 0, 0);
            }
        }
    }
    // D[ci = 16 wave + 4q + i][co = 16j + m16]
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                atomicAdd(p.dw + ((size_t)t * p.cin_total + p.ci_off + c0 + 16 * wave + 4 * q + i) * p.cout + co0 + 16 * j + m16, acc[t][j][i]);
    if (do_bias && q == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(p.dbias + co0 + 16 * j + m16, accb[j][0]);
    }
}

// k_igb_wgrad64 with eight waves per block (two per SIMD) for bf16-stored operands: the same 8 x 16-pixel tiles and 64 x 64
// channel block, wave & 3 = 16-channel slice of the input channels, wave >> 2 = half of the output channels -- 72 instead of
// 144 accumulator registers per wave, so two waves fit a SIMD: full-rate MFMA issue (one wave alone reaches 1.7 of 2.46
// PFLOP/s) and one wave's transposing reads hide behind the other's MFMAs.  Every wave still adds distinct elements to the
// gradient: the atomic traffic is unchanged.
// Two LDS buffers: the next tile (in registers since the previous iteration) is written into the other buffer while this one is
// consumed -- one barrier per tile and no wave waits for the commit.
// NJ = 16-channel output tiles per wave (2: the block covers 64 output channels).  NJ = 4 -- 128 output channels per block, the
// same X patch and nine A fragments serving twice the MFMAs (26 transposing reads per 36 MFMAs instead of 22 per 18), half as
// many split-K blocks per (ci, co) pair, so half the atomic drain and half the dY re-reads -- was built and measured in round 3:
// 253 registers (no spills), 114.5 us per launch against 94.8 and unet_big 7.73 against 6.88 ms: with 144 accumulators the
// compiler has no registers left to keep fragment reads ahead of the MFMAs.  Not instantiated.
template <int NJ>
__global__ __launch_bounds__(512, 1) void k_igb_wgrad64w(ig::WgArgs p) {
    constexpr int COB = 32 * NJ;                // output channels per block
    constexpr int GRS = COB + 16;               // bf16 per staged dY pixel row
    __shared__ __attribute__((aligned(16))) bf16_t ximg2[2][PATCH * WRS];
    __shared__ __attribute__((aligned(16))) bf16_t gimg2[2][TY * TX * GRS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 3, wn = wave >> 2;
    const int m16 = lane & 15, q = lane >> 4;
    const int gq = (lane >> 2) & 3, gp = lane & 3;
    const int c0 = blockIdx.y * 64, co0 = blockIdx.z * COB + 16 * NJ * wn;
    // bias gradient = sum of dY over the pixels: every thread sums the four channels it stages (it stages the same channel quad
    // of every tile), folded through LDS at the end -- four registers instead of the 4 NJ + 4 of an all-ones MFMA operand
    const bool do_bias = p.dbias && blockIdx.y == 0;          // block-uniform
    const int ntiles = p.tiles_x * p.tiles_y * p.B;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};

    f32x4 acc[9][NJ];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int GQ = COB / 4;                                                     // 8-byte elements per dY pixel
    constexpr int XU = (PATCH * 16 + 511) / 512, GU = TY * TX * GQ / 512;      // 6 and 4 (8) 8-byte elements per thread
    const FastDiv d_tx(p.tiles_x), d_ty(p.tiles_y);
    bf16x4 xr[XU], gr[GU];
    // Buffer loads with 32-bit byte offsets: what depends on the thread (patch pixel, channel quad) is computed once, a tile adds
    // one wave-uniform base; pixels outside the image get an out-of-range offset and come back as zeros (the staging of a
    // tile was ~150 vector instructions per wave with 64-bit address arithmetic: a fifth of the kernel).
    const size_t npix = (size_t)p.B * p.H * p.W;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (unsigned)(npix * p.cs * 2), BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc((void*)p.dz, 0, (unsigned)(npix * p.cout * 2), BUF_FLAGS);
    int xyl[XU], xoff[XU];          // (ly << 16 | lx) and the thread's byte offset inside a tile at (0, 0)
#pragma unroll
    for (int u = 0; u < XU; ++u) {
        const int i = tid + 512 * u, px = i >> 4, c4 = i & 15;
        const int ly = px / (TX + 2), lx = px - ly * (TX + 2);
        xyl[u] = px < PATCH ? (ly << 16) | lx : 0x7fff7fff;              // past the patch: never inside
        xoff[u] = ((ly * p.W + lx) * p.cs + c0 + 4 * c4) * 2;
    }
    // dY element u of this thread: pixel tid / GQ + (512 / GQ) u of the tile -- 512 / GQ pixels are GROWS whole tile rows, so
    // element u sits GROWS u rows below element 0, same column, same channel quad
    constexpr int GROWS = 512 / GQ / TX;
    static_assert(512 % GQ == 0 && (512 / GQ) % TX == 0, "dY staging: whole tile rows per element");
    const int g_ly = (tid / GQ) / TX, g_lx = (tid / GQ) % TX;
    const int goff0 = ((g_ly * p.W + g_lx) * p.cout + (int)blockIdx.z * COB + 4 * (tid % GQ)) * 2, gstep = GROWS * p.W * p.cout * 2;
    auto issue = [&](int tile) {
        const int trow = d_tx.div(tile), bx = tile - trow * p.tiles_x, b = d_ty.div(trow), by = trow - b * p.tiles_y;
        const int x0 = bx * TX, y0 = by * TY;
        const int xb = (((b * p.H + y0 - 1) * p.W + x0 - 1) * p.cs) * 2, gb = (((b * p.H + y0) * p.W + x0) * p.cout) * 2;
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int iy = y0 - 1 + (xyl[u] >> 16), ix = x0 - 1 + (xyl[u] & 0xffff);
            const bool in = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const unsigned off = in ? (unsigned)(xb + xoff[u]) : 0x80000000u;
            xr[u] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(rsx, off, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int iy = y0 + g_ly + GROWS * u, ix = x0 + g_lx;
            const bool in = iy < p.H && ix < p.W;
            const unsigned off = in ? (unsigned)(gb + goff0 + u * gstep) : 0x80000000u;
            gr[u] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(rsg, off, 0, 0));
        }
    };
    auto commit = [&](int buf) {
        bf16_t* ximg = ximg2[buf];
        bf16_t* gimg = gimg2[buf];
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int i = tid + 512 * u, px = i >> 4, c4 = i & 15;
            if (px < PATCH) *reinterpret_cast<bf16x4*>(ximg + px * WRS + 4 * c4) = xr[u];
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int i = tid + 512 * u, px = i / GQ, n4 = i % GQ;
            *reinterpret_cast<bf16x4*>(gimg + px * GRS + 4 * n4) = gr[u];
            if (do_bias) {          // (pixels outside the image were loaded as zeros)
#pragma unroll
                for (int k = 0; k < 4; ++k) bsum[k] += (float)gr[u][k];
            }
        }
    };

    int tile = blockIdx.x, buf = 0;
    if (tile < ntiles) {
        issue(tile);
        commit(0);
        if (tile + p.psplit < ntiles) issue(tile + p.psplit);
    }
    lds_barrier();
    const int xbase = (4 * q + gq) * WRS + 16 * wm + 4 * gp;
    const int gbase = (4 * q + gq) * GRS + 16 * NJ * wn + 4 * gp;
#pragma unroll 1
    for (; tile < ntiles; tile += p.psplit) {
        if (tile + p.psplit < ntiles) {
            commit(buf ^ 1);            // every wave has left the previous tile (the barrier below): its buffer is free
            if (tile + 2 * p.psplit < ntiles) issue(tile + 2 * p.psplit);
        }
        const bf16_t* ximg = ximg2[buf];
        const bf16_t* gimg = gimg2[buf];
        // the four steps of a tile unrolled: consecutive steps share two of their four patch rows per dx, and the compiler merges the
        // repeated transposing reads -- 46 reads per 72 MFMAs instead of 64 (the LDS pipe was the busier one): 79.3 -> 74.5 us per launch
#pragma unroll
        for (int s = 0; s < TY / 2; ++s) {
            bf16x8 bv[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                bv[j] = tr_frag(gimg + gbase + (2 * s * TX) * GRS + 16 * j, gimg + gbase + ((2 * s + 1) * TX) * GRS + 16 * j);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3, dx = t % 3;
                const bf16x8 av = tr_frag(ximg + xbase + ((2 * s + dy) * (TX + 2) + dx) * WRS,
                                          ximg + xbase + ((2 * s + 1 + dy) * (TX + 2) + dx) * WRS);
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv[j], acc[t][j], 0, 0, 0);
            }
        }
        lds_barrier();
        buf ^= 1;
    }
    // D[ci = 16 wm + 4q + i][co = 16 NJ wn + 16j + m16]; plain: stores into slab blockIdx.x (WgArgs::plain), else atomics into the gradient
    const size_t boff = p.plain ? (size_t)blockIdx.x * p.bucket_stride : 0;
    float* dwb = p.dw + boff + ((size_t)(p.ci_off + c0 + 16 * wm + 4 * q)) * p.cout + co0 + m16;
    if (p.plain) {          // block-uniform
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) dwb[((size_t)t * p.cin_total + i) * p.cout + 16 * j] = acc[t][j][i];
    } else {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) atomicAdd(dwb + ((size_t)t * p.cin_total + i) * p.cout + 16 * j, acc[t][j][i]);
    }
    if (do_bias) {          // fold the 512 / GQ threads of every channel quad (the tile buffers are free: the loop ended on a barrier)
        float* red = reinterpret_cast<float*>(&ximg2[0][0]);
        *reinterpret_cast<float4*>(red + 4 * tid) = make_float4(bsum[0], bsum[1], bsum[2], bsum[3]);
        __syncthreads();
        if (tid < COB) {
            const int n4 = tid >> 2, k = tid & 3;
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < 512 / GQ; ++r) a += red[4 * (n4 + GQ * r) + k];
            if (p.plain) p.dbias[boff + blockIdx.z * COB + tid] = a;
            else atomicAdd(p.dbias + blockIdx.z * COB + tid, a);
        }
    }
}

// ------------------------------------------------------------------------------------------------ bf16 transposed conv
// Conv2DTranspose(k = s = 2) under dtype bf16: four 1x1 GEMMs (one per output parity) that share their input tile.  All
// three passes are HBM passes over the 4x larger output tensor; the point of these kernels is to read / write every
// tensor once (the f32 kernels re-read the input per parity and per 16-channel chunk).
using ig::TcArgs;
using ig::tc_outpix;

// forward: block = 128 input pixels x 64 output channels x 4 parities; K = Cin in chunks of 32
template <bool X16>      // X16: the input is stored as bf16
__global__ __launch_bounds__(256, 2) void k_igb_tconv_fwd(TcArgs p, const bf16_t* __restrict__ w16) {
    __shared__ __attribute__((aligned(16))) bf16_t a_lds[128 * RS];
    __shared__ __attribute__((aligned(16))) bf16_t b_lds[4 * 64 * RS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m16 = lane & 15, q = lane >> 4;
    const int p0 = blockIdx.x * 128, co0 = blockIdx.y * 64;
    f32x4 acc[4][2][4];
#pragma unroll
    for (int ae = 0; ae < 4; ++ae)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[ae][r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int cc = 0; cc < p.cin; cc += CK) {
        lds_barrier();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + 256 * u, px = i >> 3, c4 = i & 7;
            bf16x4 h;
            if constexpr (X16) {
                h = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
                if (p0 + px < p.npix) h = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(p.in) + (size_t)(p0 + px) * p.cin + cc + 4 * c4);
            } else {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p0 + px < p.npix) v = *reinterpret_cast<const float4*>(p.in + (size_t)(p0 + px) * p.cin + cc + 4 * c4);
                h[0] = (bf16_t)v.x; h[1] = (bf16_t)v.y; h[2] = (bf16_t)v.z; h[3] = (bf16_t)v.w;
            }
            *reinterpret_cast<bf16x4*>(a_lds + px * RS + 4 * c4) = h;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + 256 * u, part = i & 3, r = i >> 2;         // r = ae * 64 + n
            const int ae = r >> 6, n = r & 63;
            *reinterpret_cast<uint4*>(b_lds + r * RS + 8 * part) =
                *reinterpret_cast<const uint4*>(w16 + ((size_t)ae * p.cout + co0 + n) * p.cin + cc + 8 * part);
        }
        lds_barrier();
        bf16x8 av[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) av[r] = *reinterpret_cast<const bf16x8*>(a_lds + (32 * wave + 16 * r + m16) * RS + 8 * q);
#pragma unroll
        for (int ae = 0; ae < 4; ++ae)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16x8 bv = *reinterpret_cast<const bf16x8*>(b_lds + (ae * 64 + 16 * j + m16) * RS + 8 * q);
#pragma unroll
                for (int r = 0; r < 2; ++r) acc[ae][r][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[r], bv, acc[ae][r][j], 0, 0, 0);
            }
    }
    float bias[4], bs[4], bq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bias[j] = p.bias[co0 + 16 * j + m16];
        bs[j] = 0.f;
        bq[j] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = p0 + 32 * wave + 16 * r + 4 * q + i;
            if (px >= p.npix) continue;
#pragma unroll
            for (int ae = 0; ae < 4; ++ae) {
                float* op = p.out + tc_outpix(px, ae >> 1, ae & 1, p.H, p.W) * p.cout + co0 + m16;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = acc[ae][r][j][i] + bias[j];
                    if (p.out_half) {          // stored as bf16 (the input of a BatchNorm): the statistics are those of the stored values
                        const hbf16 h = (hbf16)v;
                        reinterpret_cast<hbf16*>(p.out)[(size_t)(op - p.out) + 16 * j] = h;
                        v = (float)h;
                    } else {
                        op[16 * j] = v;
                    }
                    bs[j] += v;
                    bq[j] = fmaf(v, v, bq[j]);
                }
            }
        }
    if (p.bnf.tab) {        // batch statistics for the BatchNorm behind the transposed conv: a bucket row per pixel block
        __shared__ float red[4][128];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bs[j] += __shfl_xor(bs[j], 16); bs[j] += __shfl_xor(bs[j], 32);
            bq[j] += __shfl_xor(bq[j], 16); bq[j] += __shfl_xor(bq[j], 32);
            if (q == 0) { red[wave][16 * j + m16] = bs[j]; red[wave][64 + 16 * j + m16] = bq[j]; }
        }
        __syncthreads();
        if (tid < 128) {
            const float a = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
            const int half = tid >> 6, c = tid & 63;
            atomicAdd(bn_bucket(p.bnf, (int)blockIdx.x) + half * p.cout + co0 + c, (double)a);
        }
        bn_self_fold(p.bnf, gridDim.x * gridDim.y, blockIdx.y * gridDim.x + blockIdx.x);
    }
}

// forward, second generation (round 4; output stored as bf16): the same block tile (128 input pixels x 64 output channels x 4
// parities, K chunks of 32) with what the first one lacked:
//   * two LDS buffers and register prefetch -- chunk c + 1 goes to the other buffer and chunk c + 2's loads are issued while chunk c's
//     MFMAs run: one barrier per chunk (the first version loaded, waited, multiplied, chunk after chunk); two blocks per CU, so one
//     block's prologue and epilogue run beside the other's K loop;
//   * eight waves = 2 output-row parities x 4 pixel quarters; the MFMAs run channel-major (rows = output channels: A = kernel rows;
//     columns = pixels: B), so a lane holds four consecutive channels of a pixel;
//   * the results (bias added, rounded) pass a wave-private LDS tile laid out as the output is -- [pixel][column parity][64 channels]
//     -- and leave as 16-byte stores, eight lanes per 128-byte run of 64 channels: full cache lines, 1 KB per store instruction.
//     (Measured on the way, tuning builds: with 8-byte stores of four channels per lane -- 32-byte requests -- the stores ALONE took
//     half of the full-resolution launch: the store path is bound by requests, not bytes.)
//   * the batch statistics of the BatchNorm behind the layer (of the stored values) are DPP row sums over the 16 pixels of a tile.
// What was also built and measured and is NOT here (NOTES.md, round-4 log 11): persistent blocks with the kernel chunk resident in
// LDS, pixels straight from global memory as MFMA operands, pixel prefetch two items ahead behind the stores with hand-balanced
// vmcnt counts -- 53.5 us against this kernel's at the full-resolution level, slower at the three deeper ones, three times the code.
template <bool X16>      // X16: the input is stored as bf16
__global__ __launch_bounds__(512, 2) void k_igb_tconv_fwd2(TcArgs p, const bf16_t* __restrict__ w16) {
    constexpr int XB = 128 * RS, WB = 4 * 64 * RS, BUF = XB + WB;          // bf16 elements per buffer: 30 KB
    constexpr int ORS = 128 + 8;                                            // output tile row: [e][64 channels] + pad (bf16)
    static_assert(8 * 16 * ORS <= 2 * BUF, "the output tiles of one pixel half fit the staging buffers");
    __shared__ __attribute__((aligned(16))) bf16_t lds[2 * BUF];
    __shared__ float red[8 * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m16 = lane & 15, q = lane >> 4;
    const int a = wave & 1, pq = wave >> 1;
    const int p0 = blockIdx.x * 128, co0 = blockIdx.y * 64;
    const int nchunks = p.cin / CK;
    f32x4 acc[2][4][2];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) acc[e][mt][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // staging: X chunk = 128 pixels x 32 channels (thread: pixel tid / 4, eight channels), W chunk = 256 rows (ae, co) x 32 channels
    // (thread: rows tid / 4 and 128 + tid / 4, eight channels)
    const int spx = tid >> 2, part = tid & 3;
    const bool px_ok = p0 + spx < p.npix;
    u32x4 xr[X16 ? 1 : 2], wr[2];
    auto issue = [&](int c) {
        const int cc = c * CK;
        const size_t xo = (size_t)(px_ok ? p0 + spx : 0) * p.cin + cc + 8 * part;
        if constexpr (X16) {
            xr[0] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(p.in) + xo);
        } else {
            xr[0] = *reinterpret_cast<const u32x4*>(p.in + xo);
            xr[1] = *reinterpret_cast<const u32x4*>(p.in + xo + 4);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = spx + 128 * u, ae = r >> 6, n = r & 63;
            wr[u] = *reinterpret_cast<const u32x4*>(w16 + ((size_t)ae * p.cout + co0 + n) * p.cin + cc + 8 * part);
        }
    };
    auto commit = [&](bf16_t* buf) {
        u32x4 h;
        if constexpr (X16) {
            h = xr[0];
        } else {
            const f32x4 f0 = __builtin_bit_cast(f32x4, xr[0]), f1 = __builtin_bit_cast(f32x4, xr[1]);
            bf16x8 t;
#pragma unroll
            for (int i = 0; i < 4; ++i) { t[i] = (bf16_t)f0[i]; t[4 + i] = (bf16_t)f1[i]; }
            h = __builtin_bit_cast(u32x4, t);
        }
        if (!px_ok) h = u32x4{0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(buf + spx * RS + 8 * part) = h;
#pragma unroll
        for (int u = 0; u < 2; ++u) *reinterpret_cast<u32x4*>(buf + XB + (spx + 128 * u) * RS + 8 * part) = wr[u];
    };
    issue(0);
    commit(lds);
    if (nchunks > 1) issue(1);
    lds_barrier();
#pragma unroll 1
    for (int c = 0; c < nchunks; ++c) {
        const bf16_t* buf = lds + (c & 1) * BUF;
        bf16x8 xf[2];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) xf[pt] = *reinterpret_cast<const bf16x8*>(buf + (32 * pq + 16 * pt + m16) * RS + 8 * q);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(buf + XB + ((2 * a + e) * 64 + 16 * mt + m16) * RS + 8 * q);
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) acc[e][mt][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[pt], acc[e][mt][pt], 0, 0, 0);
            }
        if (c + 1 < nchunks) {
            commit(lds + ((c + 1) & 1) * BUF);
            if (c + 2 < nchunks) issue(c + 2);
        }
        lds_barrier();
    }
    // epilogue: lane (m16, q) holds channels co0 + 16 mt + 4 q .. + 3 of input pixel p0 + 32 pq + 16 pt + m16, row parity a, both
    // column parities.  Per pixel half pt the wave writes its 16 pixels into its own tile [16][e][64] (the staging buffers are free:
    // the loop ended on a barrier; a wave's DS operations execute in order) and stores it 16 bytes per lane
    bf16_t* wt = lds + wave * (16 * ORS);
    const FastDiv d_W(p.W);
    f32x4 bias[4], bs[4], bq[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        bias[mt] = *reinterpret_cast<const f32x4*>(p.bias + co0 + 16 * mt + 4 * q);
        bs[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        bq[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int pw = p0 + 32 * pq + 16 * pt;
        const bool mine = pw + m16 < p.npix;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                f32x4 v = acc[e][mt][pt] + bias[mt];
                hbf16x4 h;          // the statistics are those of the stored values
#pragma unroll
                for (int i = 0; i < 4; ++i) { h[i] = (hbf16)v[i]; v[i] = mine ? (float)h[i] : 0.f; }
                *reinterpret_cast<hbf16x4*>(reinterpret_cast<hbf16*>(wt) + m16 * ORS + e * 64 + 16 * mt + 4 * q) = h;
                bs[mt] += v;
#pragma unroll
                for (int i = 0; i < 4; ++i) bq[mt][i] = fmaf(v[i], v[i], bq[mt][i]);
            }
        // word t of lane L: pixel 4 t + L / 16 of the 16, bf16 elements 8 (L % 16) .. of its [e][64 channels] row
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int px = pw + 4 * t + (lane >> 4);
            const u32x4 v = *reinterpret_cast<const u32x4*>(wt + (4 * t + (lane >> 4)) * ORS + 8 * (lane & 15));
            if (px < p.npix) {
                const int bi = d_W.div(px), jx = px - bi * p.W;          // bi = b * H + i
                const int l16 = lane & 15, e = l16 >> 3, ch = 8 * (l16 & 7);
                const size_t o = (((size_t)bi * 2 + a) * (2 * p.W) + 2 * jx + e) * p.cout + co0 + ch;
                *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.out) + o) = v;
            }
        }
    }
    if (p.bnf.tab) {        // batch statistics for the BatchNorm behind the transposed conv: a bucket row per pixel block
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float s1 = ig::row16_sum(bs[mt][i]), s2 = ig::row16_sum(bq[mt][i]);
                if (m16 == 0) { red[wave * 128 + 16 * mt + 4 * q + i] = s1; red[wave * 128 + 64 + 16 * mt + 4 * q + i] = s2; }
            }
        __syncthreads();
        if (tid < 128) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) sum += red[w * 128 + tid];
            const int half = tid >> 6, c = tid & 63;
            atomicAdd(bn_bucket(p.bnf, (int)blockIdx.x) + half * p.cout + co0 + c, (double)sum);
        }
        bn_self_fold(p.bnf, gridDim.x * gridDim.y, blockIdx.y * gridDim.x + blockIdx.x);
    }
}

// data gradient: din[p][ci] = sum_{ae,co} dout[out(p,ae)][co] * W[ae][co][ci]; block = 128 pixels x 64 input channels,
// K = 4 x Cout in chunks of KC; w16 = [ae][ci][co] (K = co contiguous).  KC = 64 with bf16-stored dout (round 4): 16-byte loads
// of eight channels and half as many chunks -- a block's K loop is a chain of memory round trips (one chunk in flight), and
// with 32-channel chunks of 8-byte loads a block had 12 KB in flight.
template <bool G16, int KC>      // G16: dout is stored as bf16
__global__ __launch_bounds__(256, 2) void k_igb_tconv_dgrad(TcArgs p, const bf16_t* __restrict__ w16) {
    static_assert(KC == 32 || (KC == 64 && G16), "64-channel chunks: bf16-stored dout");
    constexpr int RSK = KC + 8;                      // LDS row stride (bf16)
    constexpr int AW = G16 ? KC / 16 : KC / 8;       // staging words per thread: 16 bytes (eight bf16 / four f32 channels) each
    constexpr int PPW = G16 ? KC / 8 : KC / 4;       // words per pixel
    constexpr int BW = KC / 32;                      // 16-byte kernel words per thread: 64 rows x KC / 8
    __shared__ __attribute__((aligned(16))) bf16_t a_lds[128 * RSK];
    __shared__ __attribute__((aligned(16))) bf16_t b_lds[64 * RSK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m16 = lane & 15, q = lane >> 4;
    const int p0 = blockIdx.x * 128, n0 = blockIdx.y * 64;
    f32x4 acc[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 ar[AW], br[BW];
    auto issue = [&](int kc) {
        const int ae = kc / p.cout, cc = kc - ae * p.cout;
#pragma unroll
        for (int u = 0; u < AW; ++u) {
            const int i = tid + 256 * u, px = i / PPW, part = i % PPW;
            const bool ok = p0 + px < p.npix;
            const size_t o = ok ? tc_outpix(p0 + px, ae >> 1, ae & 1, p.H, p.W) * p.cout + cc + (G16 ? 8 : 4) * part : 0;
            ar[u] = u32x4{0u, 0u, 0u, 0u};
            if (ok) ar[u] = G16 ? *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(p.dout) + o) : *reinterpret_cast<const u32x4*>(p.dout + o);
        }
#pragma unroll
        for (int u = 0; u < BW; ++u) {
            const int i = tid + 256 * u, r = i / (KC / 8), part = i % (KC / 8);
            br[u] = *reinterpret_cast<const u32x4*>(w16 + ((size_t)ae * p.cin + n0 + r) * p.cout + cc + 8 * part);
        }
    };
    issue(0);
#pragma unroll 1
    for (int kc = 0; kc < 4 * p.cout; kc += KC) {
        lds_barrier();
#pragma unroll
        for (int u = 0; u < AW; ++u) {
            const int i = tid + 256 * u, px = i / PPW, part = i % PPW;
            if constexpr (G16) {
                *reinterpret_cast<u32x4*>(a_lds + px * RSK + 8 * part) = ar[u];
            } else {
                const f32x4 f = __builtin_bit_cast(f32x4, ar[u]);
                bf16x4 h;
                h[0] = (bf16_t)f[0]; h[1] = (bf16_t)f[1]; h[2] = (bf16_t)f[2]; h[3] = (bf16_t)f[3];
                *reinterpret_cast<bf16x4*>(a_lds + px * RSK + 4 * part) = h;
            }
        }
#pragma unroll
        for (int u = 0; u < BW; ++u) {
            const int i = tid + 256 * u, r = i / (KC / 8), part = i % (KC / 8);
            *reinterpret_cast<u32x4*>(b_lds + r * RSK + 8 * part) = br[u];
        }
        if (kc + KC < 4 * p.cout) issue(kc + KC);
        lds_barrier();
#pragma unroll
        for (int ks = 0; ks < KC / 32; ++ks) {
            bf16x8 bv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const bf16x8*>(b_lds + (16 * j + m16) * RSK + 32 * ks + 8 * q);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(a_lds + (32 * wave + 16 * r + m16) * RSK + 32 * ks + 8 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv[j], acc[r][j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = p0 + 32 * wave + 16 * r + 4 * q + i;
            if (px >= p.npix) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const size_t o = (size_t)px * p.cin + n0 + 16 * j + m16;
                float v = acc[r][j][i];
                if (p.din_half) {          // the gradient arriving at a BatchNorm, stored as bf16 (ig_plan_half; never masked)
                    hbf16* dh = reinterpret_cast<hbf16*>(p.din);
                    if (p.acc) v += (float)dh[o];
                    dh[o] = (hbf16)v;
                    continue;
                }
                if (p.acc) v += p.din[o];
                if (p.mask) v *= p.mask[o] > 0.f ? 1.0f : p.alpha;
                p.din[o] = v;
            }
        }
}

// weight gradient: dW[ae][co][ci] = sum_p dout[out(p,ae)][co] * in[p][ci]; block = 64 co x 64 ci x 4 parities, persistent
// over tiles of 128 input pixels (K); same staging / transposing-read scheme as k_igb_wgrad64.
template <bool X16, bool G16>
__global__ __launch_bounds__(256, 1) void k_igb_tconv_wgrad64(TcArgs p) {
    __shared__ __attribute__((aligned(16))) bf16_t ximg[128 * WRS];
    __shared__ __attribute__((aligned(16))) bf16_t gimg[4 * 128 * WRS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m16 = lane & 15, q = lane >> 4, gq = (lane >> 2) & 3, gp = lane & 3;
    const int co0 = blockIdx.y * 64, n0 = blockIdx.z * 64;
    const bool do_bias = p.dbias && blockIdx.z == 0;
    const int ntiles = (p.npix + 127) / 128;
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (bf16_t)1.0f;
    f32x4 acc[4][4], accb[4];
#pragma unroll
    for (int ae = 0; ae < 4; ++ae) {
        accb[ae] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[ae][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    using XR = std::conditional_t<X16, bf16x4, float4>;
    using GR = std::conditional_t<G16, bf16x4, float4>;
    XR xr[8];
    GR gr[4][8];
    const bf16x4 zero4 = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
    auto issue = [&](int tile) {
        const int p0 = tile * 128;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + 256 * u, px = i >> 4, c4 = i & 15;
            const bool ok = p0 + px < p.npix;
            const size_t ox = (size_t)(p0 + px) * p.cin + n0 + 4 * c4;
            if constexpr (X16) {
                xr[u] = zero4;
                if (ok) xr[u] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(p.in) + ox);
            } else {
                xr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) xr[u] = *reinterpret_cast<const float4*>(p.in + ox);
            }
#pragma unroll
            for (int ae = 0; ae < 4; ++ae) {
                const size_t og = ok ? tc_outpix(p0 + px, ae >> 1, ae & 1, p.H, p.W) * p.cout + co0 + 4 * c4 : 0;
                if constexpr (G16) {
                    gr[ae][u] = zero4;
                    if (ok) gr[ae][u] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(p.dout) + og);
                } else {
                    gr[ae][u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ok) gr[ae][u] = *reinterpret_cast<const float4*>(p.dout + og);
                }
            }
        }
    };
    auto cvt = [](const auto& v) {
        if constexpr (std::is_same_v<std::decay_t<decltype(v)>, bf16x4>) {
            return v;
        } else {
            bf16x4 h;
            h[0] = (bf16_t)v.x; h[1] = (bf16_t)v.y; h[2] = (bf16_t)v.z; h[3] = (bf16_t)v.w;
            return h;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + 256 * u, px = i >> 4, c4 = i & 15;
            *reinterpret_cast<bf16x4*>(ximg + px * WRS + 4 * c4) = cvt(xr[u]);
#pragma unroll
            for (int ae = 0; ae < 4; ++ae) *reinterpret_cast<bf16x4*>(gimg + (ae * 128 + px) * WRS + 4 * c4) = cvt(gr[ae][u]);
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) issue(tile);
    const int xbase = (4 * q + gq) * WRS + 4 * gp;
    const int gbase = (4 * q + gq) * WRS + 16 * wave + 4 * gp;
#pragma unroll 1
    for (; tile < ntiles; tile += p.psplit) {
        lds_barrier();
        commit();
        if (tile + p.psplit < ntiles) issue(tile + p.psplit);
        lds_barrier();
#pragma unroll 1
        for (int s = 0; s < 4; ++s) {
            bf16x8 bv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = tr_frag(ximg + xbase + (32 * s) * WRS + 16 * j, ximg + xbase + (32 * s + 16) * WRS + 16 * j);
#pragma unroll
            for (int ae = 0; ae < 4; ++ae) {
                const bf16x8 av = tr_frag(gimg + gbase + (ae * 128 + 32 * s) * WRS, gimg + gbase + (ae * 128 + 32 * s + 16) * WRS);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[ae][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv[j], acc[ae][j], 0, 0, 0);
                if (do_bias) accb[ae] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, ones, accb[ae], 0, 0, 0);
            }
        }
    }
    // D[co = 16 wave + 4q + i][ci = 16j + m16]
#pragma unroll
    for (int ae = 0; ae < 4; ++ae)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                atomicAdd(p.dw + ((size_t)ae * p.cout + co0 + 16 * wave + 4 * q + i) * p.cin + n0 + 16 * j + m16, acc[ae][j][i]);
    if (do_bias && m16 == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            atomicAdd(p.dbias + co0 + 16 * wave + 4 * q + i, (accb[0][i] + accb[1][i]) + (accb[2][i] + accb[3][i]));
    }
}

}  // namespace igb

// ================================================================================================ host side
struct IgPlan {
    bool built = false;
    std::vector<ig::FlipDesc> flips;          // eager ones first (n_eager), then the convs the split-bf16 kernels cover (flipped only on demand)
    int n_eager = 0;
    std::map<size_t, int> flip_index;         // w_off -> index in flips
    ig::FlipDesc* flips_dev = nullptr;
    float* flipped = nullptr;        // same layout / offsets as the parameter vector (only conv kernels are filled)
    int max_w = 0;
    // bf16 mode: per-step bf16 copies of the conv kernels (forward layout / data-gradient layout)
    std::vector<igb::PrepDesc> preps;
    igb::PrepDesc* preps_dev = nullptr;
    igb::bf16_t* wf = nullptr;
    igb::bf16_t* wd = nullptr;
    int max_wb = 0;
    float* wg_slabs = nullptr;       // WG_BUCKETS copies of a small layer's weight + bias gradient (ig_wgrad2; left zeroed by k_wg_fold)
    std::map<std::pair<const void*, int>, std::pair<float*, size_t>> plain;      // (op, source) -> slabs of its plain-mode weight gradient, floats
    // the folds of this backward pass, launched together by ig_finish_wgrad (single-replica steps); the device copies of the tables are
    // rewritten only when the list changes (first step, another batch size)
    std::vector<ig::FoldSeg> fold_pending, fold_on_dev;
    ig::FoldSeg* fold_dev = nullptr;
    int* fold_first_dev = nullptr;
    size_t fold_cap = 0;
    double fold_bytes = 0;
    // DNNCA_FOLD_BATCH=1, read per step by ig_prepare (the tests flip it).  Opt-in: measured 0.2 - 0.4 % SLOWER on both dense
    // configurations -- the one fold is shorter than the many (159 against 252 / 347 us) but sits at the end of the backward pass,
    // where nothing overlaps it
    bool fold_batch = false;
};
constexpr int WG_BUCKETS = 16, WG_SLAB_FLOATS = 131072 + 1024;
static std::map<Model*, IgPlan> g_ig;

void ig_release(Model* m) {
    g_ig.erase(m);
    ig3x_release(m);
}

// slabs of one weight-gradient launch in plain mode (WgArgs::plain): allocated once per (op, source), grown on demand
static float* wg_plain_slabs(Model* m, const Op& o, int s, size_t floats) {
    IgPlan& pl = g_ig[m];
    const auto key = std::make_pair((const void*)&o, s);
    auto it = pl.plain.find(key);
    if (it != pl.plain.end() && it->second.second >= floats) return it->second.first;
    float* ptr = nullptr;
    if (m->alloc((void**)&ptr, floats * 4) != DNNCA_OK) return nullptr;
    pl.plain[key] = std::make_pair(ptr, floats);
    return ptr;
}
static bool wg_plain_on() {
    static const bool off = getenv("DNNCA_NO_WG_PLAIN") != nullptr;          // keep the float atomics (A/B)
    return !off;
}
// DNNCA_FOLD_BATCH=1: one fold launch per backward pass instead of one per weight-gradient launch -- unless this step sends gradient
// buckets while the backward pass runs (they need each layer's gradient final as soon as its launches are)
static bool wg_fold_batched(Model* m) { return g_ig[m].fold_batch && !m->bucketing; }

const DenseSwitches& dense_switches() {
    static const DenseSwitches sw = {getenv("DNNCA_IGCONV1") != nullptr, getenv("DNNCA_WGRAD1") != nullptr, getenv("DNNCA_NO_BN_FUSION") != nullptr,
                                     getenv("DNNCA_NO_POOL_STATS") != nullptr, getenv("DNNCA_NO_WG_BUCKETS") != nullptr,
                                     getenv("DNNCA_TCWGRAD1") != nullptr};
    return sw;
}

static inline bool dense(const View& v) { return v.C == 0 || v.ps == v.C; }

bool ig_conv_supported(const Model* m, const Op& o) {
    if (o.type != OP_CONV || o.k != 3) return false;
    if (!dense(o.inA.d) || !dense(o.inB.d) || !dense(o.out.d)) return false;
    const int CA = o.inA.d.C, CB = o.inB.d.C, CO = o.out.d.C;
    return CA % 16 == 0 && CB % 16 == 0 && CO % 16 == 0 && CA > 0;
}

static int pick_nn(int cout) { return cout % 64 == 0 ? 4 : (cout % 32 == 0 ? 2 : 1); }

// bf16 contraction: requested by the model description and possible when every K chunk holds 32 channels of one source
static bool use_bf16(const Model* m, const Op& o) {
    return m->desc.dtype == DNNCA_BF16 && o.inA.d.C % 32 == 0 && o.inB.d.C % 32 == 0 && o.out.d.C % 32 == 0;
}

bool ig_tconv_supported(const Model* m, const Op& o);
static bool use_bf16_tc(const Model* m, const Op& o) {
    return m->desc.dtype == DNNCA_BF16 && o.inA.d.C % 64 == 0 && o.out.d.C % 64 == 0;
}

// Which tensors are stored as bf16 (View::h)?  Under dtype bf16 the 64-channel conv / transposed-conv kernels round their
// operands to bf16 while staging them, so a tensor whose every reader is one of those kernels can live in HBM as bf16 with
// bit-identical results: the outputs of BatchNorm layers that feed only such kernels (and the fused max-pool, which records
// the positions of its maxima for the backward pass), and the conv-output gradients the BatchNorm backward hands to them.
int ig_plan_half(Model* m) {
    if (m->desc.dtype != DNNCA_BF16 || (m->desc.flags & 1) || getenv("DNNCA_NO_HALF")) return DNNCA_OK;
    const double MB = m->desc.max_batch;
    auto conv_ok = [&](const Op& c) {     // forward, data gradient and weight gradient all take the 64-channel bf16 kernels
        if (!ig_conv_supported(m, c) || !use_bf16(m, c)) return false;
        const int CA = c.inA.d.C, CB = c.inB.d.C, CO = c.out.d.C;
        if (CA % 64 || CB % 64 || CO % 64) return false;
        const int cmax = CA > CB ? (CA > CO ? CA : CO) : (CB > CO ? CB : CO);
        return MB * c.out.d.H * c.out.d.W * cmax * 4.0 < 2.0e9 && 9.0 * CO * (CA + CB) * 4.0 < 2.0e9;      // conv3_path()
    };
    std::map<float*, bool> hd, hg;
    for (const Op& bn : m->ops) {
        if (bn.type != OP_BN || !fast_bn_supported(m, bn) || !dense(bn.out.d) || !dense(bn.inA.g)) continue;
        bool ok = true;
        int users = 0;
        for (const Op& c : m->ops) {
            const bool a = c.inA.d.C && c.inA.d.p == bn.out.d.p, b = c.type == OP_CONV && c.inB.d.C && c.inB.d.p == bn.out.d.p;
            if (!a && !b) continue;
            ++users;
            if (c.type == OP_CONV) ok = ok && conv_ok(c);
            else if (c.type == OP_TCONV) ok = ok && ig_tconv_supported(m, c) && use_bf16_tc(m, c);
            else if (c.type == OP_POOL) ok = ok && fast_bn_pool_fusable(m, bn, c) && dense(c.inA.g) && dense(c.out.g) && !c.maskA;
            else ok = false;
        }
        if (ok && users) hd[bn.out.d.p] = true;
        // ... and so is the gradient that arrives at this BatchNorm: its writers are the backward passes of exactly those users
        // (data-gradient epilogue of the persistent conv kernel, bf16 transposed-conv data gradient, pool backward by recorded
        // positions; accumulation re-reads the bf16 value), its readers this BatchNorm's two backward passes.  DNNCA_NO_HALF_DY.
        if (ok && users && dense(bn.out.g) && !getenv("DNNCA_NO_HALF_DY")) {
            bool all_write = true;
            for (const Op& c : m->ops) {
                const bool a = c.inA.d.C && c.inA.d.p == bn.out.d.p, b = c.type == OP_CONV && c.inB.d.C && c.inB.d.p == bn.out.d.p;
                if ((a || b) && c.type == OP_CONV && !c.need_din) all_write = false;
                if ((a && c.maskA) || (b && c.maskB)) all_write = false;           // (BatchNorm nets never mask here)
            }
            if (all_write) hg[bn.out.g.p] = true;
        }
        // "bf16 activations" (BASELINE.md, configs[2]): the BatchNorm's input z -- written once by the conv / transposed conv in
        // front of it (persistent bf16 kernels: rounded in the epilogue, batch statistics taken from the rounded values), read
        // only by this BatchNorm's four passes -- is stored as bf16 as well.  Not a transparent change (z is rounded, 2^-9
        // relative): DNNCA_NO_HALF_Z keeps it in f32.
        if (!getenv("DNNCA_NO_HALF_Z") && dense(bn.inA.d)) {
            int readers = 0;
            const Op* prod = nullptr;
            for (const Op& c : m->ops) {
                if ((c.inA.d.C && c.inA.d.p == bn.inA.d.p) || (c.inB.d.C && c.inB.d.p == bn.inA.d.p)) ++readers;
                if (c.out.d.C && c.out.d.p == bn.inA.d.p) prod = &c;
            }
            const bool pre = prod && (prod->type != OP_CONV || prod->alpha < 0.f || prod->premasked);     // no g_act_bwd pass reads z in f32
            if (readers == 1 && pre &&
                ((prod->type == OP_CONV && conv_ok(*prod)) || (prod->type == OP_TCONV && ig_tconv_supported(m, *prod) && use_bf16_tc(m, *prod))))
                hd[bn.inA.d.p] = true;
        }
        // the gradient this BN's backward writes: read only by the backward of the op that produced the BN's input
        if (bn.accA) continue;
        for (const Op& c : m->ops) {
            if (c.out.d.p != bn.inA.d.p || !c.out.d.C) continue;
            const bool pre = c.type != OP_CONV || c.alpha < 0.f || c.premasked;      // nobody rewrites it in f32 (g_act_bwd)
            if ((c.type == OP_CONV && conv_ok(c) && c.need_din && pre) || (c.type == OP_TCONV && ig_tconv_supported(m, c) && use_bf16_tc(m, c)))
                hg[bn.inA.g.p] = true;
        }
    }
    // both sources of a conv are staged by the same code path
    for (bool changed = true; changed;) {
        changed = false;
        for (const Op& c : m->ops) {
            if (c.type != OP_CONV || !c.inB.d.C) continue;
            const bool a = hd.count(c.inA.d.p) && hd[c.inA.d.p], b = hd.count(c.inB.d.p) && hd[c.inB.d.p];
            if (a != b) { hd[c.inA.d.p] = false; hd[c.inB.d.p] = false; changed = true; }
        }
    }
    auto mark = [&](T& t) {
        if (!t.d.C) return;
        if (hd.count(t.d.p) && hd[t.d.p]) t.d.h = 1;
        if (hg.count(t.g.p) && hg[t.g.p]) t.g.h = 1;
    };
    for (Op& o : m->ops) {
        mark(o.inA); mark(o.inB); mark(o.out);
        if (o.type == OP_POOL && o.inA.d.h && !o.pool_idx) {     // the backward pass cannot fall back on comparing values
            void* ix = nullptr;
            DN_TRY(m->alloc(&ix, (size_t)m->desc.max_batch * o.out.d.H * o.out.d.W * o.out.d.C));
            o.pool_idx = (unsigned char*)ix;
        }
    }
    return DNNCA_OK;
}

static bool conv3_path(const ig::ConvArgs& a, int cout, bool bf16);

int ig_prepare(Model* m) {
    if (m->desc.flags & 1) return DNNCA_OK;
    IgPlan& pl = g_ig[m];
    if (!pl.built) {
        pl.built = true;
        // the data-gradient kernels of the fp32 MFMA path read flipped / transposed weights (k_ig_flip, once per backward pass); the
        // split-bf16 kernels have their own planes (ig3x_prepare), so convs they will take (the same static conditions ig3x_launch checks,
        // at max_batch) are only flipped on demand (launch_ig, should the split-bf16 launch ever decline)
        for (int pass = 0; pass < 2; ++pass) {
            for (const Op& o : m->ops) {
                if (!ig_conv_supported(m, o) || !o.need_din || use_bf16(m, o)) continue;
                const int cin = o.inA.d.C + o.inB.d.C, cout = o.out.d.C;
                ig::ConvArgs gd{};
                gd.c_src0 = cout; gd.n_dst0 = o.inA.d.C; gd.n_dst1 = o.inB.d.C;
                gd.B = m->desc.max_batch; gd.H = o.out.d.H; gd.W = o.out.d.W;
                const bool covered = ig3x_enabled(m) && conv3_path(gd, cin, false) && cin <= 1024 && 9.0 * cin * cout + 2.0 * (m->nT + 16) <= 1.0e9;
                if (covered != (pass == 1)) continue;
                ig::FlipDesc d{(int)o.w_off, cin, cout};
                pl.flip_index[o.w_off] = (int)pl.flips.size();
                pl.flips.push_back(d);
                if (pass == 0) {
                    ++pl.n_eager;
                    int n = 9 * d.cin * d.cout;
                    if (n > pl.max_w) pl.max_w = n;
                }
            }
        }
        for (const Op& o : m->ops) {
            if (!ig_conv_supported(m, o) || !use_bf16(m, o)) continue;
            igb::PrepDesc d{(int)o.w_off, o.inA.d.C + o.inB.d.C, o.out.d.C, 0};
            pl.preps.push_back(d);
            int n = 9 * d.cin * d.cout;
            if (n > pl.max_wb) pl.max_wb = n;
        }
        for (const Op& o : m->ops) {
            if (!ig_tconv_supported(m, o) || !use_bf16_tc(m, o)) continue;
            igb::PrepDesc d{(int)o.w_off, o.inA.d.C, o.out.d.C, 1};
            pl.preps.push_back(d);
            int n = 4 * d.cin * d.cout;
            if (n > pl.max_wb) pl.max_wb = n;
        }
        if (!pl.preps.empty()) {
            DN_TRY(m->alloc((void**)&pl.preps_dev, pl.preps.size() * sizeof(igb::PrepDesc)));
            DN_TRY(m->alloc((void**)&pl.wf, (size_t)m->nT * 2 + 16));
            DN_TRY(m->alloc((void**)&pl.wd, (size_t)m->nT * 2 + 16));
            HIP_TRY(hipMemcpyAsync(pl.preps_dev, pl.preps.data(), pl.preps.size() * sizeof(igb::PrepDesc), hipMemcpyHostToDevice, m->stream));
        }
        if (!pl.flips.empty()) {
            DN_TRY(m->alloc((void**)&pl.flips_dev, pl.flips.size() * sizeof(ig::FlipDesc)));
            DN_TRY(m->alloc((void**)&pl.flipped, (size_t)m->nT * 4));
            HIP_TRY(hipMemcpyAsync(pl.flips_dev, pl.flips.data(), pl.flips.size() * sizeof(ig::FlipDesc), hipMemcpyHostToDevice, m->stream));
            HIP_TRY(hipStreamSynchronize(m->stream));
        }
    }
    DN_TRY(ig3x_prepare(m));
    pl.fold_pending.clear();          // (a step that failed half-way may have left some)
    pl.fold_bytes = 0;
    pl.fold_batch = getenv("DNNCA_FOLD_BATCH") != nullptr;
    if (!pl.preps.empty()) {
        int bx = (pl.max_wb + 255) / 256;
        if (bx > 1024) bx = 1024;
        LAUNCH(m, "igb_prep", 8.0 * m->nT, 0,
               hipLaunchKernelGGL(igb::k_igb_prep, dim3(bx, (unsigned)pl.preps.size()), dim3(256), 0, m->stream, pl.preps_dev, m->p,
                                  pl.wf, pl.wd));
    }
    return DNNCA_OK;
}

// once per backward pass: the data-gradient kernels read flipped/transposed weights
int ig_begin_backward(Model* m) {
    for (Op& o : m->ops) o.bwd_sums_rode = false;          // (a pass that stopped on an error may have left one set)
    if (m->desc.flags & 1) return DNNCA_OK;
    IgPlan& pl = g_ig[m];
    if (pl.n_eager == 0) return DNNCA_OK;
    int bx = (pl.max_w + 255) / 256;
    if (bx > 1024) bx = 1024;
    LAUNCH(m, "ig_flip", 8.0 * m->nT, 0,
           hipLaunchKernelGGL(ig::k_ig_flip, dim3(bx, (unsigned)pl.n_eager), dim3(256), 0, m->stream, pl.flips_dev, m->p, pl.flipped));
    return DNNCA_OK;
}

// will launch_ig / launch_igb take the persistent kernel (k_ig_conv3 / k_igb_conv3) for these arguments?
static bool conv3_path(const ig::ConvArgs& a, int cout, bool bf16) {
    // the persistent kernels address their sources with 32-bit byte offsets (bit 31 marks "outside the image")
    const int cmax = a.c_src0 > a.c_src1 ? a.c_src0 : a.c_src1;
    const int dmax = a.n_dst0 > a.n_dst1 ? a.n_dst0 : a.n_dst1;          // (the epilogue's 32-bit element offsets)
    const bool fits = (double)a.B * a.H * a.W * cmax * 4.0 < 2.0e9 && 9.0 * cout * (a.c_src0 + a.c_src1) * 4.0 < 2.0e9 &&
                      (double)a.B * a.H * a.W * dmax < 4.0e9;
    if (bf16) return fits && cout % 64 == 0 && a.c_src0 % 32 == 0 && a.c_src1 % 32 == 0 && a.n_dst0 % 64 == 0;
    return fits && !dense_switches().igconv1;
}
// waves per block of igb::k_igb_conv3: 8 (32 x 16-pixel tiles, two waves per SIMD) once that still gives every CU a unit
static int igb_waves(const ig::ConvArgs& a, int cout) {
    static const int forced = getenv("DNNCA_IGB_NW") ? atoi(getenv("DNNCA_IGB_NW")) : 0;      // tuning aid: 4 or 8
    if (forced == 4 || forced == 8) return forced;
    const long units8 = (long)((a.W + 15) / 16) * ((a.H + 31) / 32) * a.B * (cout / 64);
    return units8 >= 256 ? 8 : 4;
}
// waves per block of ig::k_ig_conv3 (f32): 8 for 16- / 32-channel tiles (the channel tile launch_ig picks) with enough units
static int ig_nn3(const ig::ConvArgs& a) {
    const int d1 = a.n_dst1 ? a.n_dst1 : 64;
    return (a.n_dst0 % 64 == 0 && d1 % 64 == 0) ? 4 : ((a.n_dst0 % 32 == 0 && d1 % 32 == 0) ? 2 : 1);
}
static int ig_waves(const ig::ConvArgs& a, int cout) {
    static const int forced = getenv("DNNCA_IG_NW") ? atoi(getenv("DNNCA_IG_NW")) : 0;        // tuning aid: 4 or 8
    const int nn3 = ig_nn3(a);
    if (nn3 == 4 || forced == 4) return 4;
    const long units8 = (long)((a.W + 15) / 16) * ((a.H + 31) / 32) * a.B * (cout / (16 * nn3));
    return (forced == 8 || units8 >= 256) ? 8 : 4;
}

template <int MODE>
static void launch_ig(Model* m, const ig::ConvArgs& a, size_t w_off, int cout, const char* name, double bytes, double flops, bool* bnb_rode = nullptr) {
    if (bnb_rode) *bnb_rode = false;
    {   // pipelined persistent kernel: channel tile 16 nn3 must divide both destinations; 32-bit byte offsets
        const int nn3 = ig_nn3(a);
        if (conv3_path(a, cout, false)) {
            static const int x3_nn_cap = getenv("DNNCA_X3_NN") ? atoi(getenv("DNNCA_X3_NN")) : 4;          // tuning aid: channel tile at most 16 x this
            const int nnx = nn3 > x3_nn_cap && (x3_nn_cap == 1 || x3_nn_cap == 2) ? x3_nn_cap : nn3;
            // fp32 by three bf16 planes on the bf16 matrix pipe (kernels_ig3x.hip), unless switched off
            if (ig3x_launch(m, MODE, a, w_off, cout, nnx, MODE == 0 ? "ig3x_conv_fwd" : "ig3x_conv_dgrad", bytes, flops, bnb_rode))
                return;
            if (MODE == 1) {          // declined although the plan expected it: this conv's flipped weights were not made at ig_begin_backward
                IgPlan& plz = g_ig[m];
                auto it = plz.flip_index.find(w_off);
                if (it != plz.flip_index.end() && it->second >= plz.n_eager) {
                    const ig::FlipDesc& fd = plz.flips[it->second];
                    int bx = (9 * fd.cin * fd.cout + 255) / 256;
                    if (bx > 1024) bx = 1024;
                    LAUNCH(m, "ig_flip", 8.0 * 9 * fd.cin * fd.cout, 0,
                           hipLaunchKernelGGL(ig::k_ig_flip, dim3(bx, 1), dim3(256), 0, m->stream, plz.flips_dev + it->second, m->p, plz.flipped));
                }
            }
            ig::ConvArgs a2 = a;
            const int nw = ig_waves(a, cout);
            a2.tiles_x = (a.W + ig::F3T - 1) / ig::F3T;
            a2.tiles_y = (a.H + 4 * nw - 1) / (4 * nw);
            const unsigned units = (unsigned)(a2.tiles_x * a2.tiles_y * a2.B * (cout / (16 * nn3)));
            const unsigned g = units < 256u ? units : 256u;
            m->set_variant("3n%dw%d", nn3, nn3 == 4 ? 4 : nw);
            if (nn3 == 4) LAUNCH(m, name, bytes, flops, hipLaunchKernelGGL((ig::k_ig_conv3<4, MODE, 4>), dim3(g), dim3(256), 0, m->stream, a2));
            else if (nn3 == 2 && nw == 8) LAUNCH(m, name, bytes, flops, hipLaunchKernelGGL((ig::k_ig_conv3<2, MODE, 8>), dim3(g), dim3(512), 0, m->stream, a2));
            else if (nn3 == 2) LAUNCH(m, name, bytes, flops, hipLaunchKernelGGL((ig::k_ig_conv3<2, MODE, 4>), dim3(g), dim3(256), 0, m->stream, a2));
            else if (nw == 8) LAUNCH(m, name, bytes, flops, hipLaunchKernelGGL((ig::k_ig_conv3<1, MODE, 8>), dim3(g), dim3(512), 0, m->stream, a2));
            else LAUNCH(m, name, bytes, flops, hipLaunchKernelGGL((ig::k_ig_conv3<1, MODE, 4>), dim3(g), dim3(256), 0, m->stream, a2));
            return;
        }
    }
    const int nn = pick_nn(cout);
    dim3 grid(a.tiles_x * a.tiles_y * a.B, cout / (16 * nn));
    m->set_variant("n%d", nn);
    if (nn == 4) LAUNCH(m, name, bytes, flops, hipLaunchKernelGGL((ig::k_ig_conv<4, MODE>), grid, dim3(256), 0, m->stream, a));
    else if (nn == 2) LAUNCH(m, name, bytes, flops, hipLaunchKernelGGL((ig::k_ig_conv<2, MODE>), grid, dim3(256), 0, m->stream, a));
    else LAUNCH(m, name, bytes, flops, hipLaunchKernelGGL((ig::k_ig_conv<1, MODE>), grid, dim3(256), 0, m->stream, a));
}

template <int MODE>
static void launch_igb(Model* m, const ig::ConvArgs& a, const igb::bf16_t* w16, int cout, const char* name, double bytes,
                       double flops) {
    const int nn = pick_nn(cout);
    if (conv3_path(a, cout, true)) {
        ig::ConvArgs a2 = a;
        const int nw = igb_waves(a, cout);
        a2.tiles_x = (a.W + igb::T2 - 1) / igb::T2;
        a2.tiles_y = (a.H + 4 * nw - 1) / (4 * nw);
        const unsigned nblocks = (unsigned)(a2.tiles_x * a2.tiles_y * a2.B * (cout / 64));
        const unsigned g = nblocks < 256u ? nblocks : 256u;
        // the launch name carries the variant (waves per block, bf16-stored source) so that tests and profiles can tell
        // which instantiation ran: igb_conv_fwd_w8 / _w4 [+ _a16]
        char vname[64];
        snprintf(vname, sizeof(vname), "%s_w%d%s", name, nw, a.src_half ? "_a16" : "");
#define IGB3(A16v, NWv) LAUNCH(m, vname, bytes, flops, hipLaunchKernelGGL((igb::k_igb_conv3<MODE, A16v, NWv>), dim3(g), dim3(64 * NWv), 0, m->stream, a2, w16))
        if (a.src_half) { if (nw == 8) IGB3(true, 8); else IGB3(true, 4); }
        else { if (nw == 8) IGB3(false, 8); else IGB3(false, 4); }
#undef IGB3
        return;
    }
    dim3 grid(a.tiles_x * a.tiles_y * a.B, cout / (16 * nn));
    m->set_variant("n%d", nn);
    if (nn == 4) LAUNCH(m, name, bytes, flops, hipLaunchKernelGGL((igb::k_igb_conv<4, MODE>), grid, dim3(256), 0, m->stream, a, w16));
    else LAUNCH(m, name, bytes, flops, hipLaunchKernelGGL((igb::k_igb_conv<2, MODE>), grid, dim3(256), 0, m->stream, a, w16));
}

// the launch condition of k_ig_wgrad2 (second-generation fp32 weight gradient) for a source of cs channels
static bool wgrad2_ok(const Model* m, const Op& o, int B, int cs) {
    const int CO = o.out.d.C;
    return !use_bf16(m, o) && (double)B * o.out.d.H * o.out.d.W * (cs > CO ? cs : CO) * 4.0 < 2.0e9 && !dense_switches().wgrad1;
}

static ig::ConvArgs conv_fwd_geometry(int B, const Op& o) {
    ig::ConvArgs a{};
    a.c_src0 = o.inA.d.C; a.c_src1 = o.inB.d.C;
    a.n_dst0 = o.out.d.C; a.n_dst1 = 0;
    a.B = B; a.H = o.out.d.H; a.W = o.out.d.W;
    return a;
}

bool ig_norm_on_load_ok(const Model* m, int B, const Op& o) {
    static const bool off = getenv("DNNCA_NO_NORM_ON_LOAD") != nullptr;
    if (off || m->desc.dtype != DNNCA_F32 || !ig_conv_supported(m, o) || use_bf16(m, o)) return false;
    return conv3_path(conv_fwd_geometry(B, o), o.out.d.C, false) && wgrad2_ok(m, o, B, o.inA.d.C) && (!o.inB.d.C || wgrad2_ok(m, o, B, o.inB.d.C));
}

// source k of conv o for the kernels that normalise on load: the BatchNorm's input + coefficient table when its apply pass was elided
static const float* conv_source(Model* m, const Op& o, int k, const float** norm) {
    const View& v = k ? o.inB.d : o.inA.d;
    *norm = nullptr;
    if (o.src_bn[k] >= 0 && m->ops[o.src_bn[k]].elided) {
        const Op& bn = m->ops[o.src_bn[k]];
        *norm = bn.coef;
        return bn.inA.d.p;
    }
    return v.p;
}

bool ig_conv_fwd(Model* m, int B, Op& o, double bytes, double flops, Op* bn_next) {
    if (!ig_conv_supported(m, o)) return false;
    ig::ConvArgs a{};
    a.src[0] = conv_source(m, o, 0, &a.norm[0]);
    a.src[1] = o.inB.d.C ? conv_source(m, o, 1, &a.norm[1]) : o.inB.d.p;
    a.c_src0 = o.inA.d.C; a.c_src1 = o.inB.d.C;
    a.w = m->p + o.w_off;
    a.bias = m->p + o.b_off;
    a.dst[0] = o.out.d.p; a.n_dst0 = o.out.d.C; a.n_dst1 = 0;
    a.B = B; a.H = o.out.d.H; a.W = o.out.d.W;
    a.tiles_x = (a.W + ig::TX - 1) / ig::TX;
    a.tiles_y = (a.H + ig::TY - 1) / ig::TY;
    a.alpha = o.alpha;
    a.src_half = o.inA.d.h;          // ig_plan_half keeps both sources of a conv in the same format
    a.dst_half = o.out.d.h;
    if (bn_next && !dense_switches().no_bn_fusion && conv3_path(a, o.out.d.C, use_bf16(m, o))) {
        // the BatchNorm behind this conv takes its batch statistics (and coefficients) from the conv's epilogue
        (void)bn_self_fold_args(m, *bn_next, B, &a.bnf);
    }
    if (a.src_half && !(use_bf16(m, o) && conv3_path(a, o.out.d.C, true))) return false;      // cannot happen (ig_plan_half); the caller reports it
    if (use_bf16(m, o)) {
        IgPlan& pl = g_ig[m];
        launch_igb<0>(m, a, pl.wf + o.w_off, o.out.d.C, "igb_conv_fwd", bytes, flops);
        return true;
    }
    launch_ig<0>(m, a, o.w_off, o.out.d.C, "ig_conv_fwd", bytes, flops);
    return true;
}

bool ig_conv_bwd(Model* m, int B, Op& o, double out_bytes, double in_bytes, double flops) {
    if (!ig_conv_supported(m, o)) return false;
    IgPlan& pl = g_ig[m];
    const int CA = o.inA.d.C, CB = o.inB.d.C, CO = o.out.d.C;
    if (o.alpha >= 0.f && !o.premasked)
        LAUNCH(m, "g_act_bwd", 3 * out_bytes, out_bytes / 4,
               g_act_bwd(m->stream, (size_t)B * o.out.d.H * o.out.d.W * CO, o.out.g.p, o.out.d.p, o.alpha));
    const int tiles_x = (o.out.d.W + ig::TX - 1) / ig::TX, tiles_y = (o.out.d.H + ig::TY - 1) / ig::TY;
    // small gradients shared by many blocks go through bucket copies (see WgArgs::nbuckets); fp32 second-generation kernel only
    const int n_w = 9 * (CA + CB) * CO;
    bool bucketed = false;
    auto wgrad2_path = [&](int cs) { return wgrad2_ok(m, o, B, cs); };        // the launch condition of k_ig_wgrad2 below
    auto wgrad2_psplit = [&](int cs) {
        const int mw = cs % 64 == 0 ? 4 : (cs % 32 == 0 ? 2 : 1), nn = pick_nn(CO);
        const int combos = (cs / (16 * mw)) * (CO / (16 * nn));
        return (256 + combos - 1) / combos;
    };
    // (the split-bf16 kernel in plain mode needs no buckets: every block stores into its own slab)
    static const bool x3_wgrad_off = getenv("DNNCA_NO_X3_WGRAD") != nullptr;
    const bool plain3 = ig3x_enabled(m) && !x3_wgrad_off && wg_plain_on();
    if (!plain3 && wgrad2_path(CA) && (!CB || wgrad2_path(CB)) && n_w + CO <= WG_SLAB_FLOATS && wgrad2_psplit(CA) > 32 &&
        !dense_switches().no_wg_buckets) {
        if (!pl.wg_slabs && !m->dry && m->alloc((void**)&pl.wg_slabs, (size_t)WG_BUCKETS * WG_SLAB_FLOATS * 4) != DNNCA_OK) pl.wg_slabs = nullptr;
        bucketed = m->dry || pl.wg_slabs != nullptr;          // the dry run lists the launches of the real one
    }
    // weight (+bias) gradient, one launch per source -- on the side stream where the step allows (Model::wg_stream)
    auto weight_gradient = [&]() {
        hipStream_t main_stream = m->stream;
        const bool side = m->wg_side_begin();
        for (int s = 0; s < (CB ? 2 : 1); ++s) {
            ig::WgArgs w{};
            w.x = conv_source(m, o, s, &w.norm);
            w.dz = o.out.g.p;
            w.dw = bucketed ? pl.wg_slabs : m->g + o.w_off;
            w.dbias = s == 0 ? (bucketed ? pl.wg_slabs + n_w : m->g + o.b_off) : nullptr;
            w.nbuckets = bucketed ? WG_BUCKETS : 1;
            w.bucket_stride = WG_SLAB_FLOATS;
            w.cs = s == 0 ? CA : CB;
            w.ci_off = s == 0 ? 0 : CA;
            w.cin_total = CA + CB;
            w.cout = CO;
            w.B = B; w.H = o.out.d.H; w.W = o.out.d.W;
            w.tiles_x = tiles_x; w.tiles_y = tiles_y;
            const int nn = pick_nn(CO);
            const int combos = (w.cs / ig::CK) * (CO / (16 * nn));
            const int ntiles = tiles_x * tiles_y * B;
            int psplit = (1024 + combos - 1) / combos;
            if (psplit > ntiles) psplit = ntiles;
            if (psplit < 1) psplit = 1;
            w.psplit = psplit;
            dim3 grid(psplit, w.cs / ig::CK, CO / (16 * nn));
            const double bb = (out_bytes + in_bytes) / (CB ? 2 : 1), ff = flops / (CB ? 2 : 1);
            // plain mode (WgArgs::plain) for a launch of `ps` pixel-split blocks per channel-block pair: slabs instead of the gradient
            const int n_ws = 9 * w.cs * CO, pstride = (n_ws + CO + 3) / 4 * 4;
            auto go_plain = [&](int ps) {
                if (!wg_plain_on() || bucketed) return false;
                float* slabs = m->dry ? nullptr : wg_plain_slabs(m, o, s, (size_t)ps * pstride);
                if (!m->dry && !slabs) return false;
                w.nbuckets = 1;
                w.plain = 1;
                w.dw = slabs;
                w.dbias = s == 0 ? slabs + n_ws : nullptr;
                w.cin_total = w.cs;
                w.ci_off = 0;
                w.bucket_stride = pstride;
                return true;
            };
            auto fold_plain = [&](int ps) {
                const int n_b = s == 0 ? CO : 0;
                const ig::FoldSeg f{w.dw, m->g + o.w_off, m->g + o.b_off, ps, pstride, w.cs, CO, CA + CB, s == 0 ? 0 : CA, n_b, n_ws + n_b < 32768};
                if (wg_fold_batched(m)) {
                    pl.fold_pending.push_back(f);
                    pl.fold_bytes += 4.0 * ps * (n_ws + n_b);
                    return;
                }
                if (!f.g16)
                    LAUNCH(m, "wg_fold", 4.0 * ps * (n_ws + n_b), 0,
                           hipLaunchKernelGGL(ig::k_wg_fold_plain<1>, dim3((n_ws + n_b + 1023) / 1024), dim3(256), 0, m->stream, f));
                else
                    LAUNCH(m, "wg_fold", 4.0 * ps * (n_ws + n_b), 0,
                           hipLaunchKernelGGL(ig::k_wg_fold_plain<16>, dim3((n_ws + n_b + 63) / 64), dim3(256), 0, m->stream, f));
            };
            if (use_bf16(m, o) && CO % 64 == 0 && w.cs % 64 == 0) {
                const int combos64 = (w.cs / 64) * (CO / 64);
                int ps = (256 + combos64 - 1) / combos64;
                if (ps > ntiles) ps = ntiles;
                w.psplit = ps < 1 ? 1 : ps;
                const dim3 g64(w.psplit, w.cs / 64, CO / 64);
                const bool xh = (s == 0 ? o.inA.d.h : o.inB.d.h) != 0, gh = o.out.g.h != 0;
                static const int narrow = getenv("DNNCA_WGRAD64_NARROW") != nullptr;          // tuning aid
                if (xh && gh && !narrow && (double)B * w.H * w.W * (w.cs > CO ? w.cs : CO) * 2.0 < 2.0e9) {         // eight waves per block (bf16-stored operands, 32-bit byte offsets)
                    const bool plain = go_plain(w.psplit);
                    m->set_variant(plain ? "w8p" : "w8");
                    LAUNCH(m, "igb_wgrad64", bb, ff, hipLaunchKernelGGL(igb::k_igb_wgrad64w<2>, g64, dim3(512), 0, m->stream, w));
                    if (plain) fold_plain(w.psplit);
                    continue;
                }
    #define WG64(XH, GH) LAUNCH(m, "igb_wgrad64", bb, ff, hipLaunchKernelGGL((igb::k_igb_wgrad64<XH, GH>), g64, dim3(256), 0, m->stream, w))
                m->set_variant("x%dg%d", (int)xh, (int)gh);
                if (xh) { if (gh) WG64(true, true); else WG64(true, false); }
                else { if (gh) WG64(false, true); else WG64(false, false); }
    #undef WG64
            } else if (!use_bf16(m, o) && (double)B * o.out.d.H * o.out.d.W * (w.cs > CO ? w.cs : CO) * 4.0 < 2.0e9 &&
                       !dense_switches().wgrad1) {
                // fp32 by three bf16 planes on the bf16 matrix pipe (kernels_ig3x.hip), unless switched off
                {
                    const int ps3 = ig3x_wgrad_psplit(m, w, CO);
                    if (ps3 > 0) {
                        const bool plain = go_plain(ps3);
                        if (ig3x_wgrad_launch(m, w, CO, "ig3x_wgrad", bb, ff)) {
                            if (plain) fold_plain(ps3);
                            continue;
                        }
                    }
                }
                const int mw = w.cs % 64 == 0 ? 4 : (w.cs % 32 == 0 ? 2 : 1);
                const int tm = (4 / mw) < (4 / nn) ? (4 / mw) : (4 / nn);
                const int nt2 = tiles_x * ((o.out.d.H + 8 * tm - 1) / (8 * tm)) * B;
                const int combos2 = (w.cs / (16 * mw)) * (CO / (16 * nn));
                int ps = (256 + combos2 - 1) / combos2;
                if (ps > nt2) ps = nt2;
                w.psplit = ps < 1 ? 1 : ps;
                dim3 g2(w.psplit, w.cs / (16 * mw), CO / (16 * nn));
                static const int wg2_narrow = getenv("DNNCA_WGRAD2_NARROW") != nullptr;        // tuning aid
    #define WG2(MWv, NNv, NWv) LAUNCH(m, "ig_wgrad2", bb, ff, hipLaunchKernelGGL((ig::k_ig_wgrad2<MWv, NNv, NWv>), g2, dim3(64 * NWv), 0, m->stream, w))
    #define WG2N(MWv) do { if (nn == 4) WG2(MWv, 4, 4); \
                           else if (nn == 2) { if (wg2_narrow) WG2(MWv, 2, 4); else WG2(MWv, 2, 8); } \
                           else { if (wg2_narrow) WG2(MWv, 1, 4); else WG2(MWv, 1, 8); } } while (0)
                m->set_variant("m%dn%dw%d", mw, nn, nn == 4 || wg2_narrow ? 4 : 8);
                if (mw == 4) WG2N(4); else if (mw == 2) WG2N(2); else WG2N(1);
    #undef WG2N
    #undef WG2
            } else if (m->set_variant("n%d", nn), use_bf16(m, o) && nn == 4) LAUNCH(m, "igb_wgrad", bb, ff, hipLaunchKernelGGL((igb::k_igb_wgrad<4>), grid, dim3(256), 0, m->stream, w));
            else if (use_bf16(m, o) && nn == 2) LAUNCH(m, "igb_wgrad", bb, ff, hipLaunchKernelGGL((igb::k_igb_wgrad<2>), grid, dim3(256), 0, m->stream, w));
            else if (nn == 4) LAUNCH(m, "ig_wgrad", bb, ff, hipLaunchKernelGGL((ig::k_ig_wgrad<4>), grid, dim3(256), 0, m->stream, w));
            else if (nn == 2) LAUNCH(m, "ig_wgrad", bb, ff, hipLaunchKernelGGL((ig::k_ig_wgrad<2>), grid, dim3(256), 0, m->stream, w));
            else LAUNCH(m, "ig_wgrad", bb, ff, hipLaunchKernelGGL((ig::k_ig_wgrad<1>), grid, dim3(256), 0, m->stream, w));
        }
        if (bucketed)
            LAUNCH(m, "wg_fold", 4.0 * WG_BUCKETS * (n_w + CO), 0,
                   hipLaunchKernelGGL(ig::k_wg_fold, dim3((n_w + CO + 255) / 256), dim3(256), 0, m->stream, pl.wg_slabs, WG_BUCKETS,
                                      WG_SLAB_FLOATS, n_w, m->g + o.w_off, m->g + o.b_off, CO));
        if (side) m->wg_side_end(main_stream);
    };
    auto data_gradient = [&]() {
        if (o.need_din) {
            ig::ConvArgs a{};
            a.src[0] = o.out.g.p; a.c_src0 = CO; a.c_src1 = 0;
            a.src_half = o.out.g.h;
            a.w = pl.flipped + o.w_off;
            a.dst[0] = o.inA.g.p; a.dst[1] = o.inB.g.p;
            a.n_dst0 = CA; a.n_dst1 = CB;
            a.mask[0] = o.maskA ? o.inA.d.p : nullptr;
            a.mask[1] = o.maskB ? o.inB.d.p : nullptr;
            a.acc[0] = o.accA; a.acc[1] = o.accB;
            a.dsth[0] = o.inA.g.h; a.dsth[1] = o.inB.g.h;
            a.B = B; a.H = o.out.d.H; a.W = o.out.d.W;
            a.tiles_x = tiles_x; a.tiles_y = tiles_y;
            a.alpha = o.mask_alpha;
            // this launch writes ALL of the gradient arriving at the BatchNorm that feeds the conv (its only reader, nothing accumulated,
            // no act' mask): the BatchNorm's backward sums ride in the epilogue and its reduction pass is not launched (ConvArgs::bnb;
            // split-bf16 kernel, f32 tensors)
            static const bool no_bnb = getenv("DNNCA_NO_BN_BWD_RIDE") != nullptr;          // A/B
            Op* bnb_op = nullptr;
            if (!no_bnb && !use_bf16(m, o) && CB == 0 && o.src_bn[0] >= 0 && !o.accA && !o.maskA && CA <= ig3x_max_bnb_channels()) {
                Op& bn = m->ops[o.src_bn[0]];
                const bool sole = bn.out_readers.size() == 1 && bn.out_readers[0] == (int)(&o - m->ops.data());
                if (sole && bn.out.g.p == o.inA.g.p && bn.out.g.C == CA && conv3_path(a, CA, false) && ig3x_accepts(m, a, CA) &&
                    bn_bwd_fold_args(m, bn, &a.bnb))
                    bnb_op = &bn;
            }
            if (use_bf16(m, o))
                launch_igb<1>(m, a, pl.wd + o.w_off, CA + CB, "igb_conv_dgrad", out_bytes + in_bytes, flops);
            else {
                bool rode = false;
                launch_ig<1>(m, a, o.w_off, CA + CB, "ig_conv_dgrad", out_bytes + in_bytes, flops, &rode);
                if (bnb_op && !rode) bnb_op->bwd_sums_rode = false;          // (a layout without the sums: the reduction pass runs as before)
            }
        }
    };
    // the fork sits in front of the data gradient (measured: forking behind it, so that the weight gradient meets only the next
    // layer's HBM-bound BatchNorm passes, is 1.5 - 2.5 % slower on both dense configs; DNNCA_WG_LAST=1 selects that order)
    static const bool wg_last = getenv("DNNCA_WG_LAST") != nullptr;
    if (!wg_last) { weight_gradient(); data_gradient(); }
    else { data_gradient(); weight_gradient(); }
    return true;
}

// end of the backward pass: the pending folds (ig_conv_bwd, fold_plain) in one launch per 64 of them -- behind the weight-gradient
// launches on their stream (the caller joins the streams afterwards)
int ig_finish_wgrad(Model* m) {
    auto it = g_ig.find(m);
    if (it == g_ig.end() || it->second.fold_pending.empty()) return DNNCA_OK;
    IgPlan& pl = it->second;
    const size_t n = pl.fold_pending.size(), nchunks = (n + ig::kFoldBatch - 1) / ig::kFoldBatch;
    std::vector<int> first(nchunks * ig::kFoldBatch, 0), grid(nchunks, 0);
    for (size_t i = 0; i < n; ++i) {
        const ig::FoldSeg& f = pl.fold_pending[i];
        const int floats = 9 * f.cs * f.cout + f.n_b;
        first[i] = grid[i / ig::kFoldBatch];
        grid[i / ig::kFoldBatch] += f.g16 ? (floats + 63) / 64 : (floats + 1023) / 1024;
    }
    if (!m->dry) {
        const bool same = pl.fold_on_dev.size() == n && memcmp(pl.fold_on_dev.data(), pl.fold_pending.data(), n * sizeof(ig::FoldSeg)) == 0;
        if (!same) {
            HIP_TRY(hipDeviceSynchronize());          // the previous step's fold may still be reading the tables
            if (pl.fold_cap < nchunks * ig::kFoldBatch) {
                pl.fold_cap = nchunks * ig::kFoldBatch;
                DN_TRY(m->alloc((void**)&pl.fold_dev, pl.fold_cap * sizeof(ig::FoldSeg)));
                DN_TRY(m->alloc((void**)&pl.fold_first_dev, pl.fold_cap * sizeof(int)));
                HIP_TRY(hipStreamSynchronize(m->stream));          // (alloc clears on the stream)
            }
            HIP_TRY(hipMemcpy(pl.fold_dev, pl.fold_pending.data(), n * sizeof(ig::FoldSeg), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(pl.fold_first_dev, first.data(), first.size() * sizeof(int), hipMemcpyHostToDevice));
            pl.fold_on_dev = pl.fold_pending;
        }
    }
    hipStream_t main_stream = m->stream;
    if (m->wg_pending && m->wg_stream) m->stream = m->wg_stream;
    for (size_t c = 0; c < nchunks; ++c) {
        const int ns = (int)std::min<size_t>(ig::kFoldBatch, n - c * ig::kFoldBatch);
        LAUNCH(m, "wg_fold_all", pl.fold_bytes / nchunks, 0,
               hipLaunchKernelGGL(ig::k_wg_fold_batch, dim3(grid[c]), dim3(256), 0, m->stream, pl.fold_dev + c * ig::kFoldBatch,
                                  pl.fold_first_dev + c * ig::kFoldBatch, ns));
    }
    m->stream = main_stream;
    pl.fold_pending.clear();
    pl.fold_bytes = 0;
    return DNNCA_OK;
}

bool ig_tconv_supported(const Model* m, const Op& o) {
    if (o.type != OP_TCONV || o.k != 2) return false;
    if (!dense(o.inA.d) || !dense(o.out.d)) return false;
    return o.inA.d.C % 16 == 0 && o.out.d.C % 16 == 0;
}

static ig::TcArgs tc_args(Model* m, int B, Op& o) {
    ig::TcArgs a{};
    a.in = o.inA.d.p;
    a.w = m->p + o.w_off;
    a.bias = m->p + o.b_off;
    a.out = o.out.d.p;
    a.dout = o.out.g.p;
    a.din = o.inA.g.p;
    a.mask = o.maskA ? o.inA.d.p : nullptr;
    a.dw = m->g + o.w_off;
    a.dbias = m->g + o.b_off;
    a.acc = o.accA;
    a.cin = o.inA.d.C; a.cout = o.out.d.C;
    a.out_half = o.out.d.h;
    a.din_half = o.inA.g.h;
    a.H = o.inA.d.H; a.W = o.inA.d.W;
    a.npix = B * a.H * a.W;
    a.alpha = o.mask_alpha;
    return a;
}

bool ig_tconv_fwd(Model* m, int B, Op& o, double bytes, double flops, Op* bn_next) {
    if (!ig_tconv_supported(m, o)) return false;
    ig::TcArgs a = tc_args(m, B, o);
    if (use_bf16_tc(m, o)) {
        IgPlan& pl = g_ig[m];
        if (bn_next && !dense_switches().no_bn_fusion)       // batch statistics of the BatchNorm behind it ride in the epilogue
            (void)bn_self_fold_args(m, *bn_next, B, &a.bnf);
        const dim3 grid((a.npix + 127) / 128, a.cout / 64);
        static const bool gen1 = getenv("DNNCA_TCONV_FWD1") != nullptr;          // A/B: the first-generation kernel
        if (!gen1 && a.out_half && (double)a.npix * 4.0 * a.cout < 2.0e9) {
            m->set_variant("2h%d", (int)(o.inA.d.h != 0));
            if (o.inA.d.h)
                LAUNCH(m, "igb_tconv_fwd", bytes, flops, hipLaunchKernelGGL(igb::k_igb_tconv_fwd2<true>, grid, dim3(512), 0, m->stream, a, pl.wf + o.w_off));
            else
                LAUNCH(m, "igb_tconv_fwd", bytes, flops, hipLaunchKernelGGL(igb::k_igb_tconv_fwd2<false>, grid, dim3(512), 0, m->stream, a, pl.wf + o.w_off));
            return true;
        }
        m->set_variant("h%d", (int)(o.inA.d.h != 0));
        if (o.inA.d.h)
            LAUNCH(m, "igb_tconv_fwd", bytes, flops, hipLaunchKernelGGL(igb::k_igb_tconv_fwd<true>, grid, dim3(256), 0, m->stream, a, pl.wf + o.w_off));
        else
            LAUNCH(m, "igb_tconv_fwd", bytes, flops, hipLaunchKernelGGL(igb::k_igb_tconv_fwd<false>, grid, dim3(256), 0, m->stream, a, pl.wf + o.w_off));
        return true;
    }
    const int nn = pick_nn(a.cout);
    static const bool gen1f = getenv("DNNCA_TCONV_FWD1") != nullptr;          // A/B: the first-generation kernel
    if (!gen1f && !o.inA.d.h && !o.out.d.h && a.cin % ig::CK == 0 && (double)a.npix * 4.0 * a.cout < 2.0e9) {
        // batch statistics of the BatchNorm behind it ride in the epilogue
        const bool stats = bn_next && !dense_switches().no_bn_fusion && bn_self_fold_args(m, *bn_next, B, &a.bnf);
        const dim3 g2((a.npix + 127) / 128, a.cout / (16 * nn));
        m->set_variant("2n%d%s", nn, stats ? "s" : "");
        if (nn == 4) LAUNCH(m, "ig_tconv_fwd", bytes, flops, hipLaunchKernelGGL((ig::k_ig_tconv_fwd2<4>), g2, dim3(256), 0, m->stream, a));
        else if (nn == 2) LAUNCH(m, "ig_tconv_fwd", bytes, flops, hipLaunchKernelGGL((ig::k_ig_tconv_fwd2<2>), g2, dim3(256), 0, m->stream, a));
        else LAUNCH(m, "ig_tconv_fwd", bytes, flops, hipLaunchKernelGGL((ig::k_ig_tconv_fwd2<1>), g2, dim3(256), 0, m->stream, a));
        return true;
    }
    dim3 grid((a.npix + 127) / 128, a.cout / (16 * nn), 4);
    m->set_variant("n%d", nn);
    if (nn == 4) LAUNCH(m, "ig_tconv_fwd", bytes, flops, hipLaunchKernelGGL((ig::k_ig_tconv_fwd<4>), grid, dim3(256), 0, m->stream, a));
    else if (nn == 2) LAUNCH(m, "ig_tconv_fwd", bytes, flops, hipLaunchKernelGGL((ig::k_ig_tconv_fwd<2>), grid, dim3(256), 0, m->stream, a));
    else LAUNCH(m, "ig_tconv_fwd", bytes, flops, hipLaunchKernelGGL((ig::k_ig_tconv_fwd<1>), grid, dim3(256), 0, m->stream, a));
    return true;
}

bool ig_tconv_bwd(Model* m, int B, Op& o, double out_bytes, double in_bytes, double flops) {
    if (!ig_tconv_supported(m, o)) return false;
    ig::TcArgs a = tc_args(m, B, o);
    if (use_bf16_tc(m, o)) {
        IgPlan& pl = g_ig[m];
        const int ntiles = (a.npix + 127) / 128, combos = (a.cout / 64) * (a.cin / 64);
        int ps = (256 + combos - 1) / combos;
        if (ps > ntiles) ps = ntiles;
        a.psplit = ps < 1 ? 1 : ps;
        const dim3 gw(a.psplit, a.cout / 64, a.cin / 64), gd((a.npix + 127) / 128, a.cin / 64);
        const bool xh = o.inA.d.h != 0, gh = o.out.g.h != 0;
#define TW64(XH, GH) LAUNCH(m, "igb_tconv_wgrad", out_bytes + in_bytes, flops, hipLaunchKernelGGL((igb::k_igb_tconv_wgrad64<XH, GH>), gw, dim3(256), 0, m->stream, a))
        hipStream_t main_stream = m->stream;
        const bool side = m->wg_side_begin();          // a leaf of the backward pass: on the side stream where the step allows
        m->set_variant("x%dg%d", (int)xh, (int)gh);
        if (xh) { if (gh) TW64(true, true); else TW64(true, false); }
        else { if (gh) TW64(false, true); else TW64(false, false); }
        if (side) m->wg_side_end(main_stream);
#undef TW64
        static const bool k32 = getenv("DNNCA_TCONV_DGRAD_K32") != nullptr;          // A/B: 32-channel chunks
        const bool k64 = gh && !k32 && a.cout % 64 == 0;
        m->set_variant("g%d%s", (int)gh, k64 ? "k64" : "");
        if (k64)
            LAUNCH(m, "igb_tconv_dgrad", out_bytes + in_bytes, flops, hipLaunchKernelGGL((igb::k_igb_tconv_dgrad<true, 64>), gd, dim3(256), 0, m->stream, a, pl.wd + o.w_off));
        else if (gh)
            LAUNCH(m, "igb_tconv_dgrad", out_bytes + in_bytes, flops, hipLaunchKernelGGL((igb::k_igb_tconv_dgrad<true, 32>), gd, dim3(256), 0, m->stream, a, pl.wd + o.w_off));
        else
            LAUNCH(m, "igb_tconv_dgrad", out_bytes + in_bytes, flops, hipLaunchKernelGGL((igb::k_igb_tconv_dgrad<false, 32>), gd, dim3(256), 0, m->stream, a, pl.wd + o.w_off));
        return true;
    }
    hipStream_t main_stream = m->stream;
    const bool side = m->wg_side_begin();
    if ((double)a.npix * 4.0 * a.cout * 4.0 < 2.0e9 && (double)a.npix * a.cin * 4.0 < 2.0e9 && !dense_switches().tcwgrad1) {
        // second generation: (16 mw x 16 nn) channel tiles x 4 parities, 32-bit byte offsets
        const int mw = a.cout % 64 == 0 ? 4 : (a.cout % 32 == 0 ? 2 : 1), nn = pick_nn(a.cin);
        const int tm = (4 / mw) < (4 / nn) ? (4 / mw) : (4 / nn);
        const int ntiles = (a.npix + 64 * tm - 1) / (64 * tm);
        const int combos = (a.cout / (16 * mw)) * (a.cin / (16 * nn));
        int ps = (256 + combos - 1) / combos;
        if (ps > ntiles) ps = ntiles;
        a.psplit = ps < 1 ? 1 : ps;
        dim3 g2(a.psplit, a.cout / (16 * mw), a.cin / (16 * nn));
#define TW2(MWv, NNv) LAUNCH(m, "ig_tconv_wgrad2", out_bytes + in_bytes, flops, hipLaunchKernelGGL((ig::k_ig_tconv_wgrad2<MWv, NNv>), g2, dim3(256), 0, m->stream, a))
        m->set_variant("m%dn%d", mw, nn);
        if (mw == 4) { if (nn == 4) TW2(4, 4); else if (nn == 2) TW2(4, 2); else TW2(4, 1); }
        else if (mw == 2) { if (nn == 4) TW2(2, 4); else if (nn == 2) TW2(2, 2); else TW2(2, 1); }
        else { if (nn == 4) TW2(1, 4); else if (nn == 2) TW2(1, 2); else TW2(1, 1); }
#undef TW2
    } else {
        const int nn = pick_nn(a.cin);
        const int ntiles = (a.npix + 127) / 128;
        const int combos = (a.cout / 16) * 4 * (a.cin / (16 * nn));
        int psplit = (1024 + combos - 1) / combos;
        if (psplit > ntiles) psplit = ntiles;
        a.psplit = psplit < 1 ? 1 : psplit;
        dim3 grid(a.psplit, a.cout / 16, 4 * (a.cin / (16 * nn)));
        m->set_variant("n%d", nn);
        if (nn == 4) LAUNCH(m, "ig_tconv_wgrad", out_bytes + in_bytes, flops, hipLaunchKernelGGL((ig::k_ig_tconv_wgrad<4>), grid, dim3(256), 0, m->stream, a));
        else if (nn == 2) LAUNCH(m, "ig_tconv_wgrad", out_bytes + in_bytes, flops, hipLaunchKernelGGL((ig::k_ig_tconv_wgrad<2>), grid, dim3(256), 0, m->stream, a));
        else LAUNCH(m, "ig_tconv_wgrad", out_bytes + in_bytes, flops, hipLaunchKernelGGL((ig::k_ig_tconv_wgrad<1>), grid, dim3(256), 0, m->stream, a));
    }
    if (side) m->wg_side_end(main_stream);
    {
        const int nn = pick_nn(a.cin);
        dim3 grid((a.npix + 127) / 128, a.cin / (16 * nn));
        m->set_variant("n%d", nn);
        if (nn == 4) LAUNCH(m, "ig_tconv_dgrad", out_bytes + in_bytes, flops, hipLaunchKernelGGL((ig::k_ig_tconv_dgrad<4>), grid, dim3(256), 0, m->stream, a));
        else if (nn == 2) LAUNCH(m, "ig_tconv_dgrad", out_bytes + in_bytes, flops, hipLaunchKernelGGL((ig::k_ig_tconv_dgrad<2>), grid, dim3(256), 0, m->stream, a));
        else LAUNCH(m, "ig_tconv_dgrad", out_bytes + in_bytes, flops, hipLaunchKernelGGL((ig::k_ig_tconv_dgrad<1>), grid, dim3(256), 0, m->stream, a));
    }
    return true;
}

}  // namespace dnnca

#ifdef DNNCA_TUNING
extern "C" int dnnca_debug_wg_stamps(unsigned long long* out, int n) {
    if (n > 64 * 8) n = 64 * 8;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(dnnca::ig::g_wg_stamps), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
extern "C" int dnnca_debug_ig_stamps(unsigned long long* out, int n) {
    if (n > 64 * 8) n = 64 * 8;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(dnnca::igb::g_ig_stamps), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
#endif
