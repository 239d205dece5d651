// kernels_fused.hip -- block-fused forward kernels for configs/unet.yaml (3/6/12 channels).
//
// One launch per reference block instead of one per Keras layer:
//   k_fz_down   Downsample.call (components.py:77-81):  conv3x3+act -> conv3x3+act (= skip) -> MaxPool2D([2,2], 2)
//   k_fz_up     Upsample.call   (components.py:158-166): Conv2DTranspose(k = s = 2) -> concat([up, skip]) -> conv3x3+act -> conv3x3+act
// The intermediate tensors of a block never come back from HBM: a block tile lives in LDS from the first layer's input to the
// last layer's output (halo recompute: the first conv of a block is evaluated on a tile one pixel larger on every side).
// They are still WRITTEN once (the backward pass reads conv outputs for act' and as weight-gradient operands); an inference
// pass (`store_mid` = 0) skips those stores.  At 128^2 / 256^2 the per-layer kernels are pure latency chains (launch -> operand
// loads -> tile loads -> MFMA -> transpose -> store, every tile of the layer resident at once): fusing a block replaces three
// chains by one.  At 512^2 it removes the re-reads: encoder block 43 -> 31 B per pixel, decoder block 78 -> 54 B per pixel.
//
// The arithmetic is the pixel-group GEMM of kernels_mfma.hip (same prepared B operands, same fmaf order per output as
// k_pgfwd: fp32 MFMA is an fmaf chain), so logits agree with the per-layer kernels to the last bit wherever the summation
// order over K is the same (it is: K runs over (source, dy, window offset) in both).
//
// LDS tiles are NHWC rows: pixel p of a row sits at LEAD + p*C floats, LEAD chosen so that the tile's first IMAGE pixel is
// 16-byte aligned in global memory (tiles are staged and stored with 16-byte vectors over the aligned superset).  Every
// float of every tile is initialised once at kernel start (zeros) and only ever overwritten with finite values: the
// K-padding slots of the MFMA A operand read up to 3 floats past a window and multiply them by zero B rows.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fast.h"
#include "fz_dev.h"
#include "kernels.h"

namespace dnnca {
namespace fz {

// tuning builds (DNNCA_TUNING=1 python -m dnncancerannotator_amd.build): s_memtime stamps of wave 0 of every block, first two tiles
#ifdef DNNCA_TUNING
__device__ unsigned long long g_fz_stamps[1024 * 2 * 8];
__device__ __forceinline__ unsigned long long fz_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define FZ_STAMP(i)                                                                                              \
    do {                                                                                                         \
        if (wave == 0 && lane == 0 && it < 2 && blockIdx.x < 1024) g_fz_stamps[(blockIdx.x * 2 + it) * 8 + (i)] = fz_now(); \
    } while (0)
#else
#define FZ_STAMP(i) do { } while (0)
#endif

struct DownArgs {
    const float* x;          // [B, H, W, CIN]
    const float* bmat1;      // prepared B operands of conv1 / conv2 (kernels_mfma.hip k_pg_prep)
    const float* bmat2;
    const float* bias1;
    const float* bias2;
    float* y0;               // conv1 output [B, H, W, C1] (nullptr: not stored -- inference)
    float* y1;               // conv2 output = skip [B, H, W, C1]
    float* pool;             // [B, H/2, W/2, C1]
    unsigned char* pool_idx; // [B, H/2, W/2, C1] window position (0..3, row-major) of each pooled value's FIRST maximum, or nullptr:
                             // the backward pass routes by it (k_pgbwd PF) instead of running a pool-backward launch
    int B, H, W, tiles_x, tiles_y;
    float alpha1, alpha2;
    // first block of a train step: the labels [B, H, W] of every tile are read alongside (tiles partition the image) and their
    // (sum, min, max) leave the block as one row of a partials table -- utils/losses.py:87-102 without a launch of its own
    const float* labels;
    float* label_part;       // [gridDim.x][4]
};

template <int CIN, int C1, int TW, int TH, int NT, int MINW>
__global__ __launch_bounds__(NT, MINW) void k_fz_down(DownArgs p) {
    constexpr int NW = NT / 64;
    constexpr int G = 12 / C1, RG = even_up(cdiv(TW + 5, G)), RG2 = TW / G, MPR = RG2 / 16;
    static_assert(RG2 % 16 == 0, "the block tile must be whole M-tiles wide");
    using TI = Tile<CIN, G, RG, TH + 4, 2>;
    using T1 = Tile<C1, G, RG, TH + 2, 1>;
    using T2 = Tile<C1, G, RG2, TH, 0>;                   // exact rows: only the stores and the pool read it
    using CV1 = Conv3<CIN, 1, C1, RG, TH + 2, TI::LEAD, 0, T1::LEAD, NW, 0>;
    using CV2 = Conv3<C1, 1, C1, RG, TH, T1::LEAD, 0, T2::LEAD, NW, MPR>;
    using ST = Stager<CIN, G, RG, TH + 4, 2, TW + 4, TI::LS, NT>;
    constexpr int KS1 = CV1::KS, KS2 = CV2::KS;
    constexpr int T2N = up4(TH * T2::LS);
    __shared__ float4 lds4[TI::N4 + T1::N4 + T2N / 4];
    float* tin = reinterpret_cast<float*>(lds4);
    float* t1 = tin + TI::N;
    float* t2 = t1 + T1::N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntiles = p.tiles_x * p.tiles_y * p.B;
    const bool xcd_map = (ntiles & 7) == 0 && (gridDim.x & 7) == 0;

    ST st;
    // label statistics (first encoder block of a train step): a tile's TH x TW labels are TH*TW/4 16-byte vectors, one per thread,
    // prefetched with the tile and folded into per-thread (sum, min, max) when the tile is committed
    constexpr int LAB4 = TH * TW / 4;
    static_assert(LAB4 <= NT, "one label vector per thread");
    float4 ylab = make_float4(0.f, 0.f, 0.f, 0.f);
    float ysum = 0.f, ymin = INFINITY, ymax = -INFINITY;
    auto issue_labels = [&](int bb, int yy0, int xx0) {
        if (p.labels && tid < LAB4) {
            const int r = tid / (TW / 4), c4 = tid - r * (TW / 4);
            ylab = *reinterpret_cast<const float4*>(p.labels + ((size_t)bb * p.H + yy0 + r) * p.W + xx0 + 4 * c4);
        }
    };
    auto fold_labels = [&]() {
        if (p.labels && tid < LAB4) {
            ysum += (ylab.x + ylab.y) + (ylab.z + ylab.w);
            ymin = fminf(fminf(ymin, fminf(ylab.x, ylab.y)), fminf(ylab.z, ylab.w));
            ymax = fmaxf(fmaxf(ymax, fmaxf(ylab.x, ylab.y)), fmaxf(ylab.z, ylab.w));
        }
    };
    int tile = blockIdx.x;
    int b, x0, y0;
    if (tile < ntiles) {      // the first tile's loads fly during the prologue
        decode_tile(tile, ntiles, xcd_map, p.tiles_x, p.tiles_y, TW, TH, b, x0, y0);
        issue_labels(b, y0, x0);
        st.issue(p.x, b, y0, x0, p.H, p.W, tid);
    }
    float breg1[KS1], breg2[KS2];
    load_breg<KS1>(breg1, p.bmat1, lane);
    load_breg<KS2>(breg2, p.bmat2, lane);
    const int co = (lane & 15) % C1;
    float bias1 = p.bias1[co], bias2 = p.bias2[co];
    for (int i = tid; i < TI::N4 + T1::N4 + T2N / 4; i += NT) lds4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pin_breg<KS1>(breg1);
    pin_breg<KS2>(breg2);
    asm volatile("" : "+v"(bias1), "+v"(bias2));
    __syncthreads();

    // software pipeline: the registers hold the NEXT tile (loads issued one iteration ago) while LDS holds the current one; the
    // commit of the next tile sits between the last conv and the stores, so that its wait for the loads (vmcnt retires in order)
    // does not also wait for stores that were issued a moment ago
    int cb = b, cx0 = x0, cy0 = y0;
    if (tile < ntiles) {
        st.commit(tin, tid);
        fold_labels();
        tile += gridDim.x;
        if (tile < ntiles) {
            decode_tile(tile, ntiles, xcd_map, p.tiles_x, p.tiles_y, TW, TH, b, x0, y0);
            issue_labels(b, y0, x0);
            st.issue(p.x, b, y0, x0, p.H, p.W, tid);
        }
        lds_barrier();
    }
    int it = 0;
    (void)it;
#pragma unroll 1
    for (; blockIdx.x < (unsigned)ntiles; ++it) {
        const unsigned edge = tile_edge(cx0, cy0, TW, TH, p.H, p.W);
        FZ_STAMP(0);
        // conv1 on the tile enlarged by one pixel on every side -> t1
        CV1::run(tin, t1, breg1, bias1, p.alpha1, wave, lane);
        lds_barrier();
        if (edge) {            // block-uniform: border tiles only
            zero_ring<C1, G, RG, TH + 2, TW + 2, T1::LEAD, NT>(t1, edge, tid);
            lds_barrier();
        }
        FZ_STAMP(1);
        CV2::run(t1, t2, breg2, bias2, p.alpha2, wave, lane);
        lds_barrier();
        FZ_STAMP(2);
        const bool more = tile < ntiles;                  // the registers hold a tile
        const int nb_ = b, nx0 = x0, ny0 = y0;
        if (more) {
            st.commit(tin, tid);                          // tin is free: conv1 is done
            fold_labels();
            tile += gridDim.x;
            if (tile < ntiles) {
                decode_tile(tile, ntiles, xcd_map, p.tiles_x, p.tiles_y, TW, TH, b, x0, y0);
                issue_labels(b, y0, x0);
                st.issue(p.x, b, y0, x0, p.H, p.W, tid);
            }
        }
        FZ_STAMP(3);
        if (p.y0) store_interior<C1, G, RG, TH + 2, 1, TW, TH, NT>(t1, p.y0, cb, cy0, cx0, p.H, p.W, tid);
        store_interior<C1, G, RG2, TH, 0, TW, TH, NT>(t2, p.y1, cb, cy0, cx0, p.H, p.W, tid);
        {   // MaxPool2D([2,2], 2) of the tile: TH/2 rows of TW/2 pixels
            constexpr int PR4 = (TW / 2) * C1 / 4;
            const int Wp = p.W >> 1, Hp = p.H >> 1;
            float* pb = p.pool + ((size_t)cb * Hp + (cy0 >> 1)) * Wp * C1 + (size_t)(cx0 >> 1) * C1;
            for (int idx = tid; idx < (TH / 2) * PR4; idx += NT) {
                const int r = idx / PR4, c4 = idx - r * PR4;
                float o[4];
                unsigned where = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int f = 4 * c4 + e, pp = f / C1, c = f - pp * C1;
                    const float* a = t2 + (2 * r) * T2::LS + T2::LEAD + (2 * pp) * C1 + c;
                    const float a0 = a[0], a1 = a[C1], a2 = a[T2::LS], a3 = a[T2::LS + C1];
                    const float mx = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
                    o[e] = mx;
                    where |= (a0 == mx ? 0u : (a1 == mx ? 1u : (a2 == mx ? 2u : 3u))) << (8 * e);     // the order g_pool_bwd searches in
                }
                *reinterpret_cast<float4*>(pb + (size_t)r * Wp * C1 + 4 * c4) = make_float4(o[0], o[1], o[2], o[3]);
                if (p.pool_idx)
                    *reinterpret_cast<unsigned*>(p.pool_idx + ((size_t)cb * Hp + (cy0 >> 1) + r) * Wp * C1 + (size_t)(cx0 >> 1) * C1 + 4 * c4) = where;
            }
        }
        if (!more) break;
        lds_barrier();        // t1 / t2 are overwritten by the next tile only after every store has read them; tin is complete
        FZ_STAMP(4);
        cb = nb_; cx0 = nx0; cy0 = ny0;
    }
    if (p.labels) {           // block partials of the label statistics: wave shuffles, the waves through LDS, one table row
        for (int o = 32; o > 0; o >>= 1) {
            ysum += __shfl_down(ysum, o, 64);
            ymin = fminf(ymin, __shfl_down(ymin, o, 64));
            ymax = fmaxf(ymax, __shfl_down(ymax, o, 64));
        }
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds4);
        if (lane == 0) { red[wave] = ysum; red[NW + wave] = ymin; red[2 * NW + wave] = ymax; }
        __syncthreads();
        if (tid == 0) {
            float s = 0.f, mn = INFINITY, mx = -INFINITY;
            for (int w = 0; w < NW; ++w) { s += red[w]; mn = fminf(mn, red[NW + w]); mx = fmaxf(mx, red[2 * NW + w]); }
            reinterpret_cast<float4*>(p.label_part)[blockIdx.x] = make_float4(s, mn, mx, 0.f);
        }
    }
}

struct UpArgs {
    const float* low;        // transposed conv input [B, H/2, W/2, CIN]
    const float* skip;       // [B, H, W, F]
    const float* wt;         // Conv2DTranspose kernel [2][2][F][CIN] and bias [F] (parameter vector)
    const float* bt;
    const float* bmat0;      // conv0 ([up | skip] -> F) and conv1 (F -> F) B operands
    const float* bmat1;
    const float* bias0;
    const float* bias1;
    const float* upin;       // NOTC: the transposed conv's output [B, H, W, F], computed by an earlier launch (the 12-channel block's riding tconv)
    float* up;               // transposed conv output [B, H, W, F]   (nullptr: not stored)
    float* y0;               // conv0 output                          (nullptr: not stored)
    float* y1;               // conv1 output
    int B, H, W, tiles_x, tiles_y;
    float alpha0, alpha1;
    // NF > 0: the next Upsample block's Conv2DTranspose(F -> NF) rides in the epilogue (its input is this block's output tile)
    const float* nwt;        // [2][2][NF][F]
    const float* nbt;        // [NF]
    float* nup;              // [B, 2H, 2W, NF]
};

// NOTC: the block WITHOUT its transposed conv -- conv0 over [up | skip] and conv1 only; the up-sampled tensor comes from memory (upin:
// the previous decoder block's launch computed it in its epilogue, NF there).  The up-sampled tile is then staged like the skip tile,
// both are committed right after conv0, and t2 gets LDS of its own (with the transposed conv inside, t2 aliases the up-sampled tile).
template <int CIN, int F, int TW, int TH, int NT, int MINW, int NF = 0, bool NOTC = false>
__global__ __launch_bounds__(NT, MINW) void k_fz_up(UpArgs p) {
    constexpr int NW = NT / 64;
    constexpr int G = 12 / F, RG = even_up(cdiv(TW + 5, G)), RG2 = TW / G, MPR = RG2 / 16;
    static_assert(RG2 % 16 == 0, "the block tile must be whole M-tiles wide");
    // the low-resolution input tile is only read by the transposed conv (vector ALU): plain rows, halo 1 (= 2 output pixels)
    // rows of LWP pixels back to back, so that low pixel p sits at LLEAD + p*CIN: the transposed conv is a GEMM over the pixel list
    constexpr int LW = TW / 2 + 2, LH = TH / 2 + 2, LLEAD = (4 - CIN % 4) % 4, LWP = low_row_pixels(LW, CIN, LLEAD), LLS = LWP * CIN;
    constexpr int NMTL = cdiv(LH * LWP, 16), LN = up4(LLEAD + NMTL * 16 * CIN + 8);
    constexpr int NB = cdiv(4 * F, 16), KT = cdiv(CIN, 4);
    using TU = Tile<F, G, RG, TH + 4, 2>;                 // up-sampled tile and skip tile: same geometry, adjacent in LDS
    using T1 = Tile<F, G, RG, TH + 2, 1>;
    using T2 = Tile<F, G, RG2, TH, 0>;
    using CV0 = Conv3<F, 2, F, RG, TH + 2, TU::LEAD, TU::N, T1::LEAD, NW, 0>;
    using CV1 = Conv3<F, 1, F, RG, TH, T1::LEAD, 0, T2::LEAD, NW, MPR>;
    // Stager of the low tile: a "tile" of 1-pixel groups with LLS floats per row
    using STL = Stager<CIN, 1, 1, LH, 1, LW, LLS, NT>;
    using STS = Stager<F, G, RG, TH + 4, 2, TW + 4, TU::LS, NT>;
    static_assert(STL::LEAD == LLEAD, "low tile lead");
    constexpr int KS0 = CV0::KS, KS1 = CV1::KS;
    static_assert(TH * T2::LS <= TU::N, "t2 aliases the up-sampled tile");
    constexpr int T2N = up4(T2::LEAD + TH * T2::LS + 8);
    constexpr int R0 = NOTC ? T2N : LN;                   // first LDS region: the low-resolution tile, or (NOTC) t2
    __shared__ float4 lds4[R0 / 4 + 2 * TU::N4 + T1::N4];
    float* tlow = reinterpret_cast<float*>(lds4);
    float* tup = tlow + R0;
    float* tskip = tup + TU::N;
    float* t1 = tskip + TU::N;
    float* t2 = NOTC ? tlow : tup;                        // (tup is dead by the time conv1 runs: conv0 has consumed it)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntiles = p.tiles_x * p.tiles_y * p.B;
    const bool xcd_map = (ntiles & 7) == 0 && (gridDim.x & 7) == 0;

    STL stl;
    STS sts, stu;
    int tile = blockIdx.x;
    int b, x0, y0;
    auto issue_inputs = [&]() {
        if constexpr (NOTC) stu.issue(p.upin, b, y0, x0, p.H, p.W, tid);
        else stl.issue(p.low, b, y0 >> 1, x0 >> 1, p.H >> 1, p.W >> 1, tid);
        sts.issue(p.skip, b, y0, x0, p.H, p.W, tid);
    };
    auto commit_inputs = [&]() {
        if constexpr (NOTC) stu.commit(tup, tid);
        else stl.commit(tlow, tid);
        sts.commit(tskip, tid);
    };
    if (tile < ntiles) {
        decode_tile(tile, ntiles, xcd_map, p.tiles_x, p.tiles_y, TW, TH, b, x0, y0);
        issue_inputs();
    }
    float breg0[KS0], breg1[KS1];
    load_breg<KS0>(breg0, p.bmat0, lane);
    load_breg<KS1>(breg1, p.bmat1, lane);
    const int co = (lane & 15) % F;
    float bias0 = p.bias0[co], bias1 = p.bias1[co];
    // Conv2DTranspose(k = s = 2) as a GEMM on the matrix cores: M = low-resolution pixels, K = CIN, N = output rows (a, e, co);
    // B[ci][(a, e, co)] = W[a][e][co][ci] straight from the parameter vector, one register per (N block, K step)
    float tw[NB][KT], tbias[NB];
    int toff[NB];              // where output row (a, e, co) lands relative to up-tile pixel (2 li, 2 lj); -1: padding column
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int rr = nb * 16 + (lane & 15);
        const bool rv = rr < 4 * F;
        const int ae = rv ? rr / F : 0, cc = rv ? rr - ae * F : 0;
        tbias[nb] = (!NOTC && rv) ? p.bt[cc] : 0.f;          // (NOTC: no transposed conv in this launch, wt / bt are null)
        toff[nb] = rv ? (ae >> 1) * TU::LS + (ae & 1) * F + cc : -1;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const int ci = 4 * k + (lane >> 4);
            tw[nb][k] = (!NOTC && rv && ci < CIN) ? p.wt[rr * CIN + ci] : 0.f;
        }
    }

    // riding Conv2DTranspose(F -> NF): per output-row parity a one N block of columns (e, co), K = F
    constexpr int NFX = NF > 0 ? NF : 1, KT2 = F / 4;
    static_assert(NF == 0 || (2 * NF <= 16 && F % 4 == 0 && (16 * 2 * NF) % 4 == 0 && 16 * 2 * NF <= 64 * 4), "riding tconv shape");
    float tw2[2][KT2], tb2 = 0.f;
    if constexpr (NF > 0) {
        const int n = lane & 15;
        const bool nv = n < 2 * NF;
        tb2 = nv ? p.nbt[n % NFX] : 0.f;
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int k = 0; k < KT2; ++k) tw2[a2][k] = nv ? p.nwt[(a2 * 2 * NF + n) * F + 4 * k + (lane >> 4)] : 0.f;
    }

    for (int i = tid; i < R0 / 4 + 2 * TU::N4 + T1::N4; i += NT) lds4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pin_breg<KS0>(breg0);
    pin_breg<KS1>(breg1);
    asm volatile("" : "+v"(bias0), "+v"(bias1));
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        asm volatile("" : "+v"(tbias[nb]));
#pragma unroll
        for (int k = 0; k < KT; ++k) asm volatile("" : "+v"(tw[nb][k]));
    }
    if constexpr (NF > 0) {
        asm volatile("" : "+v"(tb2));
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int k = 0; k < KT2; ++k) asm volatile("" : "+v"(tw2[a2][k]));
    }
    __syncthreads();

    // software pipeline as in k_fz_down: the registers hold the next tile's low-resolution and skip inputs; they are committed
    // right after conv0 (the last reader of tlow / tskip), before this tile's stores are issued
    int cb = b, cx0 = x0, cy0 = y0;
    if (tile < ntiles) {
        commit_inputs();
        tile += gridDim.x;
        if (tile < ntiles) {
            decode_tile(tile, ntiles, xcd_map, p.tiles_x, p.tiles_y, TW, TH, b, x0, y0);
            issue_inputs();
        }
        lds_barrier();
    } else {
        return;
    }
    int it = 0;
    (void)it;
#pragma unroll 1
    for (;; ++it) {
        const unsigned edge = tile_edge(cx0, cy0, TW, TH, p.H, p.W);
        FZ_STAMP(0);
        // Conv2DTranspose(k = s = 2, no activation): up[2i + a][2j + e][co] = bias[co] + sum_ci low[i][j][ci] W[a][e][co][ci]
        // for every pixel of the (TH + 4) x (TW + 4) tile, zeros outside the image
#pragma unroll 1
        for (int mt = NOTC ? NMTL : wave; mt < NMTL; mt += NW) {          // (NOTC: the up-sampled tile was staged)
            const float* ap = tlow + LLEAD + (mt * 16 + (lane & 15)) * CIN + (lane >> 4);
            f32x4 tacc[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) tacc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const float av = ap[4 * k];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) tacc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, tw[nb][k], tacc[nb], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pp = mt * 16 + 4 * (lane >> 4) + r;
                const int li = pp / LWP, lj = pp - li * LWP;
                if (li < LH && lj < LW) {
                    const int gi = (cy0 >> 1) - 1 + li, gj = (cx0 >> 1) - 1 + lj;
                    const bool inside = (unsigned)gi < (unsigned)(p.H >> 1) && (unsigned)gj < (unsigned)(p.W >> 1);
                    float* ob = tup + TU::LEAD + (2 * li) * TU::LS + (2 * lj) * F;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        if (toff[nb] >= 0) ob[toff[nb]] = inside ? tacc[nb][r] + tbias[nb] : 0.f;
                }
            }
        }
        lds_barrier();
        FZ_STAMP(1);
        // conv0 over concat([up, skip]) (components.py:164: up-sampled first) on the tile enlarged by one pixel -> t1
        CV0::run(tup, t1, breg0, bias0, p.alpha0, wave, lane);
        lds_barrier();                                    // tlow / tskip are free, every wave has finished reading tup
        FZ_STAMP(2);
        const bool more = tile < ntiles;
        const int nb_ = b, nx0 = x0, ny0 = y0;
        if (more) {
            commit_inputs();
            tile += gridDim.x;
            if (tile < ntiles) {
                decode_tile(tile, ntiles, xcd_map, p.tiles_x, p.tiles_y, TW, TH, b, x0, y0);
                issue_inputs();
            }
        }
        if (!NOTC && p.up) store_interior<F, G, RG, TH + 4, 2, TW, TH, NT>(tup, p.up, cb, cy0, cx0, p.H, p.W, tid);
        if (edge) zero_ring<F, G, RG, TH + 2, TW + 2, T1::LEAD, NT>(t1, edge, tid);
        lds_barrier();                                    // t2 (= tup) may be overwritten; the ring is in place
        FZ_STAMP(3);
        CV1::run(t1, t2, breg1, bias1, p.alpha1, wave, lane);
        if (p.y0) store_interior<F, G, RG, TH + 2, 1, TW, TH, NT>(t1, p.y0, cb, cy0, cx0, p.H, p.W, tid);
        lds_barrier();
        FZ_STAMP(4);
        store_interior<F, G, RG2, TH, 0, TW, TH, NT>(t2, p.y1, cb, cy0, cx0, p.H, p.W, tid);
        if constexpr (NF > 0) {
            // next block's Conv2DTranspose (components.py:161) of the finished tile: out[2i + a][2j + e][co] = bias[co] +
            // sum_c t2[i][j][c] W[a][e][co][c].  Per M-tile (16 pixels of one row) and a the 16 x 2 x NF results are one contiguous
            // run of output row 2i + a: they go through a per-wave slice of t1 (free since conv1) and leave as float4.
            static_assert(G == 1 && T1::N >= (NT / 64) * 32 * NF, "riding tconv scratch");
            float* scr = t1 + wave * (32 * NF);
            const int n = lane & 15, q = lane >> 4;
#pragma unroll 1
            for (int mt = wave; mt < TH * MPR; mt += NW) {
                const int ty = mt / MPR, mx = mt - ty * MPR;
                const float* ap = t2 + T2::LEAD + ty * T2::LS + (mx * 16 + n) * F + q;
                f32x4 acc2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int k = 0; k < KT2; ++k) {
                    const float av = ap[4 * k];
#pragma unroll
                    for (int a2 = 0; a2 < 2; ++a2) acc2[a2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, tw2[a2][k], acc2[a2], 0, 0, 0);
                }
#pragma unroll
                for (int a2 = 0; a2 < 2; ++a2) {
                    if (n < 2 * NF) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) scr[(4 * q + r) * 2 * NF + n] = acc2[a2][r] + tb2;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    if (lane < 8 * NF) {
                        const float4 v = reinterpret_cast<const float4*>(scr)[lane];
                        const size_t row = (size_t)cb * (2 * p.H) + 2 * (cy0 + ty) + a2;
                        reinterpret_cast<float4*>(p.nup + (row * (2 * p.W) + 2 * (cx0 + mx * 16)) * NF)[lane] = v;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (!more) break;
        // t2 shares its floats with tup; what it leaves in the padding pixels is finite (activations), which is all the junk
        // groups and the K-padding of the A operand need
        lds_barrier();
        FZ_STAMP(5);
        cb = nb_; cx0 = nx0; cy0 = ny0;
    }
}

}  // namespace fz

// ================================================================================================ host side
const float* fast_conv_bmat(Model* m, const Op& o);       // kernels_mfma.hip: prepared forward B operand of a pixel-group conv

static inline bool dense(const View& v) { return v.C == 0 || v.ps == v.C; }

template <typename K>
static int fz_resident(K kernel, int nt) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, nt, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (per_cu > 6) per_cu = 6;
    return 256 * per_cu;
}

static bool fz_enabled() {
    static const bool on = getenv("DNNCA_NO_FUSED") == nullptr;
    return on;
}
// tuning aid: DNNCA_FZ_ONLY=down0|down1|down2|up0|up1|up2 fuses only that block (level = log2(512 / block height) at 512 x 512)
static bool fz_selected(const char* kind, int level) {
    const char* e = getenv("DNNCA_FZ_ONLY");
    if (!e) return true;
    char want[16];
    snprintf(want, sizeof(want), "%s%d", kind, level);
    return strcmp(e, want) == 0;
}

// ops[oi .. oi+2] = conv3x3(CIN -> C1), conv3x3(C1 -> C1), MaxPool2D(2) of one Downsample block without BatchNorm?
// Launches the fused kernel and returns true; false: not this shape (the caller runs the layers one by one).
bool fused_down_fwd(Model* m, int B, size_t oi, bool store_mid, const float* labels) {
    if (!fz_enabled() || (m->desc.flags & 1) || m->desc.dtype != DNNCA_F32) return false;
    if (oi + 2 >= m->ops.size()) return false;
    Op &c1 = m->ops[oi], &c2 = m->ops[oi + 1], &pl = m->ops[oi + 2];
    if (c1.type != OP_CONV || c2.type != OP_CONV || pl.type != OP_POOL || c1.k != 3 || c2.k != 3 || pl.k != 2) return false;
    if (c1.inB.d.C || c2.inB.d.C || c2.inA.d.p != c1.out.d.p || pl.inA.d.p != c2.out.d.p) return false;
    if (!dense(c1.inA.d) || !dense(c1.out.d) || !dense(c2.out.d) || !dense(pl.out.d)) return false;
    const int CIN = c1.inA.d.C, C1 = c1.out.d.C, H = c1.out.d.H, W = c1.out.d.W;
    if (c2.out.d.C != C1) return false;
    if (!fz_selected("down", C1 == 3 ? 0 : (C1 == 6 ? 1 : 2))) return false;
    const float *b1 = fast_conv_bmat(m, c1), *b2 = fast_conv_bmat(m, c2);
    if (!b1 || !b2) return false;
    fz::DownArgs a{};
    a.x = c1.inA.d.p;
    a.bmat1 = b1; a.bmat2 = b2;
    a.bias1 = m->p + c1.b_off; a.bias2 = m->p + c2.b_off;
    a.y0 = store_mid ? c1.out.d.p : nullptr;
    a.y1 = c2.out.d.p;
    a.pool = pl.out.d.p;
    pl.pool_idx_valid = false;
    if (store_mid && !m->dry) {       // a backward pass follows: record where every maximum sits
        if (!pl.pool_idx) {
            void* ix = nullptr;
            if (m->alloc(&ix, (size_t)m->desc.max_batch * pl.out.d.H * pl.out.d.W * C1 + 16) == DNNCA_OK) pl.pool_idx = (unsigned char*)ix;
        }
        a.pool_idx = pl.pool_idx;
    }
    a.B = B; a.H = H; a.W = W;
    a.alpha1 = c1.alpha; a.alpha2 = c2.alpha;
    if (labels) {         // the caller (forward of a train step) wants this step's label statistics from this launch
        m->label_part_valid = false;
        if (!m->label_part && m->alloc((void**)&m->label_part, 2048 * 16) != DNNCA_OK) return false;
        a.labels = labels;
        a.label_part = m->label_part;
    }
    const double bytes = 4.0 * B * H * W * (CIN + C1 + C1 + C1 + C1 + 0.25 * C1 + (labels ? 1 : 0));      // the three layers' algorithmic bytes
    const double flops = 2.0 * B * H * W * 9.0 * (CIN * C1 + C1 * C1);
    if (CIN == 1 && C1 == 3) {          // the full-resolution block of configs/unet.yaml: the column-strip kernel (any even size)
        int nblk = 0;
        if (fast_first3_fwd(m, B, c1, c2, pl, a.y0, a.pool_idx, a.labels, a.label_part, bytes, flops, &nblk)) {
            if (labels) { m->label_part_valid = true; m->label_part_nblk = nblk; }
            pl.pool_idx_valid = store_mid && (m->dry || a.pool_idx != nullptr);
            return true;
        }
    }
#define X(cin, c1v, tw, th, nt, mw)                                                                                     \
    if (CIN == cin && C1 == c1v && W % tw == 0 && H % th == 0) {                                                  \
        a.tiles_x = W / tw; a.tiles_y = H / th;                                                                   \
        const int ntiles = a.tiles_x * a.tiles_y * B;                                                             \
        static const int fit = fz_resident(fz::k_fz_down<cin, c1v, tw, th, nt, mw>, nt);                            \
        const int g = ntiles < fit ? ntiles : fit;                                                                \
        if (labels && (g > 2048 || W != c1.out.d.W)) return false;                                                \
        LAUNCH(m, "fz_down_" #cin "_" #c1v, bytes, flops,                                                         \
               hipLaunchKernelGGL((fz::k_fz_down<cin, c1v, tw, th, nt, mw>), dim3(g), dim3(nt), 0, m->stream, a));   \
        if (labels) { m->label_part_valid = true; m->label_part_nblk = g; }                                       \
        pl.pool_idx_valid = store_mid && (m->dry || a.pool_idx != nullptr);                                       \
        return true;                                                                                              \
    }
    static const int alt = getenv("DNNCA_FZ_NT") ? atoi(getenv("DNNCA_FZ_NT")) : 0;      // tuning aid
    if (alt == 256) {
        X(1, 3, 128, 8, 256, 2) X(3, 6, 64, 8, 256, 2) X(6, 12, 32, 8, 256, 2)
    }
    X(1, 3, 128, 8, 512, 4) X(3, 6, 64, 8, 512, 4) X(6, 12, 32, 8, 512, 4)
#undef X
    return false;
}

// ops[oi .. oi+2] = Conv2DTranspose(CIN -> F, 2x2/2), conv3x3([up | skip] -> F), conv3x3(F -> F) of one Upsample block without BatchNorm?
bool fused_up_fwd(Model* m, int B, size_t oi, bool store_mid, int* consumed) {
    if (consumed) *consumed = 3;
    if (!fz_enabled() || (m->desc.flags & 1) || m->desc.dtype != DNNCA_F32) return false;
    if (oi + 2 >= m->ops.size()) return false;
    Op &tc = m->ops[oi], &c0 = m->ops[oi + 1], &c1 = m->ops[oi + 2];
    if (tc.type != OP_TCONV || c0.type != OP_CONV || c1.type != OP_CONV || tc.k != 2 || c0.k != 3 || c1.k != 3) return false;
    if (c0.inA.d.p != tc.out.d.p || !c0.inB.d.C || c1.inB.d.C || c1.inA.d.p != c0.out.d.p) return false;
    if (!dense(tc.inA.d) || !dense(tc.out.d) || !dense(c0.inB.d) || !dense(c0.out.d) || !dense(c1.out.d)) return false;
    const int CIN = tc.inA.d.C, F = tc.out.d.C, H = tc.out.d.H, W = tc.out.d.W;
    if (c0.inB.d.C != F || c0.out.d.C != F || c1.out.d.C != F || c0.inB.d.H != H || c0.inB.d.W != W) return false;
    if (!fz_selected("up", F == 3 ? 0 : (F == 6 ? 1 : 2))) return false;
    // measured on MI355X (tools/fz_ab.py, profiles/r02_fused_ab.txt): the fused decoder block beats its three per-layer launches
    // only at 128^2 (12 channels); at 256^2 / 512^2 the block is bound by the fp32 matrix pipe (two 3x3 convs on a halo-enlarged
    // tile) and the per-layer kernels, which overlap it better with their memory traffic, stay ahead.  DNNCA_FZ_ALL=1 fuses all.
    static const bool all = getenv("DNNCA_FZ_ALL") != nullptr || getenv("DNNCA_FZ_ONLY") != nullptr;
    if (!all && F != 12) return false;
    const float *b0 = fast_conv_bmat(m, c0), *b1 = fast_conv_bmat(m, c1);
    if (!b0 || !b1) return false;
    fz::UpArgs a{};
    a.low = tc.inA.d.p; a.skip = c0.inB.d.p;
    a.wt = m->p + tc.w_off; a.bt = m->p + tc.b_off;
    a.bmat0 = b0; a.bmat1 = b1;
    a.bias0 = m->p + c0.b_off; a.bias1 = m->p + c1.b_off;
    a.up = store_mid ? tc.out.d.p : nullptr;
    a.y0 = store_mid ? c0.out.d.p : nullptr;
    a.y1 = c1.out.d.p;
    a.B = B; a.H = H; a.W = W;
    a.alpha0 = c0.alpha; a.alpha1 = c1.alpha;
    double bytes = 4.0 * B * H * W * (0.25 * CIN + F + 2 * F + F + F + F);        // tconv (in + out) + conv0 (2 in + out) + conv1 (in + out)
    double flops = 2.0 * B * H * W * (F * CIN + 9.0 * (2 * F * F + F * F));
    // the next block's Conv2DTranspose(12 -> 6) rides in the epilogue (its input is c1's output tile, still in LDS)
    if (!getenv("DNNCA_NO_TCONV_RIDE") && consumed && CIN == 12 && F == 12 && W % 32 == 0 && H % 8 == 0 && oi + 3 < m->ops.size()) {
        Op& nt2 = m->ops[oi + 3];
        if (nt2.type == OP_TCONV && nt2.k == 2 && nt2.inA.d.p == c1.out.d.p && nt2.inA.d.C == 12 && nt2.out.d.C == 6 && dense(nt2.inA.d) &&
            dense(nt2.out.d) && nt2.out.d.H == 2 * H && nt2.out.d.W == 2 * W) {
            a.nwt = m->p + nt2.w_off; a.nbt = m->p + nt2.b_off; a.nup = nt2.out.d.p;
            a.tiles_x = W / 32; a.tiles_y = H / 8;
            bytes += 4.0 * B * H * W * (F + 4 * 6);
            flops += 2.0 * B * H * W * 4 * 6 * F;
            const int ntiles = a.tiles_x * a.tiles_y * B;
            static const int fit = fz_resident(fz::k_fz_up<12, 12, 32, 8, 512, 2, 6>, 512);
            const int g = ntiles < fit ? ntiles : fit;
            LAUNCH(m, "fz_up_tc_12_12", bytes, flops,
                   hipLaunchKernelGGL((fz::k_fz_up<12, 12, 32, 8, 512, 2, 6>), dim3(g), dim3(512), 0, m->stream, a));
            *consumed = 4;
            return true;
        }
    }
#define X(cin, f, tw, th, nt, mw)                                                                                       \
    if (CIN == cin && F == f && W % tw == 0 && H % th == 0) {                                                     \
        a.tiles_x = W / tw; a.tiles_y = H / th;                                                                   \
        const int ntiles = a.tiles_x * a.tiles_y * B;                                                             \
        static const int fit = fz_resident(fz::k_fz_up<cin, f, tw, th, nt, mw>, nt);                                \
        const int g = ntiles < fit ? ntiles : fit;                                                                \
        LAUNCH(m, "fz_up_" #cin "_" #f, bytes, flops,                                                             \
               hipLaunchKernelGGL((fz::k_fz_up<cin, f, tw, th, nt, mw>), dim3(g), dim3(nt), 0, m->stream, a));       \
        return true;                                                                                              \
    }
    static const int alt = getenv("DNNCA_FZ_NT") ? atoi(getenv("DNNCA_FZ_NT")) : 0;      // tuning aid
    if (alt == 256) {
        X(12, 12, 32, 8, 256, 2) X(12, 6, 64, 8, 256, 2) X(6, 3, 128, 8, 256, 2)
    }
    X(12, 12, 32, 8, 512, 2) X(12, 6, 64, 8, 512, 2) X(6, 3, 128, 8, 512, 4)
#undef X
    return false;
}

// ops[oi], ops[oi + 1] = conv3x3([up | skip] -> F), conv3x3(F -> F) of an Upsample block (components.py:162-165) whose transposed conv has
// already run (it rode in the previous block's launch, fz_up_tc_12_12): the rest of the block in one launch (k_fz_up<..., NOTC>).
bool fused_up2_fwd(Model* m, int B, size_t oi, bool store_mid) {
    // OFF by default.  Measured on MI355X (round 4, bench.py --steps 50, same box): fz_up2_6 34.4 us against pgfwd_6x2_6 + pgfwd_6x1_6
    // 21.5 + 14.3 us under event brackets, the step 0.4059 ms (14 launches) against 0.4034 ms (15): the launch it saves is paid back by
    // the halo recompute of conv0 on the fp32 matrix pipe (+29 % MFMAs on a 64 x 8 tile), as round 2 found for the whole block.
    // DNNCA_FZ_UP2=1 enables it (read per call: the test flips it).
    if (!getenv("DNNCA_FZ_UP2") || !fz_enabled() || (m->desc.flags & 1) || m->desc.dtype != DNNCA_F32) return false;
    if (oi + 1 >= m->ops.size()) return false;
    Op &c0 = m->ops[oi], &c1 = m->ops[oi + 1];
    if (c0.type != OP_CONV || c1.type != OP_CONV || c0.k != 3 || c1.k != 3) return false;
    if (!c0.inB.d.C || c1.inB.d.C || c1.inA.d.p != c0.out.d.p) return false;
    if (!dense(c0.inA.d) || !dense(c0.inB.d) || !dense(c0.out.d) || !dense(c1.out.d)) return false;
    const int F = c0.inA.d.C, H = c0.out.d.H, W = c0.out.d.W;
    if (F != 6 || c0.inB.d.C != F || c0.out.d.C != F || c1.out.d.C != F || c0.inB.d.H != H || c0.inB.d.W != W) return false;
    if (W % 64 || H % 8) return false;
    if (c0.src_bn[0] >= 0 || c0.src_bn[1] >= 0) return false;
    const float *b0 = fast_conv_bmat(m, c0), *b1 = fast_conv_bmat(m, c1);
    if (!b0 || !b1) return false;
    fz::UpArgs a{};
    a.upin = c0.inA.d.p; a.skip = c0.inB.d.p;
    a.bmat0 = b0; a.bmat1 = b1;
    a.bias0 = m->p + c0.b_off; a.bias1 = m->p + c1.b_off;
    a.y0 = store_mid ? c0.out.d.p : nullptr;
    a.y1 = c1.out.d.p;
    a.B = B; a.H = H; a.W = W;
    a.alpha0 = c0.alpha; a.alpha1 = c1.alpha;
    a.tiles_x = W / 64; a.tiles_y = H / 8;
    const double bytes = 4.0 * B * H * W * (2 * F + F + F + F);        // conv0 (2 in + out) + conv1 (in + out)
    const double flops = 2.0 * B * H * W * 9.0 * (2 * F * F + F * F);
    const int ntiles = a.tiles_x * a.tiles_y * B;
    static const int fit = fz_resident(fz::k_fz_up<12, 6, 64, 8, 512, 2, 0, true>, 512);
    const int g = ntiles < fit ? ntiles : fit;
    LAUNCH(m, "fz_up2_6", bytes, flops, hipLaunchKernelGGL((fz::k_fz_up<12, 6, 64, 8, 512, 2, 0, true>), dim3(g), dim3(512), 0, m->stream, a));
    return true;
}

}  // namespace dnnca

// development aid (not part of include/dnnca.h): the stamps of the last fused kernel of a tuning build
extern "C" int dnnca_debug_fz_stamps(unsigned long long* out, int n) {
#ifdef DNNCA_TUNING
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(dnnca::fz::g_fz_stamps), (size_t)n * 8) == hipSuccess ? 0 : -1;
#else
    (void)out; (void)n;
    return -2;
#endif
}
