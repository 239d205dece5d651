// kernels_mfma.hip -- "pixel-group" implicit-GEMM 3x3 convolutions on the fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
// configs/unet.yaml has 3/6/12 channels: as a plain implicit GEMM (M = pixels, N = Cout, K = 9*Cin) the N dimension
// would fill 3..12 of 16 MFMA columns.  Instead G = 12/Cout horizontally adjacent output pixels share one GEMM row:
//     M = pixel groups,  N = (dx, co) = G*Cout (= 12 of 16),  K = (dy, j) over the group's 3 x (G+2)*C input window,
// with B[(dy, j), (dx, co)] = W[dy][xoff - dx][ci][co] (j = xoff*C + ci) where 0 <= xoff - dx <= 2 and 0 elsewhere.
// A group's window row is (G+2)*C *contiguous* NHWC floats, so the A operand is a plain strided LDS read and a tile
// is staged global->LDS with fully coalesced 16-byte loads.
//
//   k_pgfwd   forward conv (+bias +activation; the decoder's channel concat is two sources, never materialised)
//   k_pgbwd   backward conv, ONE pass over the data: stages dz and x once, then
//               - data gradient  (same GEMM with flipped/transposed weights; act'(x) mask and accumulate fused in the store)
//               - weight gradient D[(dy, j), (dx, co)] += x_window * dz with K = pixel groups, accumulators in registers
//                 for the whole (persistent) block; an all-ones A row yields the bias gradient
//   k_tconv_wgrad   weight gradient of the 2x2/2 transposed conv as D[(a,e,co)][ci], operands straight from global
//   k_pg_fold       folds the partial D's (and their dx-diagonals) into the flat gradient vector
//
// All kernels are persistent (grid-stride over tiles) and prefetch the next tile into registers while the matrix
// cores work on the current one (one LDS buffer, several blocks per CU).
// exact fp32: the f32 MFMA is a chain of fmaf (MI355X guide, "FP32-input MFMA"), so parity with the oracle holds to
// fp32 rounding.
#include <stdio.h>
#include <stdlib.h>

#include "fast.h"
#include "kernels.h"
#include "tconv2_dev.h"

namespace dnnca {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int TH = 8;          // rows of a block tile
static constexpr int NBUCKET = kPgBuckets;    // partial-sum slabs per weight gradient (blocks add into slab blockIdx % NBUCKET)

// geometry of one staged tensor: C channels, tile of TW pixels (+1 halo pixel on each side), rows padded to 16 B
template <int C, int TW>
struct TG {
    static constexpr int HL = (C + 3) / 4 * 4;        // staged floats in front of the tile's first pixel
    static constexpr int LEAD = HL - C;               // junk floats in front of the left-halo pixel
    static constexpr int LS = (LEAD + (TW + 2) * C + 3 + 3) / 4 * 4;   // LDS row stride (floats)
    static constexpr int LS4 = LS / 4;
    static constexpr int N4 = (TH + 2) * LS4;         // float4's of the staged tile
    static constexpr int npf(int nt) { return (N4 + nt - 1) / nt; }   // prefetch registers (float4) per thread
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory queue (s_waitcnt
// vmcnt(0)), which would make every tile wait for its epilogue stores and for the prefetch loads of the next tile.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Staging of one (TH+2)-row halo tile.  The (row, 16-byte column) a thread fetches for prefetch register k depends only
// on its thread id, so it is computed once per kernel (TileMap); per tile only the tile origin changes.
template <int C, int TW, int NT>
struct TileMap {
    using T = TG<C, TW>;
    static constexpr int NPF = T::npf(NT);
    int row[NPF], c4[NPF], off[NPF];      // off = row * rowlen4 + c4: float4 offset from the tile's first staged float4
    __device__ __forceinline__ void init(int tid, int rowlen4) {
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int idx = tid + k * NT;
            row[k] = idx < T::N4 ? idx / T::LS4 : -100000;    // out-of-tile slots never pass the row test
            c4[k] = idx - (idx / T::LS4) * T::LS4;
            off[k] = idx < T::N4 ? row[k] * rowlen4 + c4[k] : 0;
        }
    }
};

// Issue the global loads of one tile into registers.  The loads are unconditional (addresses clamped into the image)
// so the issue phase is branch-free; what lies outside the image is zeroed at commit time from the returned bit mask.
template <int C, int TW, int NT>
__device__ __forceinline__ unsigned tile_issue(float4* pre, const TileMap<C, TW, NT>& mp, const float* __restrict__ src,
                                               int b, int x0, int y0, int B, int H, int W) {
    using T = TG<C, TW>;
    const int rowlen4 = W * C / 4;
    const int g40 = (x0 * C - T::HL) / 4;
    const int tbase = (b * H + y0 - 1) * rowlen4 + g40;          // float4 index of the tile's (row 0, column 0); uniform
    const int last4 = B * H * rowlen4 - 1;
    const float4* base = reinterpret_cast<const float4*>(src);
    unsigned ok = 0;
#pragma unroll
    for (int k = 0; k < T::npf(NT); ++k) {
        const int iy = y0 - 1 + mp.row[k], g4 = g40 + mp.c4[k];
        ok |= ((unsigned)iy < (unsigned)H && (unsigned)g4 < (unsigned)rowlen4) ? (1u << k) : 0u;
        pre[k] = base[min(max(tbase + mp.off[k], 0), last4)];    // clamped: what is outside the image is masked at commit
    }
    return ok;
}

template <int C, int TW, int NT>
__device__ __forceinline__ void tile_commit(const float4* pre, unsigned ok, float4* lds4, int tid) {
    using T = TG<C, TW>;
#pragma unroll
    for (int k = 0; k < T::npf(NT); ++k) {
        const int idx = tid + k * NT;
        if (idx < T::N4) lds4[idx] = (ok >> k) & 1u ? pre[k] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// ================================================================================================ forward
struct FwdArgs {
    const float* src[2];     // dense NHWC sources, C channels each
    const float* bmat;       // prepared B operand: [KS][64]
    const float* bias;       // CO floats
    float* dst;              // dense NHWC, CO channels
    float* pool_dst;         // fused MaxPool2D([2,2], 2) of the output (components.py:54): [B, H/2, W/2, CO] or nullptr
    int B, H, W;
    int tiles_x, tiles_y;
    float alpha;             // activation slope (<0: none)
    // HEAD variant (the conv that feeds the annotator head, training): the head, the weighted BCE and the head's backward ride in
    // this conv's epilogue (unet.py:241-244, losses.py:17-37) -- the feature map is not read again
    const float* hy;         // labels [B, H, W]
    const float* hw;         // head kernel (CO weights) and bias
    const float* hb;
    float* hdfeat;           // gradient of the feature map = gradient of this conv's output [B, H, W, CO]
    float* hpartials;        // [gridDim.x][CO + 2] block partial sums (dW..., db, loss), reduced by k_pg_fold
    double* hscalars;        // scalars[0] = label sum (k_label_stats ran earlier on the stream), or, with hlabel_part:
    const float* hlabel_part;  // [hlabel_nblk][4] per-block (sum, min, max, -) of the labels from the first encoder block's launch
    int hlabel_nblk;           //   -> every block sums them; block 0 also writes scalars[0..2] for the step outputs
    dnnca_loss_cfg hcfg;
    double hn_label;
    float hgscale;
    int hmask;               // multiply dfeat by act'(feat)
    float halpha;
};

template <int C, int NSRC, int CO, int NT, bool DB, bool HEAD = false>
__global__ __launch_bounds__(NT) void k_pgfwd(FwdArgs p) {
    constexpr int G = 12 / CO, TX = 2, TW = 16 * G * TX, N = G * CO, NW = NT / 64;
    using T = TG<C, TW>;
    constexpr int WR = (G + 2) * C, SR = (WR + 3) / 4, KS = NSRC * 3 * SR, LS = T::LS;
    // DB: two stage buffers -> the next tile is committed while other waves still read the current one, one barrier per tile
    constexpr int STAGE4 = NSRC * T::N4, NBUF = DB ? 2 : 1;
    __shared__ float4 lds4[NBUF * STAGE4 + NW * 96];   // staged tiles + two 16x12 output rows per wave
    float* orow = reinterpret_cast<float*>(lds4 + NBUF * STAGE4) + (threadIdx.x >> 6) * 384;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, q = lane >> 4, n = m;
    const int co = n % CO;
    const int ntiles = p.tiles_x * p.tiles_y * p.B;
    const bool xcd_map = (ntiles & 7) == 0 && (gridDim.x & 7) == 0;

    // B operand (one register per K-step) and bias first; drain them and hide their origin from the compiler:
    // otherwise hipcc keeps `s_waitcnt vmcnt(..0)` for these registers inside the tile loop, and since vmcnt retires in
    // order that wait also drains the tile prefetch issued just before the MFMAs (load latency serialised per tile).
    // (the drain sits behind the issue of the first tile's loads: one memory round trip at the start of the kernel, not two)
    float breg[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) breg[s] = p.bmat[s * 64 + lane];
    float bias = n < N ? p.bias[co] : 0.f;
    auto drain_breg = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(breg[s]));
        asm volatile("" : "+v"(bias));
    };
    if constexpr (HEAD) drain_breg();

    // HEAD: head weights, the positive-class weight (losses.py:24-29) and this lane's partial sums
    float hwv[HEAD ? CO : 1], hbias = 0.f, hwgt = 1.f, hsum[HEAD ? CO + 2 : 1];
    if constexpr (HEAD) {
        static_assert(G * 16 == 64, "HEAD: one lane per pixel of an M-tile row (CO = 3)");
#pragma unroll
        for (int c = 0; c < CO; ++c) hwv[c] = p.hw[c];
        hbias = p.hb[0];
        double lsum;
        if (p.hlabel_part) {
            // label statistics of this step: per-block partials written by k_fz_down (no same-address atomics, no launch of their own)
            double ds = 0.0;
            float mn = INFINITY, mx = -INFINITY;
            for (int i = tid; i < p.hlabel_nblk; i += NT) {
                const float4 v = reinterpret_cast<const float4*>(p.hlabel_part)[i];
                ds += (double)v.x;
                mn = fminf(mn, v.y);
                mx = fmaxf(mx, v.z);
            }
            for (int o = 32; o > 0; o >>= 1) {
                ds += __shfl_down(ds, o, 64);
                mn = fminf(mn, __shfl_down(mn, o, 64));
                mx = fmaxf(mx, __shfl_down(mx, o, 64));
            }
            double* redd = reinterpret_cast<double*>(lds4);
            float* redf = reinterpret_cast<float*>(lds4) + 2 * NW;
            if (lane == 0) { redd[wave] = ds; redf[wave] = mn; redf[NW + wave] = mx; }
            __syncthreads();
            ds = 0.0; mn = INFINITY; mx = -INFINITY;
            for (int w = 0; w < NW; ++w) { ds += redd[w]; mn = fminf(mn, redf[w]); mx = fmaxf(mx, redf[NW + w]); }
            __syncthreads();                 // the LDS words are about to hold the first staged tile
            lsum = ds;
            if (blockIdx.x == 0 && tid == 0) { p.hscalars[0] = ds; p.hscalars[1] = (double)mn; p.hscalars[2] = (double)mx; }
        } else {
            lsum = p.hscalars[0];
        }
        if (p.hcfg.has_weight) {
            hwgt = p.hcfg.weight;
        } else {
            const float pr = (float)(lsum / p.hn_label);
            hwgt = pr > 0.f ? 1.0f / pr : 1.0f;
        }
        hwgt = p.hcfg.weight_mul * hwgt + p.hcfg.weight_add;
#pragma unroll
        for (int c = 0; c < CO + 2; ++c) hsum[c] = 0.f;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int c = 0; c < CO; ++c) asm volatile("" : "+v"(hwv[c]));
        asm volatile("" : "+v"(hbias), "+v"(hwgt));
    }

    float4 pre[NSRC][T::npf(NT)];
    TileMap<C, TW, NT> mp;
    mp.init(tid, p.W * C / 4);
    unsigned okm = 0;          // in-image mask of the prefetched tile (the same for every source)
    int tile = blockIdx.x;
    auto decode = [&](int t, int& b, int& x0, int& y0) {
        // blocks b, b + 8, b + 16 ... share an XCD (and its L2): give each XCD one contiguous eighth of the tile sequence, so
        // that the halo rows of vertically adjacent tiles are fetched into the same L2
        if (xcd_map) t = (t & 7) * (ntiles >> 3) + (t >> 3);
        const int bx = t % p.tiles_x, by = (t / p.tiles_x) % p.tiles_y;
        b = t / (p.tiles_x * p.tiles_y);
        x0 = bx * TW;
        y0 = by * TH;
    };
    if (tile < ntiles) {
        int b, x0, y0;
        decode(tile, b, x0, y0);
#pragma unroll
        for (int s = 0; s < NSRC; ++s) okm = tile_issue<C, TW, NT>(pre[s], mp, p.src[s], b, x0, y0, p.B, p.H, p.W);
    }
    if constexpr (!HEAD) drain_breg();

    int buf = 0;
    if (DB && tile < ntiles) {
#pragma unroll
        for (int s = 0; s < NSRC; ++s) tile_commit<C, TW, NT>(pre[s], okm, lds4 + s * T::N4, tid);
        lds_barrier();
    }
#pragma unroll 1
    while (tile < ntiles) {
        int b, x0, y0;
        decode(tile, b, x0, y0);
        if (!DB) {
#pragma unroll
            for (int s = 0; s < NSRC; ++s) tile_commit<C, TW, NT>(pre[s], okm, lds4 + s * T::N4, tid);
            lds_barrier();
        }
        const float* lds = reinterpret_cast<const float*>(lds4 + buf * STAGE4);
        // HEAD: this tile's labels are loaded BEFORE the next tile's prefetch is issued: vmcnt retires in order, so a label load
        // behind the prefetch would make the epilogue wait for the whole prefetch (an HBM round trip per tile)
        float zlab[2] = {0.f, 0.f};
        if constexpr (HEAD) {
            const int tx_ = wave % TX, ty0_ = wave / TX;
            const int px = x0 + tx_ * 64 + lane;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int y = y0 + 2 * ty0_ + i;
                const int pxc = px < p.W ? px : p.W - 1, yc = y < p.H ? y : p.H - 1;      // clamped: masked in the epilogue
                zlab[i] = p.hy[((size_t)b * p.H + yc) * p.W + pxc];
            }
        }
        const int next = tile + gridDim.x;
        if (next < ntiles) {
            int nb, nx0, ny0;
            decode(next, nb, nx0, ny0);
#pragma unroll
            for (int s = 0; s < NSRC; ++s) okm = tile_issue<C, TW, NT>(pre[s], mp, p.src[s], nb, nx0, ny0, p.B, p.H, p.W);
        }
        // every wave owns NCH M-tiles (same column block tx, adjacent rows NCH*ty0 + i) and interleaves their MFMA chains:
        // independent accumulators keep the matrix pipe issuing back to back (a dependent 16x16x4 f32 chain stalls
        // 8 of every 40 cycles) and put all 60 LDS reads in flight at once
        {
            constexpr int NCH = TX * TH / NW;
            const int tx = wave % TX, ty0 = wave / TX;
            f32x4 acc[NCH];
#pragma unroll
            for (int i = 0; i < NCH; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int aoff = T::LEAD + (tx * 16 + m) * (G * C) + q;
#pragma unroll
            for (int s = 0; s < NSRC; ++s)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int k = 0; k < SR; ++k)
#pragma unroll
                        for (int i = 0; i < NCH; ++i) {
                            const int ty = NCH * ty0 + i;
                            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                lds[s * (T::N4 * 4) + (ty + dy) * LS + aoff + 4 * k], breg[(s * 3 + dy) * SR + k], acc[i], 0, 0, 0);
                        }
            // D[group 4q+r][n] -> the M-tile's output row is 16 groups x 12 contiguous floats: transpose through LDS
            // (same wave writes and reads; DS operations of one wave execute in order) and store 16 B per lane
            static_assert(NCH == 2, "the fused 2x2 max-pool needs each wave to own a pair of adjacent rows");
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int y = y0 + NCH * ty0 + i;
                float* orw = orow + i * 192;
                if (n < N) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[i][r] + bias;
                        orw[(4 * q + r) * 12 + n] = p.alpha < 0.f ? v : (v > 0.f ? v : p.alpha * v);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                const int f0 = (x0 + tx * 16 * G) * CO + 4 * lane;          // float index within the image row
                if (lane < 48 && f0 < p.W * CO && y < p.H) {
                    const float4 v = reinterpret_cast<const float4*>(orw)[lane];
                    *reinterpret_cast<float4*>(p.dst + ((size_t)b * p.H + y) * p.W * CO + f0) = v;
                }
                if constexpr (HEAD) {
                    // one lane = one pixel of the 64-pixel row segment: logit -> weighted BCE -> dlogit -> dW, db, d(feature map)
                    const int px = x0 + tx * 64 + lane;
                    const bool ok = px < p.W && y < p.H;
                    float f[CO], df[CO];
#pragma unroll
                    for (int c = 0; c < CO; ++c) f[c] = orw[lane * CO + c];
                    const float z = ok ? zlab[i] : 0.f;
                    float xl = hbias;
#pragma unroll
                    for (int c = 0; c < CO; ++c) xl = fmaf(f[c], hwv[c], xl);
                    const float mk = fmaf(z, hwgt - 1.0f, 1.0f);
                    const float e = expf(-fabsf(xl));
                    const float sig = xl >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
                    const float dl = ok ? mk * (sig - z) * p.hgscale : 0.f;
                    if (ok) hsum[CO + 1] = fmaf(fmaxf(xl, 0.f) - xl * z + log1pf(e), mk, hsum[CO + 1]);
                    hsum[CO] += dl;
#pragma unroll
                    for (int c = 0; c < CO; ++c) {
                        hsum[c] = fmaf(f[c], dl, hsum[c]);
                        float d = dl * hwv[c];
                        if (p.hmask) d *= f[c] > 0.f ? 1.0f : p.halpha;
                        df[c] = d;
                    }
                    __builtin_amdgcn_wave_barrier();        // every lane has read its features: the row can be overwritten
#pragma unroll
                    for (int c = 0; c < CO; ++c) orw[lane * CO + c] = df[c];
                    __builtin_amdgcn_wave_barrier();
                    if (lane < 48 && f0 < p.W * CO && y < p.H) {
                        const float4 v = reinterpret_cast<const float4*>(orw)[lane];
                        *reinterpret_cast<float4*>(p.hdfeat + ((size_t)b * p.H + y) * p.W * CO + f0) = v;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (p.pool_dst) {
                // both rows of the pair are in LDS: pooled row = max over the 2x2 windows, 8G pixels x CO = 96 floats
                const int yp = (y0 >> 1) + ty0, Wp = p.W >> 1;
                const int fp0 = ((x0 + tx * 16 * G) >> 1) * CO + 4 * lane;   // float index within the pooled row
                if (lane < 24 && fp0 < Wp * CO && 2 * yp + 1 < p.H) {
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int f = 4 * lane + e, pp = f / CO, c = f - pp * CO;
                        const int i0 = (2 * pp) * CO + c;
                        o[e] = fmaxf(fmaxf(orow[i0], orow[i0 + CO]), fmaxf(orow[192 + i0], orow[192 + i0 + CO]));
                    }
                    *reinterpret_cast<float4*>(p.pool_dst + ((size_t)b * (p.H >> 1) + yp) * Wp * CO + fp0) = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (DB) {
            if (next < ntiles) {
#pragma unroll
                for (int s = 0; s < NSRC; ++s) tile_commit<C, TW, NT>(pre[s], okm, lds4 + (buf ^ 1) * STAGE4 + s * T::N4, tid);
            }
            buf ^= 1;
        }
        lds_barrier();
        tile = next;
    }
    if constexpr (HEAD) {
        // block partial sums: wave shuffles, the waves through LDS, one row of the partials table per block
        float* red = reinterpret_cast<float*>(lds4);
#pragma unroll
        for (int k = 0; k < CO + 2; ++k) {
            float v = hsum[k];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            if (lane == 0) red[wave * (CO + 2) + k] = v;
        }
        __syncthreads();
        if (tid < CO + 2) {
            float v = 0.f;
            for (int w = 0; w < NW; ++w) v += red[w * (CO + 2) + tid];
            p.hpartials[blockIdx.x * (CO + 2) + tid] = v;
        }
    }
}

__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
// The stamps and the phase-skipping switches are compiled in only for tuning builds (DNNCA_TUNING=1 python -m
// dnncancerannotator_amd.build): in the shipped kernel they would sit as branches in the per-tile loop.
#ifdef DNNCA_TUNING
#define STAMP(i)                                                                                        \
    do {                                                                                                \
        if (p.stamps && wave == 0 && lane == 0 && it < 4) p.stamps[((size_t)blockIdx.x * 4 + it) * 8 + (i)] = stamp(); \
    } while (0)
#define STAMP_END(i)                                                                                    \
    do {                                                                                                \
        if (p.stamps && wave == 0 && lane == 0) p.stamps[((size_t)blockIdx.x * 4 + 3) * 8 + (i)] = stamp(); \
    } while (0)
#define DBG_FLAGS(p) ((p).dbg)
#else
#define STAMP(i) do { } while (0)
#define STAMP_END(i) do { } while (0)
#define DBG_FLAGS(p) 0
#endif

// ================================================================================================ backward (fused)
struct BwdArgs {
    const float* dz;         // gradient of the conv's pre-activation output, dense NHWC, CO channels
    const float* x[2];       // the conv's input sources, dense NHWC, C channels each
    const float* bmat;       // data-gradient B operands: NPASS x [KSd][64]
    float* dx[2];            // data-gradient destinations (same geometry as x)
    float* slabs[2];         // weight-gradient partial sums per source: [NBUCKET][MT*256]
    int acc[2];              // accumulate into dx instead of overwriting
    int mask[2];             // multiply the (summed) dx by act'(x)
    int B, H, W;
    int tiles_x, tiles_y;
    float alpha;             // slope of the masked activation
    // PF (pool fold): this conv's output feeds a 2x2 max-pool and a skip connection.  dz then holds only the skip gradient; the
    // staging adds the pooled gradient at the window position the forward pass recorded and applies act'(y) -- what the
    // pool-backward launch did in place (components.py:54; max-pool gradient goes to the first maximum)
    const float* pf_y;               // this conv's output [B, H, W, CO]
    const float* pf_dpool;           // gradient of the pooled tensor [B, H/2, W/2, CO]
    const unsigned char* pf_idx;     // window position of every maximum [B, H/2, W/2, CO]
    float pf_alpha;                  // slope of act'
    // TCF (the two-source 3-channel conv whose first source is the output of a 6 -> 3 transposed conv): the gradient of that source
    // stays in LDS and the transposed conv's whole backward follows in the same launch (components.py:118-127)
    const float* tc_in;      // the transposed conv's input [B, H/2, W/2, 6]
    float* tc_din;           // its gradient
    const float* tc_w;       // kernel [2][2][3][6]
    float* tc_slabs;         // weight-gradient slabs of the transposed conv [NBUCKET][256]
    int tc_mask;             // multiply tc_din by act'(tc_in)
    float tc_alpha;
    int dbg;                 // tuning aid (DNNCA_DBG): bit 0 skip the data-gradient phase, bit 1 skip the weight-gradient phase
    unsigned long long* stamps;   // tuning aid (DNNCA_STAMPS): [block][tile slot 0..3][8] s_memtime stamps of wave 0
};

template <int C, int NSRC, int CO>
struct BW {
    static constexpr int Gw = 12 / CO;                     // wgrad: N = (dx, co)
    static constexpr int TW = 32 * Gw;
    static constexpr int CTOT = NSRC * C;
    static constexpr int NPASS = CTOT <= 12 ? 1 : NSRC;    // dgrad passes (each fills N <= 16 columns)
    static constexpr int COd = CTOT / NPASS;               // dgrad output channels per pass
    static constexpr int Gd = (COd == 3 || COd == 6 || COd == 12) ? 12 / COd : (COd == 1 ? 4 : 1);
    static constexpr int WRd = (Gd + 2) * CO, SRd = (WRd + 3) / 4, KSd = 3 * SRd;
    static constexpr int MTX = TW / (16 * Gd);             // dgrad M-tiles across the tile
    static constexpr int WRw = (Gw + 2) * C;
    static constexpr int MROWS = 3 * WRw + 1;
    static constexpr int MT = (MROWS + 15) / 16;
    using TGg = TG<CO, TW>;
    using TGx = TG<C, TW>;
    static constexpr int STAGE4 = TGg::N4 + NSRC * TGx::N4;
    static constexpr int RED4 = NSRC * MT * 256;           // 4 waves * NSRC*MT*256 floats / 4
    static constexpr int LDS4 = STAGE4 > RED4 ? STAGE4 : RED4;
};

// sum over the 64 lanes of a wave, result in lane 63 (DPP: four shifts inside the 16-lane rows, then the two row broadcasts)
#define DNNCA_DPP_ADD(v, ctrl, rmask) \
    (v) += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), (rmask), 0xf, true))
__device__ __forceinline__ float wave_total_l63(float v) {
    DNNCA_DPP_ADD(v, 0x118, 0xf);     // row_shr:8
    DNNCA_DPP_ADD(v, 0x114, 0xf);     // row_shr:4
    DNNCA_DPP_ADD(v, 0x112, 0xf);     // row_shr:2
    DNNCA_DPP_ADD(v, 0x111, 0xf);     // row_shr:1  -> lane 15 of each row holds the row's sum
    DNNCA_DPP_ADD(v, 0x142, 0xa);     // row_bcast:15 into rows 1 and 3
    DNNCA_DPP_ADD(v, 0x143, 0xc);     // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return v;
}
__device__ __forceinline__ int slab_index(int mrow, int n);

// VW (3 -> 3 channel convs, the 512^2 level): the weight gradient leaves the matrix cores.  With 3 channels the pixel-group GEMM
// spends 16 MFMAs per 64 pixels on it at a third of their capacity (structural zeros), as much as the whole data gradient, and the
// fp32 matrix pipe was the co-limiter of this kernel (15 of 28 us at full issue).  As plain outer products it is 81 + 3 FMAs per
// pixel with no waste, and the vector ALU is idle next to the matrix pipe: the block's waves SPLIT -- waves [0, NW/2) run the
// data-gradient MFMAs for the whole tile, waves [NW/2, NW) walk down pixel columns of the same staged tiles with a sliding
// 3 x 3 x C window in registers and keep dW[dy][kx][ci][co] + db[co] in 84 accumulators per lane.  Every SIMD hosts one wave of
// each kind (waves are dealt to SIMDs cyclically), so the two pipes run side by side.
//
// TCM (two-source 6- / 12-channel convs whose first source is the output of a 12 -> C transposed conv): as TCF, on the matrix cores.
// The data gradient of the first source goes to an LDS tile instead of HBM; behind one more barrier the block runs
//   T1  din[i][j][ci] = sum_(a,e,co) dup[2i+a][2j+e][co] W[a][e][co][ci]     M = 16 input pixels of a row, K = (a, e, co), N = ci
//   T2  dW[(a,e,co)][ci] += sum_pixels dup[..](a,e,co) * in[i][j][ci]         M = (a, e, co), N = ci (+ an all-ones column: the bias
//       gradient), K = the tile's input pixels, split over the waves; the accumulators live across tiles and leave through the
//       transposed conv's slabs in the D layout k_pg_fold expects (kind 1) -- the layout k_tconv_bwd leaves there.
template <int C, int NSRC, int CO, bool DGRAD, int NT, bool DB, bool VW = false, bool PF = false, bool TCF = false, bool TCM = false>
__global__ __launch_bounds__(NT, (VW && NSRC == 1) ? 4 : 1) void k_pgbwd(BwdArgs p) {
    constexpr int NW = NT / 64;
    static_assert(!TCF || (VW && NSRC == 2 && C == 3 && CO == 3), "TCF rides in the two-source 3-channel VW kernel");
    static_assert(!TCM || (!VW && !PF && !DB && !TCF && NSRC == 2 && DGRAD && C == CO && (C == 6 || C == 12)), "TCM: two-source C -> C conv");
    static_assert(!PF || (NSRC == 1 && C == CO && DGRAD && !VW && !DB), "PF: single-source C -> C conv with data gradient");
    static_assert(!VW || (C == 3 && CO == 3 && NT == 512 && DGRAD), "VW: 3 -> 3 channels, eight waves, with data gradient");
    constexpr int NWD = VW ? NW / 2 : NW;          // waves that run the data gradient
    using Wc = BW<C, NSRC, CO>;
    using TGg = typename Wc::TGg;
    using TGx = typename Wc::TGx;
    constexpr int TW = Wc::TW, Gw = Wc::Gw, Gd = Wc::Gd, COd = Wc::COd, NPASS = Wc::NPASS, SRd = Wc::SRd, KSd = Wc::KSd;
    constexpr int MT = Wc::MT, WRw = Wc::WRw, LSg = TGg::LS, LSx = TGx::LS, Nw = Gw * CO, Nd = Gd * COd;
    static_assert(TW % (16 * Gd) == 0, "tile width must hold whole dgrad M-tiles");
    // DB: two stage buffers -> the next tile is committed while other waves still read the current one, one barrier per tile
    constexpr int STAGE4 = Wc::STAGE4, NBUF = DB ? 2 : 1;
    constexpr int MAIN4 = NBUF * STAGE4 > Wc::RED4 ? NBUF * STAGE4 : Wc::RED4;
    // PF: pooled-gradient tile (floats) and window-position tile (bytes), TH/2 + 2 rows of TW/2 + 2 pixels, lead as for a halo-1 tile
    constexpr int PFR = TH / 2 + 2, PFW = Wc::TW / 2 + 2, PFLEAD = (4 - CO % 4) % 4;
    constexpr int PFLS = (PFLEAD + PFW * CO + 3) / 4 * 4, PFN4 = PF ? PFR * PFLS / 4 : 0, PFLI = PFLS / 4, PFNI = PF ? PFR * PFLI : 0;
    // TCF: gradient of the transposed conv's output for the whole tile [TH rows][4 column blocks][32 pixels x 3] + its kernel (72)
    constexpr int TCF4 = TCF ? TH * 4 * 24 + 18 : 0;
    // TCM: transposed conv 12 -> C: K = (a, e, co) in runs of KA per output-row parity a; the tile's gradient [TH][TW * C] and the
    // transposed conv's input tile [TH/2 * TW/2 pixels][12]
    constexpr int CIt = 12, KA = 2 * C, KTt = 4 * C, KS1 = KA / 4, MBt = (KTt + 15) / 16;
    constexpr int LWt = Wc::TW / 2, NLP = (TH / 2) * LWt, ROWF = Wc::TW * C;
    constexpr int DT4 = TCM ? TH * ROWF / 4 : 0, XLT4 = TCM ? NLP * CIt / 4 : 0, TCW4 = TCM ? KTt * CIt / 4 : 0;
    static_assert(!TCM || (XLT4 <= NT && LWt % 16 == 0), "TCM: one staging slot per thread, whole M-tiles per input row");
    constexpr int TCM0 = MAIN4 + NW * 48 + 1 + PFN4 + (PFNI + 3) / 4 + TCF4;       // float4 index of the TCM tiles
    constexpr int DTF = TCM0 * 4, XLTF = DTF + DT4 * 4, TCWF = XLTF + XLT4 * 4;     // their float indices (+ the transposed conv's kernel)
    // 12 channels: the kernel sits at the register limit of two waves per SIMD -- T2's sums then live in a per-wave LDS area
    // ([wave][M block][lane] float4) instead of accumulator registers that would be live across the whole tile loop
    constexpr bool TLDS = TCM && C == 12;
    constexpr int TSUM0 = TCM0 + DT4 + XLT4 + TCW4, TSUM4 = TLDS ? NW * MBt * 64 : 0;
    __shared__ float4 lds4[TSUM0 + TSUM4];       // staged tiles (reused for the final reduction) + output rows + constants
    float* orow = reinterpret_cast<float*>(lds4 + MAIN4) + (threadIdx.x >> 6) * 192;
    float4* dta4 = lds4 + MAIN4 + NW * 48 + 1 + PFN4 + (PFNI + 3) / 4;
    float* tcw = reinterpret_cast<float*>(dta4 + TH * 4 * 24);
    float* pf_dp = reinterpret_cast<float*>(lds4 + MAIN4 + NW * 48 + 1);
    unsigned* pf_ix = reinterpret_cast<unsigned*>(lds4 + MAIN4 + NW * 48 + 1 + PFN4);
    constexpr int NPS = PF ? (PFN4 + NT - 1) / NT : 1;       // prefetch slots per thread for the pooled tiles (PFNI = PFN4)
    static_assert(PFNI == PFN4, "the two pooled tiles have the same number of staging slots");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m16 = lane & 15, q = lane >> 4, n = m16;
    const int ntiles = p.tiles_x * p.tiles_y * p.B;
    const bool xcd_map = (ntiles & 7) == 0 && (gridDim.x & 7) == 0;

    // two constants in LDS (1.0 for the all-ones bias row of the weight-gradient A operand, 0.0 for padding rows and
    // padding columns): the MFMA operands are then plain LDS reads, no per-MFMA select
    float* cst = reinterpret_cast<float*>(lds4 + MAIN4 + NW * 48);
    if (threadIdx.x == 0) { cst[0] = 1.0f; cst[1] = 0.0f; }
    if constexpr (TCF) {
        if (threadIdx.x < 72) tcw[threadIdx.x] = p.tc_w[threadIdx.x];
    }
    // TCF: dWt[(a, e, co)][ci] and dbt[co] of this lane's pooled pixels (data-gradient waves)
    float tacc[TCF ? 72 : 1], tbias[TCF ? 3 : 1];
#pragma unroll
    for (int i = 0; i < (TCF ? 72 : 1); ++i) tacc[i] = 0.f;
#pragma unroll
    for (int i = 0; i < (TCF ? 3 : 1); ++i) tbias[i] = 0.f;
    const float* ldsf = reinterpret_cast<const float*>(lds4);
    constexpr int CST1 = (MAIN4 + NW * 48) * 4, CST0 = CST1 + 1;     // absolute float indices of the constants
    // wgrad A-operand addressing: row mrow = (dy, j) of the group's window; last valid row is the all-ones bias row.
    // stepA[t] is what one K-step (4 groups) adds to the address: 0 for the constant rows
    int offA[MT], stepA[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int mrow = 16 * t + m16;
        if (mrow < 3 * WRw) {
            const int dy = mrow / WRw, j = mrow - dy * WRw;
            offA[t] = dy * LSx + TGx::LEAD + j + q * (Gw * C);
            stepA[t] = 4 * Gw * C;
        } else {
            offA[t] = mrow == 3 * WRw ? CST1 : CST0;       // absolute index (window rows are relative to the x tile)
            stepA[t] = 0;
        }
    }
    f32x4 acc[NSRC][MT];
#pragma unroll
    for (int s = 0; s < NSRC; ++s)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // VW: two sources: dW[dy][kx*3 + ci][co] (81) + db[co] (3) of this lane's pixels; one source: this wave's kernel row (27) or db
    constexpr int NWACC = !VW ? 1 : (NSRC == 2 ? 84 : 27);
    float wacc[NWACC];
#pragma unroll
    for (int i = 0; i < NWACC; ++i) wacc[i] = 0.f;

    // TCM operands: T1's B operand is the transposed conv's kernel in LDS (row k = 4 s + q); T2's accumulators and LDS addressing
    f32x4 tacc2[TCM ? MBt : 1];
    if constexpr (TCM) {
        if (tid < TCW4) lds4[TCM0 + DT4 + XLT4 + tid] = reinterpret_cast<const float4*>(p.tc_w)[tid];
        if constexpr (TLDS) {
#pragma unroll
            for (int t = 0; t < MBt; ++t) lds4[TSUM0 + (wave * MBt + t) * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int t = 0; t < MBt; ++t) {
            tacc2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    if (VW && wave < NWD) __builtin_amdgcn_s_setprio(2);     // the MFMA waves' few vector instructions go first; the FMA waves fill in
    float breg[DGRAD ? NPASS * KSd : 1];
    if (DGRAD) {
#pragma unroll
        for (int s = 0; s < NPASS * KSd; ++s) breg[s] = p.bmat[s * 64 + lane];
    }
    // dgrad epilogue: column n = (dxp, co) of pass ps -> (source, channel)
    const int dxp = n / COd, cod = n % COd;

    float4 preg[TGg::npf(NT)];
    float4 prex[NSRC][TGx::npf(NT)];
    TileMap<CO, TW, NT> mpg;
    TileMap<C, TW, NT> mpx;
    mpg.init(tid, p.W * CO / 4);
    mpx.init(tid, p.W * C / 4);
    float4 prey[PF ? TGg::npf(NT) : 1];      // PF: this conv's output, same tile geometry as dz
    float4 predp[NPS];
    unsigned preix[NPS];
    auto pf_issue = [&](int b, int x0, int y0) {
        if constexpr (PF) {
            (void)tile_issue<CO, TW, NT>(prey, mpg, p.pf_y, b, x0, y0, p.B, p.H, p.W);
            const int Hp = p.H >> 1, Wp = p.W >> 1, rowf = Wp * CO;
#pragma unroll
            for (int k = 0; k < NPS; ++k) {
                const int id = tid + k * NT;
                const int r = id / PFLI, c4 = id - r * PFLI;
                const int gy = (y0 >> 1) - 1 + r, gf = ((x0 >> 1) - 1) * CO - PFLEAD + 4 * c4;
                const bool ok = id < PFN4 && (unsigned)gy < (unsigned)Hp && gf >= 0 && gf < rowf;
                const size_t off = ok ? ((size_t)b * Hp + gy) * rowf + gf : 0;
                predp[k] = *reinterpret_cast<const float4*>(p.pf_dpool + off);
                if (!ok) predp[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                preix[k] = *reinterpret_cast<const unsigned*>(p.pf_idx + off);
                if (!ok) preix[k] = 0xffffffffu;
            }
        }
    };
    // PF: commit the pooled tiles, then (after a barrier) turn the skip gradient in `preg` into dz:
    //     dz = (dskip + (window position == recorded position ? dpool : 0)) * act'(y)
    auto pf_commit_pooled = [&]() {
        if constexpr (PF) {
#pragma unroll
            for (int k = 0; k < NPS; ++k) {
                const int id = tid + k * NT;
                if (id < PFN4) {
                    reinterpret_cast<float4*>(pf_dp)[id] = predp[k];
                    pf_ix[id] = preix[k];
                }
            }
        }
    };
    auto pf_transform = [&](unsigned ok) {
        if constexpr (PF) {
            const unsigned char* ixb = reinterpret_cast<const unsigned char*>(pf_ix);
#pragma unroll
            for (int k = 0; k < TGg::npf(NT); ++k) {
                const int r = mpg.row[k], c4 = mpg.c4[k];
                float g[4] = {preg[k].x, preg[k].y, preg[k].z, preg[k].w};
                const float yv[4] = {prey[k].x, prey[k].y, prey[k].z, prey[k].w};
                const bool in = (ok >> k) & 1u;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int f = 4 * c4 + e - TGg::LEAD;               // float index from the left-halo pixel
                    if (in && f >= 0 && f < (TW + 2) * CO) {
                        const int px = f / CO, ch = f - px * CO;
                        const int pr = ((r - 1) >> 1) + 1, pc = ((px - 1) >> 1) + 1;        // row / pixel in the pooled tiles
                        const unsigned pos = (((unsigned)(r - 1) & 1u) << 1) | ((unsigned)(px - 1) & 1u);
                        const int o = PFLEAD + pc * CO + ch;
                        const float dp = ixb[pr * PFLS + o] == pos ? pf_dp[pr * PFLS + o] : 0.f;
                        g[e] = (g[e] + dp) * (yv[e] > 0.f ? 1.0f : p.pf_alpha);
                    }
                }
                preg[k] = make_float4(g[0], g[1], g[2], g[3]);
            }
        }
    };
    unsigned okg = 0, okx = 0;
    auto decode = [&](int t, int& b, int& x0, int& y0) {
        // blocks b, b + 8, b + 16 ... share an XCD (and its L2): give each XCD one contiguous eighth of the tile sequence, so
        // that the halo rows of vertically adjacent tiles are fetched into the same L2
        if (xcd_map) t = (t & 7) * (ntiles >> 3) + (t >> 3);
        const int bx = t % p.tiles_x, by = (t / p.tiles_x) % p.tiles_y;
        b = t / (p.tiles_x * p.tiles_y);
        x0 = bx * TW;
        y0 = by * TH;
    };
    int tile = blockIdx.x;
    int it = 0, buf = 0;
    (void)it;          // read by the tuning build's stamps only
    STAMP(6);
    if (tile < ntiles) {
        int b, x0, y0;
        decode(tile, b, x0, y0);
        okg = tile_issue<CO, TW, NT>(preg, mpg, p.dz, b, x0, y0, p.B, p.H, p.W);
        pf_issue(b, x0, y0);
#pragma unroll
        for (int s = 0; s < NSRC; ++s) okx = tile_issue<C, TW, NT>(prex[s], mpx, p.x[s], b, x0, y0, p.B, p.H, p.W);
    }
    if (DGRAD) {
        // drain + hide the origin of the B-operand registers (see k_pgfwd): no vmcnt waits for them inside the tile loop.  Behind the
        // issue of the first tile's loads: one memory round trip at the start of the kernel, not two.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < NPASS * KSd; ++s) asm volatile("" : "+v"(breg[s]));
    }

    if (DB && tile < ntiles) {
        tile_commit<CO, TW, NT>(preg, okg, lds4, tid);
#pragma unroll
        for (int s = 0; s < NSRC; ++s) tile_commit<C, TW, NT>(prex[s], okx, lds4 + TGg::N4 + s * TGx::N4, tid);
        lds_barrier();
    }
#pragma unroll 1
    while (tile < ntiles) {
        int b, x0, y0;
        decode(tile, b, x0, y0);
        STAMP(0);
        if (!DB) {
            if constexpr (PF) {
                pf_commit_pooled();
                lds_barrier();
                pf_transform(okg);
            }
            tile_commit<CO, TW, NT>(preg, okg, lds4, tid);
#pragma unroll
            for (int s = 0; s < NSRC; ++s) tile_commit<C, TW, NT>(prex[s], okx, lds4 + TGg::N4 + s * TGx::N4, tid);
            lds_barrier();
        }
        const int GOFF = buf * (STAGE4 * 4), XOFF = GOFF + TGg::N4 * 4;      // float indices of the dz / x tiles in use
        const float* gl = ldsf + GOFF;
        const float* xl = ldsf + XOFF;
        STAMP(1);
        // TCF: a data-gradient wave owns the column block tx = wave of the tile (32 pixels, all TH rows) = 16 x TH/2 pixels of the
        // transposed conv's input, one per lane; their 6 channels are fetched now (ahead of the prefetch: vmcnt retires in order)
        float tin[TCF ? 6 : 1];
        if constexpr (TCF) {
            if (wave < NWD) {
                const int pr = lane >> 4, pc = lane & 15;
                const float* ip = p.tc_in + ((((size_t)b * (p.H >> 1) + (y0 >> 1) + pr) * (p.W >> 1)) + (x0 >> 1) + wave * 16 + pc) * 6;
#pragma unroll
                for (int ci = 0; ci < 6; ++ci) tin[ci] = ip[ci];
            }
        }
        // TCM: this thread's float4 of the transposed conv's input tile (ahead of the prefetch, as tin)
        float4 txl = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (TCM) {
            if (tid < XLT4) {
                const int r = tid / (LWt * 3), c = tid - r * (LWt * 3);
                txl = reinterpret_cast<const float4*>(p.tc_in)[(((size_t)b * (p.H >> 1) + (y0 >> 1) + r) * (p.W >> 1) + (x0 >> 1)) * 3 + c];
            }
        }
        const int next = tile + gridDim.x;
        if (next < ntiles) {
            int nb, nx0, ny0;
            decode(next, nb, nx0, ny0);
            okg = tile_issue<CO, TW, NT>(preg, mpg, p.dz, nb, nx0, ny0, p.B, p.H, p.W);
            pf_issue(nb, nx0, ny0);
#pragma unroll
            for (int s = 0; s < NSRC; ++s) okx = tile_issue<C, TW, NT>(prex[s], mpx, p.x[s], nb, nx0, ny0, p.B, p.H, p.W);
        }

        STAMP(2);
        // ---- data gradient: conv of dz with the flipped kernel; M-tiles of 16 groups x Gd pixels
        if (DGRAD && !(DBG_FLAGS(p) & 1)) {
            // M-tiles of this wave: t = wave + 4j, j < MTX*TH/4; NCH of them are processed with interleaved MFMA chains
            constexpr int PERW = Wc::MTX * TH / NWD;
            constexpr int NCH = (VW && NSRC == 1) ? 2 : (PERW >= 4 ? 4 : PERW);      // (VW, one source: two chains keep the kernel under 128 registers)
#pragma unroll 1
            for (int j0 = 0; j0 < (VW && wave >= NWD ? 0 : PERW); j0 += NCH) {
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) {
                    f32x4 d[NCH];
                    int txs[NCH], tys[NCH];
#pragma unroll
                    for (int i = 0; i < NCH; ++i) {
                        const int t = wave + NWD * (j0 + i);
                        txs[i] = t % Wc::MTX;
                        tys[i] = t / Wc::MTX;
                        d[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int k = 0; k < SRd; ++k)
#pragma unroll
                            for (int i = 0; i < NCH; ++i)
                                d[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                    gl[(tys[i] + dy) * LSg + TGg::LEAD + (txs[i] * 16 + m16) * (Gd * CO) + q + 4 * k],
                                    breg[ps * KSd + dy * SRd + k], d[i], 0, 0, 0);
                    // transpose D through LDS: per destination the M-tile's gradient row is 16 groups x (Gd*CS)
                    // contiguous floats; then 16-byte masked / accumulated stores
                    constexpr int SPL = (NSRC == 2 && NPASS == 1) ? 2 : 1;     // destinations covered by this pass
                    constexpr int CS = COd / SPL;                              // channels per destination (== C)
                    constexpr int PER4 = 16 * Gd * CS / 4;                      // float4's per destination row segment
#pragma unroll
                    for (int i = 0; i < NCH; ++i) {
                        const int tx = txs[i], ty = tys[i], y = y0 + ty;
                        if (n < Nd) {
                            const int sp = cod / CS, cs = cod - sp * CS;
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                orow[sp * (16 * Gd * CS) + (4 * q + r) * (Gd * CS) + dxp * CS + cs] = d[i][r];
                        }
                        __builtin_amdgcn_wave_barrier();
                        const int sp = lane / PER4, i4 = lane - sp * PER4;
                        const int src = NPASS == 2 ? ps : sp;
                        // (the destination is picked with selects between kernel arguments: indexing p.dx / p.acc / p.mask with a
                        //  per-lane value made the compiler FETCH them from the kernarg segment with vector loads, and the
                        //  s_waitcnt vmcnt(0) behind each of those also drained the next tile's prefetch -- a full HBM round
                        //  trip per M-tile, 2 400 cycles around 480 cycles of MFMAs)
                        float* dxs;
                        int accs, masks;
                        if constexpr (NPASS == 2) { dxs = ps ? p.dx[1] : p.dx[0]; accs = ps ? p.acc[1] : p.acc[0]; masks = ps ? p.mask[1] : p.mask[0]; }
                        else if constexpr (SPL == 2) { dxs = sp ? p.dx[1] : p.dx[0]; accs = sp ? p.acc[1] : p.acc[0]; masks = sp ? p.mask[1] : p.mask[0]; }
                        else { dxs = p.dx[0]; accs = p.acc[0]; masks = p.mask[0]; }
                        const int f0 = (x0 + tx * 16 * Gd) * C + 4 * i4;        // float index within the image row
                        if (TCF && sp == 0) {
                            // the first source's gradient (= the transposed conv's output gradient) stays in LDS (whole tiles only)
                            if (lane < PER4) dta4[(ty * 4 + tx) * 24 + i4] = reinterpret_cast<const float4*>(orow)[lane];
                        } else if (TCM && src == 0) {
                            if (lane < PER4) lds4[TCM0 + (ty * Wc::MTX + tx) * PER4 + i4] = reinterpret_cast<const float4*>(orow)[lane];
                        } else if (lane < SPL * PER4 && f0 < p.W * C && y < p.H) {
                            float4 v = reinterpret_cast<const float4*>(orow)[lane];
                            float* dst = dxs + ((size_t)b * p.H + y) * p.W * C + f0;
                            if (accs) {
                                const float4 o = *reinterpret_cast<const float4*>(dst);
                                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                            }
                            if (masks) {
                                const float4 xv = *reinterpret_cast<const float4*>(
                                    xl + src * (TGx::N4 * 4) + (ty + 1) * LSx + TGx::HL + tx * 16 * Gd * C + 4 * i4);
                                v.x *= xv.x > 0.f ? 1.0f : p.alpha;
                                v.y *= xv.y > 0.f ? 1.0f : p.alpha;
                                v.z *= xv.z > 0.f ? 1.0f : p.alpha;
                                v.w *= xv.w > 0.f ? 1.0f : p.alpha;
                            }
                            *reinterpret_cast<float4*>(dst) = v;
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
        }
        if constexpr (TCM) {
            static_assert(Wc::MTX * (16 * Gd * C / 4) == ROWF / 4, "TCM: the M-tiles of a row tile the gradient row");
            if (tid < XLT4) lds4[TCM0 + DT4 + tid] = txl;
            lds_barrier();                                    // the tile's first-source gradient and the input tile are complete
            // LDS addressing of T1 / T2, recomputed per tile from an opaque copy of the lane id: as loop invariants these seven
            // values would be live across the whole tile loop of a kernel that sits at its register limit
            int lo = lane;
            asm volatile("" : "+v"(lo));
            const int m16 = lo & 15, q = lo >> 4;
            int offT[MBt];
#pragma unroll
            for (int t = 0; t < MBt; ++t) {
                const int k = 16 * t + m16, a = k / KA, kr = k - a * KA;
                offT[t] = k < KTt ? DTF + a * ROWF + 2 * q * C + kr : CST0;
            }
            const int boffT = m16 < CIt ? XLTF + q * CIt + m16 : (m16 == CIt ? CST1 : CST0), bstepT = m16 < CIt ? 4 * CIt : 0;
            const int woffT = m16 < CIt ? TCWF + q * CIt + m16 : CST0, wstepT = bstepT;
            // ---- T1: the transposed conv's data gradient, one M-tile = 16 input pixels of a row
#pragma unroll 1
            for (int mt = wave; mt < NLP / 16; mt += NW) {
                const int li = mt / (LWt / 16), mx = mt - li * (LWt / 16);
                f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int kk = 0; kk < KS1; ++kk)
                        d = __builtin_amdgcn_mfma_f32_16x16x4f32(ldsf[DTF + (2 * li + a) * ROWF + 2 * (mx * 16 + m16) * C + 4 * kk + q],
                                                                 ldsf[woffT + (a * KS1 + kk) * wstepT], d, 0, 0, 0);
                if (m16 < CIt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) orow[(4 * q + r) * CIt + m16] = d[r];
                }
                __builtin_amdgcn_wave_barrier();
                if (lane < 16 * CIt / 4) {
                    float4 v = reinterpret_cast<const float4*>(orow)[lane];
                    if (p.tc_mask) {
                        const float4 xv = lds4[TCM0 + DT4 + (li * LWt + mx * 16) * 3 + lane];
                        v.x *= xv.x > 0.f ? 1.0f : p.tc_alpha;
                        v.y *= xv.y > 0.f ? 1.0f : p.tc_alpha;
                        v.z *= xv.z > 0.f ? 1.0f : p.tc_alpha;
                        v.w *= xv.w > 0.f ? 1.0f : p.tc_alpha;
                    }
                    reinterpret_cast<float4*>(p.tc_din)[(((size_t)b * (p.H >> 1) + (y0 >> 1) + li) * (p.W >> 1) + (x0 >> 1) + mx * 16) * 3 + lane] = v;
                }
                __builtin_amdgcn_wave_barrier();
            }
            // ---- T2: the transposed conv's weight (+ bias) gradient, K-steps of 4 input pixels dealt to the waves
            if constexpr (TLDS) {
#pragma unroll
                for (int t = 0; t < MBt; ++t) tacc2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll 1
            for (int st = wave; st < NLP / 4; st += NW) {
                const int sd = ((4 * st) / LWt) * 2 * ROWF + ((4 * st) % LWt) * 2 * C;
                const float bv = ldsf[boffT + st * bstepT];
#pragma unroll
                for (int t = 0; t < MBt; ++t)
                    tacc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ldsf[offT[t] + (16 * (t + 1) <= KTt || offT[t] != CST0 ? sd : 0)], bv, tacc2[t], 0, 0, 0);
            }
            if constexpr (TLDS) {
#pragma unroll
                for (int t = 0; t < MBt; ++t) {
                    float4 v = lds4[TSUM0 + (wave * MBt + t) * 64 + lane];
                    v.x += tacc2[t][0]; v.y += tacc2[t][1]; v.z += tacc2[t][2]; v.w += tacc2[t][3];
                    lds4[TSUM0 + (wave * MBt + t) * 64 + lane] = v;
                }
            }
        }
        if constexpr (TCF) {
            // ---- transposed conv backward on the data-gradient waves: lane = one input pixel (pr, pc) of this wave's column block; its
            // 2 x 2 output pixels' gradients are in the LDS tile this wave has just written (DS operations of one wave execute in order)
            if (wave < NWD && !(DBG_FLAGS(p) & 5)) {          // (tuning builds: DNNCA_DBG bit 2 skips this part alone)
                __builtin_amdgcn_wave_barrier();
                const int pr = lane >> 4, pc = lane & 15;
                const float* dt = reinterpret_cast<const float*>(dta4);
                float din[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    float d6[6];                 // (e, co) of output row 2 pr + a, pixels 2 pc, 2 pc + 1
                    const float* rp = dt + (((2 * pr + a) * 4 + wave) * 24) * 4 + (2 * pc) * 3;
#pragma unroll
                    for (int i = 0; i < 6; ++i) d6[i] = rp[i];
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int co = 0; co < 3; ++co) {
                            const float d = d6[e * 3 + co];
                            tbias[co] += d;
                            const float* wr = tcw + ((a * 2 + e) * 3 + co) * 6;      // uniform address: broadcast reads
#pragma unroll
                            for (int ci = 0; ci < 6; ++ci) {
                                din[ci] = fmaf(d, wr[ci], din[ci]);
                                tacc[((a * 2 + e) * 3 + co) * 6 + ci] = fmaf(d, tin[ci], tacc[((a * 2 + e) * 3 + co) * 6 + ci]);
                            }
                        }
                }
                if (p.tc_mask) {
#pragma unroll
                    for (int ci = 0; ci < 6; ++ci) din[ci] *= tin[ci] > 0.f ? 1.0f : p.tc_alpha;
                }
                float* op = p.tc_din + ((((size_t)b * (p.H >> 1) + (y0 >> 1) + pr) * (p.W >> 1)) + (x0 >> 1) + wave * 16 + pc) * 6;
#pragma unroll
                for (int ci = 0; ci < 6; ++ci) op[ci] = din[ci];
                __builtin_amdgcn_wave_barrier();
            }
        }
        STAMP(3);
        if constexpr (VW) {
            // ---- weight gradient on the vector ALU (waves NWD .. NW-1): lane = one pixel column, sliding window down the rows
            if (wave >= NWD) {
                const int wv = wave - NWD;
                if constexpr (NSRC == 1) {
                    // one kernel row dy per wave (27 accumulators): waves 0..2 walk the whole tile, reading input row r + dy for
                    // output row r; wave 3 sums dz (the bias gradient).  Few registers: two blocks per CU stay resident.
#pragma unroll 1
                    for (int h = 0; h < 2; ++h) {
                        const int col = h * 64 + lane;                                       // TW = 128 = two wave widths
                        const float* gb = gl + TGg::HL + col * CO + LSg;                      // dz of (row 0, col)
                        if (wv < 3) {
                            const float* xb = xl + TGx::LEAD + col * C + wv * LSx;           // window row dy = wv of (row 0, col)
                            // two rows in flight: the LDS reads of the next row are issued before the 27 FMAs of this one
                            float xr[2][9], dzv[2][3];
                            auto load = [&](int b, int r) {
#pragma unroll
                                for (int j = 0; j < 9; ++j) xr[b][j] = xb[r * LSx + j];
#pragma unroll
                                for (int co = 0; co < 3; ++co) dzv[b][co] = gb[r * LSg + co];
                            };
                            auto fma27 = [&](int b) {
#pragma unroll
                                for (int j = 0; j < 9; ++j)
#pragma unroll
                                    for (int co = 0; co < 3; ++co) wacc[j * 3 + co] = fmaf(xr[b][j], dzv[b][co], wacc[j * 3 + co]);
                            };
                            load(0, 0);
#pragma unroll 1
                            for (int r = 0; r < TH; r += 2) {
                                load(1, r + 1);
                                fma27(0);
                                load(0, r + 2 < TH ? r + 2 : TH - 1);
                                fma27(1);
                            }
                        } else {
#pragma unroll
                            for (int r = 0; r < TH; ++r)
#pragma unroll
                                for (int co = 0; co < 3; ++co) wacc[co] += gb[r * LSg + co];
                        }
                    }
                } else {
                // two sources: waves 0, 1 take the columns of source 0, waves 2, 3 those of source 1; a lane walks down its
                // column with a sliding 3 x 3 x C window in registers (81 + 3 accumulators)
                const int src = wv >> 1;
                const int col = (wv & 1) * 64 + lane;                                   // TW = 128 = two wave widths
                constexpr int ROWS = TH;
                const int r0 = 0;
                const float* xb = xl + src * (TGx::N4 * 4) + TGx::LEAD + col * C;      // window of pixel (row, col): 9 floats from here
                const float* gb = gl + TGg::HL + col * CO;
                float xw[3][9];
#pragma unroll
                for (int j = 0; j < 9; ++j) { xw[0][j] = xb[r0 * LSx + j]; xw[1][j] = xb[(r0 + 1) * LSx + j]; }
#pragma unroll
                for (int rr = 0; rr < ROWS; ++rr) {
                    const int r = r0 + rr;
                    float dzv[3];
#pragma unroll
                    for (int co = 0; co < 3; ++co) dzv[co] = gb[(r + 1) * LSg + co];
#pragma unroll
                    for (int j = 0; j < 9; ++j) xw[(rr + 2) % 3][j] = xb[(r + 2) * LSx + j];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int j = 0; j < 9; ++j)
#pragma unroll
                            for (int co = 0; co < 3; ++co)
                                wacc[(dy * 9 + j) * 3 + co] = fmaf(xw[(rr + dy) % 3][j], dzv[co], wacc[(dy * 9 + j) * 3 + co]);
#pragma unroll
                    for (int co = 0; co < 3; ++co) wacc[81 + co] += dzv[co];
                }
                }
            }
        } else {
        // ---- weight gradient: K = pixel groups (4 per MFMA); every wave takes rows ty = wave, wave + 4
#pragma unroll 1
        for (int ty = (DBG_FLAGS(p) & 2) ? TH : wave; ty < TH; ty += NW) {
            const int goff = n < Nw ? GOFF + (ty + 1) * LSg + TGg::HL + q * Nw + n : CST0;
            const int gstep = n < Nw ? 4 * Nw : 0;
            int aoffs[NSRC][MT];
#pragma unroll
            for (int s = 0; s < NSRC; ++s)
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    aoffs[s][t] = stepA[t] ? XOFF + s * (TGx::N4 * 4) + ty * LSx + offA[t] : offA[t];
#pragma unroll
            for (int st = 0; st < TW / (4 * Gw); ++st) {
                const float bv = ldsf[goff + st * gstep];
#pragma unroll
                for (int s = 0; s < NSRC; ++s)
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        // tiles whose 16 rows are all window rows use a compile-time step (immediate LDS offsets)
                        const int ao = 16 * (t + 1) <= 3 * WRw ? aoffs[s][t] + st * (4 * Gw * C) : aoffs[s][t] + st * stepA[t];
                        acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ldsf[ao], bv, acc[s][t], 0, 0, 0);
                    }
            }
        }
        }
        STAMP(4);
        if (DB) {
            if (next < ntiles) {
                tile_commit<CO, TW, NT>(preg, okg, lds4 + (buf ^ 1) * STAGE4, tid);
#pragma unroll
                for (int s = 0; s < NSRC; ++s)
                    tile_commit<C, TW, NT>(prex[s], okx, lds4 + (buf ^ 1) * STAGE4 + TGg::N4 + s * TGx::N4, tid);
            }
            buf ^= 1;
        }
        lds_barrier();
        STAMP(5);
        ++it;
        tile = next;
    }

    STAMP_END(5);
    if constexpr (VW) {
        // lane sums by DPP, the block's four weight-gradient waves through LDS, then one atomic per element into slab
        // (blockIdx % NBUCKET).  The totals go where k_pg_fold expects D[(dy, j), (dx, co)]: everything in the dx = 0 entries
        // (j = kx*C + ci), zeros elsewhere (the slabs are zeroed at the top of the step), the bias in the all-ones row.
        float* red = reinterpret_cast<float*>(lds4);
        __syncthreads();                       // every wave has left the tile loop: the staged tiles are dead
        if (wave >= NWD) {
            const int wv = wave - NWD;
#pragma unroll
            for (int i = 0; i < NWACC; ++i) {
                const float t = wave_total_l63(wacc[i]);
                // one source: wave wv < 3 holds kernel row wv (elements wv*27 ..), wave 3 the bias (81 .. 83)
                if (lane == 63 && (NSRC == 2 || wv < 3 || i < 3)) red[NSRC == 2 ? wv * 84 + i : wv * 27 + i] = t;
            }
        } else if constexpr (TCF) {
            // the data-gradient waves: totals of the transposed conv's weight / bias gradient behind the four 84-float rows
#pragma unroll
            for (int i = 0; i < 72; ++i) {
                const float t = wave_total_l63(tacc[i]);
                if (lane == 63) red[4 * 84 + wave * 75 + i] = t;
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float t = wave_total_l63(tbias[i]);
                if (lane == 63) red[4 * 84 + wave * 75 + 72 + i] = t;
            }
        }
        __syncthreads();
        if constexpr (TCF) {
            if (tid < 75) {
                const float* r2 = red + 4 * 84;
                const float v = (r2[tid] + r2[75 + tid]) + (r2[150 + tid] + r2[225 + tid]);
                // rows (a, e, co), columns ci; the bias in column 6 of its (a = 0, e = 0, co) row (the fold sums that column over (a, e))
                const int mrow = tid < 72 ? tid / 6 : tid - 72, col = tid < 72 ? tid % 6 : 6;
                atomicAdd(p.tc_slabs + (size_t)(blockIdx.x % NBUCKET) * 256 + slab_index(mrow, col), v);
            }
        }
        const int bucket = blockIdx.x % NBUCKET;
        for (int i = tid; i < NSRC * 84; i += NT) {
            const int s = i / 84, e = i - s * 84;
            const float v = NSRC == 2 ? red[(2 * s) * 84 + e] + red[(2 * s + 1) * 84 + e] : red[e];
            const int mrow = e < 81 ? (e / 27) * WRw + (e / 3) % 9 : 3 * WRw, co = e < 81 ? e % 3 : e - 81;
            atomicAdd(p.slabs[s] + (size_t)bucket * (MT * 256) + slab_index(mrow, co), v);
        }
        return;
    }
    // ---- sum the 4 waves through LDS; the block adds its partial D into slab (blockIdx % NBUCKET)
    float* red = reinterpret_cast<float*>(lds4);
    constexpr int PER = NSRC * MT * 256;
    if (NW == 8) {      // waves 4..7 hand their sums to waves 0..3 first (the LDS holds four partial sets)
        if (wave >= 4) {
#pragma unroll
            for (int s = 0; s < NSRC; ++s)
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[(wave - 4) * PER + ((s * MT + t) * 4 + r) * 64 + lane] = acc[s][t][r];
        }
        __syncthreads();
        if (wave < 4) {
#pragma unroll
            for (int s = 0; s < NSRC; ++s)
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[s][t][r] += red[wave * PER + ((s * MT + t) * 4 + r) * 64 + lane];
        }
        __syncthreads();
    }
    if (wave < 4) {
#pragma unroll
        for (int s = 0; s < NSRC; ++s)
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave * PER + ((s * MT + t) * 4 + r) * 64 + lane] = acc[s][t][r];
    }
    __syncthreads();
    const int bucket = blockIdx.x % NBUCKET;
    for (int i = tid; i < PER; i += NT) {
        const float v = (red[i] + red[PER + i]) + (red[2 * PER + i] + red[3 * PER + i]);
        const int s = i / (MT * 256), e = i - s * (MT * 256);
        atomicAdd(p.slabs[s] + (size_t)bucket * (MT * 256) + e, v);
    }
    if constexpr (TCM) {
        static_assert(NW * MBt * 256 <= 4 * PER, "TCM: the waves' partial sums fit the reduction area");
        __syncthreads();
#pragma unroll
        for (int t = 0; t < MBt; ++t) {
            if constexpr (TLDS) {
                const float4 v = lds4[TSUM0 + (wave * MBt + t) * 64 + lane];
                tacc2[t] = f32x4{v.x, v.y, v.z, v.w};
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave * (MBt * 256) + (t * 4 + r) * 64 + lane] = tacc2[t][r];
        }
        __syncthreads();
        for (int i = tid; i < MBt * 256; i += NT) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += red[w * (MBt * 256) + i];
            atomicAdd(p.tc_slabs + (size_t)bucket * (MBt * 256) + i, v);
        }
    }
    STAMP_END(7);
}

// ================================================================================================ B-operand prep
// One block per descriptor.  forward:  B[(s,dy,j),(dx,co)] = W[dy][xoff-dx][s*C+ci][co]
//                            dgrad:    B[(dy,j),(dx,o)]   = W[2-dy][2-(xoff-dx)][o_off+o][ci]   (input = gradient, C = Cout)
struct PrepDesc {
    int w_off, out_off;      // floats: weight offset in the parameter vector, output offset in the bmat buffer
    int C, NSRC, CO, G, mode;
    int cin_total, cout_total, o_off;
};

// the (static) gather table: for every B-operand element the index of its weight in the parameter vector, or -1
__global__ void k_pg_prep_index(const PrepDesc* __restrict__ descs, int* __restrict__ index) {
    const PrepDesc d = descs[blockIdx.x];
    const int WR = (d.G + 2) * d.C, SR = (WR + 3) / 4;
    const int KS = d.NSRC * 3 * SR;
    for (int i = threadIdx.x; i < KS * 64; i += blockDim.x) {
        int lane = i & 63, step = i >> 6;
        int n = lane & 15, kk = lane >> 4;
        int s = step / (3 * SR), dy = (step / SR) % 3, k = step % SR;
        int j = 4 * k + kk;
        int v = -1;
        if (j < WR && n < d.G * d.CO) {
            int xoff = j / d.C, ci = j % d.C, dx = n / d.CO, co = n % d.CO;
            int kx = xoff - dx;
            if (kx >= 0 && kx <= 2) {
                if (d.mode == 0)
                    v = d.w_off + ((dy * 3 + kx) * d.cin_total + s * d.C + ci) * d.cout_total + co;
                else
                    v = d.w_off + (((2 - dy) * 3 + (2 - kx)) * d.cin_total + d.o_off + co) * d.cout_total + ci;
            }
        }
        index[d.out_off + i] = v;
    }
}

// every step: B operands <- current weights (a plain gather).  On a train step the blocks past `nprep` do k_step_init's job
// (scalar block, flat gradient vector, weight-gradient slabs) so that the loss does not need a launch for it.
struct StepInit {
    double* scalars;
    float4* a;
    float4* b;
    unsigned na4, nb4;
};

struct PrepRide {          // k_pg_prep's arguments when the preparation rides in another kernel's launch (blocks past that kernel's own)
    const int* index;
    const float* params;
    float* bmat;
    int n, nprep, nblocks;     // nblocks = nprep + zeroing blocks (0: nothing rides)
    StepInit z;
};

// block `blk` of `nblocks` preparation blocks (256 threads)
__device__ __forceinline__ void pg_prep_body(const int* __restrict__ index, const float* __restrict__ params, float* __restrict__ bmat,
                                             int n, int nprep, const StepInit& z, int blk, int nblocks) {
    if (blk >= nprep) {
        const unsigned T = (unsigned)(nblocks - nprep) * 256u, i0 = (unsigned)(blk - nprep) * 256u + threadIdx.x;
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        for (unsigned i = i0; i < z.na4; i += T) z.a[i] = zero;
        for (unsigned i = i0; i < z.nb4; i += T) z.b[i] = zero;
        if (i0 < 8) z.scalars[i0] = i0 == 1 ? (double)INFINITY : (i0 == 2 ? -(double)INFINITY : 0.0);   // see k_step_init
        return;
    }
    const int i = blk * 256 + threadIdx.x;
    if (i >= n) return;
    const int j = index[i];
    bmat[i] = j >= 0 ? params[j] : 0.f;
}

__global__ __launch_bounds__(256) void k_pg_prep(const int* __restrict__ index, const float* __restrict__ params,
                                                 float* __restrict__ bmat, int n, int nprep, StepInit z) {
    pg_prep_body(index, params, bmat, n, nprep, z, (int)blockIdx.x, (int)gridDim.x);
}

}  // namespace dnnca
#include "strip_dev.h"
namespace dnnca {

// ================================================================================================ backward, 3 -> 3 channels, vector ALU
// k_bwd3v: the whole backward of a single-source 3 -> 3 channel conv (the 512^2 level) without the matrix cores.
// With 3 channels the pixel-group GEMM fills a third of each MFMA (structural zeros): 31 thirty-two-cycle MFMAs per 64 pixels for
// data + weight gradient = 992 matrix-pipe cycles per SIMD, 72 % of k_pgbwd<3,1,3>'s time.  As plain FMAs the same work is 162 per
// pixel = 324 vector-ALU cycles per 64 pixels (v_fma_f32: 2 cycles per wave on a SIMD-32), with no waste.
// Both gradients read the SAME 3 x 3 x 3 window of dz around a pixel q:
//     dx[q][ci]            = sum_{wy,wx,co} win[wy][wx][co] * W[2-wy][2-wx][ci][co]        (81 FMAs, the weight is an SGPR operand)
//     dW[2-wy][2-wx][ci][co] += x[q][ci] * win[wy][wx][co]                                  (81 FMAs, 81 accumulators per lane)
//     db[co]               += win[1][1][co]
// so one lane per pixel does everything from 27 window registers: a wave owns 64 columns x 4 rows of a 128 x 8 tile and walks down
// its rows with a rotating three-row window (9 LDS reads per pixel).  Only dz is staged through LDS (register-prefetched halo tile,
// as in k_pgbwd); x is needed at q alone (outer product, act' mask), so every lane loads its own 12 bytes straight from global,
// software-pipelined one tile ahead row by row.  256-thread blocks, three per CU (<= 168 registers).
// Same slab / fold protocol and the same pool-fold transform (PF) as k_pgbwd.
template <bool PF, int ABL = 0>      // ABL (tuning builds): 1 = no FMAs (traffic, staging and barriers only)
__global__ __launch_bounds__(256, 2) void k_bwd3v(BwdArgs p, const float* __restrict__ wts) {
    constexpr int C = 3, CO = 3, NT = 256, TW = 128, R = 4;
    static_assert(TH == 2 * R, "two wave rows of R output rows");
    using TGg = TG<CO, TW>;
    constexpr int LS = TGg::LS, LS4 = TGg::LS4, N4 = TGg::N4, NPF = TGg::npf(NT), WRw = 18, MT = 4;   // WRw / MT: slab geometry of k_pgbwd<3,1,3>
    constexpr int PFR = TH / 2 + 2, PFW = TW / 2 + 2, PFLEAD = 1, PFLS = (PFLEAD + PFW * CO + 3) / 4 * 4, PFN4 = PF ? PFR * PFLS / 4 : 0;
    __shared__ float4 lds4[N4 + PFN4 + (PFN4 + 3) / 4 + 84];
    const float* gl = reinterpret_cast<const float*>(lds4);     // dz tile
    float* pf_dp = reinterpret_cast<float*>(lds4 + N4);
    unsigned* pf_ix = reinterpret_cast<unsigned*>(lds4 + N4 + PFN4);
    float* red = reinterpret_cast<float*>(lds4 + N4 + PFN4 + (PFN4 + 3) / 4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntiles = p.tiles_x * p.tiles_y * p.B;              // the host guarantees W % TW == 0 and H % TH == 0: whole tiles only
    const bool xcd_map = (ntiles & 7) == 0 && (gridDim.x & 7) == 0;
    const int rowlen4 = p.W * CO / 4;
    const float slope = p.mask[0] ? p.alpha : 1.0f;             // act'(x) for x <= 0 (1: no mask)

    // the 81 weights as wave-uniform values (scalar registers): w[((dy*3 + kx)*3 + ci)*3 + co], HWIO
    float w[81];
#pragma unroll
    for (int i = 0; i < 81; ++i) w[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wts[i])));

    const int col = (wave & 1) * 64 + lane, rbase = (wave >> 1) * R;     // this lane's pixel column and first output row in the tile

    float4 preg[NPF], prey[PF ? NPF : 1];
    unsigned okg = 0;
    constexpr int NPS = PF ? (PFN4 + NT - 1) / NT : 1;
    float4 predp[NPS];
    unsigned preix[NPS];
    float xq[R][3];               // x of this lane's R pixels: loaded one tile ahead, row by row
    // tile t -> (grow0, x0, y0): grow0 = b*H + y0 is the tile's first row counted through the whole batch
    auto decode = [&](int t, int& grow0, int& x0, int& y0) {
        t = xcd_map ? (t & 7) * (ntiles >> 3) + (t >> 3) : t;
        const int trow = t / p.tiles_x;
        x0 = (t - trow * p.tiles_x) * TW;
        grow0 = trow * TH;
        y0 = (trow % p.tiles_y) * TH;
    };
    // staging slot k of this thread: (tile row, 16-byte column) -- recomputed, not kept (registers)
    auto slot = [&](int k, int& row, int& c4) {
        const int idx = tid + k * NT;
        row = idx / LS4;
        c4 = idx - row * LS4;
    };
    auto issue_tile = [&](float4* pre, const float* __restrict__ src, int grow0, int x0, int y0) {
        const float4* base = reinterpret_cast<const float4*>(src);
        const int g40 = (x0 * CO - TGg::HL) / 4;
        unsigned ok = 0;
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            int row, c4;
            slot(k, row, c4);
            const int iy = y0 - 1 + row, g4 = g40 + c4;
            const bool in = tid + k * NT < N4 && (unsigned)iy < (unsigned)p.H && (unsigned)g4 < (unsigned)rowlen4;
            ok |= in ? (1u << k) : 0u;
            pre[k] = base[in ? (grow0 - 1 + row) * rowlen4 + g4 : 0];
        }
        return ok;
    };
    auto issue = [&](int grow0, int x0, int y0) {
        okg = issue_tile(preg, p.dz, grow0, x0, y0);
        if constexpr (PF) {
            (void)issue_tile(prey, p.pf_y, grow0, x0, y0);
            const int Hp = p.H >> 1, rowf = (p.W >> 1) * CO;
#pragma unroll
            for (int k = 0; k < NPS; ++k) {
                const int id = tid + k * NT;
                const int r = id / (PFLS / 4), c4 = id - r * (PFLS / 4);
                const int gy = (y0 >> 1) - 1 + r, gf = ((x0 >> 1) - 1) * CO - PFLEAD + 4 * c4;
                const bool ok = id < PFN4 && (unsigned)gy < (unsigned)Hp && gf >= 0 && gf < rowf;
                const size_t off = ok ? ((size_t)((grow0 >> 1) - 1 + r)) * rowf + gf : 0;
                predp[k] = *reinterpret_cast<const float4*>(p.pf_dpool + off);
                if (!ok) predp[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                preix[k] = *reinterpret_cast<const unsigned*>(p.pf_idx + off);
                if (!ok) preix[k] = 0xffffffffu;
            }
        }
    };
    auto issue_x = [&](int rr, int grow0, int x0) {
        const float* xp = p.x[0] + ((size_t)(grow0 + rbase + rr) * p.W + x0 + col) * C;
        xq[rr][0] = xp[0]; xq[rr][1] = xp[1]; xq[rr][2] = xp[2];
    };
    auto commit = [&]() {
        if constexpr (PF) {
#pragma unroll
            for (int k = 0; k < NPS; ++k) {
                const int id = tid + k * NT;
                if (id < PFN4) {
                    reinterpret_cast<float4*>(pf_dp)[id] = predp[k];
                    pf_ix[id] = preix[k];
                }
            }
            lds_barrier();
            const unsigned char* ixb = reinterpret_cast<const unsigned char*>(pf_ix);
#pragma unroll
            for (int k = 0; k < NPF; ++k) {
                int r, c4;
                slot(k, r, c4);
                float g[4] = {preg[k].x, preg[k].y, preg[k].z, preg[k].w};
                const float yv[4] = {prey[k].x, prey[k].y, prey[k].z, prey[k].w};
                const bool in = (okg >> k) & 1u;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int f = 4 * c4 + e - TGg::LEAD;
                    if (in && f >= 0 && f < (TW + 2) * CO) {
                        const int px = f / CO, ch = f - px * CO;
                        const int pr = ((r - 1) >> 1) + 1, pc = ((px - 1) >> 1) + 1;
                        const unsigned pos = (((unsigned)(r - 1) & 1u) << 1) | ((unsigned)(px - 1) & 1u);
                        const int o = PFLEAD + pc * CO + ch;
                        const float dp = ixb[pr * PFLS + o] == pos ? pf_dp[pr * PFLS + o] : 0.f;
                        g[e] = (g[e] + dp) * (yv[e] > 0.f ? 1.0f : p.pf_alpha);
                    }
                }
                preg[k] = make_float4(g[0], g[1], g[2], g[3]);
            }
        }
        tile_commit<CO, TW, NT>(preg, okg, lds4, tid);
    };

    float acc[84];                // acc[((wy*3 + wx)*3 + ci)*3 + co] = dW[2-wy][2-wx][ci][co] of this lane's pixels; acc[81 + co] = db[co]
#pragma unroll
    for (int i = 0; i < 84; ++i) acc[i] = 0.f;

    int tile = blockIdx.x, it = 0;
    (void)it;          // read by the tuning build's stamps only
    STAMP(6);
    if (tile < ntiles) {
        int grow0, x0, y0;
        decode(tile, grow0, x0, y0);
        issue(grow0, x0, y0);
#pragma unroll
        for (int rr = 0; rr < R; ++rr) issue_x(rr, grow0, x0);
    }
#pragma unroll 1
    while (tile < ntiles) {
        int grow0, x0, y0;
        decode(tile, grow0, x0, y0);
        STAMP(0);
        commit();
        lds_barrier();
        STAMP(1);
        // The prefetch is issued unconditionally (past the end: the last tile again, never committed): with every vector-memory
        // operation of the loop in straight-line code hipcc counts its s_waitcnt vmcnt(N) exactly; a conditional load made it wait
        // for the prefetch it had just issued (vmcnt retires in order), a full HBM round trip per tile.
        const int next = tile + gridDim.x;
        int ngrow0, nx0, ny0;
        decode(min(next, ntiles - 1), ngrow0, nx0, ny0);
        issue(ngrow0, nx0, ny0);
        STAMP(2);
        const float* gb = gl + rbase * LS + TGg::LEAD + col * CO;     // tile row rbase = output row rbase - 1; pixels col-1 .. col+1
        float* dst = p.dx[0] + ((size_t)(grow0 + rbase) * p.W + x0 + col) * C;
        float win[3][9];
#pragma unroll
        for (int i = 0; i < 9; ++i) { win[0][i] = gb[i]; win[1][i] = gb[LS + i]; }
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
#pragma unroll
            for (int i = 0; i < 9; ++i) win[(rr + 2) % 3][i] = gb[(rr + 2) * LS + i];
            const float xv[3] = {xq[rr][0], xq[rr][1], xq[rr][2]};
            float dx[3] = {0.f, 0.f, 0.f};
            if constexpr (ABL & 1) {
#pragma unroll
                for (int i = 0; i < 9; ++i) dx[i % 3] += win[(rr + 2) % 3][i] + xv[i % 3];
            } else
#pragma unroll
            for (int wy = 0; wy < 3; ++wy)
#pragma unroll
                for (int wx = 0; wx < 3; ++wx)
#pragma unroll
                    for (int co = 0; co < 3; ++co) {
                        const float d = win[(rr + wy) % 3][wx * 3 + co];
#pragma unroll
                        for (int ci = 0; ci < 3; ++ci) {
                            dx[ci] = fmaf(d, w[(((2 - wy) * 3 + (2 - wx)) * 3 + ci) * 3 + co], dx[ci]);
                            acc[((wy * 3 + wx) * 3 + ci) * 3 + co] = fmaf(xv[ci], d, acc[((wy * 3 + wx) * 3 + ci) * 3 + co]);
                        }
                    }
#pragma unroll
            for (int co = 0; co < 3; ++co) acc[81 + co] += win[(rr + 1) % 3][3 + co];
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) dx[ci] *= xv[ci] > 0.f ? 1.0f : slope;
            float* d3 = dst + (size_t)rr * p.W * C;
            d3[0] = dx[0]; d3[1] = dx[1]; d3[2] = dx[2];
            issue_x(rr, ngrow0, nx0);                              // this row's registers are free: the next tile's x row
        }
        STAMP(3);
        lds_barrier();
        STAMP(4);
        ++it;
        tile = next;
    }
    STAMP_END(5);
    // lane sums by DPP, the four waves through LDS, one atomic per element into slab (blockIdx % NBUCKET), laid out where
    // k_pg_fold expects D[(dy, j), (dx, co)]: everything in the dx = 0 entries (j = kx*3 + ci), the bias in the all-ones row
#pragma unroll
    for (int i = 0; i < 84; ++i) {
        const float t = wave_total_l63(acc[i]);
        if (lane == 63) red[wave * 84 + i] = t;
    }
    __syncthreads();
    if (tid < 84) {
        const int e = tid;
        const float v = (red[e] + red[84 + e]) + (red[168 + e] + red[252 + e]);
        int mrow, co;
        if (e < 81) {
            const int tap = e / 9, ci = (e / 3) % 3;
            co = e % 3;
            mrow = (2 - tap / 3) * WRw + (2 - tap % 3) * 3 + ci;
        } else {
            mrow = 3 * WRw;
            co = e - 81;
        }
        atomicAdd(p.slabs[0] + (size_t)(blockIdx.x % NBUCKET) * (MT * 256) + slab_index(mrow, co), v);
    }
    STAMP_END(7);
}

// ================================================================================================ fold
struct FoldDesc {
    int slab_off, nslabs, MT;      // slabs of this (op, source): floats offset into the slab buffer
    int C, G, CO;                  // geometry of the pixel-group GEMM
    int cin_total, ci_off;         // where this source's channels sit in the HWIO kernel
    int w_off, b_off;              // gradient destinations (floats into the flat gradient vector); b_off < 0: no bias
    int kind;                      // 0: 3x3 conv (dx-diagonal fold), 1: 2x2 transposed conv (plain sum; C = Cin, CO = Cout)
};

__device__ __forceinline__ int slab_index(int mrow, int n) {
    int t = mrow >> 4, r = mrow & 3, qq = (mrow & 15) >> 2;
    return (t * 4 + r) * 64 + qq * 16 + n;
}

// dW[dy][kx][ci][co] = sum_slabs sum_dx D[(dy, (dx + kx)*C + ci), (dx, co)];  db[co] = sum_dx D[ones row, (dx, co)]
// Every block first sums the NBUCKET slabs into LDS (coalesced, independent loads), then folds its 256 outputs.
// Blocks with blockIdx.x >= nfolds (y == 0) reduce the fused head's per-block partials instead (k_head_reduce's job: value
// k = blockIdx.x - nfolds of [dW (C), db, loss sum]), so that the backward pass ends in one launch.
struct HeadTail {
    const float* partials;
    int nblocks, C;
    float* dw;
    float* dbias;
    double* scalars;
};

// ad.p != nullptr (single-replica train steps whose every gradient comes out of this launch): the thread that finishes a gradient
// applies the Adam update to that parameter at once (g_adam's arithmetic), and the block that finishes the loss sum writes the
// step outputs (k_finalize_scalars' job) -- the optimizer needs no launch of its own.
struct AdamTail {
    float* p;
    float* m;
    float* v;
    float lr_t, b1, b2, eps;
    dnnca_loss_cfg cfg;
    double n_label, inv_batch_hw;
    float* out5;
};
__device__ __forceinline__ void fold_out(float* __restrict__ grads, const AdamTail& ad, int i, float gi) {
    grads[i] = gi;
    if (ad.p) {
        const float mi = ad.m[i] * ad.b1 + gi * (1.f - ad.b1);
        const float vi = ad.v[i] * ad.b2 + (gi * gi) * (1.f - ad.b2);
        ad.m[i] = mi;
        ad.v[i] = vi;
        ad.p[i] = ad.p[i] - ad.lr_t * mi / (sqrtf(vi) + ad.eps);
    }
}

__global__ __launch_bounds__(256) void k_pg_fold(const FoldDesc* __restrict__ descs, const float* __restrict__ slabs,
                                                 float* __restrict__ grads, int nfolds, HeadTail h, AdamTail ad) {
    __shared__ float Dl[7 * 256];
    if ((int)blockIdx.x >= nfolds) {
        if (blockIdx.y != 0) return;
        double* red = reinterpret_cast<double*>(Dl);
        const int k = blockIdx.x - nfolds;
        double s = 0.0;
        for (int i = threadIdx.x; i < h.nblocks; i += 256) s += (double)h.partials[i * (h.C + 2) + k];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            if (k < h.C) fold_out(grads, ad, (int)(h.dw - grads) + k, (float)red[0]);       // (h.dw / h.dbias point into grads)
            else if (k == h.C) fold_out(grads, ad, (int)(h.dbias - grads), (float)red[0]);
            else {
                h.scalars[3] = red[0];
                if (ad.p) {          // the step outputs (no L2 penalty on this path: scalars[4] == 0)
                    const double lsum = h.scalars[0];
                    float wgt;
                    if (ad.cfg.has_weight) wgt = ad.cfg.weight;
                    else { const float pr = (float)(lsum / ad.n_label); wgt = pr > 0.f ? 1.0f / pr : 1.0f; }
                    ad.out5[0] = (float)(red[0] * ad.inv_batch_hw + h.scalars[4]);
                    ad.out5[1] = (float)(lsum / ad.n_label);
                    ad.out5[2] = ad.cfg.weight_mul * wgt + ad.cfg.weight_add;
                    ad.out5[3] = (float)h.scalars[1];
                    ad.out5[4] = (float)h.scalars[2];
                }
            }
        }
        return;
    }
    const FoldDesc d = descs[blockIdx.x];
    {
        const int total = d.kind == 1 ? 4 * d.CO * d.C + d.CO : 9 * d.C * d.CO + (d.b_off >= 0 ? d.CO : 0);
        if ((int)(blockIdx.y * blockDim.x) >= total) return;     // block-uniform
    }
    const float* S = slabs + d.slab_off;
    const int stride = d.MT * 256;
    // every slab value of this fold in flight at once (MT <= 7 chunks of 256 x NBUCKET slabs): one memory round trip, not one per chunk
    {
        float v[7][NBUCKET];
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            if (c < d.MT) {
#pragma unroll
                for (int s = 0; s < NBUCKET; ++s) v[c][s] = S[(size_t)s * stride + c * 256 + threadIdx.x];
            }
        }
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            if (c < d.MT) {
#pragma unroll
                for (int w = NBUCKET / 2; w > 0; w >>= 1)
#pragma unroll
                    for (int s = 0; s < w; ++s) v[c][s] += v[c][s + w];
                Dl[c * 256 + threadIdx.x] = v[c][0];
            }
        }
    }
    __syncthreads();
    const int o = blockIdx.y * blockDim.x + threadIdx.x;
    if (d.kind == 1) {
        // rows m = (a, e, co) are exactly the [r, r, Cout, Cin] kernel rows; column Cin is the all-ones (bias) column
        const int nwt = 4 * d.CO * d.C;
        if (o < nwt) {
            fold_out(grads, ad, d.w_off + o, Dl[slab_index(o / d.C, o % d.C)]);
        } else if (o < nwt + d.CO) {
            const int co = o - nwt;
            float a = 0.f;
            for (int ae = 0; ae < 4; ++ae) a += Dl[slab_index(ae * d.CO + co, d.C)];
            fold_out(grads, ad, d.b_off + co, a);
        }
        return;
    }
    const int WR = (d.G + 2) * d.C;
    const int nw = 9 * d.C * d.CO;
    const int total = nw + (d.b_off >= 0 ? d.CO : 0);
    if (o >= total) return;
    float acc = 0.f;
    if (o < nw) {
        int co = o % d.CO, ci = (o / d.CO) % d.C, tap = o / (d.CO * d.C);
        int dy = tap / 3, kx = tap % 3;
        for (int dx = 0; dx < d.G; ++dx) acc += Dl[slab_index(dy * WR + (dx + kx) * d.C + ci, dx * d.CO + co)];
        fold_out(grads, ad, d.w_off + ((tap * d.cin_total) + d.ci_off + ci) * d.CO + co, acc);
    } else {
        int co = o - nw;
        for (int dx = 0; dx < d.G; ++dx) acc += Dl[slab_index(3 * WR, dx * d.CO + co)];
        fold_out(grads, ad, d.b_off + co, acc);
    }
}

// ================================================================================================ transposed-conv wgrad
// dW[a][e][co][ci] = sum_pixels dout[2i+a][2j+e][co] * in[i][j][ci]  as  D[(a,e,co)][ci] with K = input pixels.
// Operands come straight from global memory (A: 16 consecutive floats of an output row per pixel; B: the pixel's
// channels) -- no reuse, so no LDS.  Column n = Cin of B is all ones: D[.][Cin] is the bias gradient.
struct TwArgs {
    const float* in;
    const float* dout;
    float* slabs;
    int W;           // input width
    int nquads;      // B*H*W / 4
};

template <int CIN, int COUT>
__device__ __forceinline__ void tconv_wgrad_body(const TwArgs& p, const int bid, const int nblocks) {
    constexpr int MROWS = 4 * COUT, MT = (MROWS + 15) / 16;
    __shared__ float red[4 * MT * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m16 = lane & 15, q = lane >> 4, n = m16;
    int aoff[MT];
    bool valid[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        int mrow = 16 * t + m16;
        valid[t] = mrow < MROWS;
        int a = mrow / (2 * COUT), r = mrow % (2 * COUT);
        aoff[t] = a * (2 * p.W) * COUT + r;       // relative to the pixel's first output row / column
    }
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int gw = bid * 4 + wave, nw = nblocks * 4;
    constexpr int UQ = 4;          // quads per iteration: all their loads are in flight before the first MFMA
    for (int quad0 = gw * UQ; quad0 < p.nquads; quad0 += nw * UQ) {
        float av[UQ][MT], bv[UQ];
#pragma unroll
        for (int u = 0; u < UQ; ++u) {
            const int quad = quad0 + u;
            const bool ok = quad < p.nquads;
            const int pix = (ok ? quad : 0) * 4 + q;
            const int j = pix % p.W, bi = pix / p.W;
            bv[u] = !ok ? 0.0f : (n < CIN ? p.in[(size_t)pix * CIN + n] : (n == CIN ? 1.0f : 0.0f));
            const float* dp = p.dout + ((size_t)bi * 2 * (2 * p.W) + 2 * j) * COUT;
#pragma unroll
            for (int t = 0; t < MT; ++t) av[u][t] = (ok && valid[t]) ? dp[aoff[t]] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < UQ; ++u)
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][t], bv[u], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * MT * 4 + t * 4 + r) * 64 + lane] = acc[t][r];
    __syncthreads();
    float* slab = p.slabs + (size_t)(bid % NBUCKET) * (MT * 256);
    for (int i = tid; i < MT * 256; i += 256)
        atomicAdd(slab + i, (red[i] + red[MT * 256 + i]) + (red[2 * MT * 256 + i] + red[3 * MT * 256 + i]));
}

// the whole backward of a small transposed conv in one launch: blocks [0, nbw) accumulate the weight gradient, the rest
// compute the data gradient (both only read dout; a launch of its own costs each of them more than its work at 64^2..256^2)
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void k_tconv_bwd(TwArgs w, TdArgs d, int nbw) {
    if ((int)blockIdx.x < nbw) tconv_wgrad_body<CIN, COUT>(w, blockIdx.x, nbw);
    else tconv2_dgrad_body<CIN, COUT>(d, ((int)blockIdx.x - nbw) * 256 + threadIdx.x);
}

// ================================================================================================ host side
struct PgPlan {                      // per-model table built lazily on the first step
    bool built = false;
    std::vector<PrepDesc> descs;
    PrepDesc* descs_dev = nullptr;
    float* bmat = nullptr;
    int* bindex = nullptr;
    int bmat_n = 0;
    std::map<std::pair<const Op*, int>, int> slot;   // (op, kind) -> bmat offset;  kind 0 fwd, 1 dgrad (all passes)
    std::vector<FoldDesc> folds;
    FoldDesc* folds_dev = nullptr;
    float* slabs = nullptr;
    size_t slab_floats = 0;
    std::map<std::pair<const Op*, int>, int> wslot;  // (op, source) -> index into folds (transposed convs: source 0)
    int fold_chunks = 0;
    PrepRide prep{};                // this step's k_pg_prep arguments (fast_prepare)
    int64_t fold_outputs = 0;      // gradient elements k_pg_fold writes (all folds); with the head's C + 1: == nT when the plan covers the model
    HeadTail pending_head{};       // fast_finish_backward -> fast_fold_adam
    int nblocks_cap = 512;
    bool nblocks_forced = false;       // DNNCA_NBLOCKS (tuning aid) overrides the occupancy-derived grids
    int nthreads = 512;
    bool double_buffer = true;
    unsigned long long* stamps = nullptr;
};

static std::map<Model*, PgPlan> g_plans;

void fast_release(Model* m) { g_plans.erase(m); }
unsigned long long* fast_debug_stamps(Model* m) { return g_plans[m].stamps; }

static inline bool dense(const View& v) { return v.C == 0 || v.ps == v.C; }

static bool conv_supported(const Model* m, const Op& o) {
    if (o.type != OP_CONV || o.k != 3) return false;
    if (!dense(o.inA.d) || !dense(o.inB.d) || !dense(o.out.d)) return false;
    const int CA = o.inA.d.C, CB = o.inB.d.C, CO = o.out.d.C;
    if (CB && CB != CA) return false;
    if (o.out.d.W % 4) return false;
    const int ns = CB ? 2 : 1;
    // the instantiated (C, NSRC, CO) set: every 3x3 conv of configs/unet.yaml
    if (ns == 1)
        return (CA == 1 && CO == 3) || (CA == 3 && (CO == 3 || CO == 6)) || (CA == 6 && (CO == 6 || CO == 12)) ||
               (CA == 12 && CO == 12);
    return (CA == 3 && CO == 3) || (CA == 6 && CO == 6) || (CA == 12 && CO == 12);
}

// a MaxPool2D([2,2], 2) that directly consumes this conv's output can ride in the conv's epilogue
bool fast_pool_fusable(const Model* m, const Op& conv, const Op& pool) {
    return conv_supported(m, conv) && pool.type == OP_POOL && pool.k == 2 && pool.inA.d.p == conv.out.d.p && dense(pool.out.d) &&
           conv.out.d.H % 2 == 0 && conv.out.d.W % 2 == 0 && ((conv.out.d.W / 2) * conv.out.d.C) % 4 == 0;
}

static int build_plan(Model* m, PgPlan& pl) {
    int off = 0;
    auto add = [&](const Op* op, int kind, PrepDesc d, bool first) {
        int WR = (d.G + 2) * d.C, SR = (WR + 3) / 4;
        d.out_off = off;
        if (first) pl.slot[{op, kind}] = off;
        off += d.NSRC * 3 * SR * 64;
        pl.descs.push_back(d);
    };
    size_t slab_floats = 0;
    int max_out = 0;
    for (const Op& o : m->ops) {
        if (!conv_supported(m, o)) continue;
        const int CA = o.inA.d.C, CB = o.inB.d.C, CO = o.out.d.C, ns = CB ? 2 : 1;
        PrepDesc d{};
        d.w_off = (int)o.w_off;
        d.C = CA; d.NSRC = ns; d.CO = CO; d.G = 12 / CO; d.mode = 0;
        d.cin_total = CA + CB; d.cout_total = CO; d.o_off = 0;
        add(&o, 0, d, true);
        if (o.need_din) {
            // data gradient: input = dz (CO channels); NPASS passes of COd output channels each (see BW<>)
            const int ctot = CA + CB, npass = ctot <= 12 ? 1 : ns, cod = ctot / npass;
            for (int ps = 0; ps < npass; ++ps) {
                PrepDesc g{};
                g.w_off = (int)o.w_off; g.C = CO; g.NSRC = 1; g.mode = 1; g.cin_total = ctot; g.cout_total = CO;
                g.CO = cod; g.G = 12 / cod; g.o_off = ps * cod;
                add(&o, 1, g, ps == 0);
            }
        }
        const int Gw = 12 / CO;
        for (int sidx = 0; sidx < ns; ++sidx) {
            FoldDesc f{};
            f.kind = 0;
            f.C = CA; f.G = Gw; f.CO = CO;
            f.MT = (3 * (Gw + 2) * CA + 1 + 15) / 16;
            f.nslabs = NBUCKET;
            f.slab_off = (int)slab_floats;
            f.cin_total = CA + CB; f.ci_off = sidx * CA;
            f.w_off = (int)o.w_off;
            f.b_off = sidx == 0 ? (int)o.b_off : -1;
            slab_floats += (size_t)f.nslabs * f.MT * 256;
            pl.wslot[{&o, sidx}] = (int)pl.folds.size();
            pl.folds.push_back(f);
            int outs = 9 * CA * CO + CO;
            if (outs > max_out) max_out = outs;
            pl.fold_outputs += 9 * CA * CO + (f.b_off >= 0 ? CO : 0);
        }
    }
    for (const Op& o : m->ops) {
        if (!fast_tconv_supported(m, o)) continue;
        FoldDesc f{};
        f.kind = 1;
        f.C = o.inA.d.C; f.CO = o.out.d.C; f.G = 1;
        f.MT = (4 * f.CO + 15) / 16;
        f.nslabs = NBUCKET;
        f.slab_off = (int)slab_floats;
        f.w_off = (int)o.w_off;
        f.b_off = (int)o.b_off;
        slab_floats += (size_t)f.nslabs * f.MT * 256;
        pl.wslot[{&o, 0}] = (int)pl.folds.size();
        pl.folds.push_back(f);
        int outs = 4 * f.CO * f.C + f.CO;
        if (outs > max_out) max_out = outs;
        pl.fold_outputs += outs;
    }
    pl.built = true;
    if (const char* e = getenv("DNNCA_DB")) pl.double_buffer = atoi(e) != 0;            // tuning aid
    if (const char* e = getenv("DNNCA_NBLOCKS")) { pl.nblocks_cap = atoi(e) > 0 ? atoi(e) : pl.nblocks_cap; pl.nblocks_forced = atoi(e) > 0; }   // tuning aid
    if (pl.folds.empty() && pl.descs.empty()) return DNNCA_OK;
    pl.fold_chunks = (max_out + 255) / 256;
    pl.slab_floats = slab_floats;
    if (!pl.folds.empty()) {
        DN_TRY(m->alloc((void**)&pl.folds_dev, pl.folds.size() * sizeof(FoldDesc)));
        DN_TRY(m->alloc((void**)&pl.slabs, (slab_floats + 4) * 4));
        m->extra_zero = pl.slabs;           // accumulated with atomics: model.hip zeroes them at the top of each backward
        m->head_defer_ok = true;            // k_pg_fold ends every backward pass: it also reduces the fused head's partials
        m->extra_zero_n = slab_floats;
        HIP_TRY(hipMemcpyAsync(pl.folds_dev, pl.folds.data(), pl.folds.size() * sizeof(FoldDesc), hipMemcpyHostToDevice, m->stream));
    }
    if (!pl.descs.empty()) {
        DN_TRY(m->alloc((void**)&pl.descs_dev, pl.descs.size() * sizeof(PrepDesc)));
        DN_TRY(m->alloc((void**)&pl.bmat, (size_t)off * 4));
        DN_TRY(m->alloc((void**)&pl.bindex, (size_t)off * 4));
        pl.bmat_n = off;
        HIP_TRY(hipMemcpyAsync(pl.descs_dev, pl.descs.data(), pl.descs.size() * sizeof(PrepDesc), hipMemcpyHostToDevice, m->stream));
        hipLaunchKernelGGL(k_pg_prep_index, dim3((unsigned)pl.descs.size()), dim3(256), 0, m->stream, pl.descs_dev, pl.bindex);
    }
    HIP_TRY(hipStreamSynchronize(m->stream));
    return DNNCA_OK;
}

// the prepared forward B operand of a pixel-group conv (for the block-fused kernels of kernels_fused.hip); nullptr: not planned
const float* fast_conv_bmat(Model* m, const Op& o) {
    if (!conv_supported(m, o)) return nullptr;
    PgPlan& pl = g_plans[m];
    auto it = pl.slot.find({&o, 0});
    return it == pl.slot.end() ? nullptr : pl.bmat + it->second;
}

bool fast_pg_conv_supported(const Model* m, const Op& o) { return conv_supported(m, o); }
const float* fast_conv_bmat_dgrad(Model* m, const Op& o) {
    if (!conv_supported(m, o)) return nullptr;
    PgPlan& pl = g_plans[m];
    auto it = pl.slot.find({&o, 1});
    return it == pl.slot.end() ? nullptr : pl.bmat + it->second;
}
float* fast_wgrad_slabs(Model* m, const Op& o, int source) {
    PgPlan& pl = g_plans[m];
    auto it = pl.wslot.find({&o, source});
    return it == pl.wslot.end() ? nullptr : pl.slabs + pl.folds[it->second].slab_off;
}

// Called by model.hip at the top of every forward: (re)derive the B operands from the current weights.
int fast_prepare(Model* m) {
    if (m->desc.flags & 1) return DNNCA_OK;
    PgPlan& pl = g_plans[m];
    if (!pl.built) DN_TRY(build_plan(m, pl));
    if (pl.descs.empty()) return DNNCA_OK;
    const int nprep = (pl.bmat_n + 255) / 256;
    StepInit z{};
    int nz = 0;
    if (m->defer_head && m->merged_launches() && !m->dry) {      // dnnca_train_step: a backward pass follows this forward pass
        z.scalars = m->scalars;
        z.a = reinterpret_cast<float4*>(m->g);
        z.na4 = (unsigned)((m->nT + 8 + 3) / 4);                // both buffers are allocated with >= 16 bytes of slack
        z.b = reinterpret_cast<float4*>(m->extra_zero);
        z.nb4 = (unsigned)((m->extra_zero_n + 3) / 4);
        nz = (int)((z.na4 + z.nb4 + 1023) / 1024);
        nz = nz < 1 ? 1 : (nz > 1024 ? 1024 : nz);
        m->step_init_done = true;
    }
    pl.prep = PrepRide{pl.bindex, m->p, pl.bmat, pl.bmat_n, nprep, nprep + nz, z};
    if (nz && !getenv("DNNCA_NO_PREP_RIDE")) {
        // train step: the first launch of the forward pass decides -- the first encoder block's strip kernel takes the preparation
        // along as extra blocks (it needs none of its results); any other launch flushes it first (LAUNCH)
        m->prep_flush = [](Model* mm) {
            PgPlan& q = g_plans[mm];
            LAUNCH(mm, "pg_prep", 0, 0,
                   hipLaunchKernelGGL(k_pg_prep, dim3(q.prep.nblocks), dim3(256), 0, mm->stream, q.prep.index, q.prep.params, q.prep.bmat,
                                      q.prep.n, q.prep.nprep, q.prep.z));
        };
        return DNNCA_OK;
    }
    LAUNCH(m, "pg_prep", 0, 0,
           hipLaunchKernelGGL(k_pg_prep, dim3(nprep + nz), dim3(256), 0, m->stream, pl.bindex, m->p, pl.bmat, pl.bmat_n, nprep, z));
    return DNNCA_OK;
}

// Grid of a persistent pixel-group kernel: as many 512-thread blocks as fit the chip at once (two or three per CU for the kernels whose
// registers allow it, one for the heavy backward kernels -- launching 512 blocks of those ran them in two rounds and paid the
// per-block prologue twice: 4-6 us on each of six kernels of the unet.yaml step).
template <typename K>
static int resident_blocks(K kernel, int cap, int nt = 512) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, nt, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    static const int maxocc = getenv("DNNCA_PG_MAXOCC") ? atoi(getenv("DNNCA_PG_MAXOCC")) : 3;      // tuning aid (2: -0.3 %, 4: same)
    if (per_cu > maxocc) per_cu = maxocc;
    const int n = 256 * per_cu;
    return n < cap ? n : cap;
}

#define CONV_SHAPES(X) X(1, 1, 3) X(3, 1, 3) X(3, 2, 3) X(3, 1, 6) X(6, 1, 6) X(6, 2, 6) X(6, 1, 12) X(12, 1, 12) X(12, 2, 12)

bool fast_conv_fwd(Model* m, int B, Op& o, double bytes, double flops, Op* pool) {
    if (!conv_supported(m, o)) return false;
    PgPlan& pl = g_plans[m];
    auto it = pl.slot.find({&o, 0});
    if (it == pl.slot.end()) return false;
    FwdArgs a{};
    a.src[0] = o.inA.d.p; a.src[1] = o.inB.d.p;
    a.bmat = pl.bmat + it->second;
    a.bias = m->p + o.b_off;
    a.dst = o.out.d.p;
    a.pool_dst = pool ? pool->out.d.p : nullptr;     // the caller checked fast_pool_fusable(o, *pool)
    a.B = B; a.H = o.out.d.H; a.W = o.out.d.W;
    a.alpha = o.alpha;
    const int C = o.inA.d.C, NS = o.inB.d.C ? 2 : 1, CO = o.out.d.C;
    const int TW = 32 * (12 / CO);
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + TH - 1) / TH;
    int ntiles = a.tiles_x * a.tiles_y * B;
    int nb = ntiles < pl.nblocks_cap ? ntiles : pl.nblocks_cap;
    const bool db = pl.double_buffer && ntiles >= 3 * nb;     // LDS double buffering pays when a block walks several tiles
#define X(c, ns, co)                                                                                            \
    if (C == c && NS == ns && CO == co) {                                                                       \
        (void)db;                                                                                               \
        static const int fit = resident_blocks(k_pgfwd<c, ns, co, 512, false>, 1 << 20);                        \
        const int g = pl.nblocks_forced ? nb : (ntiles < fit ? ntiles : fit);                                   \
        LAUNCH(m, "pgfwd_" #c "x" #ns "_" #co, bytes, flops,                                                    \
               hipLaunchKernelGGL((k_pgfwd<c, ns, co, 512, false>), dim3(g), dim3(512), 0, m->stream, a));      \
        return true;                                                                                            \
    }
    CONV_SHAPES(X)
#undef X
    return false;
}

// The first conv of the network when `conv` is the second conv of the first encoder block and the pair can run its backward in
// one column-strip launch (k_first3, strip_dev.h); nullptr otherwise.
static Op* first3_conv0(Model* m, Op& conv) {
    if (getenv("DNNCA_NO_FIRST3")) return nullptr;
    const size_t i = (size_t)(&conv - m->ops.data());
    if (i < 1 || i >= m->ops.size()) return nullptr;
    Op& c0 = m->ops[i - 1];
    if (c0.type != OP_CONV || c0.need_din || c0.inB.d.C || c0.inA.d.C != 1 || c0.out.d.C != 3 || c0.k != 3 || c0.out.d.p != conv.inA.d.p) return nullptr;
    if (!conv_supported(m, c0) || !conv_supported(m, conv) || conv.inB.d.C || conv.inA.d.C != 3 || conv.out.d.C != 3 || conv.accA) return nullptr;
    if (!dense(c0.inA.d) || !dense(conv.inA.d) || !dense(conv.out.d) || !dense(conv.out.g) || (c0.alpha >= 0.f && !c0.premasked)) return nullptr;
    if ((c0.alpha >= 0.f) != (conv.maskA != 0)) return nullptr;       // the activation derivative of the first conv rides in this launch
    const int H = conv.out.d.H, W = conv.out.d.W;
    if ((H & 1) || (W & 1) || W < 8 || H < 8 || (double)m->desc.max_batch * H * W * 12.0 >= 1073741824.0) return nullptr;
    PgPlan& pl = g_plans[m];
    if (pl.wslot.find({&c0, 0}) == pl.wslot.end() || pl.wslot.find({&conv, 0}) == pl.wslot.end()) return nullptr;
    return &c0;
}

bool fast_conv_bwd(Model* m, int B, Op& o, double out_bytes, double in_bytes, double flops) {
    if (!conv_supported(m, o)) return false;
    PgPlan& pl = g_plans[m];
    const int C = o.inA.d.C, NS = o.inB.d.C ? 2 : 1, CO = o.out.d.C;
    // dz = dy * act'(y) in place unless the consumers already delivered the pre-activation gradient
    if (o.alpha >= 0.f && !o.premasked)
        LAUNCH(m, "g_act_bwd", 3 * out_bytes, out_bytes / 4,
               g_act_bwd(m->stream, (size_t)B * o.out.d.H * o.out.d.W * o.out.d.C, o.out.g.p, o.out.d.p, o.alpha));
    BwdArgs a{};
    a.dz = o.out.g.p;
    a.x[0] = o.inA.d.p; a.x[1] = o.inB.d.p;
    if (o.need_din) {
        auto it = pl.slot.find({&o, 1});
        if (it == pl.slot.end()) return false;
        a.bmat = pl.bmat + it->second;
    }
    a.dx[0] = o.inA.g.p; a.dx[1] = o.inB.g.p;
    a.acc[0] = o.accA; a.acc[1] = o.accB;
    a.mask[0] = o.maskA; a.mask[1] = o.maskB;
    a.alpha = o.mask_alpha;
    if (const char* e = getenv("DNNCA_DBG")) a.dbg = atoi(e);
    if (const char* e = getenv("DNNCA_STAMPS")) {      // e = "C,NS,CO" of the kernel to stamp
        int sc = 0, sn = 0, so = 0;
        if (sscanf(e, "%d,%d,%d", &sc, &sn, &so) == 3 && sc == C && sn == NS && so == CO) {
            if (!pl.stamps) {
                if (m->alloc((void**)&pl.stamps, 1024 * 4 * 8 * 8) != DNNCA_OK) return false;
            }
            a.stamps = pl.stamps;
        }
    }
    for (int s = 0; s < NS; ++s) {
        auto it = pl.wslot.find({&o, s});
        if (it == pl.wslot.end()) return false;
        a.slabs[s] = pl.slabs + pl.folds[it->second].slab_off;
    }
    a.B = B; a.H = o.out.d.H; a.W = o.out.d.W;
    const int TW = 32 * (12 / CO);
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + TH - 1) / TH;
    int ntiles = a.tiles_x * a.tiles_y * B;
    int nb = ntiles < pl.nblocks_cap ? ntiles : pl.nblocks_cap;
    const bool db = pl.double_buffer && ntiles >= 3 * nb;
    const double bytes = out_bytes + (o.need_din ? 2 : 1) * in_bytes;
    const double fl = (o.need_din ? 2 : 1) * flops;
    // single-source 3 -> 3 channels: the all-vector-ALU backward (k_bwd3v), with or without the pool fold
    // the first encoder block: the second conv's backward (with the pool fold) and the first conv's weight gradient in one launch
    if (m->pool_fold.conv == &o) {
        if (Op* c0 = first3_conv0(m, o)) {
            const Op& pool = *m->pool_fold.pool;
            m->pool_fold.conv = nullptr;
            FirstArgs f{};
            f.dskip = o.out.g.p; f.y1 = o.out.d.p;
            f.dpool = pool.out.g.p; f.idx = pool.pool_idx;
            f.x1 = o.inA.d.p; f.xin = c0->inA.d.p;
            f.w = m->p + o.w_off;
            f.pf_alpha = pool.mask_alpha;
            f.mask = o.maskA; f.mask_alpha = o.mask_alpha;
            f.slabs1 = a.slabs[0];
            f.slabs0 = pl.slabs + pl.folds[pl.wslot.find({c0, 0})->second].slab_off;
            f.B = B; f.H = a.H; f.W = a.W;
            f.nstrips = (a.W + STRIP - 1) / STRIP;
            int nchunks = 2048 / (B * f.nstrips);
            if (nchunks > a.H / 8) nchunks = a.H / 8;
            if (nchunks < 1) nchunks = 1;
            f.nchunks = nchunks;
            const int nblk = (B * nchunks * f.nstrips + 3) / 4;
            const double npx = (double)B * a.H * a.W;
            // algorithmic bytes of the layers this launch stands for: pool backward (y, dy in; dx in/out; pooled gradient) 3 + 3 + 0.75 + 0.75,
            // second conv backward 3 + 3 + 3, first conv weight gradient 3 + 1 floats per pixel
            LAUNCH(m, "first3_bwd", 4.0 * npx * 20.5, 2.0 * npx * (162 + 30),
                   hipLaunchKernelGGL((k_first3<2, 40>), dim3(nblk), dim3(256), 0, m->stream, f));
            m->first_done = c0;
            return true;
        }
    }
    static const bool v3_on = getenv("DNNCA_NO_BWD3V") == nullptr;
    if (v3_on && o.need_din && C == 3 && CO == 3 && NS == 1 && !o.accA && a.W % 128 == 0 && a.H % TH == 0) {
        const bool pf = m->pool_fold.conv == &o;
        double pb = 0.0;
        if (pf) {
            const Op& pool = *m->pool_fold.pool;
            m->pool_fold.conv = nullptr;
            a.pf_y = o.out.d.p;
            a.pf_dpool = pool.out.g.p;
            a.pf_idx = pool.pool_idx;
            a.pf_alpha = pool.mask_alpha;
            pb = 4.0 * (2 * (double)B * o.out.d.H * o.out.d.W * CO + 2 * (double)B * pool.out.d.H * pool.out.d.W * CO);
        }
        const float* wts = m->p + o.w_off;
        if (a.stamps && pf != (getenv("DNNCA_STAMPS_PF") != nullptr)) a.stamps = nullptr;       // tuning aid: which of the two launches is stamped
        if (pf) {
            static const int fit = resident_blocks(k_bwd3v<true>, 1 << 20, 256);
            const int g = pl.nblocks_forced ? nb : (ntiles < fit ? ntiles : fit);
            LAUNCH(m, "bwd3v_pool_3x1_3", bytes + pb, fl, hipLaunchKernelGGL(k_bwd3v<true>, dim3(g), dim3(256), 0, m->stream, a, wts));
        } else {
            static const int fit = resident_blocks(k_bwd3v<false>, 1 << 20, 256);
            const int g = pl.nblocks_forced ? nb : (ntiles < fit ? ntiles : fit);
#ifdef DNNCA_TUNING
            static const int abl = getenv("DNNCA_ABL") ? atoi(getenv("DNNCA_ABL")) : 0;
            if (abl == 1) {
                LAUNCH(m, "bwd3v_3x1_3", bytes, fl, hipLaunchKernelGGL((k_bwd3v<false, 1>), dim3(g), dim3(256), 0, m->stream, a, wts));
                return true;
            }
#endif
            LAUNCH(m, "bwd3v_3x1_3", bytes, fl, hipLaunchKernelGGL(k_bwd3v<false>, dim3(g), dim3(256), 0, m->stream, a, wts));
        }
        return true;
    }
    // this conv's output feeds a max-pool whose backward has been folded into this launch (fast_pool_fold)
    if (m->pool_fold.conv == &o) {
        const Op& pool = *m->pool_fold.pool;
        m->pool_fold.conv = nullptr;
        a.pf_y = o.out.d.p;
        a.pf_dpool = pool.out.g.p;
        a.pf_idx = pool.pool_idx;
        a.pf_alpha = pool.mask_alpha;
        const double pb = 4.0 * (2 * (double)B * o.out.d.H * o.out.d.W * CO + 2 * (double)B * pool.out.d.H * pool.out.d.W * CO);
#define PFX(c)                                                                                                          \
        if (C == c) {                                                                                                   \
            static const int fit = resident_blocks(k_pgbwd<c, 1, c, true, 512, false, false, true>, 1 << 20);           \
            const int g = pl.nblocks_forced ? nb : (ntiles < fit ? ntiles : fit);                                       \
            LAUNCH(m, "pgbwd_pool_" #c "x1_" #c, bytes + pb, fl,                                                        \
                   hipLaunchKernelGGL((k_pgbwd<c, 1, c, true, 512, false, false, true>), dim3(g), dim3(512), 0, m->stream, a)); \
            return true;                                                                                                \
        }
        PFX(3) PFX(6) PFX(12)
#undef PFX
        return false;
    }
    // 3 -> 3 channels: the variant whose weight gradient runs on the vector ALU beside the data-gradient MFMAs
    // (measured, profiles/r02_vw_ab.txt: two sources 52.4 -> 48.2 us; one source 31.4 -> 32.0 us -- there the matrix pipe is not
    //  what the data-gradient waves wait for -- so the single-source conv keeps the all-MFMA kernel unless DNNCA_VW_ALL is set)
    static const bool vw_on = getenv("DNNCA_NO_VW") == nullptr;
    static const bool vw_all = getenv("DNNCA_VW_ALL") != nullptr;
    if (vw_on && o.need_din && C == 3 && CO == 3 && (NS == 2 || vw_all)) {
        if (NS == 1) {
            static const int fit = resident_blocks(k_pgbwd<3, 1, 3, true, 512, false, true>, 1 << 20);
            const int g = pl.nblocks_forced ? nb : (ntiles < fit ? ntiles : fit);
            LAUNCH(m, "pgbwd_3x1_3", bytes, fl,
                   hipLaunchKernelGGL((k_pgbwd<3, 1, 3, true, 512, false, true>), dim3(g), dim3(512), 0, m->stream, a));
        } else {
            // the first source is the output of a 6 -> 3 transposed conv (the last decoder block): that layer's backward rides along
            const size_t oi = (size_t)(&o - m->ops.data());
            Op* tc = oi >= 1 && oi < m->ops.size() ? &m->ops[oi - 1] : nullptr;
            auto tw = tc ? pl.wslot.find({tc, 0}) : pl.wslot.end();
            if (tc && !getenv("DNNCA_NO_TCF") && tc->type == OP_TCONV && tc->k == 2 && tc->inA.d.C == 6 && tc->out.d.C == 3 &&
                tc->out.d.p == o.inA.d.p && tw != pl.wslot.end() && fast_tconv_supported(m, *tc) && !tc->accA && !o.accA && !o.maskA &&
                dense(tc->inA.d) && dense(tc->inA.g) && dense(o.inA.d) && tc->inA.d.H * 2 == a.H && tc->inA.d.W * 2 == a.W && a.W % 128 == 0 &&
                a.H % TH == 0) {
                a.tc_in = tc->inA.d.p; a.tc_din = tc->inA.g.p; a.tc_w = m->p + tc->w_off;
                a.tc_slabs = pl.slabs + pl.folds[tw->second].slab_off;
                a.tc_mask = tc->maskA; a.tc_alpha = tc->mask_alpha;
                static const int fit = resident_blocks(k_pgbwd<3, 2, 3, true, 512, false, true, false, true>, 1 << 20);
                const int g = pl.nblocks_forced ? nb : (ntiles < fit ? ntiles : fit);
                const double tb = 4.0 * ((double)B * a.H * a.W * 3 + (double)B * (a.H / 2) * (a.W / 2) * 6);      // the transposed conv's out + in
                LAUNCH(m, "pgbwd_tc_3x2_3", bytes + 2 * tb, fl + 4.0 * B * a.H * a.W * 3 * 6,
                       hipLaunchKernelGGL((k_pgbwd<3, 2, 3, true, 512, false, true, false, true>), dim3(g), dim3(512), 0, m->stream, a));
                m->tconv_done = tc;
                return true;
            }
            static const int fit = resident_blocks(k_pgbwd<3, 2, 3, true, 512, false, true>, 1 << 20);
            const int g = pl.nblocks_forced ? nb : (ntiles < fit ? ntiles : fit);
            LAUNCH(m, "pgbwd_3x2_3", bytes, fl,
                   hipLaunchKernelGGL((k_pgbwd<3, 2, 3, true, 512, false, true>), dim3(g), dim3(512), 0, m->stream, a));
        }
        return true;
    }
    // 6 / 12 channels, two sources, the first one the output of a 12 -> C transposed conv: that layer's backward rides along (TCM)
    if (o.need_din && NS == 2 && C == CO && (C == 6 || C == 12) && !getenv("DNNCA_NO_TCM")) {
        const size_t oi = (size_t)(&o - m->ops.data());
        Op* tc = oi >= 1 && oi < m->ops.size() ? &m->ops[oi - 1] : nullptr;
        auto tw = tc ? pl.wslot.find({tc, 0}) : pl.wslot.end();
        const int TWc = 32 * (12 / CO);
        if (tc && tc->type == OP_TCONV && tc->k == 2 && tc->inA.d.C == 12 && tc->out.d.C == C && tc->out.d.p == o.inA.d.p &&
            tw != pl.wslot.end() && fast_tconv_supported(m, *tc) && !tc->accA && !o.accA && !o.maskA && dense(tc->inA.d) && dense(tc->inA.g) &&
            dense(o.inA.d) && tc->inA.d.H * 2 == a.H && tc->inA.d.W * 2 == a.W && a.W % TWc == 0 && a.H % TH == 0) {
            a.tc_in = tc->inA.d.p; a.tc_din = tc->inA.g.p; a.tc_w = m->p + tc->w_off;
            a.tc_slabs = pl.slabs + pl.folds[tw->second].slab_off;
            a.tc_mask = tc->maskA; a.tc_alpha = tc->mask_alpha;
            const double tb = 4.0 * ((double)B * a.H * a.W * C + (double)B * (a.H / 2) * (a.W / 2) * 12);      // the transposed conv's out + in
            const double tfl = 4.0 * B * a.H * a.W * C * 12;
#define TCMX(c)                                                                                                         \
            if (C == c) {                                                                                               \
                static const int fit = resident_blocks(k_pgbwd<c, 2, c, true, 512, false, false, false, false, true>, 1 << 20); \
                const int g = pl.nblocks_forced ? nb : (ntiles < fit ? ntiles : fit);                                   \
                LAUNCH(m, "pgbwd_tc_" #c "x2_" #c, bytes + 2 * tb, fl + tfl,                                            \
                       hipLaunchKernelGGL((k_pgbwd<c, 2, c, true, 512, false, false, false, false, true>), dim3(g), dim3(512), 0, m->stream, a)); \
                m->tconv_done = tc;                                                                                     \
                return true;                                                                                            \
            }
            TCMX(6) TCMX(12)
#undef TCMX
        }
    }
#define X(c, ns, co)                                                                                            \
    if (C == c && NS == ns && CO == co) {                                                                       \
        (void)db;                                                                                               \
        if (o.need_din) {                                                                                       \
            static const int fit = resident_blocks(k_pgbwd<c, ns, co, true, 512, false>, 1 << 20);              \
            const int g = pl.nblocks_forced ? nb : (ntiles < fit ? ntiles : fit);                               \
            LAUNCH(m, "pgbwd_" #c "x" #ns "_" #co, bytes, fl,                                                   \
                   hipLaunchKernelGGL((k_pgbwd<c, ns, co, true, 512, false>), dim3(g), dim3(512), 0, m->stream, a)); \
        } else {                                                                                                \
            static const int fit = resident_blocks(k_pgbwd<c, ns, co, false, 512, false>, 1 << 20);             \
            const int g = pl.nblocks_forced ? nb : (ntiles < fit ? ntiles : fit);                               \
            LAUNCH(m, "pgbwd_w_" #c "x" #ns "_" #co, bytes, fl,                                                 \
                   hipLaunchKernelGGL((k_pgbwd<c, ns, co, false, 512, false>), dim3(g), dim3(512), 0, m->stream, a)); \
        }                                                                                                       \
        return true;                                                                                            \
    }
    CONV_SHAPES(X)
#undef X
    return false;
}

// The backward of a 2x2 max-pool folded into the backward launch of the conv that produced its input: possible when the forward
// pass recorded the window positions (the block-fused encoder kernel), the pool's input gradient already holds the skip gradient
// (accumulate) and the pool applies act' of that conv (ReLU: ties at zero are killed by act'(0) = 0).  The caller then skips the
// pool's own launch; the very next fast_conv_bwd call must be for `conv`.
bool fast_pool_fold(Model* m, Op& pool, Op& conv) {
    static const bool on = getenv("DNNCA_NO_POOL_FOLD") == nullptr;
    if (!on || pool.type != OP_POOL || pool.k != 2 || !pool.pool_idx_valid || !(pool.pool_idx || m->dry)) return false;
    if (!pool.accA || !pool.maskA || pool.mask_alpha != 0.f) return false;
    if (!conv_supported(m, conv) || conv.inB.d.C || !conv.need_din || conv.out.d.p != pool.inA.d.p || !conv.premasked) return false;
    const int C = conv.inA.d.C, CO = conv.out.d.C;
    if (C != CO || !(C == 3 || C == 6 || C == 12)) return false;
    const int TW = 32 * (12 / CO);
    if (!dense(pool.out.g) || !dense(pool.out.d)) return false;
    if ((conv.out.d.W % TW || conv.out.d.H % TH) && !first3_conv0(m, conv)) return false;      // the tile kernels fold whole tiles only
    pool.pool_idx_valid = false;
    m->pool_fold.conv = &conv;
    m->pool_fold.pool = &pool;
    return true;
}

bool fast_head_in_conv_possible(Model* m) {
    if (m->ops.size() < 2 || !m->head_defer_ok) return false;
    const Op& head = m->ops.back();
    const Op& o = m->ops[m->ops.size() - 2];
    if (head.type != OP_HEAD || !conv_supported(m, o)) return false;
    return o.inA.d.C == 3 && !o.inB.d.C && o.out.d.C == 3 && head.inA.d.p == o.out.d.p && head.inA.d.C == 3 && dense(head.inA.d);
}

// The conv that feeds the annotator head, in a training step: its forward launch also runs the head, the weighted BCE and the
// head's backward (what k_head_train does in a launch of its own, re-reading the feature map).  The caller has made sure that
// the label statistics of this step are already on the stream.  Returns false when the shape has no such kernel.
bool fast_conv_fwd_head(Model* m, int B, Op& o, Op& head, const float* y, const dnnca_loss_cfg& cfg, float gscale, double bytes,
                        double flops) {
    if (!conv_supported(m, o) || !m->head_defer_ok) return false;
    const int C = o.inA.d.C, NS = o.inB.d.C ? 2 : 1, CO = o.out.d.C;
    if (C != 3 || NS != 1 || CO != 3 || head.inA.d.p != o.out.d.p || head.inA.d.C != 3 || !dense(head.inA.d)) return false;
    PgPlan& pl = g_plans[m];
    auto it = pl.slot.find({&o, 0});
    if (it == pl.slot.end()) return false;
    FwdArgs a{};
    a.src[0] = o.inA.d.p; a.src[1] = nullptr;
    a.bmat = pl.bmat + it->second;
    a.bias = m->p + o.b_off;
    a.dst = o.out.d.p;
    a.pool_dst = nullptr;
    a.B = B; a.H = o.out.d.H; a.W = o.out.d.W;
    a.alpha = o.alpha;
    a.hy = y;
    a.hw = m->p + head.w_off; a.hb = m->p + head.b_off;
    a.hdfeat = head.inA.g.p;
    a.hpartials = m->head_partials;
    a.hscalars = m->scalars;
    a.hlabel_part = m->label_part_valid ? m->label_part : nullptr;
    a.hlabel_nblk = m->label_part_nblk;
    a.hcfg = cfg;
    a.hn_label = (double)B * a.H * a.W;
    a.hgscale = gscale;
    a.hmask = head.maskA; a.halpha = head.mask_alpha;
    const int TW = 32 * (12 / CO);
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + TH - 1) / TH;
    const int ntiles = a.tiles_x * a.tiles_y * B;
    static const int fit = resident_blocks(k_pgfwd<3, 1, 3, 512, false, true>, 2048);      // partials table: 2048 rows
    const int g = ntiles < fit ? ntiles : fit;
    LAUNCH(m, "pgfwd_head_3x1_3", bytes, flops,
           hipLaunchKernelGGL((k_pgfwd<3, 1, 3, 512, false, true>), dim3(g), dim3(512), 0, m->stream, a));
    // the partial sums wait for the launch that ends the backward pass (k_pg_fold)
    m->head_pending.partials = m->head_partials; m->head_pending.nblocks = g; m->head_pending.C = 3;
    m->head_pending.dw = m->g + head.w_off; m->head_pending.dbias = m->g + head.b_off;
    return true;
}

// The forward pass of the first encoder block as one column-strip launch (k_first3_fwd, strip_dev.h); called by fused_down_fwd
// (kernels_fused.hip), which owns the block's bookkeeping (pool positions, label partials).  nblocks: rows of the label table.
bool fast_first3_fwd(Model* m, int B, Op& c1, Op& c2, Op& pool, float* y0, unsigned char* pool_idx, const float* labels, float* label_part,
                     double bytes, double flops, int* nblocks) {
    if (getenv("DNNCA_NO_FIRST3F") || c1.inA.d.C != 1 || c1.out.d.C != 3 || c2.out.d.C != 3) return false;
    const int H = c1.out.d.H, W = c1.out.d.W;
    if ((H & 1) || (W & 1) || W < 8 || H < 8 || (double)B * H * W * 12.0 >= 1073741824.0 || c1.alpha > 1.f || c2.alpha > 1.f) return false;
    FirstFwdArgs a{};
    a.xin = c1.inA.d.p;
    a.w0 = m->p + c1.w_off; a.b0 = m->p + c1.b_off;
    a.w1 = m->p + c2.w_off; a.b1 = m->p + c2.b_off;
    a.alpha0 = c1.alpha; a.alpha1 = c2.alpha;
    a.y0 = y0; a.y1 = c2.out.d.p;
    a.pool = pool.out.d.p; a.pool_idx = pool_idx;
    if (const char* e = getenv("DNNCA_F3F_ABL")) {      // tuning aid (wrong results): drop output tensors
        const int abl = atoi(e);
        if (abl & 1) a.y0 = nullptr;
        if (abl & 2) a.pool_idx = nullptr;
    }
    a.labels = labels; a.label_part = label_part;
    a.B = B; a.H = H; a.W = W;
    a.nstrips = (W + STRIP - 1) / STRIP;
    static const int wps = getenv("DNNCA_F3F_WPS") ? atoi(getenv("DNNCA_F3F_WPS")) : 2;      // tuning aid: waves per SIMD
    int nchunks = 1024 * wps / (B * a.nstrips);
    if (nchunks > H / 8) nchunks = H / 8;
    if (nchunks < 1) nchunks = 1;
    a.nchunks = nchunks;
    const int nblk = (B * nchunks * a.nstrips + 3) / 4;
    if (nblk > 2048) return false;
    a.nstrip_blocks = nblk;
    int nride = 0;
    if (m->prep_flush) {            // the step's operand preparation rides in this launch (blocks nblk ..)
        m->prep_flush = nullptr;
        a.prep = g_plans[m].prep;
        nride = a.prep.nblocks;
    }
    if (nride) {
        LAUNCH(m, "first3_fwd", bytes, flops, hipLaunchKernelGGL((k_first3_fwd<0, 27>), dim3(nblk + nride), dim3(256), 0, m->stream, a));
        *nblocks = nblk;
        return true;
    }
    if (wps == 3)
        LAUNCH(m, "first3_fwd", bytes, flops, hipLaunchKernelGGL((k_first3_fwd<0, 27, 3>), dim3(nblk), dim3(256), 0, m->stream, a));
    else
        LAUNCH(m, "first3_fwd", bytes, flops, hipLaunchKernelGGL((k_first3_fwd<0, 27>), dim3(nblk), dim3(256), 0, m->stream, a));
    *nblocks = nblk;
    return true;
}

// Conv2DTranspose 6 -> 3 followed by the two-source conv 6 -> 3 of the last decoder block (forward) as one column-strip launch
// (k_up3_fwd, strip_dev.h); ops[oi], ops[oi + 1] are consumed when it returns true.
bool fast_up3_fwd(Model* m, int B, size_t oi) {
    if (getenv("DNNCA_NO_UP3F") || (m->desc.flags & 1) || oi + 1 >= m->ops.size()) return false;
    Op &tc = m->ops[oi], &c0 = m->ops[oi + 1];
    if (tc.type != OP_TCONV || c0.type != OP_CONV || tc.k != 2 || c0.k != 3) return false;
    if (tc.inA.d.C != 6 || tc.out.d.C != 3 || c0.inA.d.C != 3 || c0.inB.d.C != 3 || c0.out.d.C != 3) return false;
    if (c0.inA.d.p != tc.out.d.p || !dense(tc.inA.d) || !dense(tc.out.d) || !dense(c0.inB.d) || !dense(c0.out.d)) return false;
    if (!conv_supported(m, c0) || !fast_tconv_supported(m, tc)) return false;
    const int H = c0.out.d.H, W = c0.out.d.W;
    if (tc.out.d.H != H || tc.out.d.W != W || c0.inB.d.H != H || c0.inB.d.W != W || tc.inA.d.H * 2 != H || tc.inA.d.W * 2 != W) return false;
    if ((H & 1) || (W & 1) || W < 8 || H < 8 || (double)B * H * W * 12.0 >= 1073741824.0 || c0.alpha > 1.f) return false;
    UpFwdArgs a{};
    a.in = tc.inA.d.p; a.skip = c0.inB.d.p;
    a.wt = m->p + tc.w_off; a.bt = m->p + tc.b_off;
    a.w = m->p + c0.w_off; a.b = m->p + c0.b_off;
    a.alpha = c0.alpha;
    a.tout = tc.out.d.p; a.out = c0.out.d.p;
    a.B = B; a.H = H; a.W = W;
    a.nstrips = (W + STRIP - 1) / STRIP;
    int nchunks = 2048 / (B * a.nstrips);
    if (nchunks > H / 8) nchunks = H / 8;
    if (nchunks < 1) nchunks = 1;
    a.nchunks = nchunks;
    const int nblk = (B * nchunks * a.nstrips + 3) / 4;
    const double npx = (double)B * H * W;
    // algorithmic bytes of the two layers: transposed conv (1.5 in, 3 out), conv (3 + 3 in, 3 out) floats per pixel
    LAUNCH(m, "up3_fwd", 4.0 * npx * 13.5, 2.0 * npx * (18 + 162),
           hipLaunchKernelGGL((k_up3_fwd<2, 48>), dim3(nblk), dim3(256), 0, m->stream, a));
    return true;
}

// The conv that feeds the head in a training step, whole: forward + head + weighted BCE + head backward + the conv's own backward
// in one column-strip launch (k_tail3, strip_dev.h).  The conv's output and its gradient are never written; the caller skips the
// conv's backward launch (Model::tail_done).  Returns false when the shape has no such kernel.
bool fast_tail3(Model* m, int B, Op& o, Op& head, const float* y, const dnnca_loss_cfg& cfg, float gscale) {
    if (getenv("DNNCA_NO_TAIL3") || !conv_supported(m, o) || !m->head_defer_ok) return false;      // (read per call: the tests flip it)
    const int C = o.inA.d.C, NS = o.inB.d.C ? 2 : 1, CO = o.out.d.C;
    if (C != 3 || NS != 1 || CO != 3 || head.inA.d.p != o.out.d.p || head.inA.d.C != 3 || !dense(head.inA.d)) return false;
    if (!o.need_din || o.accA || !dense(o.inA.d) || !dense(o.inA.g) || (o.alpha >= 0.f && !o.premasked)) return false;
    const int H = o.out.d.H, W = o.out.d.W;
    if ((double)B * H * W * 12.0 >= 1073741824.0 || W < 8 || H < 8 || o.alpha > 1.f) return false;      // byte offsets below STRIP_HALF
    PgPlan& pl = g_plans[m];
    auto it = pl.wslot.find({&o, 0});
    if (it == pl.wslot.end()) return false;
    TailArgs a{};
    a.x = o.inA.d.p;
    a.w = m->p + o.w_off; a.bias = m->p + o.b_off;
    a.alpha = o.alpha;
    a.hy = y;
    a.hw = m->p + head.w_off; a.hb = m->p + head.b_off;
    a.hpartials = m->head_partials;
    a.hscalars = m->scalars;
    a.hlabel_part = m->label_part_valid ? m->label_part : nullptr;
    a.hlabel_nblk = m->label_part_nblk;
    a.hcfg = cfg;
    a.hn_label = (double)B * H * W;
    a.hgscale = gscale;
    a.hmask = head.maskA; a.halpha = head.mask_alpha;
    a.dx = o.inA.g.p;
    a.mask = o.maskA; a.mask_alpha = o.mask_alpha;
    a.slabs = pl.slabs + pl.folds[it->second].slab_off;
    a.B = B; a.H = H; a.W = W;
    a.nstrips = (W + STRIP - 1) / STRIP;
    // one round of waves: 2 per SIMD on 256 CUs = 2048 tasks at most, in chunks of at least 8 rows
    static const int slots = getenv("DNNCA_TAIL3_SLOTS") ? atoi(getenv("DNNCA_TAIL3_SLOTS")) : 2048;      // tuning aid
    int nchunks = slots / (B * a.nstrips);
    if (nchunks > H / 8) nchunks = H / 8;
    if (nchunks < 1) nchunks = 1;
    a.nchunks = nchunks;
    const int ntasks = B * nchunks * a.nstrips, nblk = (ntasks + 3) / 4;
    if (nblk > 2048) return false;                      // rows of the head's partials table
    const double npx = (double)B * H * W;
    static const int tail3_lds = getenv("DNNCA_TAIL3_LDS") ? atoi(getenv("DNNCA_TAIL3_LDS")) : 0;      // tuning aid: dynamic LDS bytes (limits blocks per CU)
    auto kern = k_tail3<3, 27, true, 0, 1>;
#ifdef DNNCA_TUNING
    static const int variant = getenv("DNNCA_TAIL3_VARIANT") ? atoi(getenv("DNNCA_TAIL3_VARIANT")) : 0;      // tuning aid
    if (variant == 1) kern = k_tail3<3, 27, true>;
    if (variant == 2) kern = k_tail3<6, 27, true>;
    if (variant == 3) kern = k_tail3<3, 0, true>;
    if (variant == 4) kern = k_tail3<3, 54, true>;
    if (variant == 5) kern = k_tail3<3, 45, true>;
    if (variant == 6) kern = k_tail3<3, 27, true, 0, 1>;
    if (variant == 7) kern = k_tail3<3, 27, false, 0, 1>;
    if (variant == 8) kern = k_tail3<6, 27, false, 0, 1>;
    if (variant == 9) kern = k_tail3<6, 27, true, 0, 1>;
    if (variant >= 100) {
        const int abl = variant - 100;
        kern = abl == 1 ? k_tail3<3, 27, true, 1> : abl == 2 ? k_tail3<3, 27, true, 2> : abl == 4 ? k_tail3<3, 27, true, 4> : abl == 8 ? k_tail3<3, 27, true, 8>
             : abl == 16 ? k_tail3<3, 27, true, 16> : abl == 7 ? k_tail3<3, 27, true, 7> : k_tail3<3, 27, true, 31>;
    }
#endif
    // algorithmic bytes / FLOPs of the layers this launch stands for (every layer reads its inputs and writes its output):
    // conv forward 3 + 3, head + loss + head backward 3 + 1 + 3, conv backward (dz, x in; dx out) 3 + 3 + 3 floats per pixel
    LAUNCH(m, "tail3_3x1_3", 4.0 * npx * 22, 2.0 * npx * (81 * 3 + 15),
           hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), tail3_lds, m->stream, a));
    m->head_pending.partials = m->head_partials; m->head_pending.nblocks = nblk; m->head_pending.C = 3;
    m->head_pending.dw = m->g + head.w_off; m->head_pending.dbias = m->g + head.b_off;
    m->tail_done = &o;
    return true;
}

bool fast_tconv_bwd(Model* m, int B, Op& o, double out_bytes, double in_bytes, double flops) {
    if (!fast_tconv_supported(m, o)) return false;
    PgPlan& pl = g_plans[m];
    auto it = pl.wslot.find({&o, 0});
    if (it == pl.wslot.end()) return false;
    const FoldDesc& f = pl.folds[it->second];
    TwArgs a;
    a.in = o.inA.d.p;
    a.dout = o.out.g.p;
    a.slabs = pl.slabs + f.slab_off;
    a.W = o.inA.d.W;
    a.nquads = B * o.inA.d.H * o.inA.d.W / 4;
    int nb = a.nquads / 64;
    nb = nb < 64 ? 64 : (nb > 1024 ? 1024 : nb);
    const int CI = o.inA.d.C, CO = o.out.d.C;
    const int H = o.inA.d.H, W = o.inA.d.W;
    const TdArgs d{o.out.g.p, m->p + o.w_off, o.inA.d.p, o.inA.g.p, B, H, W, (int)o.accA, (int)o.maskA, o.mask_alpha};
    const int nbd = (B * H * W + 255) / 256;
#define TW_CASE(ci, co)                                                                                       \
    if (CI == ci && CO == co) {                                                                               \
        LAUNCH(m, "tconv_bwd_" #ci "_" #co, 2 * (out_bytes + in_bytes), 2 * flops,                            \
               hipLaunchKernelGGL((k_tconv_bwd<ci, co>), dim3(nb + nbd), dim3(256), 0, m->stream, a, d, nb)); \
        return true;                                                                                          \
    }
    TW_CASE(12, 12) TW_CASE(12, 6) TW_CASE(6, 3)
#undef TW_CASE
    return false;
}

// after the last backward op: fold every slab set into the flat gradient vector (one launch)
int fast_finish_backward(Model* m) {
    if (m->desc.flags & 1) return DNNCA_OK;
    PgPlan& pl = g_plans[m];
    if (!pl.built || pl.folds.empty()) return DNNCA_OK;
    HeadTail h{};
    int extra = 0;
    if (m->head_pending.partials) {
        h.partials = m->head_pending.partials; h.nblocks = m->head_pending.nblocks; h.C = m->head_pending.C;
        h.dw = m->head_pending.dw; h.dbias = m->head_pending.dbias; h.scalars = m->scalars;
        extra = h.C + 2;
        m->head_pending.partials = nullptr;
    }
    // single-replica train step whose every gradient (and the loss) comes out of this launch: wait for optimizer_step and let the
    // fold apply Adam as well (fast_fold_adam)
    if (extra && !m->comm && !m->dry && m->merged_launches() && m->desc.l2 == 0.f && pl.fold_outputs + h.C + 1 == m->nT &&
        !getenv("DNNCA_NO_FOLD_ADAM")) {
        m->fold_deferred = true;
        pl.pending_head = h;
        return DNNCA_OK;
    }
    LAUNCH(m, "pg_fold", 0, 0,
           hipLaunchKernelGGL(k_pg_fold, dim3((unsigned)pl.folds.size() + extra, pl.fold_chunks), dim3(256), 0, m->stream,
                              pl.folds_dev, pl.slabs, m->g, (int)pl.folds.size(), h, AdamTail{}));
    return DNNCA_OK;
}

// the deferred fold of fast_finish_backward with the Adam update (and the step outputs) in the same launch
int fast_fold_adam(Model* m, float lr_t, const dnnca_loss_cfg& cfg, double n_label, double inv_batch_hw) {
    PgPlan& pl = g_plans[m];
    const HeadTail h = pl.pending_head;
    m->fold_deferred = false;
    AdamTail ad{m->p, m->m, m->v, lr_t, m->beta1, m->beta2, m->eps, cfg, n_label, inv_batch_hw, m->out5};
    LAUNCH(m, "pg_fold_adam", 28.0 * m->nT, 10.0 * m->nT,
           hipLaunchKernelGGL(k_pg_fold, dim3((unsigned)pl.folds.size() + h.C + 2, pl.fold_chunks), dim3(256), 0, m->stream,
                              pl.folds_dev, pl.slabs, m->g, (int)pl.folds.size(), h, ad));
    return DNNCA_OK;
}

// Activation-derivative fusion plan (see model.h Op::maskA).  Static: depends only on shapes and the generic flag.
void fast_plan_masks(Model* m) {
    if (m->desc.flags & 1) return;
    for (size_t ip = 0; ip < m->ops.size(); ++ip) {
        Op& P = m->ops[ip];
        if (P.type != OP_CONV || P.alpha < 0.f) continue;
        int jc = -1, which = 0;
        for (size_t j = ip + 1; j < m->ops.size() && jc < 0; ++j) {
            const Op& c = m->ops[j];
            if (c.inA.d.p == P.out.d.p && c.inA.d.C) { jc = (int)j; which = 0; }
            else if (c.type == OP_CONV && c.inB.d.C && c.inB.d.p == P.out.d.p) { jc = (int)j; which = 1; }
        }
        if (jc < 0) continue;
        Op& c = m->ops[jc];
        bool can = false;
        if (c.type == OP_CONV) can = (conv_supported(m, c) || ig_conv_supported(m, c)) && c.need_din;
        else if (c.type == OP_POOL) can = fast_pool_supported(m, c);
        else if (c.type == OP_TCONV) can = fast_tconv_supported(m, c) || ig_tconv_supported(m, c);
        else if (c.type == OP_HEAD) can = fast_head_supported(m, c);
        else if (c.type == OP_BN) can = fast_bn_supported(m, c);
        if (!can) continue;
        (which ? c.maskB : c.maskA) = true;
        c.mask_alpha = P.alpha;
        P.premasked = true;
    }
}

}  // namespace dnnca
