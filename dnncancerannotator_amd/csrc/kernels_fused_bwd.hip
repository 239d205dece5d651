// kernels_fused_bwd.hip -- block-fused BACKWARD kernels for the 6- and 12-channel levels of configs/unet.yaml.
//
// One launch per reference block instead of one per Keras layer (the forward counterparts are in kernels_fused.hip):
//   k_fzb<UP>     backward of Upsample.call   (components.py:158-166): second conv -> two-source first conv -> Conv2DTranspose
//   k_fzb<DOWN>   backward of Downsample.call (components.py:77-81):   MaxPool2D (+ skip gradient) -> second conv -> first conv
// At 128^2 / 256^2 a layer's backward is a latency chain (launch -> operand loads -> tile loads -> MFMAs -> reduction -> atomics)
// on a few tiles per CU; fusing a block runs two or three of those chains as one, and the gradients of the block's
// intermediates (the first conv's output, the transposed conv's output) live in LDS only -- they are never written.
//
// Per tile of TH x TW output pixels (halo recompute, as in the forward kernels):
//   stage dz1 (gradient of the second conv's pre-activation output) with a halo of 2, y0 (first conv's output) and the first
//   conv's sources with a halo of 1; DOWN: dz1 = (skip gradient + pooled gradient at the recorded window position) * act'(y1)
//   P1   dz0 = dgrad_conv1(dz1) * act'(y0) on the tile enlarged by one pixel -> LDS     |  dW1 += y0 (x) dz1
//   P2   dgrad_conv0(dz0) on the tile: UP [d_up -> LDS | d_skip -> HBM], DOWN dx -> HBM |  dW0 += sources (x) dz0
//        then, wave-locally (every data-gradient wave owns a rectangle of the tile):
//        UP: the transposed conv's data gradient, weight + bias gradient from d_up
// The block's eight waves SPLIT: waves [0, 4) run the data-gradient convolutions (left column: B operands in registers, no
// accumulator outlives a phase), waves [4, 8) the weight gradients (right column: every accumulator lives in registers across the
// whole persistent tile loop).  Every SIMD hosts one wave of each kind, so the matrix pipe always has two independent MFMA
// streams to interleave, and neither role pays for the other's registers: the two roles are two separate loops with the
// same sequence of workgroup barriers.
//
// The arithmetic is the pixel-group GEMM of kernels_mfma.hip: same prepared data-gradient B operands (k_pg_prep, mode 1), same
// weight-gradient D layout and slabs (k_pg_fold folds them unchanged).  fp32 MFMA is an fmaf chain: fp32 parity holds.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "fast.h"
#include "fz_dev.h"
#include "kernels.h"

namespace dnnca {
namespace fzb {

using namespace fz;

// 3x3 'same' convolution between LDS tiles on the fp32 matrix cores -- fz::Conv3 without bias / activation (a data gradient has
// neither) and with an optional mask tile of the output's geometry: out *= msk > 0 ? 1 : malpha (the act' of the conv whose
// output gradient this is).  C input channels, CO output channels, G = 12 / CO pixels per GEMM row group, RG input groups per row.
template <int C, int CO, int RG, int OROWS, int ILEAD, int OLEAD, int NW, int MPR>
struct DConv3 {
    static constexpr int G = 12 / CO, GC = G * C, WR = (G + 2) * C, SR = (WR + 3) / 4, KS = 3 * SR, ILS = RG * GC;
    static constexpr int NMT = MPR ? OROWS * MPR : cdiv(OROWS * RG, 16), CH = cdiv(NMT, NW);
    static_assert(MPR == 0 || (NW % MPR == 0 && NMT % NW == 0), "row-aligned M-tiles must divide evenly over the waves");
    // MPR > 0: wave w owns column block w % MPR of the CH consecutive rows (w / MPR) * CH ..: a rectangle of 16 groups x CH rows of
    // the output, which the caller may go on working with right away (same wave: DS operations execute in order)
    static constexpr int CSTEP = MPR ? RG * GC : NW * 16 * GC;

    static __device__ __forceinline__ void run(const float* in, float* out, const float* breg, const float* msk, float malpha, int wave, int lane) {
        const int m = lane & 15, q = lane >> 4, n = m;
        constexpr bool RAGGED = CH * NW > NMT;
        const bool last_ok = !RAGGED || wave + (CH - 1) * NW < NMT;           // wave-uniform
        const int g0 = MPR ? (wave / MPR) * CH * RG + (wave % MPR) * 16 : wave * 16;
        const float* a0 = in + ILEAD + q + (g0 + m) * GC;
        const float* al = RAGGED ? in + ILEAD + q + ((last_ok ? wave + (CH - 1) * NW : NMT - 1) * 16 + m) * GC : a0 + (CH - 1) * CSTEP;
        f32x4 acc[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the A operands of K-step s + 1 are read while the MFMAs of step s run (two register sets; left to itself hipcc reads each
        // operand right in front of its MFMA and waits for it: one exposed LDS round trip per chain and step)
        float av[2][CH];
        auto fetch = [&](int s, float (&a)[CH]) {
            const int dy = s / SR, k = s - dy * SR;
#pragma unroll
            for (int c = 0; c < CH; ++c) a[c] = c < CH - 1 ? a0[c * CSTEP + dy * ILS + 4 * k] : al[dy * ILS + 4 * k];
        };
        fetch(0, av[0]);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s + 1 < KS) fetch(s + 1, av[(s + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s & 1][c], breg[s], acc[c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (n < 12) {
            // M-tile index of chain c: linear (wave + c NW), or row-aligned ((w / MPR) CH + c) MPR + w % MPR
            const int o0 = OLEAD + ((MPR ? (wave / MPR) * CH * MPR + wave % MPR : wave) * 16 + 4 * q) * 12 + n;
            constexpr int OSTEP = (MPR ? MPR : NW) * 192;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (c < CH - 1 || last_ok) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[c][r];
                        if (msk) v *= msk[o0 + c * OSTEP + r * 12] > 0.f ? 1.0f : malpha;
                        out[o0 + c * OSTEP + r * 12] = v;
                    }
                }
            }
        }
    }
};

// Weight gradient of a 3x3 conv as D[(dy, j), (dx, co)] += window(x) * dz with K = pixel groups (k_pgbwd's formulation, so that
// k_pg_fold reads the slabs unchanged): C input channels of this source, CO output channels, Gw = 12 / CO pixels per group.
// The x tile has a halo of 1 (window of output pixel (ty, px) = tile rows ty .. ty+2, pixels px .. px+2); the last valid
// row of D is the all-ones row (bias gradient).
template <int C, int CO, int TW>
struct WG {
    static constexpr int Gw = 12 / CO, WRw = (Gw + 2) * C, MROWS = 3 * WRw + 1, MT = cdiv(MROWS, 16), KSTEPS = TW / (4 * Gw), ASTEP = 4 * Gw * C;
    int offA[MT];        // window rows: float offset from the x tile's row ty; constant rows: absolute LDS float index
    // xlead / xls: lead and row stride (floats) of the x tile; cst1 / cst0: LDS float indices of the constants 1.0 and 0.0
    __device__ __forceinline__ void init(int lane, int xlead, int xls, int cst1, int cst0) {
        const int m16 = lane & 15, q = lane >> 4;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int mrow = 16 * t + m16;
            if (mrow < 3 * WRw) {
                const int dy = mrow / WRw, j = mrow - dy * WRw;
                offA[t] = dy * xls + xlead + j + q * (Gw * C);
            } else {
                offA[t] = mrow == 3 * WRw ? cst1 : cst0;
            }
        }
    }
    // one tile row of NS sources that share dz: xrow[s] = LDS float index of source s' x tile row ty; grow = LDS float index of dz at
    // interior pixel (ty, 0).  The operands of K-step st + 1 are read while the MFMAs of step st run (two register sets).
    template <int NS>
    __device__ __forceinline__ void row(f32x4 (&acc)[NS][MT], const float* ldsf, const int (&xrow)[NS], int grow, int cst0, int lane) const {
        const int m16 = lane & 15, q = lane >> 4, n = m16;
        const int goff = n < 12 ? grow + q * 12 + n : cst0;
        const int gstep = n < 12 ? 48 : 0;
        int ao[NS][MT], as[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const bool win = 16 * t + m16 < 3 * WRw;
#pragma unroll
            for (int s = 0; s < NS; ++s) ao[s][t] = win ? xrow[s] + offA[t] : offA[t];
            as[t] = win ? ASTEP : 0;
        }
        float bv[2], av[2][NS][MT];
        auto fetch = [&](int st, float& b, float (&a)[NS][MT]) {
            b = ldsf[goff + st * gstep];
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int t = 0; t < MT; ++t)      // M-tiles whose 16 rows are all window rows step by a compile-time constant (immediate LDS offsets)
                    a[s][t] = ldsf[16 * (t + 1) <= 3 * WRw ? ao[s][t] + st * ASTEP : ao[s][t] + st * as[t]];
        };
        fetch(0, bv[0], av[0]);
#pragma unroll
        for (int st = 0; st < KSTEPS; ++st) {
            if (st + 1 < KSTEPS) fetch(st + 1, bv[(st + 1) & 1], av[(st + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st & 1][s][t], bv[st & 1], acc[s][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
};

// tuning builds (DNNCA_TUNING=1 python -m dnncancerannotator_amd.build): s_memtime stamps of the first wave of either role
#ifdef DNNCA_TUNING
__device__ unsigned long long g_fzb_stamps[4 * 1024 * 2 * 32];      // [kernel: up6, up12, down6, down12][block][role][stamp]
__device__ __forceinline__ unsigned long long fzb_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define FZB_STAMP(role, i)                                                                                            \
    do {                                                                                                              \
        if (lane == 0 && (wave == 0 || wave == NWD) && blockIdx.x < 1024 && (i) < 32) g_fzb_stamps[((KID * 1024 + blockIdx.x) * 2 + (role)) * 32 + (i)] = fzb_now(); \
    } while (0)
#else
#define FZB_STAMP(role, i) do { } while (0)
#endif

struct BArgs {
    const float* dz1;        // UP: gradient of the second conv's pre-activation output [B, H, W, F]
                             // DOWN: the SKIP gradient of that output; the staging adds the pooled part and applies act'(y1)
    const float* y1;         // DOWN: the second conv's output
    const float* dpool;      // DOWN: gradient of the pooled tensor [B, H/2, W/2, F]
    const unsigned char* pidx;   // DOWN: window position of every pooled maximum (k_fz_down)
    float pf_alpha;          // DOWN: slope of act'(y1)
    const float* y0;         // the first conv's output [B, H, W, F]: act' of dz0 and the second conv's weight-gradient operand
    const float* xa;         // the first conv's sources: UP [transposed conv output | skip], DOWN [block input]
    const float* xb;
    const float* bm1;        // prepared data-gradient B operands of the second conv [KS1][64] ...
    const float* bm0;        // ... and of the first conv, NPASS x [KS0][64]
    float* dxb;              // UP: gradient of the skip source [B, H, W, F]; DOWN: gradient of the block input [B, H, W, CA]
    int mask0;               // dz0 = (data gradient of the second conv) * act'(y0)
    float alpha0;
    float* slabs1;           // weight-gradient slabs [kPgBuckets][MT*256]: second conv, first conv per source
    float* slabs0[2];
    // UP: the transposed conv (12 -> F) that produced the first source
    const float* tc_in;      // its input [B, H/2, W/2, 12]
    float* tc_din;           // gradient of that input
    const float* tc_w;       // kernel [2][2][F][12]
    float* tc_slabs;         // [kPgBuckets][MBt*256]
    int tc_mask;             // multiply tc_din by act'(tc_in)
    float tc_alpha;
    int B, H, W, tiles_x, tiles_y;
    int dbg;                 // tuning builds (DNNCA_FZB_DBG): bit 0 skip the final slab atomics (wrong results; timing only)
};

// UP:   CA == F; NSRC = 2; the first conv's data gradient has 2F channels = NPASS passes of 12 ([up | skip])
// DOWN: CA = the block's input channels; NSRC = 1
template <bool UP, int CA, int F, int TW, int NT, int ISSUE_DG = 2, int ISSUE_WG = 2>
__global__ __launch_bounds__(NT, 1) void k_fzb(BArgs p) {
    constexpr int TH = 8, NW = NT / 64, NWD = NW / 2, NWW = NW - NWD, NSRC = UP ? 2 : 1, CT = 12;
    constexpr int KID = (UP ? 0 : 2) + (F == 12 ? 1 : 0);      // tuning builds: which stamp table
    (void)KID;
    static_assert(!UP || CA == F, "decoder block: both sources of the first conv have F channels");
    constexpr int G = 12 / F, RG = even_up(cdiv(TW + 5, G)), PR = RG * G;
    using TDZ1 = Tile<F, G, RG, TH + 4, 2>;
    using TY0 = Tile<F, G, RG, TH + 2, 1>;              // also the geometry of dz0
    using TXA = Tile<CA, G, RG, TH + 2, 1>;
    // second-stage data gradient: F -> CO2 channels per pass, G2 pixels per group, output tile of exact rows
    constexpr int CO2 = UP ? 12 : CA, G2 = 12 / CO2, NPASS = UP ? (2 * F) / 12 : 1, RG2I = PR / G2, MPR2 = (TW / G2) / 16, OUT2 = TH * TW * CO2;
    static_assert(PR % G2 == 0 && (TW / G2) % 16 == 0, "second-stage groups must tile the rows");
    using DC1 = DConv3<F, F, RG, TH + 2, TDZ1::LEAD, TY0::LEAD, NWD, 0>;
    using DC0 = DConv3<F, CO2, RG2I, TH, TY0::LEAD, 0, NWD, MPR2>;
    using W1 = WG<F, F, TW>;
    using W0 = WG<CA, F, TW>;
    using SDZ = Stager<F, G, RG, TH + 4, 2, TW + 4, TDZ1::LS, NT>;
    using SY0 = Stager<F, G, RG, TH + 2, 1, TW + 2, TY0::LS, NT>;
    using SXA = Stager<CA, G, RG, TH + 2, 1, TW + 2, TXA::LS, NT>;
    // UP: the transposed conv's input tile (TH/2 x TW/2 pixels of 12 channels, dense) and kernel
    constexpr int LWt = TW / 2, NLP = (TH / 2) * LWt, KA = 2 * F, KS1t = KA / 4, KTt = 4 * F, MBt = cdiv(KTt, 16), ROWF = TW * 12;
    using SLO = Stager<CT, 1, 1, TH / 2, 0, TW / 2, LWt * CT, NT>;
    static_assert(!UP || (LWt % 16 == 0 && KA % 4 == 0), "UP: whole M-tiles per low-resolution row");
    // DOWN: pooled-gradient tile (floats) and window-position tile (bytes): TH/2 + 2 rows of TW/2 + 2 pixels, halo 1
    constexpr int PFR = TH / 2 + 2, PFW = TW / 2 + 2, PFLEAD = (4 - F % 4) % 4, PFLS = up4(PFLEAD + PFW * F), PFLI = PFLS / 4, PFN4 = PFR * PFLI;
    constexpr int NPS = UP ? 1 : cdiv(PFN4, NT);
    // LDS map (float indices)
    constexpr int O_DZ1 = 0, O_Y0 = O_DZ1 + TDZ1::N, O_DZ0 = O_Y0 + TY0::N, O_XA = O_DZ0 + TY0::N, O_EXT = O_XA + NSRC * TXA::N;
    constexpr int O_LOW = O_EXT, O_TCW = O_LOW + NLP * CT;                       // UP
    constexpr int O_PDP = O_EXT, O_PIX = O_PDP + PFN4 * 4;                       // DOWN (O_PIX: PFN4 words of 4 bytes)
    constexpr int O_CST = UP ? O_TCW + KTt * CT : O_PIX + PFN4, O_ROW = O_CST + 4;
    // the data-gradient B operands wait in LDS and are fetched into registers per phase (81 registers across the tile loop otherwise)
    constexpr int O_BM1 = O_ROW + NWD * 192, O_BM0 = O_BM1 + DC1::KS * 64, LDSN = O_BM0 + NPASS * DC0::KS * 64;
    constexpr int O_OUT = O_DZ1;                                                  // stage-2 output tiles alias dz1 (dead after P1)
    static_assert(NPASS * OUT2 <= TDZ1::N, "stage-2 output tiles must fit the dz1 tile");
    constexpr int MT1 = W1::MT, MT0 = W0::MT;
    constexpr int ACCF = (MT1 + NSRC * MT0) * 256;                                // floats of one weight-gradient wave's sums
    constexpr int TACCF = UP ? MBt * 256 : 0;                                     // ... of one data-gradient wave's (transposed conv)
    static_assert(NWW * ACCF + NWD * TACCF <= LDSN, "the final reduction must fit the LDS");
    // a data-gradient wave's rectangle of the stage-2 output: RPW rows x 16 groups, i.e. (UP) 8 x RPW/2 transposed-conv input pixels
    constexpr int RPW = TH / (NWD / MPR2), LPW = 8 * (RPW / 2);
    static_assert(NWD % MPR2 == 0 && TH % (NWD / MPR2) == 0 && (!UP || (RPW % 2 == 0 && LPW % 16 == 0)), "wave rectangles");
    static_assert(LDSN * 4 <= 160 * 1024, "LDS budget");
    __shared__ float4 lds4[(LDSN + 3) / 4];
    float* lds = reinterpret_cast<float*>(lds4);
    const float* ldsf = lds;
    constexpr int CST1 = O_CST, CST0 = O_CST + 1;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = p.tiles_x * p.tiles_y * p.B;
#ifdef DNNCA_TUNING
    const bool xcd_map = (ntiles & 7) == 0 && (gridDim.x & 7) == 0 && !(p.dbg & 4);
#else
    const bool xcd_map = (ntiles & 7) == 0 && (gridDim.x & 7) == 0;
#endif

    SDZ sdz;
    SY0 sy0;
    SXA sxa[NSRC];
    SDZ sy1;                                   // DOWN: the second conv's output, same geometry as the skip gradient
    SLO slo;
    float4 predp[NPS];
    unsigned preix[NPS];
    auto issue = [&](int b, int x0, int y0) {
        sdz.issue(p.dz1, b, y0, x0, p.H, p.W, tid);
        if constexpr (!UP) {
            sy1.issue(p.y1, b, y0, x0, p.H, p.W, tid);
            const int Hp = p.H >> 1, Wp = p.W >> 1, rowf = Wp * F;
#pragma unroll
            for (int k = 0; k < NPS; ++k) {
                const int id = tid + k * NT;
                const int r = id / PFLI, c4 = id - r * PFLI;
                const int gy = (y0 >> 1) - 1 + r, gf = ((x0 >> 1) - 1) * F - PFLEAD + 4 * c4;
                const bool ok = id < PFN4 && (unsigned)gy < (unsigned)Hp && gf >= 0 && gf < rowf;
                const size_t off = ok ? ((size_t)b * Hp + gy) * rowf + gf : 0;
                predp[k] = *reinterpret_cast<const float4*>(p.dpool + off);
                if (!ok) predp[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                preix[k] = *reinterpret_cast<const unsigned*>(p.pidx + off);
                if (!ok) preix[k] = 0xffffffffu;
            }
        }
        sy0.issue(p.y0, b, y0, x0, p.H, p.W, tid);
        sxa[0].issue(p.xa, b, y0, x0, p.H, p.W, tid);
        if constexpr (UP) {
            sxa[1].issue(p.xb, b, y0, x0, p.H, p.W, tid);
            slo.issue(p.tc_in, b, y0 >> 1, x0 >> 1, p.H >> 1, p.W >> 1, tid);
        }
    };
    // writes the registers' tile into LDS (DOWN: one workgroup barrier inside -- every thread must call it)
    auto commit = [&]() {
        if constexpr (!UP) {
#pragma unroll
            for (int k = 0; k < NPS; ++k) {
                const int id = tid + k * NT;
                if (id < PFN4) {
                    reinterpret_cast<float4*>(lds + O_PDP)[id] = predp[k];
                    reinterpret_cast<unsigned*>(lds + O_PIX)[id] = preix[k];
                }
            }
            lds_barrier();
            // dz1 = (skip gradient + (window position == recorded position ? pooled gradient : 0)) * act'(y1)
            const unsigned char* ixb = reinterpret_cast<const unsigned char*>(lds + O_PIX);
            const float* pdp = lds + O_PDP;
            static_assert(SDZ::LEAD == 0 && F % 2 == 0 && PFLS % 2 == 0 && (PFLEAD % 2) == 0, "PF transform works on aligned channel pairs");
#pragma unroll
            for (int k = 0; k < SDZ::NPF; ++k) {
                const int idx = tid + k * NT;
                const int r = idx / SDZ::W4, c4 = idx - r * SDZ::W4;
                float g[4] = {sdz.pre[k].x, sdz.pre[k].y, sdz.pre[k].z, sdz.pre[k].w};
                const float yv[4] = {sy1.pre[k].x, sy1.pre[k].y, sy1.pre[k].z, sy1.pre[k].w};
                const int f0 = 4 * c4;                                        // float index from the first halo pixel (LEAD = 0)
                if (((sdz.ok >> k) & 1u) && f0 < (TW + 4) * F) {
                    const int px0 = f0 / F, ch0 = f0 - px0 * F;
                    const int pr = ((r - 2) >> 1) + 1;                         // row in the pooled tiles
                    const unsigned rowbit = ((unsigned)r & 1u) << 1;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {                             // two channels of one pixel per step (F is even)
                        int ch = ch0 + 2 * h, px = px0;
                        if (ch >= F) { ch -= F; ++px; }
                        const int pc = ((px - 2) >> 1) + 1;
                        const unsigned pos = rowbit | ((unsigned)px & 1u);
                        const int o = pr * PFLS + PFLEAD + pc * F + ch;       // even: 8-byte aligned pair
                        const float2 dp = *reinterpret_cast<const float2*>(pdp + o);
                        const unsigned ix = *reinterpret_cast<const unsigned short*>(ixb + o);
                        g[2 * h] = (g[2 * h] + ((ix & 0xffu) == pos ? dp.x : 0.f)) * (yv[2 * h] > 0.f ? 1.0f : p.pf_alpha);
                        g[2 * h + 1] = (g[2 * h + 1] + ((ix >> 8) == pos ? dp.y : 0.f)) * (yv[2 * h + 1] > 0.f ? 1.0f : p.pf_alpha);
                    }
                }
                sdz.pre[k] = make_float4(g[0], g[1], g[2], g[3]);
            }
        }
        sdz.commit(lds + O_DZ1, tid);
        sy0.commit(lds + O_Y0, tid);
        sxa[0].commit(lds + O_XA, tid);
        if constexpr (UP) {
            sxa[1].commit(lds + O_XA + TXA::N, tid);
            slo.commit(lds + O_LOW, tid);
        }
    };

    FZB_STAMP(wave >= NWD, 0);
    int tile = blockIdx.x;
    int b = 0, x0 = 0, y0 = 0;
    if (tile < ntiles) {          // the first tile's loads fly during the prologue
        decode_tile(tile, ntiles, xcd_map, p.tiles_x, p.tiles_y, TW, TH, b, x0, y0);
        issue(b, x0, y0);
    }
#ifdef DNNCA_TUNING
    if (p.dbg & 8) {          // tuning aid: how long do the first tile's loads alone take?
        FZB_STAMP(wave >= NWD, 25);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FZB_STAMP(wave >= NWD, 26);
    }
#endif
    // the data-gradient B operands ([K-step][64 lanes], bm1 then bm0 back to back in LDS as in the plan's buffer) travel through
    // registers, in flight together with the tile: wave w fetches K-steps w, w + NW, ... one dword per lane (as float4's from all
    // threads at once these loads took 15 us to come back on the 12-channel kernels)
    constexpr int KSA = DC1::KS + NPASS * DC0::KS, NBR = cdiv(KSA, NW);
    static_assert(O_BM0 == O_BM1 + DC1::KS * 64, "bm1 and bm0 adjacent in LDS");
    float pbm[NBR];
#pragma unroll
    for (int k = 0; k < NBR; ++k) {
        const int s = min(wave + k * NW, KSA - 1);
        pbm[k] = s < DC1::KS ? p.bm1[s * 64 + lane] : p.bm0[(s - DC1::KS) * 64 + lane];
    }
    constexpr int NTW = UP ? cdiv(KTt * CT, NT) : 1;      // the transposed conv's kernel, one dword per thread and round
    float ptw[NTW];
    if constexpr (UP) {
#pragma unroll
        for (int k = 0; k < NTW; ++k) ptw[k] = p.tc_w[min(tid + k * NT, KTt * CT - 1)];
    }
    FZB_STAMP(wave >= NWD, 27);
    for (int i = tid; i < (LDSN + 3) / 4; i += NT) lds4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    FZB_STAMP(wave >= NWD, 28);
    __syncthreads();                     // (also drains the loads above: vmcnt(0))
    FZB_STAMP(wave >= NWD, 29);
    if (tid == 0) lds[CST1] = 1.0f;
#pragma unroll
    for (int k = 0; k < NBR; ++k)
        if (wave + k * NW < KSA) lds[O_BM1 + (wave + k * NW) * 64 + lane] = pbm[k];
    if constexpr (UP) {
#pragma unroll
        for (int k = 0; k < NTW; ++k)
            if (tid + k * NT < KTt * CT) lds[O_TCW + tid + k * NT] = ptw[k];
    }
    if (tile >= ntiles) return;          // (grid <= ntiles: never taken; keeps the barrier counts below uniform by construction)

    int cb = b, cx0 = x0, cy0 = y0;
    FZB_STAMP(wave >= NWD, 1);
    commit();
    tile += gridDim.x;
    lds_barrier();
    FZB_STAMP(wave >= NWD, 2);

    // ---- the two roles: identical barrier sequences, disjoint register sets
    auto role = [&](auto tag) {
        constexpr bool DG = decltype(tag)::value;
        const int wv = DG ? wave : wave - NWD;
        // weight-gradient role: accumulators and A-operand tables
        f32x4 acc1[1][MT1], acc0[NSRC][MT0], tacc[UP ? MBt : 1];
        W1 w1;
        W0 w0;
        if constexpr (DG) {
#pragma unroll
            for (int t = 0; t < (UP ? MBt : 1); ++t) tacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (!DG) {
#pragma unroll
            for (int t = 0; t < MT1; ++t) acc1[0][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NSRC; ++s)
#pragma unroll
                for (int t = 0; t < MT0; ++t) acc0[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            w1.init(lane, TY0::LEAD, TY0::LS, CST1, CST0);
            w0.init(lane, TXA::LEAD, TXA::LS, CST1, CST0);
        }
        int it = 0;
        (void)it;
#pragma unroll 1
        for (;; ++it) {
            const unsigned edge = tile_edge(cx0, cy0, TW, TH, p.H, p.W);
            FZB_STAMP(!DG, it < 3 ? 3 + 8 * it : 99);
            // the next tile's loads are issued at the top of phase ISSUE_AT: a thread's prefetch registers are live from there to the commit
            // at the end of the iteration -- the data-gradient waves can afford them from the top of the tile, the weight-gradient waves
            // (whose accumulators never die) only across the short last phase
            // (unconditional -- past the block's last tile the same tile again, never committed: under a branch hipcc's vmcnt bookkeeping
            //  turns pessimistic at the join and the next phase starts by waiting for these loads)
            auto prefetch = [&]() {
                decode_tile(tile < ntiles ? tile : tile - (int)gridDim.x, ntiles, xcd_map, p.tiles_x, p.tiles_y, TW, TH, b, x0, y0);
                issue(b, x0, y0);
            };
            constexpr int ISSUE_AT = DG ? ISSUE_DG : ISSUE_WG;
            if constexpr (ISSUE_AT == 1) prefetch();
            // ---- P1 (the data-gradient waves are this phase's long pole: they go first on the SIMD both kinds share)
            __builtin_amdgcn_s_setprio(DG ? 2 : 0);
            if constexpr (DG) {
                float breg[DC1::KS];
                load_breg<DC1::KS>(breg, lds + O_BM1, lane);
                DC1::run(lds + O_DZ1, lds + O_DZ0, breg, p.mask0 ? lds + O_Y0 : nullptr, p.alpha0, wv, lane);
            } else {
#pragma unroll 1
                for (int ty = wv; ty < TH; ty += NWW) {
                    const int xr[1] = {O_Y0 + ty * TY0::LS};
                    w1.template row<1>(acc1, ldsf, xr, O_DZ1 + TDZ1::LEAD + (ty + 2) * TDZ1::LS + 2 * F, CST0, lane);
                }
            }
            FZB_STAMP(!DG, it < 3 ? 4 + 8 * it : 99);
            lds_barrier();
            FZB_STAMP(!DG, it < 3 ? 5 + 8 * it : 99);
            // dz0 only exists inside the image: a ReLU mask zeroes the ring by itself (the staged y0 is zero out there)
            if (edge && !(p.mask0 && p.alpha0 == 0.f)) {          // block-uniform
                zero_ring<F, G, RG, TH + 2, TW + 2, TY0::LEAD, NT>(lds + O_DZ0, edge, tid);
                lds_barrier();
            }
            // ---- P2
            __builtin_amdgcn_s_setprio(0);           // (the other way round in P2 only swapped the two roles' finishing order)
            if constexpr (ISSUE_AT == 2) prefetch();
            if constexpr (DG) {
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) {
                    float breg[DC0::KS];
                    load_breg<DC0::KS>(breg, lds + O_BM0 + ps * DC0::KS * 64, lane);
                    DC0::run(lds + O_DZ0, lds + O_OUT + ps * OUT2, breg, nullptr, 0.f, wv, lane);
                }
                // ---- the wave's rectangle of the tile leaves (and, UP, feeds the transposed conv's backward): rows [r0, r0 + RPW) x
                // column block cbk of 16 groups -- written by this wave just now, so no workgroup barrier is needed
                __builtin_amdgcn_wave_barrier();
                const int r0 = (wv / MPR2) * RPW, cbk = wv % MPR2;
                if constexpr (UP) {
                    // skip gradient: channels [F, 2F) of the first conv's data gradient, 16 pixels per row
                    float* base = p.dxb + ((size_t)cb * p.H + cy0 + r0) * p.W * F + (size_t)(cx0 + cbk * 16) * F;
                    if constexpr (NPASS == 2) {           // pass 1 is the skip half: 16 x F floats = 48 float4 per row
                        static_assert(F == 12, "two passes: 12 channels per pass");
                        if (lane < 48) {
#pragma unroll
                            for (int rr = 0; rr < RPW; ++rr)
                                *reinterpret_cast<float4*>(base + (size_t)rr * p.W * F + 4 * lane) =
                                    *reinterpret_cast<const float4*>(lds + O_OUT + OUT2 + (r0 + rr) * ROWF + cbk * 192 + 4 * lane);
                        }
                    } else {                              // one pass: 12 floats per pixel = [up (F) | skip (F)]; 16 x F floats = 48 float2 per row
                        static_assert(F == 6, "one pass: [up | skip] of 6 channels each");
                        if (lane < 48) {
                            const int px = (2 * lane) / F, ch = 2 * lane - px * F;
#pragma unroll
                            for (int rr = 0; rr < RPW; ++rr)
                                *reinterpret_cast<float2*>(base + (size_t)rr * p.W * F + 2 * lane) =
                                    *reinterpret_cast<const float2*>(lds + O_OUT + (r0 + rr) * ROWF + (cbk * 16 + px) * 12 + F + ch);
                        }
                    }
                    // the transposed conv's backward on the wave's 8 x RPW/2 input pixels; the gradient of its output is the `up` half of the
                    // stage-2 tile: pixel (r, c) channel co at O_OUT + r * ROWF + c * 12 + co
                    const int m16 = lane & 15, q = lane >> 4;
                    const int lr0 = r0 / 2, lc0 = cbk * 8;                  // the rectangle's first input pixel (tile coordinates)
                    // T1: din[i][j][ci] = sum_(a,e,co) dup[2i+a][2j+e][co] W[a][e][co][ci];  M = 16 input pixels (8 wide, 2 rows), N = ci
                    int koff[KS1t];
#pragma unroll
                    for (int kk = 0; kk < KS1t; ++kk) {
                        const int k = 4 * kk + q, e = k / F;
                        koff[kk] = e * 12 + (k - e * F);
                    }
                    const int woff = m16 < CT ? O_TCW + q * CT + m16 : CST0, wstep = m16 < CT ? 4 * CT : 0;
                    float* orow = lds + O_ROW + wv * 192;
#pragma unroll
                    for (int mt = 0; mt < LPW / 16; ++mt) {
                        const int li = lr0 + 2 * mt + (m16 >> 3), lj = lc0 + (m16 & 7);
                        float ta[2][KS1t], tb[2][KS1t];
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int kk = 0; kk < KS1t; ++kk) {
                                ta[a][kk] = ldsf[O_OUT + (2 * li + a) * ROWF + 2 * lj * 12 + koff[kk]];
                                tb[a][kk] = ldsf[woff + (a * KS1t + kk) * wstep];
                            }
                        __builtin_amdgcn_sched_barrier(0);
                        f32x4 d2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                        for (int kk = 0; kk < KS1t; ++kk)
#pragma unroll
                            for (int a = 0; a < 2; ++a) d2[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[a][kk], tb[a][kk], d2[a], 0, 0, 0);
                        const f32x4 d = d2[0] + d2[1];
                        if (m16 < CT) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) orow[(4 * q + r) * CT + m16] = d[r];      // row = input pixel 4q + r of the M-tile
                        }
                        __builtin_amdgcn_wave_barrier();
                        if (lane < 16 * CT / 4) {
                            const int px = lane / 3, c4 = lane - 3 * px;                     // input pixel (px >> 3, px & 7) of the M-tile
                            const int oi = lr0 + 2 * mt + (px >> 3), oj = lc0 + (px & 7);
                            float4 v = reinterpret_cast<const float4*>(orow)[lane];
                            if (p.tc_mask) {
                                const float4 xv = reinterpret_cast<const float4*>(lds + O_LOW)[(oi * LWt + oj) * 3 + c4];
                                v.x *= xv.x > 0.f ? 1.0f : p.tc_alpha;
                                v.y *= xv.y > 0.f ? 1.0f : p.tc_alpha;
                                v.z *= xv.z > 0.f ? 1.0f : p.tc_alpha;
                                v.w *= xv.w > 0.f ? 1.0f : p.tc_alpha;
                            }
                            reinterpret_cast<float4*>(p.tc_din)[(((size_t)cb * (p.H >> 1) + (cy0 >> 1) + oi) * (p.W >> 1) + (cx0 >> 1) + oj) * 3 + c4] = v;
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                    // T2: dW[(a,e,co)][ci] += sum_pixels dup[..](a,e,co) * in[i][j][ci];  M = (a, e, co), N = ci (+ an all-ones column: the
                    // bias gradient), K = the rectangle's input pixels, four of one row per step
                    int offT[MBt];
                    bool valT[MBt];
#pragma unroll
                    for (int t = 0; t < MBt; ++t) {
                        const int k = 16 * t + m16, a = k / KA, kr = k - a * KA, e = kr / F;
                        valT[t] = k < KTt;
                        offT[t] = valT[t] ? O_OUT + a * ROWF + e * 12 + (kr - e * F) : CST0;
                    }
                    constexpr int KST = LPW / 4;
                    float tbv[KST], tav[KST][MBt];
#pragma unroll
                    for (int i = 0; i < KST; ++i) {
                        const int li = lr0 + i / 2, lj = lc0 + (i & 1) * 4 + q;          // this lane's input pixel of K-step i
                        tbv[i] = m16 < CT ? ldsf[O_LOW + (li * LWt + lj) * CT + m16] : (m16 == CT ? 1.0f : 0.0f);
                        const int sd = 2 * li * ROWF + 2 * lj * 12;
#pragma unroll
                        for (int t = 0; t < MBt; ++t) tav[i][t] = ldsf[offT[t] + (valT[t] ? sd : 0)];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < KST; ++i)
#pragma unroll
                        for (int t = 0; t < MBt; ++t) tacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(tav[i][t], tbv[i], tacc[t], 0, 0, 0);
                } else {
                    // gradient of the block's input: the wave's RPW full-width rows of TW * CA floats (= 48 float4)
                    static_assert(MPR2 == 1 && TW * CA == 192, "DOWN: a wave stores whole rows of 48 float4");
                    float* base = p.dxb + ((size_t)cb * p.H + cy0 + r0) * p.W * CA + (size_t)cx0 * CA;
                    if (lane < 48) {
#pragma unroll
                        for (int rr = 0; rr < RPW; ++rr)
                            *reinterpret_cast<float4*>(base + (size_t)rr * p.W * CA + 4 * lane) =
                                *reinterpret_cast<const float4*>(lds + O_OUT + (r0 + rr) * (TW * CA) + 4 * lane);
                    }
                }
            } else {
#pragma unroll 1
                for (int ty = wv; ty < TH; ty += NWW) {
                    int xr[NSRC];
#pragma unroll
                    for (int s = 0; s < NSRC; ++s) xr[s] = O_XA + s * TXA::N + ty * TXA::LS;
                    w0.template row<NSRC>(acc0, ldsf, xr, O_DZ0 + TY0::LEAD + (ty + 1) * TY0::LS + F, CST0, lane);
                }
            }
            FZB_STAMP(!DG, it < 3 ? 8 + 8 * it : 99);
            const bool more = tile < ntiles;                  // the registers hold a tile
            if (!more) break;
            lds_barrier();                                    // every reader of every tile is done
            FZB_STAMP(!DG, it < 3 ? 9 + 8 * it : 99);
            cb = b; cx0 = x0; cy0 = y0;
            commit();
            tile += gridDim.x;
            // nothing of this tile is pending when the next one starts: without this the next prefetch's address arithmetic, which reuses
            // the registers the tile's global stores were fed from, waits for those stores at the top of P1 (the builtin, not an asm
            // wait: hipcc's counter bookkeeping must see it)
            if constexpr (DG) __builtin_amdgcn_s_waitcnt(0x0070);
            lds_barrier();
            FZB_STAMP(!DG, it < 3 ? 10 + 8 * it : 99);
        }
        lds_barrier();                                        // the tiles are dead: their floats become the reduction area
        if constexpr (!DG) {
            float* red = lds + wv * ACCF;
#pragma unroll
            for (int t = 0; t < MT1; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(t * 4 + r) * 64 + lane] = acc1[0][t][r];
#pragma unroll
            for (int s = 0; s < NSRC; ++s)
#pragma unroll
                for (int t = 0; t < MT0; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[(MT1 + s * MT0 + t) * 256 + r * 64 + lane] = acc0[s][t][r];
        } else if constexpr (UP) {
            float* red = lds + NWW * ACCF + wv * TACCF;           // the transposed conv's sums: behind the weight-gradient waves' areas
#pragma unroll
            for (int t = 0; t < MBt; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[t * 256 + r * 64 + lane] = tacc[t][r];
        }
    };
    if (wave < NWD) role(std::true_type{});
    else role(std::false_type{});
    __syncthreads();
    FZB_STAMP(wave >= NWD, 30);
    // ---- the block adds its partial D's into slab (blockIdx % kPgBuckets): element e of a D is (M-tile t, register r, lane) =
    // row 16 t + 4 (lane >> 4) + r, column lane & 15; rows / columns past the valid ones are exact zeros and are skipped
#ifdef DNNCA_TUNING
    if (p.dbg & 1) return;
#endif
    const int bucket = blockIdx.x % kPgBuckets;
    // first: float offset of the D inside a wave's area; nw waves, `stride` floats apart, hold partial sums of it
    auto flush = [&](int first, int nw, int stride, int mt, float* slab, int rows, int cols) {
        for (int i = tid; i < mt * 256; i += NT) {
            const int t = i >> 8, r = (i >> 6) & 3, ln = i & 63;
            if ((ln & 15) < cols && 16 * t + 4 * (ln >> 4) + r < rows) {
                float v = 0.f;
                for (int w = 0; w < nw; ++w) v += lds[first + w * stride + i];
                atomicAdd(slab + (size_t)bucket * (mt * 256) + i, v);
            }
        }
    };
    flush(0, NWW, ACCF, MT1, p.slabs1, W1::MROWS, 12);
#pragma unroll
    for (int s = 0; s < NSRC; ++s) flush((MT1 + s * MT0) * 256, NWW, ACCF, MT0, p.slabs0[s], W0::MROWS, 12);
    if constexpr (UP) flush(NWW * ACCF, NWD, TACCF, MBt, p.tc_slabs, KTt, CT + 1);
    FZB_STAMP(wave >= NWD, 31);
}

}  // namespace fzb

// ================================================================================================ host side
static inline bool dense(const View& v) { return v.C == 0 || v.ps == v.C; }

static bool fzb_enabled() { return getenv("DNNCA_NO_FUSED_BWD") == nullptr; }      // read per call: the tests flip it
// tuning aid: DNNCA_FZB_ONLY=down1|down2|up0|up1 fuses only that block's backward (the number is the level: log2(512 / height) at 512 x 512)
static bool fzb_selected(const char* kind, int F) {
    const char* e = getenv("DNNCA_FZB_ONLY");
    if (!e) return true;
    char want[16];
    snprintf(want, sizeof(want), "%s%d", kind, F == 6 ? 1 : 2);
    return strcmp(e, want) == 0;
}

template <typename K>
static int fzb_grid(K kernel, int nt, int ntiles) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, nt, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    const int fit = 256 * (per_cu > 2 ? 2 : per_cu);
    return ntiles < fit ? ntiles : fit;
}

// ops[oi - 2 .. oi] = Conv2DTranspose(12 -> F, 2x2/2), conv3x3([up | skip] -> F), conv3x3(F -> F) of one Upsample block (no BatchNorm):
// the block's whole backward in one launch.  false: not this shape / these flags (the caller runs the layers one by one).
bool fused_up_bwd(Model* m, int B, size_t oi) {
    if (!fzb_enabled() || (m->desc.flags & 1) || m->desc.dtype != DNNCA_F32 || oi < 2 || oi >= m->ops.size()) return false;
    Op &tc = m->ops[oi - 2], &c0 = m->ops[oi - 1], &c1 = m->ops[oi];
    if (tc.type != OP_TCONV || c0.type != OP_CONV || c1.type != OP_CONV || tc.k != 2 || c0.k != 3 || c1.k != 3) return false;
    if (!fast_pg_conv_supported(m, c0) || !fast_pg_conv_supported(m, c1) || !fast_tconv_supported(m, tc)) return false;
    if (c0.inA.d.p != tc.out.d.p || !c0.inB.d.C || c1.inB.d.C || c1.inA.d.p != c0.out.d.p) return false;
    const int F = c1.out.d.C, H = c1.out.d.H, W = c1.out.d.W;
    if (!(F == 6 || F == 12) || tc.inA.d.C != 12 || tc.out.d.C != F || c0.inA.d.C != F || c0.inB.d.C != F || c0.out.d.C != F || c1.inA.d.C != F) return false;
    if (tc.inA.d.H * 2 != H || tc.inA.d.W * 2 != W || c0.out.d.H != H || c0.out.d.W != W || c0.inB.d.H != H || c0.inB.d.W != W) return false;
    const int TW = 32 * (12 / F);
    if (W % TW || H % 8) return false;
    if (!dense(tc.inA.d) || !dense(tc.inA.g) || !dense(tc.out.d) || !dense(c0.inB.d) || !dense(c0.inB.g) || !dense(c0.out.d) || !dense(c1.out.g)) return false;
    // gradients: c1's output gradient arrives pre-activation; c0's output gradient is formed here (act' of c0 applied by c1's data
    // gradient); the skip gradient is overwritten unmasked (the pool behind the skip applies its act'); nothing accumulates
    if (c1.alpha >= 0.f && !c1.premasked) return false;
    if (!c1.need_din || !c0.need_din || c1.accA || c0.accA || c0.accB || c0.maskA || c0.maskB || tc.accA) return false;
    if ((c0.alpha >= 0.f) != (c1.maskA != 0) || (c0.alpha >= 0.f && !c0.premasked)) return false;
    if (!fzb_selected("up", F)) return false;
    fzb::BArgs a{};
    a.dz1 = c1.out.g.p;
    a.y0 = c0.out.d.p;
    a.xa = c0.inA.d.p; a.xb = c0.inB.d.p;
    a.bm1 = fast_conv_bmat_dgrad(m, c1); a.bm0 = fast_conv_bmat_dgrad(m, c0);
    a.dxb = c0.inB.g.p;
    a.mask0 = c1.maskA; a.alpha0 = c1.mask_alpha;
    a.slabs1 = fast_wgrad_slabs(m, c1, 0);
    a.slabs0[0] = fast_wgrad_slabs(m, c0, 0); a.slabs0[1] = fast_wgrad_slabs(m, c0, 1);
    a.tc_in = tc.inA.d.p; a.tc_din = tc.inA.g.p; a.tc_w = m->p + tc.w_off; a.tc_slabs = fast_wgrad_slabs(m, tc, 0);
    a.tc_mask = tc.maskA; a.tc_alpha = tc.mask_alpha;
    if (!a.bm1 || !a.bm0 || !a.slabs1 || !a.slabs0[0] || !a.slabs0[1] || !a.tc_slabs) return false;
    a.B = B; a.H = H; a.W = W;
    a.tiles_x = W / TW; a.tiles_y = H / 8;
    if (const char* e = getenv("DNNCA_FZB_DBG")) a.dbg = atoi(e);
    const int ntiles = a.tiles_x * a.tiles_y * B;
    const double npx = (double)B * H * W;
    // algorithmic bytes / FLOPs of the three layers' backward (out-gradient + 2 x inputs per layer)
    const double bytes = 4.0 * npx * ((F + 2 * F) + (F + 2 * 2 * F) + 2 * (F + 0.25 * 12));
    const double flops = 2.0 * (2.0 * npx * 9.0 * (F * F + 2 * F * F)) + 2.0 * (2.0 * npx * F * 12);
#define X(f, tw)                                                                                                   \
    if (F == f) {                                                                                                  \
        const int g = fzb_grid(fzb::k_fzb<true, f, f, tw, 512>, 512, ntiles);                                      \
        if (a.dbg & 2) {        /* tuning aid: the same launch first without its slab atomics (what does a warm start look like?) */ \
            fzb::BArgs a2 = a;                                                                                     \
            a2.dbg |= 1;                                                                                           \
            LAUNCH(m, "fzb_up_pre_" #f, bytes, flops,                                                              \
                   hipLaunchKernelGGL((fzb::k_fzb<true, f, f, tw, 512>), dim3(g), dim3(512), 0, m->stream, a2));   \
        }                                                                                                          \
        LAUNCH(m, "fzb_up_" #f, bytes, flops,                                                                      \
               hipLaunchKernelGGL((fzb::k_fzb<true, f, f, tw, 512>), dim3(g), dim3(512), 0, m->stream, a));        \
        return true;                                                                                               \
    }
    X(12, 32) X(6, 64)
#undef X
    return false;
}

// ops[oi - 2 .. oi] = conv3x3(CA -> F), conv3x3(F -> F), MaxPool2D(2) of one Downsample block (no BatchNorm) whose forward pass
// recorded the pool's window positions (k_fz_down): pool backward + both convs' backward in one launch.
bool fused_down_bwd(Model* m, int B, size_t oi) {
    if (!fzb_enabled() || (m->desc.flags & 1) || m->desc.dtype != DNNCA_F32 || oi < 2 || oi >= m->ops.size()) return false;
    Op &c1 = m->ops[oi - 2], &c2 = m->ops[oi - 1], &pl = m->ops[oi];
    if (c1.type != OP_CONV || c2.type != OP_CONV || pl.type != OP_POOL || c1.k != 3 || c2.k != 3 || pl.k != 2) return false;
    if (!fast_pg_conv_supported(m, c1) || !fast_pg_conv_supported(m, c2)) return false;
    if (c1.inB.d.C || c2.inB.d.C || c2.inA.d.p != c1.out.d.p || pl.inA.d.p != c2.out.d.p) return false;
    const int CA = c1.inA.d.C, F = c1.out.d.C, H = c1.out.d.H, W = c1.out.d.W;
    if (!((CA == 6 && F == 12) || (CA == 3 && F == 6)) || c2.out.d.C != F) return false;
    const int TW = 32 * (12 / F);
    if (W % TW || H % 8) return false;
    if (!dense(c1.inA.d) || !dense(c1.inA.g) || !dense(c1.out.d) || !dense(c2.out.d) || !dense(c2.out.g) || !dense(pl.out.g) || !dense(pl.out.d)) return false;
    // the pool fold's conditions (fast_pool_fold): recorded positions; the pool adds to the skip gradient and applies the ReLU mask of c2
    if (!pl.pool_idx_valid || !(pl.pool_idx || m->dry) || !pl.accA || !pl.maskA || pl.mask_alpha != 0.f || !c2.premasked) return false;
    if (!c1.need_din || !c2.need_din || c1.accA || c1.maskA || c2.accA) return false;
    if ((c1.alpha >= 0.f) != (c2.maskA != 0) || (c1.alpha >= 0.f && !c1.premasked)) return false;
    if (!fzb_selected("down", F)) return false;
    fzb::BArgs a{};
    a.dz1 = c2.out.g.p;
    a.y1 = c2.out.d.p;
    a.dpool = pl.out.g.p; a.pidx = pl.pool_idx; a.pf_alpha = pl.mask_alpha;
    a.y0 = c1.out.d.p;
    a.xa = c1.inA.d.p;
    a.bm1 = fast_conv_bmat_dgrad(m, c2); a.bm0 = fast_conv_bmat_dgrad(m, c1);
    a.dxb = c1.inA.g.p;
    a.mask0 = c2.maskA; a.alpha0 = c2.mask_alpha;
    a.slabs1 = fast_wgrad_slabs(m, c2, 0);
    a.slabs0[0] = fast_wgrad_slabs(m, c1, 0);
    if (!a.bm1 || !a.bm0 || !a.slabs1 || !a.slabs0[0]) return false;
    a.B = B; a.H = H; a.W = W;
    a.tiles_x = W / TW; a.tiles_y = H / 8;
    if (const char* e = getenv("DNNCA_FZB_DBG")) a.dbg = atoi(e);
    const int ntiles = a.tiles_x * a.tiles_y * B;
    const double npx = (double)B * H * W;
    // pool backward (y, dy in; dx in/out; pooled gradient), second conv backward, first conv backward
    const double bytes = 4.0 * npx * ((2 * F + 0.5 * F) + (F + 2 * F) + (F + 2 * CA));
    const double flops = 2.0 * (2.0 * npx * 9.0 * (F * F + CA * F));
    pl.pool_idx_valid = false;
#define X(ca, f, tw)                                                                                               \
    if (CA == ca && F == f) {                                                                                      \
        const int g = fzb_grid(fzb::k_fzb<false, ca, f, tw, 512>, 512, ntiles);                                    \
        LAUNCH(m, "fzb_down_" #ca "_" #f, bytes, flops,                                                            \
               hipLaunchKernelGGL((fzb::k_fzb<false, ca, f, tw, 512>), dim3(g), dim3(512), 0, m->stream, a));      \
        return true;                                                                                               \
    }
    X(6, 12, 32) X(3, 6, 64)
#undef X
    return false;
}

}  // namespace dnnca

// development aid (not part of include/dnnca.h): the stamps of the last block-fused backward kernel of a tuning build
extern "C" int dnnca_debug_fzb_stamps(unsigned long long* out, int n) {
#ifdef DNNCA_TUNING
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(dnnca::fzb::g_fzb_stamps), (size_t)n * 8) == hipSuccess ? 0 : -1;
#else
    (void)out; (void)n;
    return -2;
#endif
}
