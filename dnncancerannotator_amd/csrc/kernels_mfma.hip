// kernels_mfma.hip -- "pixel-group" implicit-GEMM 3x3 convolutions on the fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
// configs/unet.yaml has 3/6/12 channels: as a plain implicit GEMM (M = pixels, N = Cout, K = 9*Cin) the N dimension
// would fill 3..12 of 16 MFMA columns.  Instead G = 12/Cout horizontally adjacent output pixels share one GEMM row:
//     M = pixel groups,  N = (dx, co) = G*Cout (= 12 of 16),  K = (dy, j) over the group's 3 x (G+2)*C input window,
// with B[(dy, j), (dx, co)] = W[dy][xoff - dx][ci][co] (j = xoff*C + ci) where 0 <= xoff - dx <= 2 and 0 elsewhere.
// A group's window row is (G+2)*C *contiguous* NHWC floats, so the A operand is a plain strided LDS read and the tile
// is staged global->LDS with fully coalesced 16-byte loads.  The same kernel does the data gradient (weights flipped
// and transposed by the prep kernel, activation-derivative mask and gradient accumulation fused into the store).
// The weight gradient uses the same grouping with K = groups (k_pgwgrad below).
//
// exact fp32: the f32 MFMA is a chain of fmaf (MI355X guide, "FP32-input MFMA"), so parity with the oracle holds to
// fp32 rounding.
#include "fast.h"
#include "kernels.h"

namespace dnnca {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int TX = 2;    // M-tiles (16 groups each) across a block tile
static constexpr int TH = 8;    // rows of a block tile

template <int C, int G>
struct PG {
    static constexpr int WR = (G + 2) * C;            // floats of one window row
    static constexpr int SR = (WR + 3) / 4;           // K-steps (4 floats) per window row
    static constexpr int HL = (C + 3) / 4 * 4;        // staged floats in front of the tile's first pixel
    static constexpr int LEAD = HL - C;
    static constexpr int TW = 16 * G * TX;            // tile width in pixels
    static constexpr int LS = (LEAD + (TW + 2) * C + 3 + 3) / 4 * 4;   // LDS row stride (floats)
    static constexpr int LS4 = LS / 4;
};

struct PgArgs {
    const float* src[2];     // dense NHWC sources, C channels each
    const float* bmat;       // prepared B operand: [KS][64]
    const float* bias;       // forward: CO floats
    float* dst[2];           // forward: dst[0] (CO channels).  dgrad: dst[0] gets channels [0, C1), dst[1] [C1, CO)
    const float* mask[2];    // dgrad: data tensors whose activation derivative multiplies the gradient (or nullptr)
    int acc[2];              // dgrad: accumulate into dst instead of overwriting
    int C1;                  // dgrad: channels of dst[0]
    int B, H, W;
    float alpha;             // activation slope: forward epilogue / dgrad mask (<0: none)
};

// MODE 0: forward (bias + activation).  MODE 1: data gradient (mask, accumulate, split destination).
template <int C, int NSRC, int CO, int G, int MODE>
__global__ __launch_bounds__(256) void k_pgconv(PgArgs p) {
    using P = PG<C, G>;
    constexpr int SR = P::SR, LS = P::LS, LS4 = P::LS4, TW = P::TW, N = G * CO;
    constexpr int KS = NSRC * 3 * SR;
    static_assert(N <= 16, "G*CO must fit the 16 MFMA columns");
    __shared__ float4 lds4[NSRC * (TH + 2) * LS4];
    float* lds = reinterpret_cast<float*>(lds4);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, b = blockIdx.z;
    const int rowlen4 = p.W * C / 4;
    const int g40 = (x0 * C - P::HL) / 4;

    // ---- stage (TH+2) x (TW+2) pixels of every source: coalesced 16-B loads, zero padding outside the image
#pragma unroll
    for (int s = 0; s < NSRC; ++s) {
        const float4* base = reinterpret_cast<const float4*>(p.src[s] + (size_t)b * p.H * p.W * C);
        for (int idx = tid; idx < (TH + 2) * LS4; idx += 256) {
            int row = idx / LS4, c4 = idx - row * LS4;
            int iy = y0 - 1 + row, g4 = g40 + c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < p.H && g4 >= 0 && g4 < rowlen4) v = base[(size_t)iy * rowlen4 + g4];
            lds4[s * (TH + 2) * LS4 + idx] = v;
        }
    }
    // ---- B operand: one register per K-step, resident for the whole block
    float breg[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) breg[s] = p.bmat[s * 64 + lane];
    __syncthreads();

    const int m = lane & 15, q = lane >> 4;      // A: row m, k = q.  D: col n = m, rows 4q + r
    const int n = m;
    const int co = n % CO, dx = n / CO;
    float bias = 0.f;
    if (MODE == 0 && n < N) bias = p.bias[co];

#pragma unroll 1
    for (int t = wave; t < TX * TH; t += 4) {
        const int tx = t % TX, ty = t / TX;
        const int y = y0 + ty;
        if (y >= p.H) break;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int aoff = P::LEAD + (tx * 16 + m) * (G * C) + q;
#pragma unroll
        for (int s = 0; s < NSRC; ++s)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const float* ap = lds + (s * (TH + 2) + ty + dy) * LS + aoff;
#pragma unroll
                for (int k = 0; k < SR; ++k)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * k], breg[(s * 3 + dy) * SR + k], acc, 0, 0, 0);
            }
        if (n >= N) continue;
        const size_t prow = ((size_t)b * p.H + y) * p.W;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int px = x0 + (tx * 16 + 4 * q + r) * G + dx;
            if (px >= p.W) continue;
            float v = acc[r];
            if (MODE == 0) {
                v += bias;
                v = p.alpha < 0.f ? v : (v > 0.f ? v : p.alpha * v);
                p.dst[0][(prow + px) * CO + co] = v;
            } else {
                const int which = co >= p.C1;
                const int cw = which ? CO - p.C1 : p.C1;
                const size_t o = (prow + px) * cw + (which ? co - p.C1 : co);
                const float* mk = p.mask[which];
                float* d = p.dst[which];
                if (p.acc[which]) v += d[o];
                if (mk) v *= (mk[o] > 0.f ? 1.0f : p.alpha);
                d[o] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ B-operand prep
// One block per descriptor.  forward:  B[(s,dy,j),(dx,co)] = W[dy][xoff-dx][s*C+ci][co]
//                            dgrad:    B[(dy,j),(dx,o)]   = W[2-dy][2-(xoff-dx)][o_off+o][ci]   (input = gradient, C = Cout)
struct PrepDesc {
    int w_off, out_off;      // floats: weight offset in the parameter vector, output offset in the bmat buffer
    int C, NSRC, CO, G, mode;
    int cin_total, cout_total, o_off;
};

__global__ void k_pg_prep(const PrepDesc* __restrict__ descs, const float* __restrict__ params, float* __restrict__ bmat) {
    const PrepDesc d = descs[blockIdx.x];
    const int WR = (d.G + 2) * d.C, SR = (WR + 3) / 4;
    const int KS = d.NSRC * 3 * SR;
    const float* w = params + d.w_off;
    for (int i = threadIdx.x; i < KS * 64; i += blockDim.x) {
        int lane = i & 63, step = i >> 6;
        int n = lane & 15, kk = lane >> 4;
        int s = step / (3 * SR), dy = (step / SR) % 3, k = step % SR;
        int j = 4 * k + kk;
        float v = 0.f;
        if (j < WR && n < d.G * d.CO) {
            int xoff = j / d.C, ci = j % d.C, dx = n / d.CO, co = n % d.CO;
            int kx = xoff - dx;
            if (kx >= 0 && kx <= 2) {
                if (d.mode == 0)
                    v = w[((dy * 3 + kx) * d.cin_total + s * d.C + ci) * d.cout_total + co];
                else
                    v = w[(((2 - dy) * 3 + (2 - kx)) * d.cin_total + d.o_off + co) * d.cout_total + ci];
            }
        }
        bmat[d.out_off + i] = v;
    }
}


// ------------------------------------------------------------------------------------------------ weight gradient
// D[(dy, j), (dx, co)] = sum over pixel groups of X_window[dy][j] * dz[group*G + dx][co]   (+ one all-ones A row -> bias)
// M = 3*WR + 1 rows (MT tiles of 16), N = G*CO, K = groups (4 per MFMA).  Every wave keeps all MT accumulators in
// registers while its block walks tiles (persistent grid); the 4 waves are summed through LDS at the end and each
// block writes ONE partial slab -- no atomics, bit-reproducible.  k_pg_fold then sums the slabs and the dx-diagonals:
//     dW[dy][kx][ci][co] = sum_slabs sum_dx D[(dy, (dx + kx)*C + ci), (dx, co)]
struct WgArgs {
    const float* x;          // dense NHWC source, C channels
    const float* dz;         // dense NHWC gradient of the conv's pre-activation output, CO channels
    float* slabs;            // [gridDim.x][MT*256]
    int B, H, W;
    int tiles_x, tiles_y;
};

template <int C, int CO, int G>
struct WG {
    using P = PG<C, G>;
    static constexpr int N = G * CO;
    static constexpr int MROWS = 3 * P::WR + 1;
    static constexpr int MT = (MROWS + 15) / 16;
    static constexpr int LG = P::TW * CO;                 // floats of one dz tile row
    static constexpr int XS4 = (TH + 2) * P::LS4;
    static constexpr int GS4 = TH * LG / 4;
    static constexpr int RED4 = 4 * MT * 64;              // cross-wave reduction buffer (float4 units: 4 waves * MT*256/4)
    static constexpr int LDS4 = (XS4 + GS4) > RED4 ? (XS4 + GS4) : RED4;
};

template <int C, int CO, int G>
__global__ __launch_bounds__(256) void k_pgwgrad(WgArgs p) {
    using P = PG<C, G>;
    using Wg = WG<C, CO, G>;
    constexpr int LS = P::LS, LS4 = P::LS4, TW = P::TW, N = Wg::N, MT = Wg::MT, LG = Wg::LG, WR = P::WR;
    __shared__ float4 lds4[Wg::LDS4];
    float* xl = reinterpret_cast<float*>(lds4);
    float* gl = reinterpret_cast<float*>(lds4 + Wg::XS4);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m16 = lane & 15, q = lane >> 4, n = m16;
    const int rowlen4x = p.W * C / 4, rowlen4g = p.W * CO / 4;

    int offA[MT];
    int kind[MT];      // 0: LDS read, 1: constant one (bias row), 2: zero (padding rows)
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        int mrow = 16 * t + m16;
        if (mrow < 3 * WR) {
            int dy = mrow / WR, j = mrow - dy * WR;
            offA[t] = dy * LS + P::LEAD + j + q * (G * C);
            kind[t] = 0;
        } else {
            offA[t] = 0;
            kind[t] = mrow == 3 * WR ? 1 : 2;
        }
    }
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ntiles = p.tiles_x * p.tiles_y * p.B;
#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int bx = tile % p.tiles_x, by = (tile / p.tiles_x) % p.tiles_y, b = tile / (p.tiles_x * p.tiles_y);
        const int x0 = bx * TW, y0 = by * TH;
        __syncthreads();      // the previous tile's LDS reads are complete
        {
            const float4* base = reinterpret_cast<const float4*>(p.x + (size_t)b * p.H * p.W * C);
            const int g40 = (x0 * C - P::HL) / 4;
            for (int idx = tid; idx < (TH + 2) * LS4; idx += 256) {
                int row = idx / LS4, c4 = idx - row * LS4;
                int iy = y0 - 1 + row, g4 = g40 + c4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (iy >= 0 && iy < p.H && g4 >= 0 && g4 < rowlen4x) v = base[(size_t)iy * rowlen4x + g4];
                lds4[idx] = v;
            }
            const float4* gbase = reinterpret_cast<const float4*>(p.dz + (size_t)b * p.H * p.W * CO);
            const int gg0 = x0 * CO / 4;
            for (int idx = tid; idx < TH * (LG / 4); idx += 256) {
                int row = idx / (LG / 4), c4 = idx - row * (LG / 4);
                int iy = y0 + row, g4 = gg0 + c4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (iy < p.H && g4 < rowlen4g) v = gbase[(size_t)iy * rowlen4g + g4];
                lds4[Wg::XS4 + idx] = v;
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int ty = wave; ty < TH; ty += 4) {
            const float* xr = xl + ty * LS;
            const float* gr = gl + ty * LG + q * N + n;
#pragma unroll
            for (int s = 0; s < TW / (4 * G); ++s) {
                float bv = n < N ? gr[s * 4 * N] : 0.f;
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    float av = kind[t] == 0 ? xr[offA[t] + s * 4 * G * C] : (kind[t] == 1 ? 1.0f : 0.0f);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[t], 0, 0, 0);
                }
            }
        }
    }
    // ---- sum the 4 waves through LDS, one slab per block
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds4);
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * MT * 4 + t * 4 + r) * 64 + lane] = acc[t][r];
    __syncthreads();
    float* slab = p.slabs + (size_t)blockIdx.x * (MT * 256);
    for (int i = tid; i < MT * 256; i += 256)
        slab[i] = (red[i] + red[MT * 256 + i]) + (red[2 * MT * 256 + i] + red[3 * MT * 256 + i]);
}

// fold: one block per (descriptor, chunk of 256 outputs)
struct FoldDesc {
    int slab_off, nslabs, MT;      // slabs of this (op, source): floats offset into the slab buffer
    int C, G, CO;                  // geometry of the pixel-group GEMM
    int cin_total, ci_off;         // where this source's channels sit in the HWIO kernel
    int w_off, b_off;              // gradient destinations (floats into the flat gradient vector); b_off < 0: no bias
};

__device__ __forceinline__ int slab_index(int mrow, int n) {
    int t = mrow >> 4, r = mrow & 3, qq = (mrow & 15) >> 2;
    return (t * 4 + r) * 64 + qq * 16 + n;
}

// stage 1: dsum[desc][i] += sum over a split of the slabs (coalesced, many waves in flight; 1 atomic per element per split)
static constexpr int FOLD_SPLIT = 8;
__global__ __launch_bounds__(64) void k_pg_slabsum(const FoldDesc* __restrict__ descs, const float* __restrict__ slabs,
                                                   float* __restrict__ dsum, int dsum_stride) {
    const FoldDesc d = descs[blockIdx.x];
    const int i = blockIdx.y * 64 + threadIdx.x;
    const int n = d.MT * 256;
    if (i >= n) return;
    const int per = (d.nslabs + FOLD_SPLIT - 1) / FOLD_SPLIT;
    const int s0 = blockIdx.z * per, s1 = s0 + per < d.nslabs ? s0 + per : d.nslabs;
    const float* sl = slabs + d.slab_off + i;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int s = s0;
    for (; s + 3 < s1; s += 4) {
        a0 += sl[(size_t)s * n];
        a1 += sl[(size_t)(s + 1) * n];
        a2 += sl[(size_t)(s + 2) * n];
        a3 += sl[(size_t)(s + 3) * n];
    }
    for (; s < s1; ++s) a0 += sl[(size_t)s * n];
    if (s1 > s0) atomicAdd(dsum + (size_t)blockIdx.x * dsum_stride + i, (a0 + a1) + (a2 + a3));
}

// stage 2: dW[dy][kx][ci][co] = sum_dx D[(dy, (dx + kx)*C + ci), (dx, co)];  db[co] = sum_dx D[ones row, (dx, co)]
__global__ void k_pg_fold(const FoldDesc* __restrict__ descs, const float* __restrict__ dsum, int dsum_stride,
                          float* __restrict__ grads) {
    const FoldDesc d = descs[blockIdx.x];
    const int WR = (d.G + 2) * d.C;
    const int nw = 9 * d.C * d.CO;
    const int total = nw + (d.b_off >= 0 ? d.CO : 0);
    const int o = blockIdx.y * blockDim.x + threadIdx.x;
    if (o >= total) return;
    const float* D = dsum + (size_t)blockIdx.x * dsum_stride;
    float acc = 0.f;
    if (o < nw) {
        int co = o % d.CO, ci = (o / d.CO) % d.C, tap = o / (d.CO * d.C);
        int dy = tap / 3, kx = tap % 3;
        for (int dx = 0; dx < d.G; ++dx) acc += D[slab_index(dy * WR + (dx + kx) * d.C + ci, dx * d.CO + co)];
        grads[d.w_off + ((tap * d.cin_total) + d.ci_off + ci) * d.CO + co] = acc;
    } else {
        int co = o - nw;
        for (int dx = 0; dx < d.G; ++dx) acc += D[slab_index(3 * WR, dx * d.CO + co)];
        grads[d.b_off + co] = acc;
    }
}

// ------------------------------------------------------------------------------------------------ host side
struct PgPlan {                      // per-model table built lazily on the first step
    bool built = false;
    std::vector<PrepDesc> descs;
    PrepDesc* descs_dev = nullptr;
    float* bmat = nullptr;
    std::map<std::pair<const Op*, int>, int> slot;   // (op, kind) -> bmat offset;  kind 0 fwd, 1 dgrad(A or both), 2 dgrad B
    // weight gradient: per (op, source) slab region + fold descriptor
    std::vector<FoldDesc> folds;
    FoldDesc* folds_dev = nullptr;
    float* slabs = nullptr;
    float* dsum = nullptr;
    int dsum_stride = 0, max_mt = 0;
    std::map<std::pair<const Op*, int>, int> wslot;  // (op, source) -> index into folds
    int fold_chunks = 0;
};

static std::map<Model*, PgPlan> g_plans;

void fast_release(Model* m) { g_plans.erase(m); }

static inline bool dense(const View& v) { return v.C == 0 || v.ps == v.C; }

static int pick_g(int CO) { return CO == 3 ? 4 : CO == 6 ? 2 : CO == 12 ? 1 : 0; }

static bool conv_supported(const Model* m, const Op& o) {
    if (o.type != OP_CONV || o.k != 3 || m->desc.dtype != DNNCA_F32) return false;
    if (!dense(o.inA.d) || !dense(o.inB.d) || !dense(o.out.d)) return false;
    const int CA = o.inA.d.C, CB = o.inB.d.C, CO = o.out.d.C;
    if (CB && CB != CA) return false;
    if (!(CA == 1 || CA == 3 || CA == 6 || CA == 12)) return false;
    int G = pick_g(CO);
    if (!G) return false;
    if (o.out.d.W % 4) return false;
    if (!((CA == 1 && CO == 3) || (CA == 3 && (CO == 3 || CO == 6)) || (CA == 6 && (CO == 6 || CO == 12)) || (CA == 12 && CO == 12)))
        return false;
    return true;
}

static int build_plan(Model* m, PgPlan& pl) {
    int off = 0;
    auto add = [&](const Op* op, int kind, PrepDesc d) {
        int WR = (d.G + 2) * d.C, SR = (WR + 3) / 4;
        d.out_off = off;
        pl.slot[{op, kind}] = off;
        off += d.NSRC * 3 * SR * 64;
        pl.descs.push_back(d);
    };
    for (const Op& o : m->ops) {
        if (!conv_supported(m, o)) continue;
        const int CA = o.inA.d.C, CB = o.inB.d.C, CO = o.out.d.C;
        PrepDesc d{};
        d.w_off = (int)o.w_off;
        d.C = CA; d.NSRC = CB ? 2 : 1; d.CO = CO; d.G = pick_g(CO); d.mode = 0;
        d.cin_total = CA + CB; d.cout_total = CO; d.o_off = 0;
        add(&o, 0, d);
        if (!o.need_din) continue;
        // data gradient: input = dz (CO channels); destinations A (CA) and B (CB)
        const int Ctot = CA + CB;
        PrepDesc g{};
        g.w_off = (int)o.w_off; g.C = CO; g.NSRC = 1; g.mode = 1; g.cin_total = Ctot; g.cout_total = CO;
        if (pick_g(Ctot) && (CO == 3 || CO == 6 || CO == 12)) {
            g.CO = Ctot; g.G = pick_g(Ctot); g.o_off = 0;
            add(&o, 1, g);
        } else if (pick_g(CA) && (CO == 3 || CO == 6 || CO == 12)) {   // one launch per destination
            g.CO = CA; g.G = pick_g(CA); g.o_off = 0;
            add(&o, 1, g);
            if (CB) { g.o_off = CA; add(&o, 2, g); }
        }
    }
    if (pl.descs.empty()) { pl.built = true; return DNNCA_OK; }
    // weight-gradient slabs
    size_t slab_floats = 0;
    int max_out = 0;
    for (const Op& o : m->ops) {
        if (!conv_supported(m, o)) continue;
        const int CA = o.inA.d.C, CB = o.inB.d.C, CO = o.out.d.C, G = pick_g(CO);
        const int TWp = 16 * G * TX;
        const int tiles = ((o.out.d.W + TWp - 1) / TWp) * ((o.out.d.H + TH - 1) / TH) * m->desc.max_batch;
        for (int sidx = 0; sidx < (CB ? 2 : 1); ++sidx) {
            FoldDesc f{};
            f.C = CA; f.G = G; f.CO = CO;
            f.MT = (3 * (G + 2) * CA + 1 + 15) / 16;
            f.nslabs = tiles < 512 ? tiles : 512;
            f.slab_off = (int)slab_floats;
            f.cin_total = CA + CB; f.ci_off = sidx * CA;
            f.w_off = (int)o.w_off;
            f.b_off = sidx == 0 ? (int)o.b_off : -1;
            slab_floats += (size_t)f.nslabs * f.MT * 256;
            pl.wslot[{&o, sidx}] = (int)pl.folds.size();
            pl.folds.push_back(f);
            if (f.MT > pl.max_mt) pl.max_mt = f.MT;
            int outs = 9 * CA * CO + CO;
            if (outs > max_out) max_out = outs;
        }
    }
    pl.fold_chunks = (max_out + 255) / 256;
    DN_TRY(m->alloc((void**)&pl.folds_dev, pl.folds.size() * sizeof(FoldDesc)));
    DN_TRY(m->alloc((void**)&pl.slabs, slab_floats * 4));
    pl.dsum_stride = pl.max_mt * 256;
    DN_TRY(m->alloc((void**)&pl.dsum, pl.folds.size() * (size_t)pl.dsum_stride * 4));
    HIP_TRY(hipMemcpyAsync(pl.folds_dev, pl.folds.data(), pl.folds.size() * sizeof(FoldDesc), hipMemcpyHostToDevice, m->stream));
    DN_TRY(m->alloc((void**)&pl.descs_dev, pl.descs.size() * sizeof(PrepDesc)));
    DN_TRY(m->alloc((void**)&pl.bmat, (size_t)off * 4));
    HIP_TRY(hipMemcpyAsync(pl.descs_dev, pl.descs.data(), pl.descs.size() * sizeof(PrepDesc), hipMemcpyHostToDevice, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    pl.built = true;
    return DNNCA_OK;
}

// Called by model.hip at the top of every forward: (re)derive the B operands from the current weights.
int fast_prepare(Model* m) {
    if (m->desc.flags & 1) return DNNCA_OK;
    PgPlan& pl = g_plans[m];
    if (!pl.built) DN_TRY(build_plan(m, pl));
    if (pl.descs.empty()) return DNNCA_OK;
    LAUNCH(m, "pg_prep", 0, 0,
           hipLaunchKernelGGL(k_pg_prep, dim3((unsigned)pl.descs.size()), dim3(256), 0, m->stream, pl.descs_dev, m->p, pl.bmat));
    return DNNCA_OK;
}

template <int C, int NSRC, int CO, int G, int MODE>
static void launch_pg(Model* m, const PgArgs& a, const char* name, double bytes, double flops) {
    using P = PG<C, G>;
    dim3 grid((a.W + P::TW - 1) / P::TW, (a.H + TH - 1) / TH, a.B);
    LAUNCH(m, name, bytes, flops, hipLaunchKernelGGL((k_pgconv<C, NSRC, CO, G, MODE>), grid, dim3(256), 0, m->stream, a));
}

#define PG_DISPATCH(MODE, c, ns, co, NAME)                                                          \
    if (C == c && NS == ns && CO == co) {                                                           \
        launch_pg<c, ns, co, (co == 3 ? 4 : co == 6 ? 2 : 1), MODE>(m, a, NAME "_" #c "x" #ns "_" #co, bytes, flops); \
        return true;                                                                                \
    }

bool fast_conv_fwd(Model* m, int B, Op& o, double bytes, double flops) {
    if (!conv_supported(m, o)) return false;
    PgPlan& pl = g_plans[m];
    auto it = pl.slot.find({&o, 0});
    if (it == pl.slot.end()) return false;
    PgArgs a{};
    a.src[0] = o.inA.d.p; a.src[1] = o.inB.d.p;
    a.bmat = pl.bmat + it->second;
    a.bias = m->p + o.b_off;
    a.dst[0] = o.out.d.p;
    a.B = B; a.H = o.out.d.H; a.W = o.out.d.W;
    a.alpha = o.alpha;
    const int C = o.inA.d.C, NS = o.inB.d.C ? 2 : 1, CO = o.out.d.C;
    PG_DISPATCH(0, 1, 1, 3, "pgconv_fwd")
    PG_DISPATCH(0, 3, 1, 3, "pgconv_fwd")
    PG_DISPATCH(0, 3, 2, 3, "pgconv_fwd")
    PG_DISPATCH(0, 3, 1, 6, "pgconv_fwd")
    PG_DISPATCH(0, 6, 1, 6, "pgconv_fwd")
    PG_DISPATCH(0, 6, 2, 6, "pgconv_fwd")
    PG_DISPATCH(0, 6, 1, 12, "pgconv_fwd")
    PG_DISPATCH(0, 12, 1, 12, "pgconv_fwd")
    PG_DISPATCH(0, 12, 2, 12, "pgconv_fwd")
    return false;
}

// data gradient of a conv through the same kernel.  Returns false when the shape has no specialisation.
static bool pg_dgrad(Model* m, PgPlan& pl, int B, Op& o, double bytes, double flops) {
    auto it = pl.slot.find({&o, 1});
    if (it == pl.slot.end()) return false;
    const int CA = o.inA.d.C, CB = o.inB.d.C;
    const PrepDesc* d1 = nullptr;
    for (auto& d : pl.descs)
        if (d.out_off == it->second) d1 = &d;
    const int nlaunch = (CB && d1->CO == CA) ? 2 : 1;
    for (int li = 0; li < nlaunch; ++li) {
        PgArgs a{};
        a.src[0] = o.out.g.p;       // dz (already multiplied by the activation derivative of this conv's output)
        a.B = B; a.H = o.out.d.H; a.W = o.out.d.W;
        a.alpha = o.mask_alpha;
        if (nlaunch == 1) {
            a.bmat = pl.bmat + it->second;
            a.dst[0] = o.inA.g.p; a.dst[1] = o.inB.g.p;
            a.acc[0] = o.accA; a.acc[1] = o.accB;
            a.mask[0] = o.maskA ? o.inA.d.p : nullptr;
            a.mask[1] = o.maskB ? o.inB.d.p : nullptr;
            a.C1 = CA;
        } else {
            a.bmat = pl.bmat + pl.slot[{&o, li == 0 ? 1 : 2}];
            a.dst[0] = li == 0 ? o.inA.g.p : o.inB.g.p;
            a.acc[0] = li == 0 ? o.accA : o.accB;
            a.mask[0] = li == 0 ? (o.maskA ? o.inA.d.p : nullptr) : (o.maskB ? o.inB.d.p : nullptr);
            a.C1 = CA;
        }
        const int C = o.out.d.C, NS = 1, CO = d1->CO;
        double bb = bytes / nlaunch, ff = flops / nlaunch;
        bool done = false;
        do {
            auto run = [&]() -> bool {
                double bytes = bb, flops = ff;
                PG_DISPATCH(1, 3, 1, 3, "pgconv_dgrad")
                PG_DISPATCH(1, 3, 1, 6, "pgconv_dgrad")
                PG_DISPATCH(1, 6, 1, 6, "pgconv_dgrad")
                PG_DISPATCH(1, 6, 1, 12, "pgconv_dgrad")
                PG_DISPATCH(1, 12, 1, 12, "pgconv_dgrad")
                PG_DISPATCH(1, 12, 1, 6, "pgconv_dgrad")
                PG_DISPATCH(1, 6, 1, 3, "pgconv_dgrad")
                return false;
            };
            done = run();
        } while (0);
        if (!done) return false;
    }
    return true;
}

template <int C, int CO, int G>
static void launch_wg(Model* m, const WgArgs& a, int nblocks, const char* name, double bytes, double flops) {
    LAUNCH(m, name, bytes, flops, hipLaunchKernelGGL((k_pgwgrad<C, CO, G>), dim3(nblocks), dim3(256), 0, m->stream, a));
}

static bool pg_wgrad(Model* m, PgPlan& pl, int B, Op& o, double bytes, double flops) {
    const int CA = o.inA.d.C, CB = o.inB.d.C, CO = o.out.d.C, G = pick_g(CO);
    const int ns = CB ? 2 : 1;
    for (int sidx = 0; sidx < ns; ++sidx) {
        auto it = pl.wslot.find({&o, sidx});
        if (it == pl.wslot.end()) return false;
        const FoldDesc& f = pl.folds[it->second];
        WgArgs a{};
        a.x = sidx == 0 ? o.inA.d.p : o.inB.d.p;
        a.dz = o.out.g.p;
        a.slabs = pl.slabs + f.slab_off;
        a.B = B; a.H = o.out.d.H; a.W = o.out.d.W;
        const int TWp = 16 * G * TX;
        a.tiles_x = (a.W + TWp - 1) / TWp;
        a.tiles_y = (a.H + TH - 1) / TH;
        double bb = bytes / ns, ff = flops / ns;
#define WG_CASE(c, co, g, NAME)                                              \
        if (CA == c && CO == co) { launch_wg<c, co, g>(m, a, f.nslabs, NAME, bb, ff); continue; }
        WG_CASE(1, 3, 4, "pgwgrad_1_3")
        WG_CASE(3, 3, 4, "pgwgrad_3_3")
        WG_CASE(3, 6, 2, "pgwgrad_3_6")
        WG_CASE(6, 6, 2, "pgwgrad_6_6")
        WG_CASE(6, 12, 1, "pgwgrad_6_12")
        WG_CASE(12, 12, 1, "pgwgrad_12_12")
#undef WG_CASE
        return false;
    }
    return true;
}

static bool wgrad_supported(const Op& o) {
    const int CA = o.inA.d.C, CO = o.out.d.C;
    return (CA == 1 && CO == 3) || (CA == 3 && (CO == 3 || CO == 6)) || (CA == 6 && (CO == 6 || CO == 12)) || (CA == 12 && CO == 12);
}

// after the last backward op: fold every conv's slabs into the flat gradient vector (one launch)
int fast_finish_backward(Model* m) {
    if (m->desc.flags & 1) return DNNCA_OK;
    PgPlan& pl = g_plans[m];
    if (!pl.built || pl.folds.empty()) return DNNCA_OK;
    if (!m->dry) HIP_TRY(hipMemsetAsync(pl.dsum, 0, pl.folds.size() * (size_t)pl.dsum_stride * 4, m->stream));
    LAUNCH(m, "pg_slabsum", 0, 0,
           hipLaunchKernelGGL(k_pg_slabsum, dim3((unsigned)pl.folds.size(), pl.max_mt * 4, FOLD_SPLIT), dim3(64), 0,
                              m->stream, pl.folds_dev, pl.slabs, pl.dsum, pl.dsum_stride));
    LAUNCH(m, "pg_fold", 0, 0,
           hipLaunchKernelGGL(k_pg_fold, dim3((unsigned)pl.folds.size(), pl.fold_chunks), dim3(256), 0, m->stream,
                              pl.folds_dev, pl.dsum, pl.dsum_stride, m->g));
    return DNNCA_OK;
}

bool fast_conv_bwd(Model* m, int B, Op& o, double out_bytes, double in_bytes, double flops) {
    if (!conv_supported(m, o)) return false;
    PgPlan& pl = g_plans[m];
    if (o.need_din && pl.slot.find({&o, 1}) == pl.slot.end()) return false;
    // dz = dy * act'(y) in place, then the weight gradient (generic for now) and the MFMA data gradient
    if (o.alpha >= 0.f && !o.premasked)
        LAUNCH(m, "g_act_bwd", 3 * out_bytes, out_bytes / 4,
               g_act_bwd(m->stream, (size_t)B * o.out.d.H * o.out.d.W * o.out.d.C, o.out.g.p, o.out.d.p, o.alpha));
    if (!wgrad_supported(o) || !pg_wgrad(m, pl, B, o, out_bytes + in_bytes, flops)) {
        set_error("internal: pixel-group wgrad has no instance for a conv it planned");
        return false;
    }
    if (o.need_din && !pg_dgrad(m, pl, B, o, out_bytes + in_bytes, flops)) {
        LAUNCH(m, "g_conv_dgrad", out_bytes + in_bytes, flops,
               g_conv_dgrad(m->stream, B, o.out.g, m->p + o.w_off, o.inA.g, o.accA, o.inB.g, o.accB, o.k));
    }
    return true;
}

bool fast_tconv_bwd(Model* m, int B, Op& o, double out_bytes, double in_bytes, double flops) {
    if (!fast_tconv_supported(m, o)) return false;
    LAUNCH(m, "g_tconv_wgrad", out_bytes + in_bytes, flops,
           g_tconv_wgrad(m->stream, B, o.inA.d, o.out.g, m->g + o.w_off, m->g + o.b_off, o.k));
    return fast_tconv_dgrad(m, B, o, out_bytes + in_bytes, flops);
}

static bool dgrad_supported(const Model* m, const Op& o) {
    if (!conv_supported(m, o) || !o.need_din) return false;
    const int CA = o.inA.d.C, CB = o.inB.d.C;
    return pick_g(CA + CB) != 0 || pick_g(CA) != 0;
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
        if (c.type == OP_CONV) can = dgrad_supported(m, c);
        else if (c.type == OP_POOL) can = fast_pool_supported(m, c);
        else if (c.type == OP_TCONV) can = fast_tconv_supported(m, c);
        else if (c.type == OP_HEAD) can = fast_head_supported(m, c);
        if (!can) continue;
        (which ? c.maskB : c.maskA) = true;
        c.mask_alpha = P.alpha;
        P.premasked = true;
    }
}

}  // namespace dnnca
