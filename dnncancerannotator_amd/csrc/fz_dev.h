// fz_dev.h -- device building blocks shared by the block-fused kernels (kernels_fused.hip forward, kernels_fused_bwd.hip backward):
// LDS tile geometry, register-prefetched staging, the pixel-group 3x3 convolution between LDS tiles on the fp32 matrix cores.
#pragma once
#include <hip/hip_runtime.h>

namespace dnnca {
namespace fz {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int up4(int a) { return (a + 3) / 4 * 4; }

// ---- LDS tile geometry ----------------------------------------------------------------------------------------------------
// All tiles of one block level share G = 12/CO (pixels per GEMM row group) and RG (groups per tile row, even): a tile of C
// channels has rows of exactly RG*G pixels = RG*G*C floats, stored back to back behind LEAD floats.  Pixel (row r, column c) sits
// at LEAD + r*RG*G*C + c*C.  With that, group g = r*RG + gc of ANY tile starts at LEAD + g*G*C: a conv's A-operand address and its
// output address are linear in the group index -- no per-row arithmetic, no divisions, straight-line code.  The groups past the
// real width of a row (and past the last row, up to a whole M-tile) are computed too: they read finite junk (the next row's
// pixels) and their outputs land in padding pixels that no real window ever covers.
// LEAD makes the tile's first IMAGE pixel 16-byte aligned (tiles are staged / stored as 16-byte vectors over the aligned superset).
template <int C, int G, int RG, int ROWS, int HALO>
struct Tile {
    static constexpr int LEAD = (4 - (HALO * C) % 4) % 4;
    static constexpr int LS = RG * G * C;                    // floats per row; multiple of 4 (RG is even)
    static constexpr int LS4 = LS / 4;
    static_assert(LS % 4 == 0, "tile rows must be whole float4's");
    // rows + what the last M-tile's junk groups read (16 groups, two rows down, one window) rounded up
    static constexpr int N = up4(LEAD + (ROWS + 2) * LS + 20 * G * C + 16);
    static constexpr int N4 = N / 4;
};

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float act(float v, float alpha) { return alpha < 0.f ? v : (v > 0.f ? v : alpha * v); }


// Staging of ROWS rows of WPX pixels (+ the alignment superset) of a dense NHWC image (C channels, width W) into an LDS tile whose
// pixel (0, 0) is image pixel (y0 - HALO, x0 - HALO), in two halves so that the global loads of the NEXT tile fly while the matrix
// cores work on the current one: issue() loads into registers (unconditional loads from clamped addresses: the issue phase is
// branch-free), commit() writes them to LDS and zeroes what lies outside the image ('same' padding) from the returned bit mask.
template <int C, int G, int RG, int ROWS, int HALO, int WPX, int LSF, int NT>
struct Stager {
    static constexpr int LEAD = (4 - (HALO * C) % 4) % 4;
    static constexpr int W4 = (LEAD + WPX * C + 3) / 4;      // float4's per row that hold real pixels
    static constexpr int LS4 = LSF / 4;
    static_assert(W4 <= LS4 && LSF % 4 == 0, "tile row narrower than the staged width");
    static constexpr int NPF = cdiv(ROWS * W4, NT);
    float4 pre[NPF];
    unsigned ok;
    __device__ __forceinline__ void issue(const float* __restrict__ src, int b, int y0, int x0, int H, int W, int tid) {
        const int rowlen4 = W * C / 4;
        const int f40 = ((x0 - HALO) * C - LEAD) / 4;         // exact: the numerator is a multiple of 4 (may be negative: -4 / 4)
        const float4* base = reinterpret_cast<const float4*>(src) + (size_t)b * H * rowlen4;
        const int last = H * rowlen4 - 1;
        ok = 0;
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int idx = tid + k * NT;
            const int r = idx / W4, c4 = idx - r * W4;
            const int gy = y0 - HALO + r, g4 = f40 + c4;
            ok |= (idx < ROWS * W4 && (unsigned)gy < (unsigned)H && (unsigned)g4 < (unsigned)rowlen4) ? (1u << k) : 0u;
            pre[k] = base[min(max(gy * rowlen4 + g4, 0), last)];
        }
    }
    __device__ __forceinline__ void commit(float* lds, int tid) const {
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int idx = tid + k * NT;
            const int r = idx / W4, c4 = idx - r * W4;
            if (idx < ROWS * W4) reinterpret_cast<float4*>(lds)[r * LS4 + c4] = (ok >> k) & 1u ? pre[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
};

// Store the interior (rows [HALO, HALO + TH), pixels [HALO, HALO + TW)) of an LDS tile to the dense NHWC image.
template <int C, int G, int RG, int ROWS, int HALO, int TW, int TH, int NT>
__device__ __forceinline__ void store_interior(const float* lds, float* __restrict__ dst, int b, int y0, int x0, int H, int W, int tid) {
    using TL = Tile<C, G, RG, ROWS, HALO>;
    constexpr int R4 = TW * C / 4;                       // float4's per tile row (TW * C is a multiple of 4: TW is a multiple of 16)
    constexpr int OFF = TL::LEAD + HALO * C;             // multiple of 4 by construction of LEAD
    static_assert(OFF % 4 == 0 && (TW * C) % 4 == 0, "unaligned tile interior");
    float* base = dst + ((size_t)b * H + y0) * W * C + (size_t)x0 * C;
    for (int idx = tid; idx < TH * R4; idx += NT) {
        const int r = idx / R4, c4 = idx - r * R4;
        const float4 v = *reinterpret_cast<const float4*>(lds + (HALO + r) * TL::LS + OFF + 4 * c4);
        *reinterpret_cast<float4*>(base + (size_t)r * W * C + 4 * c4) = v;
    }
}

// 'same' padding for the NEXT conv: the outermost ring of a conv output tile (OROWS x OW pixels of C channels) is that conv's
// halo; where the ring lies outside the image it must hold zeros, not conv values.  Only border tiles pay for this pass.
// edge: bit 0 top, 1 bottom, 2 left, 3 right ring outside the image (block-uniform).
template <int C, int G, int RG, int OROWS, int OW, int LEAD, int NT>
__device__ __forceinline__ void zero_ring(float* t, unsigned edge, int tid) {
    constexpr int LS = RG * G * C;
    if (edge & 1u) for (int i = tid; i < OW * C; i += NT) t[LEAD + i] = 0.f;
    if (edge & 2u) for (int i = tid; i < OW * C; i += NT) t[LEAD + (OROWS - 1) * LS + i] = 0.f;
    if (edge & 4u) for (int i = tid; i < OROWS * C; i += NT) t[LEAD + (i / C) * LS + (i % C)] = 0.f;
    if (edge & 8u) for (int i = tid; i < OROWS * C; i += NT) t[LEAD + (i / C) * LS + (OW - 1) * C + (i % C)] = 0.f;
}

// One 3x3 'same' convolution (+bias +activation) from LDS tile(s) to an LDS tile on the fp32 matrix cores (pixel-group GEMM of
// kernels_mfma.hip: M = groups of G = 12/CO adjacent pixels, N = (dx, co) = 12, K = (source, dy, window slot)).
//   input  : NSRC tiles (ISTRIDE floats apart) of C channels, row stride RG*G*C; input pixel (r + dy, c + kx) feeds output (r, c)
//   output : groups [0, NMT*16) written at out[OLEAD + g*12 + n] -- the tile of the next conv (row stride RG*12)
// Wave w owns M-tiles w, w + NW, ... (CH of them) and interleaves their MFMA chains: independent accumulators keep the matrix pipe
// issuing back to back and put all their LDS reads in flight together.
// MPR = 0: M-tiles run linearly over the RG-wide rows (junk groups at the row ends and after the last row included);
// MPR > 0: the output is exactly MPR*16 groups wide (the block's last conv: TW/G = 32 groups) -- M-tile mt covers groups
//          [(mt % MPR)*16, +16) of row mt / MPR, no junk, and the output tile's rows are MPR*16 groups long.
template <int C, int NSRC, int CO, int RG, int OROWS, int ILEAD, int ISTRIDE, int OLEAD, int NW, int MPR>
struct Conv3 {
    static constexpr int G = 12 / CO, GC = G * C, WR = (G + 2) * C, SR = (WR + 3) / 4, KS = NSRC * 3 * SR, ILS = RG * GC;
    static constexpr int NMT = MPR ? OROWS * MPR : cdiv(OROWS * RG, 16), CH = cdiv(NMT, NW);
    static_assert(MPR == 0 || (NW % MPR == 0 && NMT % NW == 0), "row-aligned M-tiles must divide evenly over the waves");
    static constexpr int CSTEP = MPR ? (NW / MPR) * RG * GC : NW * 16 * GC;      // A-operand float offset between a wave's chains

    static __device__ __forceinline__ void run(const float* in, float* out, const float* breg, float bias, float alpha, int wave, int lane) {
        const int m = lane & 15, q = lane >> 4, n = m;
        // chain c works on M-tile wave + c*NW; in the last chain the waves past the region redo M-tile NMT-1 and drop the result
        constexpr bool RAGGED = CH * NW > NMT;
        const bool last_ok = !RAGGED || wave + (CH - 1) * NW < NMT;           // wave-uniform
        const int g0 = MPR ? (wave / MPR) * RG + (wave % MPR) * 16 : wave * 16;          // first group of this wave's chain 0
        const float* a0 = in + ILEAD + q + (g0 + m) * GC;
        const float* al = RAGGED ? in + ILEAD + q + ((last_ok ? wave + (CH - 1) * NW : NMT - 1) * 16 + m) * GC : a0 + (CH - 1) * CSTEP;
        f32x4 acc[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the A operands of K-step i + 1 are read while the MFMAs of step i run (two register sets, the order pinned by scheduling
        // barriers): left to itself hipcc reads each operand right in front of its MFMA and waits for it -- an exposed LDS round
        // trip per chain and step (tools/micro/mfma_f32_lds.hip: this form issues at 95 % of the matrix pipe's peak from one wave)
        float av[2][CH];
        auto fetch = [&](int i, float (&a)[CH]) {
            const int s = i / (3 * SR), dy = (i / SR) % 3, k = i % SR;
#pragma unroll
            for (int c = 0; c < CH; ++c) a[c] = c < CH - 1 ? a0[s * ISTRIDE + c * CSTEP + dy * ILS + 4 * k] : al[s * ISTRIDE + dy * ILS + 4 * k];
        };
        fetch(0, av[0]);
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            if (i + 1 < KS) fetch(i + 1, av[(i + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i & 1][c], breg[i], acc[c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (n < 12) {
            float* o0 = out + OLEAD + (wave * 16 + 4 * q) * 12 + n;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (c < CH - 1 || last_ok) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o0[c * (NW * 192) + r * 12] = act(acc[c][r] + bias, alpha);
                }
            }
        }
    }
};

template <int KS>
__device__ __forceinline__ void load_breg(float (&breg)[KS], const float* __restrict__ bmat, int lane) {
#pragma unroll
    for (int s = 0; s < KS; ++s) breg[s] = bmat[s * 64 + lane];
}
// hide the origin of the operand registers from the compiler: otherwise it keeps an s_waitcnt vmcnt(0) for them inside the tile
// loop, which (vmcnt retires in order) would also drain the loads of the tile being staged
template <int KS>
__device__ __forceinline__ void pin_breg(float (&breg)[KS]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(breg[s]));
}

// tile order: blocks b, b + 8, ... share an XCD and its L2 -- give every XCD one contiguous eighth of the tile sequence so that
// the halo rows of vertically adjacent tiles come out of that L2
__device__ __forceinline__ void decode_tile(int t, int ntiles, bool xcd_map, int tiles_x, int tiles_y, int TW, int TH, int& b, int& x0, int& y0) {
    if (xcd_map) t = (t & 7) * (ntiles >> 3) + (t >> 3);
    const int bx = t % tiles_x, by = (t / tiles_x) % tiles_y;
    b = t / (tiles_x * tiles_y);
    x0 = bx * TW;
    y0 = by * TH;
}
__device__ __forceinline__ unsigned tile_edge(int x0, int y0, int TW, int TH, int H, int W) {
    return (y0 == 0 ? 1u : 0u) | (y0 + TH == H ? 2u : 0u) | (x0 == 0 ? 4u : 0u) | (x0 + TW == W ? 8u : 0u);
}

constexpr int even_up(int a) { return (a + 1) / 2 * 2; }
// pixels per row of the low-resolution tile: >= w, rows of whole float4's that also hold the lead
constexpr int low_row_pixels(int w, int c, int lead) {
    int n = w;
    while ((n * c) % 4 != 0 || n * c < lead + w * c) ++n;
    return n;
}

}  // namespace fz
}  // namespace dnnca
