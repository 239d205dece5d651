// strip_dev.h -- "column strip" kernels for the 3-channel, full-resolution level of configs/unet.yaml (device code; included by
// kernels_mfma.hip, which owns the launch plan).
//
// The 512^2 level moves 25 MB per tensor and the step was paying for each of them several times over: conv forward writes the
// feature map, the head reads it and writes its gradient, the conv backward reads that gradient and the conv input again ...
// Every one of those passes ran at the speed of a cold HBM copy (~3-4 TB/s): the step is bound by bytes and launches, not by
// arithmetic.  These kernels remove passes by CHAINING layers inside one wave:
//
//   one wave = a strip of 64 pixel columns (60 owned + 2 halo columns on each side), walked top to bottom over a chunk of rows;
//   a lane owns one pixel column.  Everything a lane needs from its horizontal neighbours comes from the adjacent lanes by DPP
//   wave shifts; vertical neighbours are the rows the wave has just walked over (rotating three-row windows in registers).
//   No LDS tile, no workgroup barrier, no staging pass: the only memory traffic is each lane's own 12 bytes per tensor row,
//   PFD rows in flight, through buffer loads / stores whose out-of-range lanes read zeros / drop the store (zero padding, strip
//   and chunk edges cost no branches, and hipcc counts its s_waitcnt vmcnt(N) exactly: straight-line vector memory traffic).
//   The 81 conv weights are scalar-register operands of the FMAs.
//
// k_tail3: the conv that feeds the annotator head, in a train step -- conv forward (+ activation), the 1x1 head, the weighted BCE
//   (unet.py:241-244, losses.py:17-37), the head's backward and the conv's whole backward (data, weight and bias gradient) in ONE
//   pass: reads the conv input and the labels, writes the gradient of the conv input.  The feature map, the logits and their
//   gradients never exist in memory (4 of 7 full-resolution tensor passes of the two launches it replaces).
#pragma once
#include <type_traits>

namespace dnnca {

typedef float f3 __attribute__((ext_vector_type(3)));

constexpr int STRIP = 60;                       // owned pixel columns per wave
constexpr unsigned STRIP_HALF = 0x40000000u;    // "not there" part of a byte offset: past every buffer (their sizes are below it), and
                                                // the sum of two of them still is (no wrap-around): loads give 0, stores are dropped
constexpr unsigned STRIP_RSRC = 0x00020000u;    // raw buffer, 32-bit data format

// lane l <- lane l - 1 (lane 0 <- 0) / lane l <- lane l + 1 (lane 63 <- 0)
__device__ __forceinline__ float from_left(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float from_right(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}
// one row of a lane's 3 x 3 x 3 window: [left pixel | own pixel | right pixel] x 3 channels
// SHIFT: 0 = DPP wave shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1 -- measured at the price of ~6 FMAs each), 1 = the LDS crossbar
// (ds_bpermute_b32: no LDS memory, runs beside the vector ALU; la / ra = byte addresses of the left / right neighbour lane),
// 2 = none (tuning builds)
template <int SHIFT = 0>
__device__ __forceinline__ void strip_expand(const float (&v)[3], float (&o)[9], int la = 0, int ra = 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (SHIFT == 0) {
            o[c] = from_left(v[c]);
            o[6 + c] = from_right(v[c]);
        } else if (SHIFT == 1) {
            o[c] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(la, __builtin_bit_cast(int, v[c])));
            o[6 + c] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(ra, __builtin_bit_cast(int, v[c])));
        } else {
            o[c] = v[c] * 0.5f;
            o[6 + c] = v[c] * 0.25f;
        }
        o[3 + c] = v[c];
    }
}
__device__ __forceinline__ f3 strip_load3(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __builtin_bit_cast(f3, __builtin_amdgcn_raw_buffer_load_b96(rs, off, 0, 0));
}

struct TailArgs {
    const float* x;          // conv input [B, H, W, 3]
    const float* w;          // conv kernel, HWIO (81 floats), and bias
    const float* bias;
    float alpha;             // slope of the conv's activation (< 0: none)
    const float* hy;         // labels [B, H, W]
    const float* hw;         // head kernel (3) and bias
    const float* hb;
    float* hpartials;        // [gridDim.x][5] block partial sums (head dW (3), db, loss), reduced by k_pg_fold
    double* hscalars;        // scalars[0] = label sum, or, with hlabel_part: written by block 0
    const float* hlabel_part;  // [hlabel_nblk][4] per-block (sum, min, max, -) of the labels from the first encoder block's launch
    int hlabel_nblk;
    dnnca_loss_cfg hcfg;
    double hn_label;
    float hgscale;
    int hmask;               // the head's gradient is multiplied by act'(feature map)
    float halpha;
    float* dx;               // gradient of the conv input [B, H, W, 3]
    int mask;                // ... multiplied by act'(x)
    float mask_alpha;
    float* slabs;            // weight-gradient partial sums [NBUCKET][4*256] (k_pg_fold layout)
    int B, H, W;
    int nstrips, nchunks;    // strips of STRIP columns; row chunks per image
};

template <int PFD, int WSCALAR, bool MIDBAR, int ABL = 0, int SHIFT = 0>      // ABL (tuning builds): bit mask of parts left out
__global__ __launch_bounds__(256, 2) void k_tail3(TailArgs p) {
    static_assert(PFD == 6 || PFD == 3, "the row loop is unrolled lcm(3 window slots, PFD prefetch slots) times");
    constexpr int NACC = 84, NH = 5, NRED = NACC + NH, WRw = 18, MT = 4, NBK = kPgBuckets;      // slab geometry of k_pgbwd<3,1,3>
    __shared__ float red[4 * NRED + 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- this wave's task: blocks b, b + 8, ... share an XCD (and its L2): each XCD gets one contiguous eighth of the task list;
    // consecutive tasks are the adjacent strips of one row chunk (whole image rows are in flight together).  A wave past the end
    // of the list repeats the last task and owns nothing.
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int ntasks = p.B * p.nchunks * p.nstrips;
    const int t0 = __builtin_amdgcn_readfirstlane(bid * 4 + wave);
    const int t = t0 < ntasks ? t0 : ntasks - 1;
    const int strip = t % p.nstrips, ck = (t / p.nstrips) % p.nchunks, b = t / (p.nstrips * p.nchunks);
    const int r0 = (int)((long long)ck * p.H / p.nchunks), r1 = (int)((long long)(ck + 1) * p.H / p.nchunks);
    const int c = strip * STRIP - 2 + lane;
    const bool col_ok = (unsigned)c < (unsigned)p.W;
    const bool lane_own = lane >= 2 && lane < 2 + STRIP && col_ok && t0 < ntasks;
    const unsigned npix = (unsigned)p.B * p.H * p.W;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, npix * 12u, STRIP_RSRC);
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc((void*)p.hy, 0, npix * 4u, STRIP_RSRC);
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)p.dx, 0, npix * 12u, STRIP_RSRC);
    // A byte offset = (per-lane column part) + (per-row part, uniform).  Whatever is not there -- a column outside the image or
    // the strip, a row outside the image or the chunk -- contributes STRIP_HALF instead: the sum then lies past the buffer
    // (buffers are smaller than STRIP_HALF), the load returns zeros and the store is dropped.  No predicates, no branches.
    unsigned colx = col_ok ? (unsigned)c * 12u : STRIP_HALF, coll = col_ok ? (unsigned)c * 4u : STRIP_HALF;
    unsigned cold = lane_own ? (unsigned)c * 12u : STRIP_HALF;
    float colf = col_ok ? 1.0f : 0.f, ownf = lane_own ? 1.0f : 0.f;
    int la = ((lane + 63) & 63) * 4, ra = ((lane + 1) & 63) * 4;      // neighbour lanes (the wrap-around lands in halo lanes)
    asm volatile("" : "+v"(colx), "+v"(coll), "+v"(cold), "+v"(colf), "+v"(ownf), "+v"(la), "+v"(ra));
    const unsigned img0 = (unsigned)b * p.H;                  // first row of this image, counted through the batch
    // (uniform row conditions as bit masks: as `cond ? a : b` hipcc turned them into branches around duplicated loads, with
    //  s_waitcnt vmcnt(0) on the joins)
    auto inside = [](int row, int lo, int hi) -> unsigned {                  // all ones when lo <= row < hi
        return ~(unsigned)(((row - lo) | (hi - 1 - row)) >> 31);
    };
    auto rowpart = [&](int row, unsigned ok, unsigned bytes) -> unsigned {   // uniform
        return ((img0 + (unsigned)row) * (unsigned)p.W * bytes & ok) | (STRIP_HALF & ~ok);
    };
    auto fmask = [](float v, unsigned m) -> float { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & m); };

    // ---- everything the kernel waits for at its start is issued up front: the first rows of the strip, the weights, the label
    // statistics (three dependent memory round trips otherwise)
    f3 xp[PFD];
    unsigned lp[PFD];
#pragma unroll
    for (int k = 0; k < PFD; ++k) {
        const int row = r0 - 2 + k;
        const unsigned ok = inside(row, 0, p.H);
        xp[k] = strip_load3(rsx, colx + rowpart(row, ok, 12u));
        lp[k] = __builtin_amdgcn_raw_buffer_load_b32(rsy, coll + rowpart(row, ok, 4u), 0, 0);
    }
    // the 81 weights: the first kernel row as scalar-register operands, the other two rows in vector registers (all 81 as scalars
    // plus three buffer descriptors overflow the scalar file: 150 spills, a v_readlane in front of every third FMA); the rarely
    // used uniforms live in vector registers as well
    float w[81];
#pragma unroll
    for (int i = 0; i < 81; ++i) w[i] = p.w[i];
    float hwv[3] = {p.hw[0], p.hw[1], p.hw[2]}, hbias = p.hb[0], bs[3] = {p.bias[0], p.bias[1], p.bias[2]};

    // ---- this step's positive-class weight (losses.py:24-29) from the label statistics
    float hwgt;
    {
        double lsum;
        if (p.hlabel_part) {
            double ds = 0.0;
            float mn = INFINITY, mx = -INFINITY;
            for (int i = tid; i < p.hlabel_nblk; i += 256) {
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
            double* redd = reinterpret_cast<double*>(red);
            float* redf = red + 8;
            if (lane == 0) { redd[wave] = ds; redf[wave] = mn; redf[4 + wave] = mx; }
            __syncthreads();
            ds = 0.0; mn = INFINITY; mx = -INFINITY;
            for (int k = 0; k < 4; ++k) { ds += redd[k]; mn = fminf(mn, redf[k]); mx = fmaxf(mx, redf[4 + k]); }
            __syncthreads();
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
    }
    float hslope = p.hmask ? p.halpha : 1.0f, xslope = p.mask ? p.mask_alpha : 1.0f, gsc = p.hgscale;
    float aslope = p.alpha < 0.f ? 1.0f : p.alpha, hwgt1 = hwgt - 1.0f;
    // (nothing pending from here on: hipcc's vmcnt counts inside the row loop are then the exact steady-state ones -- it otherwise
    //  merges the prologue's order of operations in and waits for loads that have just been issued.  As the builtin, so that its
    //  wait counting sees it)
    __builtin_amdgcn_s_waitcnt(0x0070);                           // vmcnt(0) lgkmcnt(0)
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) asm volatile("" : "+v"(hwv[cc]), "+v"(bs[cc]));
    asm volatile("" : "+v"(hbias), "+v"(hwgt1), "+v"(hslope), "+v"(xslope), "+v"(gsc), "+v"(aslope));
#pragma unroll
    for (int i = 0; i < 81; ++i) {
        if (i < WSCALAR) w[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, w[i])));
        else asm volatile("" : "+v"(w[i]));
    }

    float acc[NACC], hsum[NH];    // acc[((wy*3 + wx)*3 + ci)*3 + co] = dW[2-wy][2-wx][ci][co], acc[81 + co] = db[co]; hsum = head dW, db, loss
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
#pragma unroll
    for (int i = 0; i < NH; ++i) hsum[i] = 0.f;
    float xw[3][9], dzw[3][9], lab_prev = 0.f;
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int i = 0; i < 9; ++i) { xw[s][i] = 0.f; dzw[s][i] = 0.f; }

    // One step of the walk.  Step s (u = s mod PFD, compile time) takes x row i = r0 - 2 + s out of the prefetch ring, then runs
    // the conv forward + head of row i - 1 (FWD) and the conv backward of row i - 2 (BWD).  The first steps of a chunk only fill
    // the windows: steps 0, 1 run neither, steps 2, 3 no backward.
    auto step = [&](auto uc, auto fwdc, auto bwdc, int i) {
        constexpr int u = decltype(uc)::value;
        constexpr bool FWD = decltype(fwdc)::value, BWD = decltype(bwdc)::value;
        // ---- x row i (and its labels) out of the prefetch ring; the slot takes row i + PFD
        // (the slot is refilled only after its last reader, and the values leave it through real copies: a value that is still
        //  live when its slot's next load is issued gets a second register, and the copies hipcc then places on the loop's back
        //  edge wait for the loads that have just been issued -- a memory round trip per iteration)
        float xo[3], lab_i;
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) asm volatile("v_mov_b32 %0, %1" : "=v"(xo[c3]) : "v"(xp[u][c3]));
        strip_expand<(ABL & 16) ? 2 : SHIFT>(xo, xw[u % 3], la, ra);                      // window slots: row i-2 -> (u+1)%3, i-1 -> (u+2)%3, i -> u%3
        asm volatile("v_mov_b32 %0, %1" : "=v"(lab_i) : "v"(lp[u]));
        {
            const int row = i + PFD;
            const unsigned ok = inside(row, 0, p.H);
            xp[u] = strip_load3(rsx, colx + rowpart(row, ok, 12u));
            lp[u] = __builtin_amdgcn_raw_buffer_load_b32(rsy, coll + rowpart(row, ok, 4u), 0, 0);
        }
        if constexpr (FWD) {
            // ---- conv forward of row i - 1, activation, head, weighted BCE, head backward -> dz of row i - 1
            // (two partial sums per output channel: six independent FMA chains -- with three, a wave's next FMA waits for the
            //  previous one of its chain: 1.7 instead of 1.1 ns per FMA, tools/micro/valu_dep.hip)
            const int rf = i - 1;
            float f[3] = {bs[0], bs[1], bs[2]}, f2[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < ((ABL & 1) ? 1 : 3); ++dy)
#pragma unroll
                for (int k = 0; k < ((ABL & 1) ? 3 : 9); ++k) {
                    const float xv = xw[(u + 1 + dy) % 3][k];
#pragma unroll
                    for (int co = 0; co < 3; ++co) {
                        if ((dy * 9 + k) & 1) f2[co] = fmaf(xv, w[(dy * 9 + k) * 3 + co], f2[co]);
                        else f[co] = fmaf(xv, w[(dy * 9 + k) * 3 + co], f[co]);
                    }
                }
#pragma unroll
            for (int co = 0; co < 3; ++co) {
                f[co] += f2[co];
                f[co] = fmaxf(f[co], f[co] * aslope);       // aslope in [0, 1] (1: no activation)
            }
            // imgf: a pixel of the image (else: the zero padding of dz); ownm: ... that this wave counts
            const float imgf = fmask(colf, inside(rf, 0, p.H));
            const float ownm = fmask(ownf, inside(rf, r0, r1));
            const float z = lab_prev;
            float xl = hbias;
#pragma unroll
            for (int co = 0; co < 3; ++co) xl = fmaf(f[co], hwv[co], xl);
            const float mk = fmaf(z, hwgt1, 1.0f);
            // e = exp(-|x|) in (0, 1]; sigmoid and log(1 + e) from the hardware exp2 / log2 / rcp (1 ulp each; 1 + e is exact to
            // 6e-8, far below the float32 noise of the 2M-term loss sum)
            const float e = (ABL & 8) ? 0.5f : __builtin_amdgcn_exp2f(-1.44269504f * fabsf(xl));
            const float r1e = (ABL & 8) ? 0.66f : __builtin_amdgcn_rcpf(1.0f + e);
            const float sig = xl >= 0.f ? r1e : e * r1e;
            const float dl = mk * (sig - z) * gsc * imgf;
            const float dlo = dl * ownm;
            hsum[4] = fmaf(fmaxf(xl, 0.f) - xl * z + 0.693147181f * ((ABL & 8) ? 0.58f : __builtin_amdgcn_logf(1.0f + e)), mk * ownm, hsum[4]);
            hsum[3] += dlo;
            float dz[3];
#pragma unroll
            for (int co = 0; co < 3; ++co) {
                hsum[co] = fmaf(f[co], dlo, hsum[co]);
                dz[co] = dl * hwv[co] * (f[co] > 0.f ? 1.0f : hslope);
            }
            if constexpr (MIDBAR) __builtin_amdgcn_sched_barrier(0);      // keep the forward and the backward halves of a step apart
            strip_expand<(ABL & 16) ? 2 : SHIFT>(dz, dzw[u % 3], la, ra);                     // dz window slots: row j-1 -> (u+1)%3, j -> (u+2)%3, j+1 -> u%3
        }
        lab_prev = lab_i;
        if constexpr (BWD) {
            // ---- conv backward of row j = i - 2: data gradient, weight gradient, bias gradient from the dz window
            const int j = i - 2;
            const unsigned rowj = inside(j, r0, r1);          // uniform
            const float ownj = fmask(ownf, rowj);
            const float xc[3] = {xw[(u + 1) % 3][3], xw[(u + 1) % 3][4], xw[(u + 1) % 3][5]};
            const float xv[3] = {xc[0] * ownj, xc[1] * ownj, xc[2] * ownj};
            float dx[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int wy = 0; wy < 3; ++wy)
#pragma unroll
                for (int wx = 0; wx < 3; ++wx)
#pragma unroll
                    for (int co = 0; co < 3; ++co) {
                        const float d = dzw[(u + 1 + wy) % 3][wx * 3 + co];
#pragma unroll
                        for (int ci = 0; ci < 3; ++ci) {
                            if (!(ABL & 2) || (wy == 1 && wx == 1)) dx[ci] = fmaf(d, w[(((2 - wy) * 3 + (2 - wx)) * 3 + ci) * 3 + co], dx[ci]);
                            if (!(ABL & 4) || (wy == 1 && wx == 1))
                                acc[((wy * 3 + wx) * 3 + ci) * 3 + co] = fmaf(xv[ci], d, acc[((wy * 3 + wx) * 3 + ci) * 3 + co]);
                        }
                    }
#pragma unroll
            for (int co = 0; co < 3; ++co) acc[81 + co] = fmaf(dzw[(u + 2) % 3][3 + co], ownj, acc[81 + co]);
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) dx[ci] *= xc[ci] > 0.f ? 1.0f : xslope;
            const f3 dv = {dx[0], dx[1], dx[2]};
            __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b96(rsd, 0, 0, 0)), dv), rsd,
                                                  cold + rowpart(j, rowj, 12u), 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using Yes = std::true_type;
    using No = std::false_type;
    const int nsteps = r1 - r0 + 4;
    int s0 = 0;
    if constexpr (PFD == 3) {
        // the first two groups, specialised: steps 0, 1 fill the x window; 2, 3 add the forward pass; from 4 on the whole step
        step(I0{}, No{}, No{}, r0 - 2);
        step(I1{}, No{}, No{}, r0 - 1);
        step(I2{}, Yes{}, No{}, r0);
        step(I0{}, Yes{}, No{}, r0 + 1);
        step(I1{}, Yes{}, Yes{}, r0 + 2);
        step(I2{}, Yes{}, Yes{}, r0 + 3);
        s0 = 6;
    }
#pragma unroll 1
    for (; s0 < nsteps; s0 += PFD) {
        step(I0{}, Yes{}, Yes{}, r0 - 2 + s0);
        step(I1{}, Yes{}, Yes{}, r0 - 1 + s0);
        step(I2{}, Yes{}, Yes{}, r0 + s0);
        if constexpr (PFD == 6) {
            step(std::integral_constant<int, 3>{}, Yes{}, Yes{}, r0 + 1 + s0);
            step(std::integral_constant<int, 4>{}, Yes{}, Yes{}, r0 + 2 + s0);
            step(std::integral_constant<int, 5>{}, Yes{}, Yes{}, r0 + 3 + s0);
        }
    }
    // ---- lane sums by DPP, the four waves through LDS; weight gradient: one atomic per element into slab (blockIdx % NBUCKET), laid
    // out where k_pg_fold expects D[(dy, j), (dx, co)] (everything in the dx = 0 entries, the bias in the all-ones row); head: one row
    // of the partials table per block
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        const float s = wave_total_l63(acc[i]);
        if (lane == 63) red[wave * NRED + i] = s;
    }
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const float s = wave_total_l63(hsum[i]);
        if (lane == 63) red[wave * NRED + NACC + i] = s;
    }
    __syncthreads();
    if (tid < NRED) {
        const int e = tid;
        const float v = (red[e] + red[NRED + e]) + (red[2 * NRED + e] + red[3 * NRED + e]);
        if (e < NACC) {
            int mrow, co;
            if (e < 81) {
                const int tap = e / 9, ci = (e / 3) % 3;
                co = e % 3;
                mrow = (2 - tap / 3) * WRw + (2 - tap % 3) * 3 + ci;
            } else {
                mrow = 3 * WRw;
                co = e - 81;
            }
            atomicAdd(p.slabs + (size_t)(blockIdx.x % NBK) * (MT * 256) + slab_index(mrow, co), v);
        } else {
            p.hpartials[blockIdx.x * NH + (e - NACC)] = v;
        }
    }
}

// k_first3: the backward of the first encoder block in one pass -- the second conv's whole backward with the max-pool's backward
// folded into its staging (dz = (skip gradient + pooled gradient at the recorded window position) * act'(y), components.py:54) AND the
// first conv's weight / bias gradient (its input gradient is needed by nobody: it reads the network input).  The gradient of the
// first conv's output never exists in memory: 25 MB less written, 25 MB less read and one launch less than k_bwd3v<PF> followed by the
// weight-gradient pass of the 1 -> 3 channel conv.  Reads per row and lane: skip gradient, conv output, pooled gradient + position,
// conv input (12 bytes each), network input (4 bytes); writes nothing but the two sets of weight-gradient slabs.
struct FirstArgs {
    const float* dskip;      // gradient of the second conv's output from the skip connection [B, H, W, 3]
    const float* y1;         // the second conv's output [B, H, W, 3]
    const float* dpool;      // gradient of the pooled tensor [B, H/2, W/2, 3]
    const unsigned char* idx;  // window position of every pooled maximum [B, H/2, W/2, 3]
    const float* x1;         // the second conv's input = the first conv's output [B, H, W, 3]
    const float* xin;        // network input [B, H, W, 1]
    const float* w;          // the second conv's kernel, HWIO (81 floats)
    float pf_alpha;          // slope of act'(y1)
    int mask;                // the first conv's activation: the gradient of its output is multiplied by act'(x1)
    float mask_alpha;
    float* slabs1;           // weight-gradient slabs of the second conv [NBUCKET][4*256] (k_pgbwd<3,1,3> geometry)
    float* slabs0;           // ... of the first conv [NBUCKET][2*256] (k_pgbwd<1,1,3> geometry: window row of 6 floats)
    int B, H, W;
    int nstrips, nchunks;
};

template <int PFD, int WSCALAR>
__global__ __launch_bounds__(256, 2) void k_first3(FirstArgs p) {
    static_assert(PFD == 3 || PFD == 2, "ring slots; the row loop is unrolled 6 times (3 window slots x 2)");
    constexpr int NA1 = 84, NA0 = 30, NRED = NA1 + NA0, NBK = kPgBuckets;
    __shared__ float red[4 * NRED];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int ntasks = p.B * p.nchunks * p.nstrips;
    const int t0 = __builtin_amdgcn_readfirstlane(bid * 4 + wave);
    const int t = t0 < ntasks ? t0 : ntasks - 1;
    const int strip = t % p.nstrips, ck = (t / p.nstrips) % p.nchunks, b = t / (p.nstrips * p.nchunks);
    const int r0 = (int)((long long)ck * p.H / p.nchunks), r1 = (int)((long long)(ck + 1) * p.H / p.nchunks);
    const int c = strip * STRIP - 2 + lane;
    const bool col_ok = (unsigned)c < (unsigned)p.W;
    const bool lane_own = lane >= 2 && lane < 2 + STRIP && col_ok && t0 < ntasks;
    const unsigned npix = (unsigned)p.B * p.H * p.W, Hp = p.H >> 1, Wp = p.W >> 1, npool = (unsigned)p.B * Hp * Wp;
    const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc((void*)p.dskip, 0, npix * 12u, STRIP_RSRC);
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc((void*)p.y1, 0, npix * 12u, STRIP_RSRC);
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, npix * 12u, STRIP_RSRC);
    const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc((void*)p.xin, 0, npix * 4u, STRIP_RSRC);
    const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc((void*)p.dpool, 0, npool * 12u, STRIP_RSRC);
    const __amdgpu_buffer_rsrc_t rsk = __builtin_amdgcn_make_buffer_rsrc((void*)p.idx, 0, npool * 3u, STRIP_RSRC);
    // byte offset = column part (per lane) + row part (uniform); what is not there contributes STRIP_HALF (see k_tail3)
    unsigned col12 = col_ok ? (unsigned)c * 12u : STRIP_HALF, col4 = col_ok ? (unsigned)c * 4u : STRIP_HALF;
    unsigned colp12 = col_ok ? (unsigned)(c >> 1) * 12u : STRIP_HALF, colp3 = col_ok ? (unsigned)(c >> 1) * 3u : STRIP_HALF;
    float ownf = lane_own ? 1.0f : 0.f;
    int la = ((lane + 63) & 63) * 4, ra = ((lane + 1) & 63) * 4, cpar = c & 1;
    asm volatile("" : "+v"(col12), "+v"(col4), "+v"(colp12), "+v"(colp3), "+v"(ownf), "+v"(la), "+v"(ra), "+v"(cpar));
    const unsigned img0 = (unsigned)b * p.H, imgp0 = (unsigned)b * Hp;
    auto inside = [](int row, int lo, int hi) -> unsigned { return ~(unsigned)(((row - lo) | (hi - 1 - row)) >> 31); };
    auto rowpart = [&](unsigned row0, int row, unsigned ok, unsigned rowbytes) -> unsigned {   // uniform
        return ((row0 + (unsigned)row) * rowbytes & ok) | (STRIP_HALF & ~ok);
    };
    auto fmask = [](float v, unsigned m) -> float { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & m); };

    // ring slot k: what step s (s mod PFD == k) consumes -- rows i = r0 - 1 + s of dskip / y1 / dpool / idx / xin, row i - 1 of x1
    f3 gq[PFD], yq[PFD], pq[PFD], xq[PFD];
    unsigned kq[PFD][3], iq[PFD];
    auto issue = [&](auto kc, int i) {
        constexpr int k = decltype(kc)::value;
        const unsigned ok = inside(i, 0, p.H), okx = inside(i - 1, 0, p.H);
        const unsigned r12 = rowpart(img0, i, ok, (unsigned)p.W * 12u);
        gq[k] = strip_load3(rsg, col12 + r12);
        yq[k] = strip_load3(rsy, col12 + r12);
        xq[k] = strip_load3(rsx, col12 + rowpart(img0, i - 1, okx, (unsigned)p.W * 12u));
        iq[k] = __builtin_amdgcn_raw_buffer_load_b32(rsi, col4 + rowpart(img0, i, ok, (unsigned)p.W * 4u), 0, 0);
        pq[k] = strip_load3(rsp, colp12 + rowpart(imgp0, i >> 1, ok, Wp * 12u));
        const unsigned o3 = colp3 + rowpart(imgp0, i >> 1, ok, Wp * 3u);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) kq[k][ch] = __builtin_amdgcn_raw_buffer_load_b8(rsk, o3 + ch, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    issue(I0{}, r0 - 1);
    issue(I1{}, r0);
    if constexpr (PFD == 3) issue(I2{}, r0 + 1);
    float w[81];
#pragma unroll
    for (int i = 0; i < 81; ++i) w[i] = p.w[i];
    float yslope = p.pf_alpha, xslope = p.mask ? p.mask_alpha : 1.0f;
    __builtin_amdgcn_s_waitcnt(0x0070);                           // vmcnt(0) lgkmcnt(0): nothing pending at the loop's entry (see k_tail3)
    asm volatile("" : "+v"(yslope), "+v"(xslope));
#pragma unroll
    for (int i = 0; i < 81; ++i) {
        if (i < WSCALAR) w[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, w[i])));
        else asm volatile("" : "+v"(w[i]));
    }
    float acc[NA1], acc0[NA0];    // acc as in k_tail3; acc0[(dy*3 + kx)*3 + co] = dW0[dy][kx][0][co], acc0[27 + co] = db0[co]
#pragma unroll
    for (int i = 0; i < NA1; ++i) acc[i] = 0.f;
#pragma unroll
    for (int i = 0; i < NA0; ++i) acc0[i] = 0.f;
    float dzw[3][9], iw[3][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
        for (int i = 0; i < 9; ++i) dzw[s][i] = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) iw[s][i] = 0.f;
    }
    // step s (u = s mod 6): rows i = r0 - 1 + s arrive -> dz of row i, network-input row i into the windows (slot u % 3);
    // backward of row j = i - 1 (BWD: not in the first two steps of a chunk)
    auto step = [&](auto uc, auto bwdc, int i) {
        constexpr int u = decltype(uc)::value, k = u % PFD;
        constexpr bool BWD = decltype(bwdc)::value;
        float g[3], y[3], dp[3], xc[3], xi;
        unsigned kk[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {      // real copies out of the ring (see k_tail3)
            asm volatile("v_mov_b32 %0, %1" : "=v"(g[ch]) : "v"(gq[k][ch]));
            asm volatile("v_mov_b32 %0, %1" : "=v"(y[ch]) : "v"(yq[k][ch]));
            asm volatile("v_mov_b32 %0, %1" : "=v"(dp[ch]) : "v"(pq[k][ch]));
            asm volatile("v_mov_b32 %0, %1" : "=v"(xc[ch]) : "v"(xq[k][ch]));
            asm volatile("v_mov_b32 %0, %1" : "=v"(kk[ch]) : "v"(kq[k][ch]));
        }
        asm volatile("v_mov_b32 %0, %1" : "=v"(xi) : "v"(iq[k]));
        issue(std::integral_constant<int, k>{}, i + PFD);
        // dz of row i: (skip gradient + pooled gradient where this pixel was the window's first maximum) * act'(y)
        const int pos = ((i & 1) << 1) | cpar;
        float dz[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) dz[ch] = (g[ch] + ((int)kk[ch] == pos ? dp[ch] : 0.f)) * (y[ch] > 0.f ? 1.0f : yslope);
        strip_expand<1>(dz, dzw[u % 3], la, ra);              // rows j-1 -> (u+1)%3, j -> (u+2)%3, j+1 -> u%3
        iw[u % 3][0] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(la, __builtin_bit_cast(int, xi)));
        iw[u % 3][1] = xi;
        iw[u % 3][2] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(ra, __builtin_bit_cast(int, xi)));
        if constexpr (BWD) {
            const int j = i - 1;
            const float ownj = fmask(ownf, inside(j, r0, r1));
            const float xv[3] = {xc[0] * ownj, xc[1] * ownj, xc[2] * ownj};
            float dx[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int wy = 0; wy < 3; ++wy)
#pragma unroll
                for (int wx = 0; wx < 3; ++wx)
#pragma unroll
                    for (int co = 0; co < 3; ++co) {
                        const float d = dzw[(u + 1 + wy) % 3][wx * 3 + co];
#pragma unroll
                        for (int ci = 0; ci < 3; ++ci) {
                            dx[ci] = fmaf(d, w[(((2 - wy) * 3 + (2 - wx)) * 3 + ci) * 3 + co], dx[ci]);
                            acc[((wy * 3 + wx) * 3 + ci) * 3 + co] = fmaf(xv[ci], d, acc[((wy * 3 + wx) * 3 + ci) * 3 + co]);
                        }
                    }
#pragma unroll
            for (int co = 0; co < 3; ++co) acc[81 + co] = fmaf(dzw[(u + 2) % 3][3 + co], ownj, acc[81 + co]);
            // gradient of the first conv's pre-activation output at this pixel, and its weight / bias gradient from the input window
            float d0[3];
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) d0[ci] = dx[ci] * (xc[ci] > 0.f ? 1.0f : xslope) * ownj;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int co = 0; co < 3; ++co)
                        acc0[(dy * 3 + kx) * 3 + co] = fmaf(iw[(u + 1 + dy) % 3][kx], d0[co], acc0[(dy * 3 + kx) * 3 + co]);
#pragma unroll
            for (int co = 0; co < 3; ++co) acc0[27 + co] += d0[co];
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    using Yes = std::true_type;
    using No = std::false_type;
    const int nsteps = r1 - r0 + 2;
    step(I0{}, No{}, r0 - 1);
    step(I1{}, No{}, r0);                     // (its row j = r0 - 1 belongs to the chunk above)
    step(I2{}, Yes{}, r0 + 1);
    step(std::integral_constant<int, 3>{}, Yes{}, r0 + 2);
    step(std::integral_constant<int, 4>{}, Yes{}, r0 + 3);
    step(std::integral_constant<int, 5>{}, Yes{}, r0 + 4);
#pragma unroll 1
    for (int s0 = 6; s0 < nsteps; s0 += 6) {
        step(I0{}, Yes{}, r0 - 1 + s0);
        step(I1{}, Yes{}, r0 + s0);
        step(I2{}, Yes{}, r0 + 1 + s0);
        step(std::integral_constant<int, 3>{}, Yes{}, r0 + 2 + s0);
        step(std::integral_constant<int, 4>{}, Yes{}, r0 + 3 + s0);
        step(std::integral_constant<int, 5>{}, Yes{}, r0 + 4 + s0);
    }
    // ---- lane sums by DPP, the four waves through LDS, one atomic per element into the two slab sets (k_pg_fold layouts)
#pragma unroll
    for (int i = 0; i < NA1; ++i) {
        const float s = wave_total_l63(acc[i]);
        if (lane == 63) red[wave * NRED + i] = s;
    }
#pragma unroll
    for (int i = 0; i < NA0; ++i) {
        const float s = wave_total_l63(acc0[i]);
        if (lane == 63) red[wave * NRED + NA1 + i] = s;
    }
    __syncthreads();
    if (tid < NRED) {
        const int e = tid;
        const float v = (red[e] + red[NRED + e]) + (red[2 * NRED + e] + red[3 * NRED + e]);
        if (e < NA1) {
            int mrow, co;
            if (e < 81) {
                const int tap = e / 9, ci = (e / 3) % 3;
                co = e % 3;
                mrow = (2 - tap / 3) * 18 + (2 - tap % 3) * 3 + ci;
            } else {
                mrow = 3 * 18;
                co = e - 81;
            }
            atomicAdd(p.slabs1 + (size_t)(blockIdx.x % NBK) * (4 * 256) + slab_index(mrow, co), v);
        } else {
            const int q = e - NA1;                        // (dy*3 + kx)*3 + co, or 27 + co
            const int mrow = q < 27 ? (q / 9) * 6 + (q / 3) % 3 : 18, co = q % 3;
            atomicAdd(p.slabs0 + (size_t)(blockIdx.x % NBK) * (2 * 256) + slab_index(mrow, co), v);
        }
    }
}

// k_first3_fwd: the forward pass of the first encoder block (components.py:46-61,77-81: conv 1 -> 3, conv 3 -> 3, 2x2 max-pool) in one
// column-strip pass, with the label statistics of a train step (utils/losses.py:87-102) riding along: reads the network input (and
// the labels), writes the two conv outputs, the pooled tensor and the window position of every pooled maximum.  Chunks start on even
// rows, and strips on even columns, so that a 2x2 pooling window never straddles two waves: its two rows are consecutive steps of one
// lane pair (the partner's values come through a quad-permute DPP).
struct FirstFwdArgs {
    const float* xin;        // network input [B, H, W, 1]
    const float* w0;         // first conv: kernel HWIO [3][3][1][3] and bias
    const float* b0;
    const float* w1;         // second conv: kernel HWIO (81) and bias
    const float* b1;
    float alpha0, alpha1;    // activation slopes (< 0: none)
    float* y0;               // first conv's output [B, H, W, 3] or nullptr (no backward pass follows)
    float* y1;               // second conv's output
    float* pool;             // [B, H/2, W/2, 3]
    unsigned char* pool_idx; // [B, H/2, W/2, 3] window position (0..3, row-major) of each pooled value's FIRST maximum, or nullptr
    const float* labels;     // [B, H, W] or nullptr
    float* label_part;       // [nstrip_blocks][4] per-block (sum, min, max, -)
    int B, H, W;
    int nstrips, nchunks;
    int nstrip_blocks;       // blocks [0, nstrip_blocks) walk strips; the blocks behind them run the step's operand preparation
    PrepRide prep;           // (k_pg_prep's job: the first launch of a train step takes it along; prep.nblocks == 0: nothing rides)
};

template <int WS0, int WS1, int WPS = 2>      // how many of the 27 / 81 kernel weights are scalar-register operands (the rest: vector registers); waves per SIMD
__global__ __launch_bounds__(256, WPS) void k_first3_fwd(FirstFwdArgs p) {
    constexpr int PFD = 6;           // rows in flight per lane (8 bytes each: the steps are short, three rows ahead did not cover the latency)
    __shared__ float red[32];
    if ((int)blockIdx.x >= p.nstrip_blocks) {
        pg_prep_body(p.prep.index, p.prep.params, p.prep.bmat, p.prep.n, p.prep.nprep, p.prep.z, (int)blockIdx.x - p.nstrip_blocks, p.prep.nblocks);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = blockIdx.x;
    const int nblk = p.nstrip_blocks;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int ntasks = p.B * p.nchunks * p.nstrips;
    const int t0 = __builtin_amdgcn_readfirstlane(bid * 4 + wave);
    const int t = t0 < ntasks ? t0 : ntasks - 1;
    const int strip = t % p.nstrips, ck = (t / p.nstrips) % p.nchunks, b = t / (p.nstrips * p.nchunks);
    const int Hp = p.H >> 1, Wp = p.W >> 1;
    const int r0 = 2 * (int)((long long)ck * Hp / p.nchunks), r1 = 2 * (int)((long long)(ck + 1) * Hp / p.nchunks);     // even rows
    const int c = strip * STRIP - 2 + lane;
    const bool col_ok = (unsigned)c < (unsigned)p.W;
    const bool lane_own = lane >= 2 && lane < 2 + STRIP && col_ok && t0 < ntasks;
    const unsigned npix = (unsigned)p.B * p.H * p.W, npool = (unsigned)p.B * Hp * Wp;
    const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc((void*)p.xin, 0, npix * 4u, STRIP_RSRC);
    const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc((void*)p.labels, 0, p.labels ? npix * 4u : 0u, STRIP_RSRC);
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.y0, 0, p.y0 ? npix * 12u : 0u, STRIP_RSRC);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.y1, 0, npix * 12u, STRIP_RSRC);
    const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc((void*)p.pool, 0, npool * 12u, STRIP_RSRC);
    const __amdgpu_buffer_rsrc_t rsk = __builtin_amdgcn_make_buffer_rsrc((void*)p.pool_idx, 0, p.pool_idx ? npool * 3u : 0u, STRIP_RSRC);
    unsigned col4 = col_ok ? (unsigned)c * 4u : STRIP_HALF, own12 = lane_own ? (unsigned)c * 12u : STRIP_HALF;
    // pooled tensor: the even lane of a pair stores the window's maximum
    const bool pool_lane = lane_own && !(lane & 1);
    unsigned ownp12 = pool_lane ? (unsigned)(c >> 1) * 12u : STRIP_HALF, ownp3 = pool_lane ? (unsigned)(c >> 1) * 3u : STRIP_HALF;
    float colf = col_ok ? 1.0f : 0.f, ownf = lane_own ? 1.0f : 0.f;
    int la = ((lane + 63) & 63) * 4, ra = ((lane + 1) & 63) * 4;
    asm volatile("" : "+v"(col4), "+v"(own12), "+v"(ownp12), "+v"(ownp3), "+v"(colf), "+v"(ownf), "+v"(la), "+v"(ra));
    const unsigned img0 = (unsigned)b * p.H, imgp0 = (unsigned)b * Hp;
    auto inside = [](int row, int lo, int hi) -> unsigned { return ~(unsigned)(((row - lo) | (hi - 1 - row)) >> 31); };
    auto rowpart = [&](unsigned row0, int row, unsigned ok, unsigned rowbytes) -> unsigned {   // uniform
        return ((row0 + (unsigned)row) * rowbytes & ok) | (STRIP_HALF & ~ok);
    };
    auto fmask = [](float v, unsigned m) -> float { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & m); };

    unsigned xq[PFD], lq[PFD];       // ring: network-input row i and label row i of the step that consumes the slot
    auto issue = [&](auto kc, int i) {
        constexpr int k = decltype(kc)::value;
        const unsigned off = col4 + rowpart(img0, i, inside(i, 0, p.H), (unsigned)p.W * 4u);
        xq[k] = __builtin_amdgcn_raw_buffer_load_b32(rsi, off, 0, 0);
        lq[k] = __builtin_amdgcn_raw_buffer_load_b32(rsl, off, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;
    using I5 = std::integral_constant<int, 5>;
    issue(I0{}, r0 - 2);
    issue(I1{}, r0 - 1);
    issue(I2{}, r0);
    issue(I3{}, r0 + 1);
    issue(I4{}, r0 + 2);
    issue(I5{}, r0 + 3);
    float w0[27], w1[81];
#pragma unroll
    for (int i = 0; i < 27; ++i) w0[i] = p.w0[i];
#pragma unroll
    for (int i = 0; i < 81; ++i) w1[i] = p.w1[i];
    float b0[3] = {p.b0[0], p.b0[1], p.b0[2]}, b1[3] = {p.b1[0], p.b1[1], p.b1[2]};
    float a0 = p.alpha0 < 0.f ? 1.0f : p.alpha0, a1 = p.alpha1 < 0.f ? 1.0f : p.alpha1;
    __builtin_amdgcn_s_waitcnt(0x0070);                           // vmcnt(0) lgkmcnt(0): nothing pending at the loop's entry (see k_tail3)
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) asm volatile("" : "+v"(b0[cc]), "+v"(b1[cc]));
    asm volatile("" : "+v"(a0), "+v"(a1));
#pragma unroll
    for (int i = 0; i < 27; ++i) {
        if (i < WS0) w0[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, w0[i])));
        else asm volatile("" : "+v"(w0[i]));
    }
#pragma unroll
    for (int i = 0; i < 81; ++i) {
        if (i < WS1) w1[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, w1[i])));
        else asm volatile("" : "+v"(w1[i]));
    }
    float iw[3][3], yw[3][9], prev[3] = {0.f, 0.f, 0.f};
    float lsum = 0.f, lmin = INFINITY, lmax = -INFINITY;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
        for (int i = 0; i < 3; ++i) iw[s][i] = 0.f;
#pragma unroll
        for (int i = 0; i < 9; ++i) yw[s][i] = 0.f;
    }
    // step s (u = s mod 6; window slot u mod 3): input row i = r0 - 2 + s arrives; C0: first conv of row i - 1; C1: second conv of row i - 2 (+ pool)
    auto step = [&](auto uc, auto c0c, auto c1c, int i) {
        constexpr int uu = decltype(uc)::value, u = uu % 3;
        constexpr bool C0 = decltype(c0c)::value, C1 = decltype(c1c)::value;
        float xi, lab;
        asm volatile("v_mov_b32 %0, %1" : "=v"(xi) : "v"(xq[uu]));      // real copies out of the ring (see k_tail3)
        asm volatile("v_mov_b32 %0, %1" : "=v"(lab) : "v"(lq[uu]));
        issue(std::integral_constant<int, uu>{}, i + PFD);
        iw[u][0] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(la, __builtin_bit_cast(int, xi)));
        iw[u][1] = xi;
        iw[u][2] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(ra, __builtin_bit_cast(int, xi)));
        if (p.labels) {                // uniform; a pixel's label is counted by the wave that owns the pixel
            const unsigned mine = inside(i, r0, r1);
            const float lo = fmask(ownf, mine);
            lsum = fmaf(lab, lo, lsum);
            lmin = fminf(lmin, lo > 0.f ? lab : INFINITY);
            lmax = fmaxf(lmax, lo > 0.f ? lab : -INFINITY);
        }
        if constexpr (C0) {
            // ---- first conv of row i - 1 (input rows i-2 .. i = window slots (u+1)%3, (u+2)%3, u); zero outside the image (it is
            // the second conv's zero padding there)
            const int rf = i - 1;
            float f[3] = {b0[0], b0[1], b0[2]};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int co = 0; co < 3; ++co) f[co] = fmaf(iw[(u + 1 + dy) % 3][kx], w0[(dy * 3 + kx) * 3 + co], f[co]);
            const float imgf = fmask(colf, inside(rf, 0, p.H));
#pragma unroll
            for (int co = 0; co < 3; ++co) f[co] = fmaxf(f[co], f[co] * a0) * imgf;
            const unsigned mine = inside(rf, r0, r1);
            __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b96(rs0, 0, 0, 0)), f3{f[0], f[1], f[2]}),
                                                  rs0, own12 + rowpart(img0, rf, mine, (unsigned)p.W * 12u), 0, 0);
            strip_expand<1>(f, yw[u], la, ra);            // y0 window slots: row i-3 -> (u+1)%3, i-2 -> (u+2)%3, i-1 -> u
        }
        if constexpr (C1) {
            // ---- second conv of row j = i - 2 (y0 rows i-3 .. i-1), activation, store; 2x2 max-pool over the row pair (j - 1, j)
            const int j = i - 2;
            float g[3] = {b1[0], b1[1], b1[2]}, g2[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const float xv = yw[(u + 1 + dy) % 3][k];
#pragma unroll
                    for (int co = 0; co < 3; ++co) {
                        if ((dy * 9 + k) & 1) g2[co] = fmaf(xv, w1[(dy * 9 + k) * 3 + co], g2[co]);
                        else g[co] = fmaf(xv, w1[(dy * 9 + k) * 3 + co], g[co]);
                    }
                }
#pragma unroll
            for (int co = 0; co < 3; ++co) {
                g[co] += g2[co];
                g[co] = fmaxf(g[co], g[co] * a1);
            }
            const unsigned mine = inside(j, r0, r1);
            __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b96(rs1, 0, 0, 0)), f3{g[0], g[1], g[2]}),
                                                  rs1, own12 + rowpart(img0, j, mine, (unsigned)p.W * 12u), 0, 0);
            // pool: on odd rows the window (rows j-1, j; this lane's column and its pair partner's) is complete.  First maximum in
            // row-major order: (j-1, even) = 0, (j-1, odd) = 1, (j, even) = 2, (j, odd) = 3.  (On even rows the values are computed and
            // dropped: the store offset is out of range.)
            float pm[3];
            unsigned pos[3];
#pragma unroll
            for (int co = 0; co < 3; ++co) {
                const float p01 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, prev[co]), 0xB1, 0xf, 0xf, false));
                const float p11 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, g[co]), 0xB1, 0xf, 0xf, false));
                const float m = fmaxf(fmaxf(prev[co], p01), fmaxf(g[co], p11));
                pm[co] = m;
                pos[co] = prev[co] == m ? 0u : (p01 == m ? 1u : (g[co] == m ? 2u : 3u));
                prev[co] = g[co];
            }
            const unsigned podd = mine & (unsigned)-(j & 1);          // an odd row of this chunk
            __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b96(rsp, 0, 0, 0)), f3{pm[0], pm[1], pm[2]}),
                                                  rsp, ownp12 + rowpart(imgp0, j >> 1, podd, (unsigned)Wp * 12u), 0, 0);
            const unsigned o3 = ownp3 + rowpart(imgp0, j >> 1, podd, (unsigned)Wp * 3u);
#pragma unroll
            for (int co = 0; co < 3; ++co) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)pos[co], rsk, o3 + co, 0, 0);
        }
        if constexpr (WPS == 2) __builtin_amdgcn_sched_barrier(0);
    };
    using Yes = std::true_type;
    using No = std::false_type;
    const int nsteps = r1 - r0 + 4;
    step(I0{}, No{}, No{}, r0 - 2);
    step(I1{}, No{}, No{}, r0 - 1);
    step(I2{}, Yes{}, No{}, r0);
    step(I3{}, Yes{}, No{}, r0 + 1);
    step(I4{}, Yes{}, Yes{}, r0 + 2);
    step(I5{}, Yes{}, Yes{}, r0 + 3);
#pragma unroll 1
    for (int s0 = 6; s0 < nsteps; s0 += 6) {
        step(I0{}, Yes{}, Yes{}, r0 - 2 + s0);
        step(I1{}, Yes{}, Yes{}, r0 - 1 + s0);
        step(I2{}, Yes{}, Yes{}, r0 + s0);
        step(I3{}, Yes{}, Yes{}, r0 + 1 + s0);
        step(I4{}, Yes{}, Yes{}, r0 + 2 + s0);
        step(I5{}, Yes{}, Yes{}, r0 + 3 + s0);
    }
    if (p.labels) {           // block partials of the label statistics: wave shuffles, the waves through LDS, one table row
        for (int o = 32; o > 0; o >>= 1) {
            lsum += __shfl_down(lsum, o, 64);
            lmin = fminf(lmin, __shfl_down(lmin, o, 64));
            lmax = fmaxf(lmax, __shfl_down(lmax, o, 64));
        }
        if (lane == 0) { red[wave] = lsum; red[4 + wave] = lmin; red[8 + wave] = lmax; }
        __syncthreads();
        if (tid == 0)
            reinterpret_cast<float4*>(p.label_part)[blockIdx.x] =
                make_float4((red[0] + red[1]) + (red[2] + red[3]), fminf(fminf(red[4], red[5]), fminf(red[6], red[7])),
                            fmaxf(fmaxf(red[8], red[9]), fmaxf(red[10], red[11])), 0.f);
    }
}

// k_up3_fwd: the first two layers of the last decoder block's forward pass (components.py:118-127,158-166: Conv2DTranspose 6 -> 3,
// 2x2 stride 2; concat [up, skip]; conv 6 -> 3) in one column-strip pass: a lane computes the transposed conv's output pixel itself
// (one input pixel of the half-resolution tensor, the kernel slice of its row / column parity), writes it (the backward pass reads it)
// and feeds it, beside the skip tensor's pixel, into the two three-row windows of the 3x3 conv.  Chunks start on even rows and strips
// on even columns, so a row's parity is the step's parity (compile time) and a lane's column parity is its lane parity.
struct UpFwdArgs {
    const float* in;         // half-resolution input [B, H/2, W/2, 6]
    const float* skip;       // skip tensor [B, H, W, 3]
    const float* wt;         // transposed conv: kernel [2][2][3][6] (a, e, co, ci) and bias (3)
    const float* bt;
    const float* w;          // conv: kernel HWIO [3][3][6][3] (input channels: 3 up, 3 skip) and bias
    const float* b;
    float alpha;             // the conv's activation slope (< 0: none)
    float* tout;             // transposed conv's output [B, H, W, 3]
    float* out;              // conv's output [B, H, W, 3]
    int B, H, W;
    int nstrips, nchunks;
};

template <int PFD, int WS>      // WS of the conv's 162 weights are scalar-register operands, the rest sit in vector registers
__global__ __launch_bounds__(256, 2) void k_up3_fwd(UpFwdArgs p) {
    static_assert(PFD == 2 || PFD == 3, "ring slots; the row loop is unrolled 6 times");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int ntasks = p.B * p.nchunks * p.nstrips;
    const int t0 = __builtin_amdgcn_readfirstlane(bid * 4 + wave);
    const int t = t0 < ntasks ? t0 : ntasks - 1;
    const int strip = t % p.nstrips, ck = (t / p.nstrips) % p.nchunks, b = t / (p.nstrips * p.nchunks);
    const int Hp = p.H >> 1, Wp = p.W >> 1;
    const int r0 = 2 * (int)((long long)ck * Hp / p.nchunks), r1 = 2 * (int)((long long)(ck + 1) * Hp / p.nchunks);     // even rows
    const int c = strip * STRIP - 2 + lane;
    const bool col_ok = (unsigned)c < (unsigned)p.W;
    const bool lane_own = lane >= 2 && lane < 2 + STRIP && col_ok && t0 < ntasks;
    const unsigned npix = (unsigned)p.B * p.H * p.W, npool = (unsigned)p.B * Hp * Wp;
    const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, npool * 24u, STRIP_RSRC);
    const __amdgpu_buffer_rsrc_t rss = __builtin_amdgcn_make_buffer_rsrc((void*)p.skip, 0, npix * 12u, STRIP_RSRC);
    const __amdgpu_buffer_rsrc_t rst = __builtin_amdgcn_make_buffer_rsrc((void*)p.tout, 0, npix * 12u, STRIP_RSRC);
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, npix * 12u, STRIP_RSRC);
    unsigned col12 = col_ok ? (unsigned)c * 12u : STRIP_HALF, colp24 = col_ok ? (unsigned)(c >> 1) * 24u : STRIP_HALF;
    unsigned own12 = lane_own ? (unsigned)c * 12u : STRIP_HALF;
    float colf = col_ok ? 1.0f : 0.f;
    int la = ((lane + 63) & 63) * 4, ra = ((lane + 1) & 63) * 4;
    asm volatile("" : "+v"(col12), "+v"(colp24), "+v"(own12), "+v"(colf), "+v"(la), "+v"(ra));
    const unsigned img0 = (unsigned)b * p.H, imgp0 = (unsigned)b * Hp;
    auto inside = [](int row, int lo, int hi) -> unsigned { return ~(unsigned)(((row - lo) | (hi - 1 - row)) >> 31); };
    auto rowpart = [&](unsigned row0, int row, unsigned ok, unsigned rowbytes) -> unsigned {   // uniform
        return ((row0 + (unsigned)row) * rowbytes & ok) | (STRIP_HALF & ~ok);
    };
    auto fmask = [](float v, unsigned m) -> float { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & m); };

    f3 sq[PFD], iq0[PFD], iq1[PFD];      // ring: skip row i, half-resolution input row i/2 (6 channels) of the step that consumes the slot
    auto issue = [&](auto kc, int i) {
        constexpr int k = decltype(kc)::value;
        const unsigned ok = inside(i, 0, p.H);
        sq[k] = strip_load3(rss, col12 + rowpart(img0, i, ok, (unsigned)p.W * 12u));
        const unsigned o = colp24 + rowpart(imgp0, i >> 1, ok, (unsigned)Wp * 24u);
        iq0[k] = strip_load3(rsi, o);
        iq1[k] = strip_load3(rsi, o + 12u);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    issue(I0{}, r0 - 1);
    issue(I1{}, r0);
    if constexpr (PFD == 3) issue(I2{}, r0 + 1);
    // transposed-conv kernel slices of this lane's column parity e: wt[a][co][ci] = Wt[a][e][co][ci]
    float wt[2][3][6], bt[3], bc[3], w[162];
    const int e = c & 1;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int co = 0; co < 3; ++co)
#pragma unroll
            for (int ci = 0; ci < 6; ++ci) wt[a][co][ci] = p.wt[((a * 2 + e) * 3 + co) * 6 + ci];
#pragma unroll
    for (int co = 0; co < 3; ++co) { bt[co] = p.bt[co]; bc[co] = p.b[co]; }
#pragma unroll
    for (int i = 0; i < 162; ++i) w[i] = p.w[i];
    float aslope = p.alpha < 0.f ? 1.0f : p.alpha;
    __builtin_amdgcn_s_waitcnt(0x0070);                           // vmcnt(0) lgkmcnt(0): nothing pending at the loop's entry (see k_tail3)
#pragma unroll
    for (int co = 0; co < 3; ++co) asm volatile("" : "+v"(bt[co]), "+v"(bc[co]));
    asm volatile("" : "+v"(aslope));
#pragma unroll
    for (int i = 0; i < 162; ++i) {
        if (i < WS) w[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, w[i])));
        else asm volatile("" : "+v"(w[i]));
    }
    float ta[3][9], sk[3][9];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int i = 0; i < 9; ++i) { ta[s][i] = 0.f; sk[s][i] = 0.f; }
    // step s (uu = s mod 6; window slot u = uu mod 3): rows i = r0 - 1 + s arrive -> transposed conv of row i (row parity a = (uu + 1) & 1),
    // both rows into the windows; CONV: the 3x3 conv of row j = i - 1
    auto step = [&](auto uc, auto convc, int i) {
        constexpr int uu = decltype(uc)::value, u = uu % 3, k = uu % PFD, a = (uu + 1) & 1;
        constexpr bool CONV = decltype(convc)::value;
        float s3[3], x6[6];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {      // real copies out of the ring (see k_tail3)
            asm volatile("v_mov_b32 %0, %1" : "=v"(s3[ch]) : "v"(sq[k][ch]));
            asm volatile("v_mov_b32 %0, %1" : "=v"(x6[ch]) : "v"(iq0[k][ch]));
            asm volatile("v_mov_b32 %0, %1" : "=v"(x6[3 + ch]) : "v"(iq1[k][ch]));
        }
        issue(std::integral_constant<int, k>{}, i + PFD);
        // transposed conv at (i, c); zero outside the image (the conv's zero padding)
        const float imgf = fmask(colf, inside(i, 0, p.H));
        float tv[3];
#pragma unroll
        for (int co = 0; co < 3; ++co) {
            float acc = bt[co];
#pragma unroll
            for (int ci = 0; ci < 6; ++ci) acc = fmaf(x6[ci], wt[a][co][ci], acc);
            tv[co] = acc * imgf;
        }
        __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b96(rst, 0, 0, 0)), f3{tv[0], tv[1], tv[2]}),
                                              rst, own12 + rowpart(img0, i, inside(i, r0, r1), (unsigned)p.W * 12u), 0, 0);
        strip_expand<1>(tv, ta[u], la, ra);             // rows j-1 -> (u+1)%3, j -> (u+2)%3, j+1 -> u
        strip_expand<1>(s3, sk[u], la, ra);
        if constexpr (CONV) {
            const int j = i - 1;
            float g[3] = {bc[0], bc[1], bc[2]}, g2[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int kk = 0; kk < 9; ++kk) {
                    const int kx = kk / 3, ci = kk % 3;
                    const float xa = ta[(u + 1 + dy) % 3][kk], xb = sk[(u + 1 + dy) % 3][kk];
#pragma unroll
                    for (int co = 0; co < 3; ++co) {
                        g[co] = fmaf(xa, w[((dy * 3 + kx) * 6 + ci) * 3 + co], g[co]);
                        g2[co] = fmaf(xb, w[((dy * 3 + kx) * 6 + 3 + ci) * 3 + co], g2[co]);
                    }
                }
#pragma unroll
            for (int co = 0; co < 3; ++co) {
                g[co] += g2[co];
                g[co] = fmaxf(g[co], g[co] * aslope);
            }
            __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b96(rso, 0, 0, 0)), f3{g[0], g[1], g[2]}),
                                                  rso, own12 + rowpart(img0, j, inside(j, r0, r1), (unsigned)p.W * 12u), 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    using Yes = std::true_type;
    using No = std::false_type;
    const int nsteps = r1 - r0 + 2;
    step(I0{}, No{}, r0 - 1);
    step(I1{}, No{}, r0);                     // (its row j = r0 - 1 belongs to the chunk above)
    step(I2{}, Yes{}, r0 + 1);
    step(std::integral_constant<int, 3>{}, Yes{}, r0 + 2);
    step(std::integral_constant<int, 4>{}, Yes{}, r0 + 3);
    step(std::integral_constant<int, 5>{}, Yes{}, r0 + 4);
#pragma unroll 1
    for (int s0 = 6; s0 < nsteps; s0 += 6) {
        step(I0{}, Yes{}, r0 - 1 + s0);
        step(I1{}, Yes{}, r0 + s0);
        step(I2{}, Yes{}, r0 + 1 + s0);
        step(std::integral_constant<int, 3>{}, Yes{}, r0 + 2 + s0);
        step(std::integral_constant<int, 4>{}, Yes{}, r0 + 3 + s0);
        step(std::integral_constant<int, 5>{}, Yes{}, r0 + 4 + s0);
    }
}

}  // namespace dnnca
