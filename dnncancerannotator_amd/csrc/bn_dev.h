// BatchNorm batch statistics that fold themselves (device side).
//
// The kernels that see every value a BatchNormalization normalises (the conv in front of it, components.py:56-59 -- or the
// plain reduction pass where no producer can) leave per-block partial sums.  Folding those used to be a launch of its own
// (k_bn_fold_stats: 8 .. 128 blocks, 8 us, plus a dependent launch on the step's critical chain, 24 / 48 times per step of the
// dense configurations).  Here the producer folds: every block ADDS its partial sums to one of R bucket rows of a small table
// (double atomics, global_atomic_add_f64: the order the blocks arrive in moves the sum by double rounding only, ~1e-16 relative,
// far below the float the result is rounded to; a row meets blocks / R adders), draws a ticket when it is done, and
// the block that draws the last ticket folds the R rows, writes mean / variance / moving statistics / coefficients exactly as
// g_bn_finalize does, and leaves table and ticket zeroed for the next user.  Table: Model::bn_tab (kBnTab doubles,
// R * 2C <= kBnTab, R <= kBnRows), uses are stream-ordered.
//
// Ordering (round 4: made explicit; the workgroup-scope release fence that stood here compiles to NO wait on gfx950 outside
// tgsplit mode, so nothing ordered a block's in-flight bucket adds before its ticket):
//   1. every thread waits for its OWN outstanding memory operations with `s_waitcnt vmcnt(0)` -- a no-return
//      global_atomic_add_f64 leaves vmcnt when the memory side has acknowledged (= performed) it;
//   2. the workgroup barrier makes that hold for the whole block;
//   3. only then thread 0 draws the ticket (a device-scope returning atomic, performed at the same place as the adds);
//   4. the block that draws the last ticket reads the rows with device-scope loads, issued behind its ticket's return.
// So every add of every block is performed before the last ticket is, and the fold (and its zeroing) comes behind that ticket.
// No device-scope FENCE: on gfx950 a release / acquire fence at agent scope is an L2 write-back / invalidate, and one per block
// costs more than the reduction it guards (measured on k_bn_bwd_reduce_fast: 74 us against 25).  Nothing is published through
// plain stores: the partial sums travel as device-scope atomic adds.  tests/test_isa_bn_order.py checks the compiled code of
// every kernel that calls bn_last_block for step 1 (vmcnt(0) between the last bucket add and the barrier in front of the ticket).
#pragma once
#include <hip/hip_runtime.h>

namespace dnnca {

struct BnSelfFold {
    double* tab;             // [R][2C] bucket rows: sums, then sums of squares; nullptr: no statistics wanted
    unsigned* ticket;
    int R, C;
    float momentum, eps;
    double n;                // values per channel
    const float* gamma;
    const float* beta;
    float* mmean;
    float* mvar;
    float* coef;             // [4][C]: scale, shift, mean, 1 / sqrt(var + eps)
};

// bucket row of a contribution (key: anything spread evenly over the contributors, e.g. the block or tile index)
constexpr int kBnRows = 16;          // bucket rows at most: the last block loads them all at once
__device__ __forceinline__ double* bn_bucket(const BnSelfFold& f, int key) { return f.tab + (size_t)(key % f.R) * 2 * f.C; }

// the R bucket values of column `col`, summed (device-scope loads, all in flight; rows past R: the last row again, not added)
__device__ __forceinline__ double bn_fold_column(const double* tab, int R, int stride, int col) {
    double v[kBnRows];
#pragma unroll
    for (int r = 0; r < kBnRows; ++r) v[r] = __hip_atomic_load(tab + (size_t)min(r, R - 1) * stride + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double a = 0.0;
#pragma unroll
    for (int r = 0; r < kBnRows; ++r)
        if (r < R) a += v[r];
    return a;
}

// Is this block the last one of the launch to get here?  (block-uniform; all threads of every block call it once, after the
// block's last bucket add; bid: linear block index, nblocks: blocks of the launch.)  Two levels of counters: the blocks of a
// persistent kernel finish together, and same-address atomics execute one after the other (~56 ns each: 256 blocks on one counter
// held the last one up for 14 us) -- so a block draws from one of kBnGroups group counters, and only the last of a group draws
// from the top counter.  ticket[0]: top, ticket[1 + g]: groups; whoever draws the last ticket of a counter zeroes it.
constexpr int kBnGroups = 32;          // (the ticket array behind the table has 64 slots)
__device__ __forceinline__ bool bn_last_block(unsigned* ticket, unsigned nblocks, unsigned bid) {
    __shared__ unsigned bn_last;
#ifndef DNNCA_AB_NO_BN_WAIT          // (A/B arm of tools: the round-3 code without the wait; never shipped)
    __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0) lgkmcnt(0): this thread's bucket adds have been performed (acknowledged)
#endif
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned G = nblocks < (unsigned)kBnGroups ? nblocks : (unsigned)kBnGroups, g = bid % G;
        const unsigned members = (nblocks - g + G - 1) / G;
        unsigned last = 0;
        if (__hip_atomic_fetch_add(ticket + 1 + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
            ticket[1 + g] = 0u;
            if (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G - 1) {
                ticket[0] = 0u;
                last = 1u;
            }
        }
        bn_last = last;
    }
    __syncthreads();
    return bn_last != 0u;
}

// All threads of every block call this once, after the block's last bn_bucket() add.  nblocks / bid as for bn_last_block.
__device__ __forceinline__ void bn_self_fold(const BnSelfFold& f, unsigned nblocks, unsigned bid) {
    if (!bn_last_block(f.ticket, nblocks, bid)) return;
    const int C = f.C, R = f.R;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        // everything this channel needs is requested at once: ONE memory round trip (as dependent load -> store pairs the moving
        // statistics and gamma / beta cost four more, 2 us each, and held the producer's last block for 15 us)
        const float gamma = f.gamma[c], beta = f.beta[c], mm = f.mmean[c], mv = f.mvar[c];
        const double sum = bn_fold_column(f.tab, R, 2 * C, c), sumsq = bn_fold_column(f.tab, R, 2 * C, C + c);
        for (int r = 0; r < R; ++r) { f.tab[(size_t)r * 2 * C + c] = 0.0; f.tab[(size_t)r * 2 * C + C + c] = 0.0; }
        // raw moments -> coefficients and moving statistics: the contract of g_bn_finalize
        const double mean_d = sum / f.n;
        double var_d = sumsq / f.n - mean_d * mean_d;
        if (var_d < 0.0) var_d = 0.0;
        const float mean = (float)mean_d, var = (float)var_d;
        const float unbiased = (float)(var_d * (f.n > 1.0 ? f.n / (f.n - 1.0) : 1.0));
        f.mmean[c] = mm * f.momentum + mean * (1.f - f.momentum);
        f.mvar[c] = mv * f.momentum + unbiased * (1.f - f.momentum);
        const float inv = 1.0f / sqrtf(var + f.eps);
        const float sc = gamma * inv;
        f.coef[c] = sc;
        f.coef[C + c] = beta - mean * sc;
        f.coef[2 * C + c] = mean;
        f.coef[3 * C + c] = inv;
    }
}

// The sums of a BatchNorm BACKWARD (dgamma = sum dy xhat, dbeta = sum dy) folding themselves the same way: the contributors (the
// reduction pass k_bn_bwd_reduce_fast, or the data-gradient launch that produces dy: ConvArgs::bnb) add to bucket rows laid out
// [C sums of dy xhat | C sums of dy]; the block that draws the last ticket adds the folded rows to dgamma / dbeta.
struct BnBwdFold {
    double* tab;             // nullptr: not wanted
    unsigned* ticket;
    int R, C;
    float* dgamma;
    float* dbeta;
    const float* x;          // the BatchNorm's input (what xhat is made of), same layout as dy; contributors that are not the reduction pass
    const float* coef;       // the BatchNorm's [4][C] coefficients: mean at 2C, 1 / sqrt(var + eps) at 3C
};

// All threads of every block call this once, after the block's last bucket add.
__device__ __forceinline__ void bn_bwd_self_fold(const BnBwdFold& f, unsigned nblocks, unsigned bid) {
    if (!bn_last_block(f.ticket, nblocks, bid)) return;
    const int C = f.C, R = f.R;
    for (int o = threadIdx.x; o < 2 * C; o += blockDim.x) {
        float* dst = o < C ? f.dgamma + o : f.dbeta + (o - C);
        const float before = *dst;          // requested together with the rows: one memory round trip
        const double a = bn_fold_column(f.tab, R, 2 * C, o);
        for (int r = 0; r < R; ++r) f.tab[(size_t)r * 2 * C + o] = 0.0;
        *dst = before + (float)a;
    }
}

}  // namespace dnnca
