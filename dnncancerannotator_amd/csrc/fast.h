// fast.h -- hooks through which model.hip routes an op to a tuned gfx950 kernel.  Each returns false when it has no
// specialisation for the op's shape, in which case the caller falls through to the generic kernel.
#pragma once
#include "model.h"

namespace dnnca {

int fast_prepare(Model* m);     // per-step operand preparation (weights -> MFMA B operands)
int fast_finish_backward(Model* m);   // folds the weight-gradient slabs into the flat gradient vector (or defers: Model::fold_deferred)
int fast_fold_adam(Model* m, float lr_t, const dnnca_loss_cfg& cfg, double n_label, double inv_batch_hw);   // the deferred fold + Adam + step outputs in one launch
void fast_release(Model* m);
unsigned long long* fast_debug_stamps(Model* m);   // tuning aid: in-kernel s_memtime stamps (DNNCA_STAMPS)    // drop the per-model plan
// `pool`: a max-pool op fused into the conv's epilogue (fast_pool_fusable), or nullptr
bool fast_conv_fwd(Model* m, int B, Op& o, double bytes, double flops, Op* pool);
bool fast_pool_fusable(const Model* m, const Op& conv, const Op& pool);
// the conv that feeds the head, training step: head + loss + head backward ride in its epilogue (labels' statistics already on the stream)
bool fast_conv_fwd_head(Model* m, int B, Op& o, Op& head, const float* y, const dnnca_loss_cfg& cfg, float gscale, double bytes, double flops);
// kernels_fused.hip: a whole Downsample / Upsample block (components.py:77-81, 158-166) of configs/unet.yaml in one launch;
// ops[oi .. oi+2] are consumed when these return true.  store_mid: also write the block's intermediate tensors (a backward pass follows)
bool fused_down_fwd(Model* m, int B, size_t oi, bool store_mid, const float* labels = nullptr);   // labels: also reduce them (first block only)
// ... or, where the shape allows, that conv's forward, the head, the loss and the conv's whole backward in one launch (sets Model::tail_done)
bool fast_tail3(Model* m, int B, Op& o, Op& head, const float* y, const dnnca_loss_cfg& cfg, float gscale);
bool fast_first3_fwd(Model* m, int B, Op& c1, Op& c2, Op& pool, float* y0, unsigned char* pool_idx, const float* labels, float* label_part,
                     double bytes, double flops, int* nblocks);      // first encoder block forward as one column-strip launch
bool fast_up3_fwd(Model* m, int B, size_t oi);     // last decoder block: transposed conv + two-source conv forward in one column-strip launch
bool fast_head_in_conv_possible(Model* m);      // would fast_conv_fwd_head take the conv that feeds the head?
bool fused_up_fwd(Model* m, int B, size_t oi, bool store_mid, int* consumed = nullptr);
bool fused_up2_fwd(Model* m, int B, size_t oi, bool store_mid);     // the two convs of a decoder block whose transposed conv has already run: ops[oi], ops[oi + 1]
// kernels_fused_bwd.hip: the whole BACKWARD of a Downsample / Upsample block of configs/unet.yaml (6- and 12-channel levels) in one
// launch; `oi` is the block's LAST op (the max-pool / the second conv): ops[oi - 2 .. oi] are consumed when these return true
bool fused_down_bwd(Model* m, int B, size_t oi);
bool fused_up_bwd(Model* m, int B, size_t oi);
// kernels_mfma.hip: what the block-fused kernels need from the pixel-group plan
constexpr int kPgBuckets = 16;                                 // partial-sum slabs per weight gradient
bool fast_pg_conv_supported(const Model* m, const Op& o);      // a 3x3 conv of the pixel-group plan
const float* fast_conv_bmat_dgrad(Model* m, const Op& o);      // prepared data-gradient B operands (all passes), or nullptr
float* fast_wgrad_slabs(Model* m, const Op& o, int source);    // weight-gradient slabs of (op, source) -- transposed convs: source 0 -- or nullptr
bool fast_conv_bwd(Model* m, int B, Op& o, double out_bytes, double in_bytes, double flops);
// kernels_first.hip: the one-channel-input 3x3 convs (first layer of every encoder)
bool fast_first_conv_fwd(Model* m, int B, Op& o, double bytes, double flops, Op* bn_next);   // bn_next as for ig_conv_fwd
bool fast_first_conv_bwd(Model* m, int B, Op& o, double out_bytes, double in_bytes, double flops);
bool fast_pool_fwd(Model* m, int B, Op& o, double bytes);
bool fast_pool_bwd(Model* m, int B, Op& o, double bytes);
bool fast_pool_fold(Model* m, Op& pool, Op& conv);      // the pool's backward rides in conv's backward launch (conv produced the pool's input)
bool fast_tconv_fwd(Model* m, int B, Op& o, double bytes, double flops);
bool fast_tconv_bwd(Model* m, int B, Op& o, double out_bytes, double in_bytes, double flops);
bool fast_pool_supported(const Model* m, const Op& o);
bool fast_tconv_supported(const Model* m, const Op& o);
bool fast_head_supported(const Model* m, const Op& o);
bool fast_head_train(Model* m, int B, Op& o, const float* y, const dnnca_loss_cfg& cfg, float gscale, double bytes);
// utils/losses.py:62-67 label_smoothing: Gaussian blur of the labels (false: unsupported filter size / image smaller than the pad)
bool fast_label_smooth(Model* m, int B, int H, int W, const float* y, float* out, int k, float sigma);
bool fast_label_stats(Model* m, size_t n, const float* y);
// pool: a 2x2 max-pool of the output that rides in the apply pass; pool_bn: the BatchNorm of the pooled tensor (its batch statistics ride along too)
bool fast_bn_fwd(Model* m, int B, Op& o, bool training, float momentum, float eps, Op* pool, Op* pool_bn);
bool fast_bn_pool_fusable(const Model* m, const Op& bn, const Op& pool);
bool fast_bn_bwd(Model* m, int B, Op& o);
bool fast_pool_into_bn(Model* m, Op& pool, Op& bn);      // the pool's backward rides in the backward passes of the BatchNorm in front of it
bool fast_bn_supported(const Model* m, const Op& o);
void fast_plan_masks(Model* m);
// Path-selecting switches of the dense kernels (tuning aids / A-B arms), read ONCE per process at their first use: the plan made at
// model creation (which BatchNorm apply passes are elided, ig_norm_on_load_ok) and every later launch decision must see the same
// values -- a conv must never read a normalised tensor that was never written.  (The per-model switches DNNCA_NO_HALF* are read when
// a model is built; the unet.yaml fusion switches are read per call: each of those paths falls back on the launches it replaced.)
struct DenseSwitches {
    bool igconv1, wgrad1, no_bn_fusion, no_pool_stats, no_wg_buckets, tcwgrad1;
};
const DenseSwitches& dense_switches();
// implicit-GEMM MFMA path for channel counts that are multiples of 16 (kernels_igemm.hip)
bool ig_conv_supported(const Model* m, const Op& o);
int ig_prepare(Model* m);
int ig_finish_wgrad(Model* m);      // the backward pass's weight-gradient slab folds in one launch (before the streams join)
int ig_plan_half(Model* m);        // dtype bf16: decides View::h for every tensor (static per model; after fast_plan_masks)
int ig_begin_backward(Model* m);
void ig_release(Model* m);
bool ig_conv_fwd(Model* m, int B, Op& o, double bytes, double flops, Op* bn_next);   // bn_next: BatchNorm of the output whose statistics may ride in the epilogue
bool ig_conv_bwd(Model* m, int B, Op& o, double out_bytes, double in_bytes, double flops);
// will this conv's forward and weight-gradient launches (batch B) be the kernels that can normalise a BatchNorm's input while staging it?
bool ig_norm_on_load_ok(const Model* m, int B, const Op& o);
struct BnSelfFold;
struct BnBwdFold;
// batch statistics of BatchNorm `bn` folded by the kernel that produces its input (bn_dev.h): fills *f and marks the statistics as
// taken care of (Op::fused_stats_rows); false: not available (the BatchNorm then runs its own reduction pass)
bool bn_self_fold_args(Model* m, Op& bn, int B, BnSelfFold* f);
bool bn_bwd_fold_args(Model* m, Op& bn, BnBwdFold* f);      // the BatchNorm backward sums ride in the launch that produces dy (ConvArgs::bnb)
// kernels_ig3x.hip: the fp32 3x3 convs on the bf16 matrix pipe (three bf16 planes per operand, fp32-accurate)
namespace ig { struct ConvArgs; struct WgArgs; }
bool ig3x_enabled(const Model* m);
int ig3x_prepare(Model* m);
void ig3x_release(Model* m);
bool ig3x_accepts(Model* m, const ig::ConvArgs& a, int cout);
int ig3x_max_bnb_channels();
// bnb_rode: (data gradient with ConvArgs::bnb filled) did the chosen kernel take the BatchNorm backward sums along?
bool ig3x_launch(Model* m, int mode, const ig::ConvArgs& a, size_t w_off, int cout, int nn, const char* name, double bytes, double flops,
                 bool* bnb_rode = nullptr);
bool ig3x_wgrad_launch(Model* m, ig::WgArgs w, int co, const char* name, double bytes, double flops);
int ig3x_wgrad_psplit(const Model* m, const ig::WgArgs& w, int co);      // pixel-split blocks of that launch; 0: not this path
bool ig_tconv_supported(const Model* m, const Op& o);
bool ig_tconv_fwd(Model* m, int B, Op& o, double bytes, double flops, Op* bn_next);
bool ig_tconv_bwd(Model* m, int B, Op& o, double out_bytes, double in_bytes, double flops);   // decides maskA/maskB/premasked for every op (static per model)

}  // namespace dnnca
