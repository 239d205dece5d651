// model.h -- the layer plan of one annotator model and its device state.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "common.h"

typedef struct ncclComm* ncclComm_t;

namespace dnnca {

struct T {          // a tensor = data view + gradient view of identical geometry
    View d, g;
};

inline T tslice(const T& t, int c0, int c) { return T{slice(t.d, c0, c), slice(t.g, c0, c)}; }

enum OpType { OP_CONV = 0, OP_BN, OP_POOL, OP_TCONV, OP_HEAD };

struct Op {
    OpType type;
    std::string name;
    T inA, inB, out;
    int64_t w_off = -1, b_off = -1;      // trainable offsets: kernel/bias, or gamma/beta for BN
    int64_t mm_off = -1, mv_off = -1;    // state offsets (BN moving mean / variance)
    float alpha = -1.f;                  // conv activation: <0 none, 0 relu, >0 leaky
    int k = 0;                           // conv kernel size / pool+tconv rate
    bool need_din = true;                // false for the convs that read the network input
    bool accA = false, accB = false;     // backward: accumulate into (instead of overwrite) the input gradients
    // activation-derivative fusion: the LAST gradient contributor of a conv+activation output multiplies the summed
    // gradient by act'(output) when it stores it (maskA/maskB on that consumer), and the producing conv is then
    // `premasked`: its backward takes out.g as the pre-activation gradient directly.
    bool maskA = false, maskB = false, premasked = false;
    float mask_alpha = 0.f;
    float* coef = nullptr;               // BN: [scale, shift, mean, inv] x C
    double* ws = nullptr;                // BN: [sum, centred sumsq] x C
    // BN batch statistics that rode in the producing conv's epilogue this step: rows of float partials waiting in the
    // model's partials table (0: none)
    int fused_stats_rows = 0;
    // BatchNorm apply elided (f32 dense path): when every reader of a BatchNorm's output is a 3x3 conv that normalises while it
    // stages its operands (k_ig_conv3 forward, k_ig_wgrad2), the apply pass and the normalised tensor are skipped for the step
    // (`elided`, decided by the forward pass, valid until the next one); the convs read the BatchNorm's INPUT and its coefficients.
    // src_bn[k]: the BatchNorm op that produces conv input k (-1: none); out_readers: the ops that read this op's output tensor
    bool elided = false;
    // BN: the max-pool behind this BatchNorm handed its backward pass over (fast_pool_into_bn): the BatchNorm's backward passes route
    // the pooled gradient themselves; pool_grad_acc: dy already holds another reader's gradient (a skip connection)
    const Op* pool_grad = nullptr;
    bool pool_grad_acc = false;
    // BN: this step's dgamma / dbeta sums rode in the data-gradient launch that produced dy (bn_bwd_fold_args): no reduction pass
    bool bwd_sums_rode = false;
    int src_bn[2] = {-1, -1};
    std::vector<int> out_readers;
    // pool: position (0..3, row-major in the 2x2 window) of each output's first maximum, written by the fused BN-apply + pool
    // pass of this step; the backward pass routes by it instead of re-reading the input and output tensors
    unsigned char* pool_idx = nullptr;
    bool pool_idx_valid = false;
};

struct ParamInfo {
    std::string name;
    int64_t shape[4];
    int ndim;
    int trainable;
    int64_t offset;
    int64_t size;
};

struct KStat {
    std::string name;
    int64_t launches = 0;
    double total_ms = 0, bytes = 0, flops = 0;
};

constexpr float kBnMomentum = 0.99f;   // Keras BatchNormalization defaults [TF-2.6]
constexpr float kBnEps = 1e-3f;

struct Model {
    dnnca_model_desc desc;
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<ParamInfo> params;
    std::vector<Op> ops;
    std::vector<void*> allocs;
    int64_t nT = 0, nS = 0;
    float *p = nullptr, *g = nullptr, *m = nullptr, *v = nullptr, *state = nullptr;
    double* scalars = nullptr;           // kScalars doubles
    float* out5 = nullptr;               // = g + nT : [loss, positive_rate, weight, ymin, ymax]
    float *x_stage = nullptr, *y_stage = nullptr, *logits = nullptr, *dlogits = nullptr, *prob = nullptr;
    float* thr_dev = nullptr;
    // the fused head's partial sums wait for the launch that ends the backward pass (k_pg_fold reduces them: one launch less)
    struct HeadPending { const float* partials = nullptr; int nblocks = 0, C = 0; float* dw = nullptr; float* dbias = nullptr; } head_pending;
    bool head_defer_ok = false;           // set by the pixel-group plan when a k_pg_fold launch exists
    bool step_init_done = false;      // train step: the launch that prepares the pixel-group B operands already zeroed scalars / gradients / slabs
    // single-replica training steps: the step outputs are written by the Adam launch instead of a launch of their own
    struct FinalizePending { bool on = false; dnnca_loss_cfg cfg; double n_label = 0, inv_batch_hw = 0; } fin_pending;
    float* y_smooth = nullptr;           // smoothed labels of the step (label_smoothing, utils/losses.py:62-67)
    void* warp_scratch = nullptr;        // control points + spline weights of dnnca_warp_f32
    size_t warp_scratch_bytes = 0;
    void* aug_scratch = nullptr;         // per-image augmentation draws + channel sums (kernels_aug.hip)
    size_t aug_scratch_bytes = 0;
    // the caller's draws wait for their asynchronous upload in a small ring of pinned rows, so that dnnca_augment_u8 returns
    // without synchronising the stream (the exam-file train loop keeps a step queued behind the running one)
    static constexpr int kAugRing = 4;
    void* aug_pin = nullptr;             // kAugRing x aug_pin_bytes, pinned
    size_t aug_pin_bytes = 0;
    hipEvent_t aug_ev[kAugRing] = {nullptr, nullptr, nullptr, nullptr};      // recorded behind the upload that read row k
    int aug_k = 0;
    float* first_slabs = nullptr;        // bucket copies of the first-layer weight gradient (kernels_first.hip; kept zeroed)
    // bucket rows of the BN reductions that fold themselves (bn_dev.h): kBnTab doubles, then the ticket counter; kept zeroed
    static constexpr int kBnTab = 4096;
    double* bn_tab = nullptr;
    float* extra_zero = nullptr;         // buffer the tuned kernels need zeroed at the top of every backward pass
    size_t extra_zero_n = 0;
    float* head_partials = nullptr;      // [2048][8] block partials of the fused head kernel
    double* conf_dev = nullptr;
    T xin;                               // network input view (points at the current batch)
    int outH = 0, outW = 0;
    int64_t iterations = 0;
    float beta1 = 0.9f, beta2 = 0.999f, eps = 1e-7f;
    int last_batch = 0;
    bool defer_head = false, head_deferred = false;   // train step: the head runs fused with the loss and its backward
    // train step on the pixel-group plan: the head rides in the epilogue of the conv that feeds it (fast_conv_fwd_head); forward()
    // then needs the step's labels and loss configuration, and loss_and_backward() finds label statistics and head already done
    // label statistics of a train step as per-block partials (sum, min, max, -) written by the first encoder block's fused launch
    // (kernels_fused.hip) and consumed by the head-in-conv kernel: no launch and no atomics of their own
    float* label_part = nullptr;
    int label_part_nblk = 0;
    bool label_part_valid = false;
    const Op* tconv_done = nullptr;      // the transposed conv whose backward rode in the launch of the conv behind it (k_pgbwd TCF)
    const Op* first_done = nullptr;      // the first conv whose weight gradient rode in the backward launch of the conv behind it (k_first3)
    // the step's operand preparation (k_pg_prep: B operands + zeroing) waits for the first launch of the forward pass: the first
    // encoder block's strip kernel takes it along as extra blocks; any other launch flushes it first (LAUNCH)
    void (*prep_flush)(Model*) = nullptr;
    bool fold_deferred = false;          // the slab fold waits for optimizer_step: fold + Adam + step outputs in one launch (fast_fold_adam)
    const Op* tail_done = nullptr;       // the conv whose backward already ran inside the forward pass (fast_tail3)
    struct PoolFold { const Op* conv = nullptr; const Op* pool = nullptr; } pool_fold;     // fast_pool_fold -> fast_conv_bwd hand-over
    struct HeadInConv { const float* y = nullptr; dnnca_loss_cfg cfg; bool requested = false, done = false, labels_done = false; } head_in_conv;
    // data parallel
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    // Gradient vectors above 1 MB (unet_big: 63 MB, mulmo_unet: 6.9 MB) are all-reduced in buckets while the backward pass is
    // still running: the backward pass finalises the flat gradient vector from its END (head, decoder ... first encoder), so
    // a bucket is the newest finalised suffix; it goes to RCCL on a second stream behind an event.  Sums over ranks are
    // elementwise, so the result is bit-identical to the single call.
    hipStream_t comm_stream = nullptr;
    // Weight gradients of the dense (igemm) convs are leaves of the backward pass: nothing reads them before the optimizer step.
    // Train steps launch them on a second, low-priority stream (fork: an event behind the kernel that produced
    // the conv's output gradient; join: one event at the end of the backward pass), where they run beside the HBM-bound
    // BatchNorm passes and data-gradient convs of the main chain instead of in front of them.
    hipStream_t wg_stream = nullptr;
    hipEvent_t wg_fork = nullptr, wg_join = nullptr, wg_bucket = nullptr;
    bool wg_pending = false;             // the side stream holds work of this backward pass
    bool wg_side_begin();                // fork: later launches on `stream` go to the side stream; false: not in this mode
    void wg_side_end(hipStream_t main);  // back to the main stream
    int wg_side_join();                  // the main stream waits for the side stream (end of the backward pass)
    hipEvent_t ev_bucket = nullptr, ev_comm_done = nullptr;
    int bucket_state = 0;                // 0 undecided, 1 the op order finalises a suffix (bucketing possible), -1 it does not
    bool bucketing = false;              // this step's backward pass sends buckets
    int64_t bucket_hi = 0, bucket_fin = 0;   // [bucket_fin, bucket_hi) is final and not yet sent
    size_t bucket_bytes = 8u << 20;
    int collectives_last_step = 0;       // gradient all-reduce calls issued by the last train step
    int send_bucket(int64_t lo, int64_t hi);
    // Input pipeline (dnnca_stage_*): a batch travels host -> HBM on its own copy stream into one of a ring of staging slots
    // while the main stream is still working on the previous step; the main stream waits for the slot's `uploaded` event, runs
    // the step, sends the step outputs to a pinned host ring and records `done` (the next upload into the slot waits for it,
    // and the host reads the outputs behind it -- normally one step late, so that it never stalls the launch queue).
    static constexpr int kStageSlots = 8;     // e.g. four for the train feeder + four for the validation inside train()
    struct StageSlot {
        float *x = nullptr, *y = nullptr;
        hipEvent_t uploaded = nullptr, done = nullptr;
        bool has_done = false;           // `done` has been recorded at least once (main thread; handed over with the slot)
        bool is_eval = false;            // the step that last ran on the slot was an evaluation step (rank-local loss)
        int batch = 0;
    } stage[kStageSlots];
    int stage_slots = 0;
    size_t stage_bytes = 0;              // capacity of one slot
    hipStream_t copy_stream = nullptr;
    // staged evaluation (dnnca_eval_begin .. dnnca_eval_end): the pixel-confusion histogram stays on the device and keeps adding
    // up over the batches (exact integer counts); it is read once, at the end
    bool eval_active = false;
    std::vector<int> eval_order;         // thresholds: sorted position -> the caller's position
    float* out_ring = nullptr;           // pinned host memory: kStageSlots x 8 floats (out5 of the step that used the slot)
    // measurement
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int prof_mode = 0;                   // 0 off, 1 every launch, 2 only `focus`, 3 every launch keyed by kernel@layer
    const std::string* cur_op = nullptr;
    int prof_period = 1;                 // mode 2: bracket the focus kernel in one train step out of prof_period
    // launches that ride in another kernel's launch keep doing so -- the profile table of mode 1 is the schedule that really runs
    // (their work is then inside the carrying launch's row) -- except in the per-layer table of mode 3
    bool merged_launches() const { return prof_mode != 3; }
    bool dry = false;
    std::string focus, plan_text;
    // the template variant of the next launch ("n4w8", "m2n4w4", ...): appended to the launch name as `name#variant` in the dry
    // plan only (dnnca_plan_dump) -- the kernel-coverage test tells variants apart, the profile tables keep aggregating by name
    std::string variant;
    void set_variant(const char* fmt, ...);
    std::map<std::string, int> kid;
    std::vector<KStat> kstats;
    struct Rec {
        int id;
        hipEvent_t a, b;
    };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> evpool;

    // mulmo: the encoders (unet.py:152-165) are identical in shape and independent until the bottleneck concat; their ops sit one
    // encoder after the other in `ops` (enc_ops each, n_enc of them, from index 0).  Both passes walk them in LOCKSTEP (op j of every
    // encoder, then op j + 1, ...) so that one launch can serve the twin ops of all encoders (fast_bn_bwd_mp, ...): at op j of the
    // first encoder visited, the inputs of every encoder's op j are ready.  enc_ops = 0: no such structure.
    int enc_ops = 0, n_enc = 0;
    bool lockstep() const;
    std::vector<int> pass_order(bool backward) const;      // op indices in the order a pass visits them
    std::vector<char> op_done;                              // per pass: this op's work rode in an earlier launch of the pass

    ~Model();
    int build();
    int alloc(void** ptr, size_t bytes);
    int forward(const float* x_dev, int B, bool training);
    int loss_and_backward(const float* y_dev, int B, const dnnca_loss_cfg& cfg, bool backward);
    int optimizer_step(float lr);
    int flush_profile();

    // launch accounting ------------------------------------------------------------------------------------
    bool begin(const char* name, double bytes, double flops);
    void end();
};

}  // namespace dnnca

// A launch is only "open" for profiling between begin() and end(); focus mode leaves other launches untouched.
#define LAUNCH(m, name, bytes, flops, call)                \
    do {                                                   \
        if ((m)->prep_flush) {      /* a deferred operand preparation that no launch has taken along: it goes first */ \
            auto flush_ = (m)->prep_flush;                 \
            (m)->prep_flush = nullptr;                     \
            flush_(m);                                     \
        }                                                  \
        bool prof_open_ = false;                           \
        if ((m)->dry || (m)->prof_mode) {                  \
            size_t before_ = (m)->recs.size();             \
            bool go_ = (m)->begin(name, bytes, flops);     \
            prof_open_ = (m)->recs.size() != before_;      \
            if (!go_) break;                               \
        }                                                  \
        call;                                              \
        if (prof_open_) (m)->end();                        \
        (m)->variant.clear();                              \
    } while (0)

