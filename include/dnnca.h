/*
 * dnnca.h -- C ABI of libdnnca.so: the MI355X-native (gfx950, hand-written HIP) U-Net train/evaluate engine that
 * stands in for the TensorFlow/Keras numeric back-end of yoshihikoueno/DNNCancerAnnotator's hot path.
 *
 * The reference has no FFI boundary of its own on this path (it is Python on top of TensorFlow); each entry point below
 * cites the reference interface (file:line under annotator/) whose work it takes over.  The Python host
 * (dnncancerannotator_amd/engine.py, a mirror of annotator/engine.py:36-288) binds these with ctypes; INTEGRATION.md
 * shows the stub.
 *
 * Conventions: every function returns 0 on success or a negative DNNCA_E* code (message via dnnca_last_error());
 * nothing throws across the ABI; the caller owns host buffers; the library owns device memory; tensors are NHWC float32;
 * a model handle is bound to one HIP device + one stream and is not thread-safe; data parallel = one process per GPU,
 * each with its own handle, joined by dnnca_comm_init (RCCL over xGMI).
 */
#ifndef DNNCA_H
#define DNNCA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DNNCA_OK 0
#define DNNCA_EINVAL (-1)      /* bad argument / unsupported configuration */
#define DNNCA_EHIP (-2)        /* HIP runtime error (no device, launch failure, out of memory) */
#define DNNCA_ECOMM (-3)       /* RCCL error */
#define DNNCA_ESTATE (-4)      /* call sequence error (e.g. step before init) */
#define DNNCA_EASSERT (-5)     /* a reference-side tf.debugging.assert_* would have fired (utils/losses.py:30,91-99) */

enum { DNNCA_ARCH_UNET = 0, DNNCA_ARCH_MULMO = 1 };     /* models/tf_models/unet.py:194 UNetAnnotator, :285 MulmoUNetAnnotator */
enum { DNNCA_PAD_VALID = 0, DNNCA_PAD_SAME = 1 };       /* model_options.padding (configs/unet.yaml:9) */
/* arithmetic type of the conv contractions.  DNNCA_BF16: the operands of the 3x3 convolutions with >= 32 channels and of the
   transposed convolutions with channel counts that are multiples of 64 are rounded to bfloat16 (round to nearest even) on
   their way into the matrix cores; weights, optimizer state, BatchNorm statistics and every accumulation stay float32.
   Activations ("bf16 acts", BASELINE.md configs[2]): tensors whose every reader rounds them anyway are kept as bfloat16 in
   HBM (bit-identical results), and so are -- this does round -- the input of a BatchNorm written by a 64-channel-multiple
   conv / transposed conv (batch statistics are those of the stored values) and the gradient arriving at a BatchNorm whose
   users are all such layers or a 2x2 max-pool.  DNNCA_NO_HALF_Z / DNNCA_NO_HALF_DY in the environment keep those in float32 */
enum { DNNCA_F32 = 0, DNNCA_BF16 = 1 };
enum { DNNCA_UNIQUE_ID_BYTES = 128 };

/* model_options of configs/{unet,unet_big,mulmo_unet}.yaml (unet.py:195-207) + the input element spec
 * (engine.py:93 model.build(dataset.element_spec[0].shape)). */
typedef struct dnnca_model_desc {
    int32_t arch;             /* DNNCA_ARCH_* */
    int32_t in_channels;      /* C of x[B,H,W,C]; mulmo builds one encoder per channel (unet.py:151-165) */
    int32_t height, width;    /* must be divisible by rate^n_downsample */
    int32_t max_batch;        /* activations are allocated once for this batch size */
    int32_t n_filters_first;  /* unet.yaml:3 */
    int32_t n_downsample;     /* unet.yaml:4 */
    int32_t rate;             /* unet.yaml:5 (pool window/stride, transposed-conv kernel/stride) */
    int32_t kernel_size;      /* unet.yaml:6 */
    int32_t conv_stride;      /* unet.yaml:7 (only 1 is supported; the reference configs use 1) */
    int32_t bn;               /* unet.yaml:8 */
    int32_t padding;          /* DNNCA_PAD_* (only SAME is supported; every reference config uses same) */
    int32_t reference_index;  /* unet.py:114 (mulmo: which encoder's skips feed the decoder) */
    int32_t n_conv;           /* components.py:25 (2) */
    float leaky_alpha;        /* 0 = relu; 0.3 = configs/additionals/leakyReLU.yaml */
    float l2;                 /* 0 = none; 0.01 = configs/additionals/kernel_regularizer.yaml */
    int32_t dtype;            /* DNNCA_F32 | DNNCA_BF16 */
    int32_t flags;            /* bit 0: force the generic (untuned) kernels everywhere -- used by the parity tests */
} dnnca_model_desc;

/* deploy_options.loss.config of configs/additionals/deploy_options.yaml:4-7 (utils/losses.py:41-56) */
typedef struct dnnca_loss_cfg {
    int32_t has_weight;       /* 0: weight = 1/positive_rate (or 1 when there are no positives), losses.py:24-27 */
    float weight;
    float weight_add;         /* losses.py:29 */
    float weight_mul;
    /* utils/losses.py:62-67: y_true = tfa.image.gaussian_filter2d(y_true, filter_shape, sigma) before everything else (positive
       rate, assertions, loss): REFLECT padding of (k-1)/2 rows/columns before and k-1-(k-1)/2 after, separable kernel
       softmax(-u^2 / (2 sigma^2)) over u = -k/2+1 .. k/2 (tensorflow-addons, absent third-party dependency: restated from its
       published algorithm) */
    int32_t label_smoothing;  /* 0 = off (the default, losses.py:46) */
    int32_t label_smoothing_filter_size;
    float label_smoothing_sigma;
} dnnca_loss_cfg;

typedef struct dnnca_step_out {
    float loss;               /* scalar Keras loss of the step (mean over batch [+ L2]); mean over ranks under DP */
    float positive_rate;      /* utils/losses.py:87-102 (rank-local) */
    float weight;             /* the positive-class weight actually applied (rank-local) */
    float label_min, label_max;
} dnnca_step_out;

/* pixel confusion counts at one threshold (prob > threshold: Keras Precision/Recall convention, metrics.yaml:2-6) */
typedef struct dnnca_confusion {
    double tp, fp, fn, tn;
} dnnca_confusion;

/* ---- process / device ---------------------------------------------------------------------------------------- */
const char* dnnca_version(void);
const char* dnnca_last_error(void);
int dnnca_device_count(int* count);
int dnnca_init(int device_ordinal);                 /* hipSetDevice; engine.py:260-263 (strategy creation) */

/* ---- model life-cycle: engine.py:254-288 from_config + engine.py:93 model.build ------------------------------- */
int dnnca_model_create(const dnnca_model_desc* desc, void** model_out);
int dnnca_model_destroy(void* model);

/* variables in Keras creation order (components.py:69-75,226-233,292-312; unet.py:166-176,273-277).
 * Conv2D kernels are HWIO, Conv2DTranspose kernels [kh,kw,Cout,Cin]. offset = position in the trainable (or state) flat vector. */
int dnnca_param_count(void* model, int* count);
int dnnca_param_info(void* model, int index, char* name, size_t name_cap, int64_t shape[4], int* ndim,
                     int* trainable, int64_t* offset);
int dnnca_num_trainable(void* model, int64_t* n);   /* floats in the trainable flat vector */
int dnnca_num_state(void* model, int64_t* n);       /* floats in the non-trainable flat vector (BN moving statistics) */

/* model.load_weights / save_weights (engine.py:197,224-231, ModelCheckpoint engine.py:103-106) */
int dnnca_set_params(void* model, const float* flat, int64_t n);
int dnnca_get_params(void* model, float* flat, int64_t n);
int dnnca_set_state(void* model, const float* flat, int64_t n);
int dnnca_get_state(void* model, float* flat, int64_t n);
int dnnca_get_grads(void* model, float* flat, int64_t n);          /* gradients of the last train step (parity tests) */
/* Adam slots + iteration counter (engine.py:276-284); checkpoint / auto-resume (engine.py:67-78) */
int dnnca_set_opt_state(void* model, const float* m, const float* v, int64_t n, int64_t iterations);
int dnnca_get_opt_state(void* model, float* m, float* v, int64_t n, int64_t* iterations);
int dnnca_set_adam(void* model, float beta1, float beta2, float epsilon);

/* ---- the hot path, host buffers ------------------------------------------------------------------------------ */
/* UNetAnnotator.call (unet.py:279-282): probabilities [B,H,W,1]; logits = the tensor Keras caches as _keras_logits */
int dnnca_forward(void* model, const float* x_nhwc, int batch, int training, float* prob_out, float* logit_out);
/* keras Model.train_step under engine.py:126-135: forward(training=True) + TFWeightedCrossentropy (losses.py:60-72)
 * + backward + [RCCL all-reduce] + Adam apply with learning rate lr (LearningRateScheduler, engine.py:97-100) */
int dnnca_train_step(void* model, const float* x_nhwc, const float* y_hw, int batch, float lr,
                     const dnnca_loss_cfg* cfg, dnnca_step_out* out);
/* keras Model.test_step under engine.py:198-203: forward(training=False) + loss; optional probabilities */
int dnnca_eval_step(void* model, const float* x_nhwc, const float* y_hw, int batch,
                    const dnnca_loss_cfg* cfg, dnnca_step_out* out, float* prob_out);

/* ---- the hot path, device-resident batches (what bench.py times) ---------------------------------------------- */
int dnnca_dev_alloc(void** dev_ptr, size_t bytes);
int dnnca_dev_free(void* dev_ptr);
int dnnca_memcpy_h2d(void* dev_dst, const void* host_src, size_t bytes);
int dnnca_memcpy_d2h(void* host_dst, const void* dev_src, size_t bytes);
/* ---- train-time augmentation on the device (annotator/data.py:62-111 train_ds): crop + flip + contrast + feature/label split
   of a uint8 batch [batch, hs, ws, cs] resident in HBM (upload it with dnnca_memcpy_h2d as stored in the TFRecords), written as
   x [batch, ho, wo, cs-1] and y [batch, ho, wo] float32.  Per image the host passes its random draws:
     dy, dx    crop jitter added to the centre offsets (data.py:677-689 random_crop: clip(int(N(0, 4)), -6, 6))
     flip      1 = tf.image.random_flip_left_right took the flip branch (data.py:620-625)
     contrast  factor of tf.image.random_contrast, U[0.8, 1.2) (data.py:586-609); applied to the source channels whose bit is
               set in contrast_mask (never to the label channel); 1.0 = identity
   Fails with DNNCA_EINVAL when a crop window leaves the source image (tf.image.crop_to_bounding_box asserts the same).
   Asynchronous on the model's stream: params_host has been copied when the call returns (it may be reused at once); x_dev / y_dev
   are complete for later work on that stream (a train step), a host read needs dnnca_sync first. */
typedef struct { int32_t dy, dx, flip; float contrast; } dnnca_aug_param;
int dnnca_augment_u8(void* model, const void* src_dev, int batch, int hs, int ws, int cs, int label_index, unsigned contrast_mask,
                     const dnnca_aug_param* params_host, int ho, int wo, float* x_dev, float* y_dev);
/* random_warp (annotator/data.py:725-763 -> tfa.image.sparse_image_warp, interpolation order 2, no boundary points): dense part.
   ctrl_host [batch, n_points, 2] = the destination control points (row, column); wv_host [batch, n_points + 3, 2] = the solution
   (w; v) of the polyharmonic-spline system for the control-point flows (dest - source), solved by the caller
   (dnncancerannotator_amd/augment.py solve_warp).  Every output pixel q evaluates flow(q) and samples x [batch, h, w, c] and
   y [batch, h, w] bilinearly at q - flow(q) (tfa dense_image_warp); outputs must not alias the inputs. */
int dnnca_warp_f32(void* model, const float* x_dev, const float* y_dev, int batch, int h, int w, int c, int n_points,
                   const double* ctrl_host, const double* wv_host, float* x_out_dev, float* y_out_dev);
/* same as dnnca_train_step with x/y already in HBM; asynchronous on the model's stream; out may be NULL (no sync) */
int dnnca_train_step_dev(void* model, const float* x_dev, const float* y_dev, int batch, float lr,
                         const dnnca_loss_cfg* cfg, dnnca_step_out* out);
int dnnca_forward_dev(void* model, const float* x_dev, int batch, int training);   /* results stay on the device */
int dnnca_last_step_out(void* model, dnnca_step_out* out);        /* synchronises, then reads the last step's scalars */
int dnnca_sync(void* model);

/* ---- host-side helper of the exam-file reader (dnncancerannotator_amd/tfrecord.py; no device involved) ------------------------
 * CRC-32C (Castagnoli) of `n` bytes: tf.data.TFRecordDataset (annotator/data.py:448-470) checks the masked CRC-32C of every record
 * length and payload; the SSE4.2 crc32 instruction where the host has it */
int dnnca_crc32c(const void* data, size_t n, uint32_t* crc_out);

/* ---- input pipeline: what `ds.prefetch(AUTOTUNE)` (annotator/data.py:110,143) + Keras fit's asynchronous input feeding
 * (engine.py:126-135) do for the reference.  A ring of `slots` (<= 8) staging slots in HBM and a copy stream: the next
 * batch travels host -> HBM while the main stream still works on the previous step, and the step outputs come back through a
 * pinned host ring, so the host never has to wait for the step it has just enqueued.
 *   dnnca_stage_init           once per model; bytes_per_slot 0 = one float batch (x, y) at max_batch
 *   dnnca_stage_upload         copies host_a (and host_b right behind it, 256-byte aligned) into the slot on the copy stream, first
 *                              waiting for the step that consumed the slot's previous content; returns the device addresses.
 *                              May run on other host threads beside the thread that enqueues steps (each on slots of its own).
 *   dnnca_stage_uploaded       blocks the calling host thread until the slot's upload has completed: the host buffers may then be
 *                              reused or freed (hipMemcpyAsync from pageable memory gives no such promise on return)
 *   dnnca_stage_wait           the model's stream waits for the slot's upload (for work other than the step, e.g. dnnca_augment_u8)
 *   dnnca_train_step_staged    = dnnca_stage_wait + dnnca_train_step_dev(x_dev, y_dev) + outputs to the pinned ring; asynchronous
 *   dnnca_staged_out           waits for the step that last ran on the slot and reads its scalars (label / weight assertions of
 *                              utils/losses.py:30,91-92 surface here, i.e. one fetch later than with dnnca_train_step) */
int dnnca_stage_init(void* model, int slots, size_t bytes_per_slot);
int dnnca_stage_upload(void* model, int slot, const void* host_a, size_t bytes_a, const void* host_b, size_t bytes_b,
                       void** a_dev, void** b_dev);
int dnnca_stage_uploaded(void* model, int slot);
int dnnca_stage_wait(void* model, int slot);
int dnnca_train_step_staged(void* model, int slot, const float* x_dev, const float* y_dev, int batch, float lr,
                            const dnnca_loss_cfg* cfg);
int dnnca_staged_out(void* model, int slot, dnnca_step_out* out);
/* keras Model.evaluate (engine.py:198-203) over the staging ring: test steps (forward with training=False + loss) on staged
 * batches; the pixel TP/FP/FN/TN histogram of the n thresholds (all metrics' thresholds at once, any order, n may be 0) stays
 * on the device and keeps adding up over the batches -- exact integer counts, read once by dnnca_eval_end.  Per-batch losses come
 * back through dnnca_staged_out (rank-local, like dnnca_eval_step).  dnnca_pixel_confusion* must not be called in between. */
int dnnca_eval_begin(void* model, const float* thresholds, int n);
int dnnca_eval_step_staged(void* model, int slot, const float* x_dev, const float* y_dev, int batch, const dnnca_loss_cfg* cfg);
int dnnca_eval_end(void* model, dnnca_confusion* out /* n entries, the caller's threshold order */);

/* pixel TP/FP/FN/TN of the last forward/eval probabilities against y at n thresholds (metrics.yaml:2-23 pixel metrics;
 * utils/metrics.py:37-77 FBetaScore builds on them). y_hw is a host buffer [B,H,W]. */
int dnnca_pixel_confusion(void* model, const float* y_hw, int batch, const float* thresholds, int n, dnnca_confusion* out);
/* the same counts for probabilities the caller supplies: tf.keras.metrics.{Precision,Recall,AUC}.update_state(y_true, y_pred)
 * as utils/metrics.py:53-56 calls it.  Both host buffers hold n_pixels floats (at most max_batch * H * W); any number of
 * thresholds up to 1024 in any order (AUC(num_thresholds=150), metrics.yaml:8-15); counts are exact integers. */
int dnnca_pixel_confusion_of(void* model, const float* prob_hw, const float* y_hw, int64_t n_pixels, const float* thresholds,
                             int n, dnnca_confusion* out);

/* ---- data parallel: tf.distribute.MirroredStrategy (engine.py:260-263) re-done as one process per GPU + RCCL ---- */
int dnnca_comm_unique_id(void* id_out /* DNNCA_UNIQUE_ID_BYTES */);
int dnnca_comm_init(void* model, int rank, int world, const void* unique_id, size_t id_len);   /* world == 1: no-op */
int dnnca_comm_world(void* model, int* rank, int* world);
/* gradient all-reduce calls the last train step issued: 1, or several when a gradient vector above 1 MB was sent in buckets
 * (reverse layer order, second stream) while the backward pass was still running -- bit-identical to the single call */
int dnnca_comm_collectives(void* model, int* count);
int dnnca_comm_broadcast_weights(void* model, int root);   /* weights, BN statistics and Adam slots of `root` on every rank */
int dnnca_comm_average_state(void* model);          /* BN moving statistics: mean over ranks before a checkpoint */
/* small host-side reductions (validation loss sums, metric counts: MirroredStrategy's metric aggregation); doubles, any length */
int dnnca_comm_allreduce_host(void* model, double* values, int64_t n, int op /* 0 sum, 1 max */);

/* ---- measurement: HIP events on the model's stream ------------------------------------------------------------- */
int dnnca_timer_start(void* model);
int dnnca_timer_stop(void* model, float* elapsed_ms);  /* synchronises */
/* per-kernel accounting: when enabled every launch is bracketed by HIP events on the model's stream */
int dnnca_profile_enable(void* model, int mode /* 0 off, 1 all kernels, 2 only the kernel named by dnnca_profile_focus */);
int dnnca_profile_focus(void* model, const char* kernel_name);
/* mode 2 only: bracket the focus kernel in one train step out of `period` (>= 1), so that the brackets of a timed region cost
   next to nothing; every bracketed launch is a full HIP-event measurement on the launch stream */
int dnnca_profile_sample(void* model, int period);
int dnnca_profile_reset(void* model);
int dnnca_profile_count(void* model, int* count);
int dnnca_profile_get(void* model, int index, char* name, size_t name_cap, int64_t* launches, double* total_ms,
                      double* algorithmic_bytes, double* flops);   /* bytes/flops are per launch (mean) */
/* the launch schedule of one train step: name + algorithmic bytes/flops per launch (for DESIGN.md and bench.py) */
int dnnca_plan_dump(void* model, char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* DNNCA_H */
