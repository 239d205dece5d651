// kernels.h -- host-side launchers of the HIP kernels.  All tensors are NHWC float32 views (common.h).
#pragma once
#include "common.h"

namespace dnnca {

// ----------------------------------------------------------------------------- generic (any shape, untuned) kernels
// y = act(conv_kxk(concat(A, Bv)) + bias); Bv.C may be 0.  Weight layout HWIO with I = A.C + Bv.C.
// alpha < 0 -> no activation, alpha == 0 -> relu, alpha > 0 -> leaky relu.
void g_conv_fwd(hipStream_t s, int B, View A, View Bv, const float* w, const float* bias, View out, int K, float alpha);
// dz = dy * act'(y) in place on dy (dense tensors)
void g_act_bwd(hipStream_t s, size_t n, float* dy, const float* y, float alpha);
// dA / dB (beta 0 = overwrite, 1 = accumulate) from dz (view of the conv output gradient, C = Cout)
void g_conv_dgrad(hipStream_t s, int B, View dz, const float* w, View dA, int accA, View dB, int accB, int K);
// dW (HWIO) += ..., dbias += ...   (atomic accumulation into pre-zeroed buffers)
void g_conv_wgrad(hipStream_t s, int B, View A, View Bv, View dz, float* dw, float* dbias, int K);

void g_pool_fwd(hipStream_t s, int B, View in, View out, int r);
// din = (acc ? din : 0) + route(dout) ; first maximum in row-major window order receives the gradient
void g_pool_bwd(hipStream_t s, int B, View in, View out, View dout, View din, int acc, int r);

// Conv2DTranspose k = s = r; weight [r, r, Cout, Cin]
void g_tconv_fwd(hipStream_t s, int B, View in, const float* w, const float* bias, View out, int r);
void g_tconv_dgrad(hipStream_t s, int B, View dout, const float* w, View din, int acc, int r);
void g_tconv_wgrad(hipStream_t s, int B, View in, View dout, float* dw, float* dbias, int r);

// BatchNormalization.  ws: 4*C doubles of scratch (sum, sumsq-centred, dgamma, dbeta) zeroed by the caller;
// coef: [scale(C), shift(C), mean(C), inv(C)] floats.
void g_bn_stats_mean(hipStream_t s, int B, View x, double* ws);
void g_bn_stats_var(hipStream_t s, int B, View x, double* ws);   // uses mean = ws[c]/n
void g_bn_finalize(hipStream_t s, int C, double n, const double* ws, const float* gamma, const float* beta,
                   float* mmean, float* mvar, float* coef, int training, float momentum, float eps);
void g_bn_apply(hipStream_t s, int B, View x, View y, const float* coef);
void g_bn_bwd_reduce(hipStream_t s, int B, View x, View dy, const float* coef, float* dgamma, float* dbeta);
void g_bn_bwd_apply(hipStream_t s, int B, View x, View dy, View dx, int acc, const float* coef, const float* gamma,
                    const float* dgamma, const float* dbeta, double n);

// head: logits[b,y,x] = sum_c feat*w[c] + bias  (Conv2D(1, 1), unet.py:241-244)
void g_head_fwd(hipStream_t s, int B, View feat, const float* w, const float* bias, float* logits);
void g_head_bwd(hipStream_t s, int B, View feat, const float* w, const float* dlogits, View dfeat, float* dw, float* dbias);

// scalars layout (doubles): see model.hip kScalar*
void g_label_stats(hipStream_t s, size_t n, const float* y, double* scalars);
// loss + dlogits + probabilities.  inv_count = 1/(H*W*B*replicas) for dlogits; per-pixel loss summed into scalars.
void g_loss(hipStream_t s, size_t n, const float* logits, const float* y, const dnnca_loss_cfg cfg, double n_label,
            double* scalars, float* dlogits, float* prob, float grad_scale);
void g_sigmoid(hipStream_t s, size_t n, const float* logits, float* prob);

void g_l2(hipStream_t s, size_t n, const float* w, float* g, float l2, double* scalars);   // g += 2*l2*w ; penalty += l2*sum w^2
// writes [loss, positive_rate, weight, ymin, ymax] floats to out5 (device) from the scalar block
void g_finalize_scalars(hipStream_t s, double* scalars, const dnnca_loss_cfg cfg, double n_label, double inv_batch_hw,
                        float* out5);
void g_adam(hipStream_t s, size_t n, float* p, const float* g, float* m, float* v, float lr_t, float b1, float b2,
            float eps, float gscale);
void g_adam_finalize(hipStream_t s, size_t n, float* p, const float* g, float* m, float* v, float lr_t, float b1, float b2, float eps,
                     float gscale, double* scalars, const dnnca_loss_cfg cfg, double n_label, double inv_batch_hw, float* out5);
constexpr int DNNCA_CONF_MAX_THR = 1024;
void g_confusion_hist(hipStream_t s, size_t n, const float* prob, const float* y, const float* thr_sorted, int nthr,
                      unsigned long long* hist /* [2][nthr + 1], zeroed */);
void g_scale(hipStream_t s, size_t n, float* p, float a);
// zeroes a[0..na) and b[0..nb) (either may be empty) and resets the scalar block, in one launch
void g_step_init(hipStream_t s, double* scalars, float* a, size_t na, float* b, size_t nb);

}  // namespace dnnca
