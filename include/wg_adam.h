/* wg_adam.h -- C-ABI of the fused Adam step over all of a model's parameter tensors ("fused Adam ... over P").
 *
 * Replaces, as an opt-in for callers, `self.model.optimizer.step()` of the training loop (wildgaussians/method.py:2019) for the
 * optimizer the reference builds (`torch.optim.Adam(l, lr=1.0, eps=1e-15)` with one parameter group per Gaussian attribute,
 * method.py:1030-1049): torch's default implementation makes about ten passes over every tensor per step, this makes one
 * (16 B read, 12 B written per element), in one launch for up to WG_ADAM_MAX_TENSORS tensors.  Per element, torch's update
 * (torch/optim/adam.py, _single_tensor_adam / the fused kernel's arithmetic), in float32:
 *
 *     g      = grad + weight_decay * param            (L2 regularisation, as torch.optim.Adam; not AdamW)
 *     m      = m + (g - m) * one_minus_beta1
 *     v      = beta2 * v + one_minus_beta2 * g * g
 *     param -= step_size * (m / (sqrt(v) / bias_correction2_sqrt + eps))
 *
 * The per-step scalars are the CALLER's, computed in double precision and rounded once, as torch's Python does (torch/optim/adam.py):
 * one_minus_beta{1,2} = 1 - beta{1,2}, step_size = lr / (1 - beta1^step), bias_correction2_sqrt = sqrt(1 - beta2^step); the step
 * counter lives in the optimizer's state on the
 * host).  amsgrad / maximize are not implemented.  All pointers are float32 device pointers of `numel` elements; `stream` is a
 * hipStream_t.  Returns 0 or a negative wg_status (wg_rasterizer.h).
 */
#ifndef WG_ADAM_H
#define WG_ADAM_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define WG_ADAM_MAX_TENSORS 24 /* per launch; longer lists are processed in several launches */

typedef struct wg_adam_tensor {
    float* param;            /* updated in place */
    const float* grad;
    float* exp_avg;          /* m, updated in place */
    float* exp_avg_sq;       /* v, updated in place */
    size_t numel;
    float beta2, one_minus_beta1, one_minus_beta2;
    float step_size;              /* lr / (1 - beta1^step) */
    float bias_correction2_sqrt;  /* sqrt(1 - beta2^step) */
    float eps, weight_decay;
} wg_adam_tensor;

int wg_fused_adam(int n_tensors, const wg_adam_tensor* tensors /* host array */, void* stream);

#ifdef __cplusplus
}
#endif
#endif
