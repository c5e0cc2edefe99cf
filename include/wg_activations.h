/* wg_activations.h -- C-ABI of the fused Gaussian activations + 3-D smoothing filter ("the step before").
 *
 * Replaces, as an opt-in for callers, GaussianModel.get_gaussians (wildgaussians/method.py:1060-1086): rotation
 * normalisation (F.normalize, eps 1e-12), exp of the log-scales, sigmoid of the opacity logits, the Mip-Splatting 3-D filter
 * (scales = sqrt(s^2 + f^2), opacity *= sqrt(prod s^2 / prod (s^2 + f^2))) -- about 15 elementwise kernels forward and 30
 * backward in PyTorch, one kernel each here.  float32 device pointers, explicit HIP stream; 0 or a negative wg_status.
 */
#ifndef WG_ACTIVATIONS_H
#define WG_ACTIVATIONS_H
#ifdef __cplusplus
extern "C" {
#endif

/* raw_rotations[P,4], raw_scales[P,3] (log), raw_opacities[P] (logit), filter_3D[P]  ->  rotations[P,4], scales[P,3], opacities[P] */
int wg_activations_forward(int P, const float* raw_rotations, const float* raw_scales, const float* raw_opacities,
                           const float* filter_3D, float* rotations, float* scales, float* opacities, void* stream);

/* gradients w.r.t. the three outputs -> gradients w.r.t. the three raw inputs (overwritten); filter_3D is a constant.
 * Any of the three incoming gradients may be NULL (= zero). */
int wg_activations_backward(int P, const float* raw_rotations, const float* raw_scales, const float* raw_opacities,
                            const float* filter_3D, const float* dL_drotations, const float* dL_dscales, const float* dL_dopacities,
                            float* dL_draw_rotations, float* dL_draw_scales, float* dL_draw_opacities, void* stream);

#ifdef __cplusplus
}
#endif
#endif
