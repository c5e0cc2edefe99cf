/* wg_densify.h -- C-ABI of the fused densification statistics ("the step after").
 *
 * Replaces, as an opt-in for callers, what the training loop does with the operator's outputs after every backward pass
 * while densification is on (wildgaussians/method.py:1995-1998 and GaussianModel.add_densification_stats, :1470-1477):
 *
 *     visibility_filter = radii > 0
 *     max_radii2D[vis]              = max(max_radii2D[vis], radii[vis])
 *     xyz_grad[vis]                += norm(viewspace_points.grad[vis, :2])
 *     xyz_gradient_accum_abs[vis]  += norm(viewspace_points.grad[vis, 2:])          (use_gof_abs_gradient)
 *     xyz_gradient_accum_abs_max[vis] = max(xyz_gradient_accum_abs_max[vis], norm(viewspace_points.grad[vis, 2:]))
 *     denom[vis]                   += 1
 *
 * -- six boolean-mask index operations (each a nonzero() with a host synchronisation, a gather and an index_put) in
 * PyTorch, one streaming kernel here: 16 B in and up to 40 B read-modify-write per visible Gaussian, no synchronisation.
 * float32 / int32 device pointers, all arrays of P elements except viewspace_grad (P x 3); the two *_abs arrays may both be
 * NULL (use_gof_abs_gradient off), max_radii2D may be NULL.  Returns 0 or a negative wg_status.
 */
#ifndef WG_DENSIFY_H
#define WG_DENSIFY_H
#ifdef __cplusplus
extern "C" {
#endif

int wg_densification_stats(int P, const int* radii, const float* viewspace_grad, float* xyz_grad, float* xyz_gradient_accum_abs,
                           float* xyz_gradient_accum_abs_max, float* denom, float* max_radii2D, void* stream);

#ifdef __cplusplus
}
#endif
#endif
