/* wg_sh_eval.h -- C-ABI of the fused spherical-harmonics colour evaluation ("the step before the operator").
 *
 * Replaces, as an opt-in for callers, the reference's `eval_sh(deg, sh, dirs)` (wildgaussians/method.py:493-548) as the caller uses
 * it before each rasterizer call (method.py:1555-1565, :1596-1598): in PyTorch that is ~60 elementwise kernels over strided
 * `sh[..., k]` slices forward and, through autograd, a P x 3 x K zero-fill + slice-add per coefficient backward; here one streaming
 * kernel each way.
 *
 *   sh    [P, 3, K] float32, K >= (deg + 1)^2 coefficients per channel (the reference's `features.view(-1, K, 3).transpose(1, 2)`)
 *   dirs  [P, 3]    float32 (unit) view directions
 *   out   [P, 3]    out[p][c] = sum_k basis_k(dirs[p]) * sh[p][c][k],  k < (deg + 1)^2,  deg in 0..3 (the reference's hard-coded
 *                   real SH polynomials, same constants and sign conventions)
 * Backward: grad_sh [P, 3, K] is fully overwritten (zeros for k >= (deg + 1)^2, as autograd of the slices gives);
 * grad_dirs [P, 3] (may be NULL) is the derivative of the polynomials with x, y, z as independent variables, as autograd of the
 * reference's expression gives it.  Device pointers, explicit HIP stream.  Returns 0 or a negative wg_status (wg_rasterizer.h).
 */
#ifndef WG_SH_EVAL_H
#define WG_SH_EVAL_H
#ifdef __cplusplus
extern "C" {
#endif

int wg_eval_sh_forward(int P, int deg, int K, const float* sh, const float* dirs, float* out, void* stream);
int wg_eval_sh_backward(int P, int deg, int K, const float* sh, const float* dirs, const float* grad_out, float* grad_sh, float* grad_dirs,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif
