/* wg_ssim.h -- C-ABI of the fused SSIM map ("the step after" the rasterizer).
 *
 * Replaces, as an opt-in for callers, the reference's `ssim()` (wildgaussians/method.py:644-673): five depthwise 11x11
 * Gaussian-window convolutions (sigma 1.5, zero padding 5) of img1, img2, img1^2, img2^2, img1*img2, the SSIM formula with
 * C1 = 0.01^2, C2 = 0.03^2, and -- in the reference -- the autograd graph through all of it.  Here: one forward kernel that
 * writes the per-channel SSIM map and the three partial-derivative maps the backward needs, one backward kernel.
 * Layout: planar float32 [C, H, W], device pointers, explicit HIP stream.  Returns 0 or a negative wg_status (wg_rasterizer.h).
 */
#ifndef WG_SSIM_H
#define WG_SSIM_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ssim_map[C*H*W] (required); dm_dmu1, dm_dsigma1_sq, dm_dsigma12 [C*H*W] each, all three or none (NULL when no gradient
 * will be asked for). */
int wg_ssim_forward(int C, int H, int W, const float* img1, const float* img2, float* ssim_map, float* dm_dmu1,
                    float* dm_dsigma1_sq, float* dm_dsigma12, void* stream);

/* dL_dmap[C*H*W] = gradient w.r.t. ssim_map; dL_dimg1[C*H*W] is overwritten.  img2 is treated as a constant (the ground
 * truth), as in the reference's use (method.py:1949). */
int wg_ssim_backward(int C, int H, int W, const float* img1, const float* img2, const float* dL_dmap, const float* dm_dmu1,
                     const float* dm_dsigma1_sq, const float* dm_dsigma12, float* dL_dimg1, void* stream);

/*
 * The whole image loss of the reference's training step in two launches each way (wildgaussians/method.py:1948-1965):
 *
 *     loss = (1 - lambda) * mean(|img_l1 - gt| * mult) + lambda * mean((1 - ssim_map(img_ssim, gt)) * mult)
 *
 * with both means over all C*H*W elements (`Ll1` is [C,H,W]; `ssim(..., size_average=False)` is the channel mean, so its pixel
 * mean is the same sum / (C*H*W)), `mult` the reference's per-pixel `loss_mult` ([H*W], broadcast over channels; NULL = 1).  The
 * reference applies L1 to the appearance-toned render and SSIM to the raw one: img_l1 and img_ssim may be different images or the
 * same pointer.  Forward: loss_out[3] (device) = {loss, mean |img_l1 - gt| * mult, 1 - mean((1 - ssim) * mult)}; `scratch` must hold
 * wg_l1_ssim_loss_scratch_floats(C, H, W) floats (per-workgroup partial sums, added in a fixed order: the loss is bit-reproducible);
 * dm_* as in wg_ssim_forward (all three or none).  Backward: dL_dloss is a DEVICE scalar (autograd's grad_output); dL_dimg_l1 and
 * dL_dimg_ssim [C*H*W] are overwritten -- pass the same pointer for both exactly when img_l1 == img_ssim, and the two terms'
 * gradients are summed into it.  gt and mult are constants.
 */
size_t wg_l1_ssim_loss_scratch_floats(int C, int H, int W);
int wg_l1_ssim_loss_forward(int C, int H, int W, const float* img_l1, const float* img_ssim, const float* gt, const float* mult,
                            float lambda, float* scratch, float* loss_out, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12,
                            void* stream);
int wg_l1_ssim_loss_backward(int C, int H, int W, const float* img_l1, const float* img_ssim, const float* gt, const float* mult,
                             float lambda, const float* dL_dloss, const float* dm_dmu1, const float* dm_dsigma1_sq,
                             const float* dm_dsigma12, float* dL_dimg_l1, float* dL_dimg_ssim, void* stream);

#ifdef __cplusplus
}
#endif
#endif
