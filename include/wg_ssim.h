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

#ifdef __cplusplus
}
#endif
#endif
