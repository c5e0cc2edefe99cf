/*
 * wg_rasterizer.h -- C-ABI of the MI355X-native differentiable Gaussian-splat rasterizer.
 *
 * This is the drop-in boundary for the hot path of jkulhanek/wild-gaussians: it replaces the static
 * C++ interface CudaRasterizer::Rasterizer::{forward,backward,markVisible}
 * (submodules/diff-gaussian-rasterization/cuda_rasterizer/rasterizer.h:24-88), i.e. what the
 * reference's torch binding (rasterize_points.cu:35-225) calls.  Differences from that interface,
 * all forced by "plain C, no C++/torch types in the signature":
 *
 *   - std::function<char*(size_t)> allocators become (function pointer, void* user) pairs;
 *   - every entry point takes the HIP stream to launch on (the reference uses the legacy default
 *     stream, rasterizer_impl.cu:148,292,...); pass NULL for the default stream;
 *   - errors are returned as negative wg_status codes instead of C++ exceptions
 *     (std::runtime_error at rasterizer_impl.cu:244-247 and auxiliary.h:166-173);
 *   - bool becomes int / unsigned char.
 *
 * All pointers are DEVICE pointers to contiguous float32 / int32 data unless stated otherwise.  A NULL
 * pointer means "not provided" exactly as in the reference (shs vs colors_precomp, scales+rotations
 * vs cov3D_precomp; forward.cu:217,253, backward.cu:426,430).  Matrices are the reference's layout:
 * viewmatrix = W2C transposed, projmatrix = (P*W2C) transposed, row-major (auxiliary.h:58-77).
 *
 * The three scratch buffers (geometry / binning / image state) are opaque; their layout is private to
 * this library (it differs from the reference's GeometryState/BinningState/ImageState).  The one
 * documented property: the image-state buffer starts, at its first 256-byte-aligned address, with
 * final_T as float[H*W] (the transmittance left at each pixel), so that
 * accumulation = 1 - final_T can be read back by the caller like the reference's Python wrapper does
 * (diff_gaussian_rasterization/__init__.py:101-113) -- and, at the next 256-byte-aligned address behind
 * final_T (wg_image_accumulation_offset()), holds that very array, accumulation as float[H*W], written by
 * the forward pass itself: a caller can hand out a view instead of running an elementwise kernel.
 */
#ifndef WG_RASTERIZER_H_INCLUDED
#define WG_RASTERIZER_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WG_TILE_X 16 /* config.h:15 */
#define WG_TILE_Y 16 /* config.h:16 */
#define WG_NUM_CHANNELS 3 /* config.h:14 */

typedef enum wg_status {
    WG_OK = 0,
    WG_ERR_INVALID_ARGUMENT = -1, /* bad sizes / missing mandatory pointer / both-or-neither optional inputs */
    WG_ERR_ALLOC = -2,            /* an allocator callback returned NULL */
    WG_ERR_HIP = -3,              /* a HIP runtime call or kernel launch failed (wg_last_hip_error() has the text) */
    WG_ERR_OVERFLOW = -4,         /* more than 2^31-1 (tile, Gaussian) instances */
    WG_ERR_SPECULATION = -5       /* option "speculative_forward" = 2 only: the calling thread's PREVIOUS forward call did not fit the binning
                                     buffer it had predicted; that call's image is NaN and its gradients are zero -- repeat the step */
} wg_status;

/* Replaces std::function<char*(size_t N)> (rasterizer.h:34-36, rasterize_points.cu:27-33): must return a
 * device pointer to at least `bytes` bytes that stays valid until the matching backward call. */
typedef char* (*wg_alloc_fn)(size_t bytes, void* user);

/* Scratch sizes (bytes), for callers that preallocate (the binning size is an upper bound). */
size_t wg_geometry_buffer_size(int P);
size_t wg_image_buffer_size(int width, int height);
size_t wg_binning_buffer_size(int num_rendered);
/* Byte offset of accumulation[H*W] from final_T (= from the buffer's first 256-byte-aligned address). */
size_t wg_image_accumulation_offset(int width, int height);

/*
 * Rasterizer::forward (rasterizer.h:33-59, rasterizer_impl.cu:198-340).
 * Returns num_rendered (>= 0) = number of (tile, Gaussian) instances, or a negative wg_status.
 * out_color: float[3*H*W] planar CHW.  radii: int[P] or NULL.  subpixel_offset: float[H*W*2] or NULL (beyond the reference:
 * NULL = all zero, nothing is read; the same in wg_rasterize_backward).  One host<->device rendezvous (the read-back
 * of num_rendered that sizes the binning buffer, as rasterizer_impl.cu:284) -- but, by default ("speculative_forward"),
 * behind the call's LAST launch, not in its middle: the count is predicted from the calling thread's recent frames of the same
 * shape, the binning buffer is requested with a margin and every kernel behind the count is enqueued at once, guarded on the
 * device by the verdict the tile scan leaves; a frame that does not fit runs none of them and the call re-issues the tail with
 * the real sizes.  binning_alloc may therefore be called TWICE in one forward call; only the buffer returned last is used (and
 * must be the one handed to wg_rasterize_backward).  Results are identical either way.
 */
int wg_rasterize_forward(wg_alloc_fn geometry_alloc, void* geometry_user,
                         wg_alloc_fn binning_alloc, void* binning_user,
                         wg_alloc_fn image_alloc, void* image_user,
                         int P, int D, int M,
                         const float* background,
                         int width, int height,
                         const float* means3D,
                         const float* shs,
                         const float* colors_precomp,
                         const float* opacities,
                         const float* scales,
                         float scale_modifier,
                         const float* rotations,
                         const float* cov3D_precomp,
                         const float* viewmatrix,
                         const float* projmatrix,
                         const float* cam_pos,
                         float tan_fovx, float tan_fovy,
                         float kernel_size,
                         const float* subpixel_offset,
                         int prefiltered,
                         float* out_color,
                         int* radii,
                         int debug,
                         void* stream);

/*
 * Rasterizer::backward (rasterizer.h:61-88, rasterizer_impl.cu:344-443).
 * The reference requires all nine gradient outputs zero-filled by the caller (rasterize_points.cu:157-165).  Here, by default
 * (option "grad_record" = 1), ALL nine are fully overwritten (zeros for culled Gaussians): the per-tile pass accumulates into a
 * 48-byte record per Gaussian inside geom_buffer, which this call clears itself, and the per-Gaussian kernel writes dL_dmean2D,
 * dL_dconic, dL_dopacity and dL_dcolor from it -- a caller following the reference's protocol (zeroed buffers) gets the same
 * values.  dL_dconic (an intermediate of the reference) and, with SH colours, dL_dcolor (the gradient of the evaluated RGB, an
 * intermediate there) may then be NULL: they are not written.  With "grad_record" = 0 those four ARE the accumulation targets,
 * none may be NULL and all must be zero on entry.
 * dL_dconic is float[P*4], 16-byte aligned (2x2 per Gaussian; [0],[1],[3] used), dL_dmean2D float[P*3]
 * (x, y in NDC-scaled units, z = abs-gradient, backward.cu:590-595).  dL_dsh may be NULL when M == 0.
 */
int wg_rasterize_backward(int P, int D, int M, int R,
                          const float* background,
                          int width, int height,
                          const float* means3D,
                          const float* shs,
                          const float* colors_precomp,
                          const float* scales,
                          float scale_modifier,
                          const float* rotations,
                          const float* cov3D_precomp,
                          const float* viewmatrix,
                          const float* projmatrix,
                          const float* campos,
                          float tan_fovx, float tan_fovy,
                          float kernel_size,
                          const float* subpixel_offset,
                          const int* radii,
                          char* geom_buffer,
                          char* binning_buffer,
                          char* image_buffer,
                          const float* dL_dpix,
                          float* dL_dmean2D,
                          float* dL_dconic,
                          float* dL_dopacity,
                          float* dL_dcolor,
                          float* dL_dmean3D,
                          float* dL_dcov3D,
                          float* dL_dsh,
                          float* dL_dscale,
                          float* dL_drot,
                          int debug,
                          void* stream);

/*
 * Beyond the reference ("the appearance MLP output feeds SH eval directly"): an optional per-Gaussian affine
 * on the SH coefficients, applied inside the preprocess kernel and differentiated inside the preprocess-backward kernel:
 *
 *     x = min(shs[i][k][c], pre_clamp_max);   t = x * mul[i][c] + (k == 0 ? offset[i][c] : 0);   used = min(t, post_clamp_max)
 *
 * (separate multiply and add, so that the values equal what the PyTorch chain it replaces computes:
 * `(features.clamp_max(1) * mul.repeat(1, 16) + cat(offset / C0, 0)).clamp_max(1)`, wildgaussians/method.py:890-900, 1590-1595,
 * with `offset` here = the caller's offset / C0).  INFINITY switches a clamp off; mul / offset may be NULL (1 / 0).
 * The *_toned entry points take the same arguments as the plain ones plus this block; tone == NULL is the plain call.
 * Backward: dL_dsh is then the gradient w.r.t. the RAW coefficients, and dL_dmul / dL_doffset ([P,3], overwritten, zeros for
 * culled Gaussians) receive the gradients of the two affine inputs.  Only with SH colours (shs != NULL).
 */
typedef struct wg_sh_tone {
    const float* mul;      /* [P,3] or NULL */
    const float* offset;   /* [P,3] or NULL */
    float pre_clamp_max;
    float post_clamp_max;
    float* dL_dmul;        /* backward only; required when mul != NULL */
    float* dL_doffset;     /* backward only; required when offset != NULL */
} wg_sh_tone;

int wg_rasterize_forward_toned(wg_alloc_fn geometry_alloc, void* geometry_user, wg_alloc_fn binning_alloc, void* binning_user,
                               wg_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background, int width,
                               int height, const float* means3D, const float* shs, const float* colors_precomp,
                               const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                               const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                               float tan_fovx, float tan_fovy, float kernel_size, const float* subpixel_offset, int prefiltered,
                               float* out_color, int* radii, int debug, void* stream, const wg_sh_tone* tone);

int wg_rasterize_backward_toned(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                                const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                                const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                                const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, float kernel_size,
                                const float* subpixel_offset, const int* radii, char* geom_buffer, char* binning_buffer,
                                char* image_buffer, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                                float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                                float* dL_drot, int debug, void* stream, const wg_sh_tone* tone);

/*
 * Beyond the reference ("fuse the step before the operator"): the caller's `get_gaussians()` -- rotation
 * normalisation, exp / sigmoid activations and the 3-D filter on scales and opacities (wildgaussians/method.py:1060-1086) -- evaluated
 * INSIDE the preprocess kernels instead of by a chain of P-sized elementwise passes in front of the operator (and their backward passes
 * behind it).  The *_raw entry points take the arguments of the *_toned ones (tone may be NULL) plus this block; then
 *   forward : `opacities` [P], `scales` [P,3], `rotations` [P,4] are the RAW parameters (logit, log-scale, unnormalised quaternion);
 *             the kernels use  q = r / max(|r|, 1e-12),  s = sqrt(exp(ls)^2 + f^2),  o = sigmoid(lo) sqrt(prod exp(ls)^2 / prod s^2).
 *   backward: `scales`, `rotations` raw as in the forward call, raw_opacities the forward call's `opacities`; dL_dopacity, dL_dscale,
 *             dL_drot receive the gradients of the RAW parameters.  Needs the gradient record ("grad_record" = 1, the default).
 * Scale / rotation pairs only (cov3D_precomp == NULL).  Same device functions as wg_activations_forward / _backward
 * (include/wg_activations.h), compiled with the same flags: the frame equals the one of activations + plain call, bit for bit.
 */
typedef struct wg_raw_gaussians {
    const float* filter_3D;       /* [P] */
    const float* raw_opacities;   /* [P], backward only */
} wg_raw_gaussians;

int wg_rasterize_forward_raw(wg_alloc_fn geometry_alloc, void* geometry_user, wg_alloc_fn binning_alloc, void* binning_user,
                             wg_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background, int width,
                             int height, const float* means3D, const float* shs, const float* colors_precomp,
                             const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                             const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                             float tan_fovx, float tan_fovy, float kernel_size, const float* subpixel_offset, int prefiltered,
                             float* out_color, int* radii, int debug, void* stream, const wg_sh_tone* tone, const wg_raw_gaussians* raw);

int wg_rasterize_backward_raw(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                              const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                              const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                              const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, float kernel_size,
                              const float* subpixel_offset, const int* radii, char* geom_buffer, char* binning_buffer,
                              char* image_buffer, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                              float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                              float* dL_drot, int debug, void* stream, const wg_sh_tone* tone, const wg_raw_gaussians* raw);

/*
 * Beyond the reference: TWO colour sets composited in ONE call -- one projection, one binning, one forward walk and one backward walk for
 * both.  WildGaussians rasterizes raw and toned colours over identical geometry in every training step
 * (wildgaussians/method.py:1573-1611: 2 forward + 2 backward passes of the reference); the per-pixel decisions (alpha, transmittance,
 * n_contrib) do not depend on the colours, so the second set only adds three sums per pixel forward and three per (tile, Gaussian)
 * instance backward.  Precomputed colours only (shs == NULL).  The *_dual entry points take the arguments of the plain ones plus this
 * block; both images sit on the same background.
 *   forward : colors_precomp2 [P,3] in, out_color2 float[3*H*W] out (fully written, like out_color).
 *   backward: dL_dpix2 float[3*H*W] in (the cotangent of out_color2; pass zeros if it took none), dL_dcolor2 [P,3] out (fully written).
 *             dL_dmean2D / dL_dopacity / dL_dmean3D / dL_dcov3D / dL_dscale / dL_drot are the gradients of BOTH images' losses: what the
 *             reference's two calls give after autograd adds them, up to float rounding.  Needs the gradient record ("grad_record" = 1,
 *             the default, or "deterministic_backward" = 1: bit-reproducible, fourteen-float slots); WG_ERR_INVALID_ARGUMENT otherwise.
 */
typedef struct wg_second_colors {
    const float* colors_precomp2;   /* forward */
    float* out_color2;              /* forward */
    const float* dL_dpix2;          /* backward */
    float* dL_dcolor2;              /* backward */
} wg_second_colors;

int wg_rasterize_forward_dual(wg_alloc_fn geometry_alloc, void* geometry_user, wg_alloc_fn binning_alloc, void* binning_user,
                              wg_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background, int width,
                              int height, const float* means3D, const float* shs, const float* colors_precomp,
                              const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                              const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                              float tan_fovx, float tan_fovy, float kernel_size, const float* subpixel_offset, int prefiltered,
                              float* out_color, int* radii, int debug, void* stream, const wg_second_colors* second);

int wg_rasterize_backward_dual(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                               const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                               const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                               const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, float kernel_size,
                               const float* subpixel_offset, const int* radii, char* geom_buffer, char* binning_buffer,
                               char* image_buffer, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                               float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                               float* dL_drot, int debug, void* stream, const wg_second_colors* second);

/*
 * Beyond the reference: the two colour sets of one call, BOTH evaluated from the same SH coefficients -- WildGaussians' training step
 * whole (wildgaussians/method.py:1573-1611 with the appearance toning in the operator): `out_color` is the coefficients through `tone`,
 * `out_color2` the same coefficients through `tone2` (either may be NULL: no affine, no clamp), one projection, one binning, one
 * forward walk, one backward walk, one read of the coefficients per pass.  SH colours only (shs != NULL, colors_precomp == NULL); `raw`
 * (may be NULL) as in the *_raw entry points.  The frame equals the one two wg_rasterize_forward_toned (_raw) calls give, bit for bit.
 *   backward: dL_dpix2 = the cotangent of out_color2 (zeros if it took none); dL_dsh is the gradient of BOTH images' losses w.r.t. the raw
 *             coefficients, tone->dL_dmul / dL_doffset and tone2->dL_dmul / dL_doffset those of each tone's inputs; dL_dcolor and
 *             dL_dcolor2 (the gradients of the two evaluated RGB sets) are intermediates and may be NULL; the geometry gradients are
 *             those of both losses.  Needs the gradient record ("grad_record" = 1, the default, or "deterministic_backward" = 1).
 * Bits 3-5 of the per-Gaussian colour-clamp byte of the geometry buffer hold the second set's flags.
 */
int wg_rasterize_forward_two_tone(wg_alloc_fn geometry_alloc, void* geometry_user, wg_alloc_fn binning_alloc, void* binning_user,
                                  wg_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background, int width,
                                  int height, const float* means3D, const float* shs, const float* colors_precomp,
                                  const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                                  const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                                  float tan_fovx, float tan_fovy, float kernel_size, const float* subpixel_offset, int prefiltered,
                                  float* out_color, int* radii, int debug, void* stream, const wg_sh_tone* tone, const wg_sh_tone* tone2,
                                  const wg_raw_gaussians* raw, float* out_color2);

int wg_rasterize_backward_two_tone(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                                   const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                                   const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                                   const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, float kernel_size,
                                   const float* subpixel_offset, const int* radii, char* geom_buffer, char* binning_buffer,
                                   char* image_buffer, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                                   float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                                   float* dL_drot, int debug, void* stream, const wg_sh_tone* tone, const wg_sh_tone* tone2,
                                   const wg_raw_gaussians* raw, const float* dL_dpix2, float* dL_dcolor2);

/*
 * Beyond the reference: a further rasterization of the SAME Gaussians through the SAME camera with other precomputed colours
 * (WildGaussians renders raw and toned colours over identical geometry in every step, wildgaussians/method.py:1573-1611; the
 * reference projects, bins and sorts twice).  parent_*: the three scratch buffers a wg_rasterize_forward call over that geometry
 * returned (R = its return value) -- they must stay alive and unmodified by the caller; they are read, and the image state is
 * rewritten with the identical per-pixel values.  The call allocates ONE new geometry buffer (its backward pass accumulates into
 * records of its own), copies the projected state into it with the new colours and composites along the parent's sorted lists.
 * out_color, radii (optional) as in wg_rasterize_forward; the image equals what wg_rasterize_forward would give for these colours,
 * bit for bit.  Backward: wg_rasterize_backward with the NEW geometry buffer and the parent's binning and image buffers.
 * Returns R or a negative wg_status.
 */
int wg_rasterize_forward_recolor(wg_alloc_fn geometry_alloc, void* geometry_user, char* parent_geom_buffer, char* parent_binning_buffer,
                                 char* parent_image_buffer, int P, int R, const float* background, int width, int height,
                                 const float* colors_precomp, const float* subpixel_offset, float* out_color, int* radii, void* stream);

/*
 * Beyond the reference: a forward pass WITHOUT any host<->device rendezvous, for callers that capture the step in a hipGraph (or
 * simply must not block).  The caller supplies the binning capacity (instances); the call requests a buffer of that size, enqueues
 * every kernel and returns binning_capacity (hand it to wg_rasterize_backward as R) -- it never learns num_rendered.  Whether the frame
 * fit is decided on the device: when it has more instances than the capacity, none of the kernels behind the count runs, out_color
 * and the accumulation are filled with NaN, and the backward pass of such a frame returns zeros.  wg_forward_status() (a copy + a
 * stream synchronise, to be called outside the capture, e.g. once per step or per epoch) reports num_rendered and the verdict; size
 * the capacity from it with a margin.  Results of a frame that fits are those of wg_rasterize_forward, bit for bit.  Restrictions:
 * frames of at most 36864 tiles, options "lazy_sort" on and "force_global_sort" off (WG_ERR_INVALID_ARGUMENT otherwise); no debug
 * mode; the deterministic backward mode needs the exact count and is not available behind it.
 */
int wg_rasterize_forward_fixed(wg_alloc_fn geometry_alloc, void* geometry_user, wg_alloc_fn binning_alloc, void* binning_user,
                               wg_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background, int width,
                               int height, const float* means3D, const float* shs, const float* colors_precomp,
                               const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                               const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                               float tan_fovx, float tan_fovy, float kernel_size, const float* subpixel_offset, int prefiltered,
                               float* out_color, int* radii, void* stream, const wg_sh_tone* tone, int binning_capacity);
/* num_rendered and fits (1 / 0) of the forward call that produced image_buffer; synchronises the stream. */
int wg_forward_status(char* image_buffer, int width, int height, int* num_rendered, int* fits, void* stream);

/* Rasterizer::markVisible (rasterizer.h:26-31, rasterizer_impl.cu:141-153). present: unsigned char[P]. */
int wg_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                    unsigned char* present, void* stream);

/* ---- introspection used by the parity tests (device pointers into the opaque buffers) ---- */
typedef struct wg_geometry_view {
    const float* depths;          /* [P]   view-space z (forward.cu:262) */
    const int* radii;             /* [P]   internal copy */
    const float* splats;          /* [P*12] 48-byte records: mx,my,conic.x,conic.y | conic.z,opacity*coef,r2,r | g,b,g2,b2  (r2,g2,b2: the second
                                     colour set of a *_dual call; otherwise 0 and two internal floats) */
    const float* cov3D;           /* [P*6] */
    const unsigned char* clamped; /* [P]   bit c set <=> SH colour channel c was clamped at 0 (forward.cu:67-69) */
    const uint32_t* tiles_touched;/* [P] */
    const uint32_t* point_offsets;/* [P]   inclusive prefix sum */
} wg_geometry_view;

typedef struct wg_binning_view {
    const uint32_t* point_list;       /* [R] Gaussian ids sorted by (tile | depth), stable */
} wg_binning_view;

typedef struct wg_image_view {
    const float* final_T;       /* [H*W] */
    const float* accumulation;  /* [H*W] 1 - final_T */
    const uint32_t* n_contrib;  /* [H*W] */
    const uint32_t* ranges;     /* [tiles*2] (start,end) */
    const uint32_t* tile_last;  /* [tiles] max n_contrib over the tile's pixels */
    const uint32_t* tile_near;  /* [tiles] near / far split: near instances of the tile (meaningful only when the split was attempted) */
    const uint32_t* split;      /* [2] {depth-code threshold of the split, 0xffffffff = off; bit b = a tile of XCD band b needed its far instances} */
} wg_image_view;

int wg_view_geometry(char* geom_buffer, int P, wg_geometry_view* out);
int wg_view_binning(char* binning_buffer, int R, wg_binning_view* out);
int wg_view_image(char* image_buffer, int width, int height, wg_image_view* out);

/* ---- per-stage timing with HIP events, recorded on the caller's stream (used by bench.py's roofline) ----
 * wg_profile_enable(1): every subsequent forward/backward call brackets each stage with a pair of events.
 * wg_profile_read(): synchronises the recorded events, adds their durations to the running totals and
 * returns them; wg_profile_reset() clears the totals. */
enum { WG_STAGE_PREPROCESS = 0, WG_STAGE_SCAN, WG_STAGE_DUPLICATE_KEYS, WG_STAGE_SORT, WG_STAGE_TILE_RANGES,
       WG_STAGE_RENDER_FORWARD, WG_STAGE_RENDER_BACKWARD, WG_STAGE_PREPROCESS_BACKWARD, WG_STAGE_RENDER_FIXUP, WG_STAGE_COUNT };
typedef struct wg_stage_times {
    double total_ms[WG_STAGE_COUNT];
    long long launches[WG_STAGE_COUNT];
} wg_stage_times;
int wg_profile_enable(int enable);
int wg_profile_read(wg_stage_times* out);
int wg_profile_reset(void);
const char* wg_stage_name(int stage);

/* Tuning / test switches (process-wide; set them before, not during, calls).  "force_global_sort" (0/1): bin with the
 * rocPRIM global radix sort of 64-bit (tile|depth) keys (the reference's scheme; the automatic fallback for frames of more than
 * 36864 tiles, and for lists longer than 8192 when "lazy_sort" is off) instead of the default counting sort by tile + per-tile
 * register sort.  Both give identical results.
 * "host_mailbox" (1/0, default 1): read num_rendered back through a pinned host mailbox that the device writes and the
 * host polls, instead of a device-to-host copy followed by a stream synchronise.
 * "staged_scatter" (-1 auto / 0 / 1, default auto: on from 1500 instances per tile): lay a workgroup's instances out
 * tile-major in LDS and copy them to the tile buckets in runs, instead of one 4-byte store per instance.  Same buckets.
 * From "band_list_min_p" Gaussians (default 2000000) both scatter kernels read per-XCD-band candidate lists written by the counting kernel
 * (16 B per Gaussian more geometry scratch) instead of scanning whole chunks; set between frames only.
 * ("staged_scatter_cap", n > 0, shrinks the staging area to n entries so that tests reach the multi-pass path; 0 = automatic.)
 * "lazy_sort" (1/0, default 1): when some tile lists more than 5/4 of "lazy_min_len" (256..2048, default 1024) instances, sort only
 * a depth-nearest front of about "lazy_target" (default 820) instances of each long list -- at most "lazy_cap" (default 2048)
 * -- and extend it per tile, in order, only where the forward pass runs past it.  Images, radii, n_contrib and gradients are
 * those of the fully sorted lists; the unsorted tails of the internal lists are simply never read.
 * "depth_codes" (1/0, default 1): with at most 2^24 Gaussians the lazy sort's bucket entries carry a coarse depth code (8 to 12
 * bits, what the ids leave free) above the id, so that the front extraction fetches exact depths only near its bounds; 0
 * exercises the uncoded path, 8..12 force a width (not wider than the ids allow). */
/* "near_split" (-1 automatic / 0 off / 1 whenever possible, default -1): dense frames of large scenes (from "band_list_min_p"
 * Gaussians on or after a dense frame, at 1100 or more instances per tile) first bin, scatter and front-sort only the NEAR instances -- those of the
 * Gaussians below a frame-wide depth-code threshold picked on the device so that about "near_per_tile" (0 = 1.1 x "lazy_target")
 * instances per tile qualify -- and scatter the far ones afterwards only into tiles whose pixels are still accumulating when their
 * near instances are used up.  num_rendered, radii, images, n_contrib and gradients are those of the full binning.
 * In automatic mode a frame in which more than 2 % of the tiles needed the far phase (pixels that do not saturate: low opacities)
 * switches the attempt off for the calling thread's next 64 frames.
 */
/* "deterministic_backward" (0/1, default 0): the per-tile backward pass adds a Gaussian's per-tile terms with float atomics, so
 * their order -- and the last bits of the gradients -- vary from run to run (as in the reference, whose atomics are per pixel).
 * With 1 every (tile, Gaussian) instance stores its ten wave-reduced sums into a slot of its own and a per-Gaussian kernel adds
 * the slots in a fixed order: bit-identical gradients run to run, at 41 B of stream-ordered scratch per tile instance and about
 * 12 % of the train step (985 -> 869 iter/s at the headline scene; 27 % in round 2).  Values agree with the default mode to rounding
 * (2e-6 of an array's largest magnitude). */
/* "box_count" (-1 automatic / 0 / 1, default -1: on for large scenes and after a dense frame, like the split): the per-tile instance
 * counts are made from a difference grid (four LDS atomics per Gaussian: its rectangle's corners) and two prefix passes instead of
 * one atomic per (Gaussian, tile) instance.  Identical counts. */
/* "speculative_forward" = 2 (opt-in): as 1, but the call does not look at its frame's verdict at all before it returns -- the host is
 * back as soon as its launches are queued (at the headline scene 0.08 ms instead of 0.15 ms into a 0.5 ms forward pass) and may run
 * any number of calls ahead.  The return value is then the predicted capacity (an upper bound of num_rendered; hand it to
 * wg_rasterize_backward as R).  The verdict is read by the thread's NEXT forward call, or by the frame's own backward call (on whatever
 * thread: it is found by its image buffer), whichever comes first: when the deferred frame did not fit, that call returns WG_ERR_SPECULATION, the deferred frame's image is NaN and its gradients are zero (as with
 * wg_rasterize_forward_fixed), and the history has learnt the frame's size.  Use a generous "spec_margin_pct" with it.  Frames the
 * history cannot predict (the first of a shape) take the synchronous flow. */
/* "speculative_forward" (1/0, default 1): see wg_rasterize_forward; "spec_margin_pct" (default 25): the binning buffer of a
 * speculative frame holds the recent frames' largest instance count plus this margin.  Setting "speculative_forward" also clears the
 * calling thread's frame history and the read-only counters wg_get_option reports for it: "spec_frames", "spec_misses",
 * "forward_polls", "forward_polls_waited", "forward_wait_us_total", "forward_wait_us_last".
 * "geometry_reuse" (0/1, default 0 since round 4): read by the torch binding only (wg_rasterize_forward_recolor is always available);
 * opt-in because a write through `tensor.data` is invisible to the binding's identity check (diff_gaussian_rasterization/_C.py).
 * "fused_scan" (0/1, default 0): the column scan and the tile scan of the binning in one launch (measured slower on MI355X: the
 * device-scope hand-over costs more than the launch it saves). */
/* "exact_compositing" (1/0, default 1): the render kernels take every skip / stop decision of forward.cu:356-372 and backward.cu:536-546
 * (power > 0, alpha < 1/255, T (1 - alpha) < 1e-4) on values computed with the reference's own float32 operations, in its order, unfused,
 * with the float32 `exp` expansion hipcc emits for the reference's sources: n_contrib, final_T and the set of blended instances are
 * bit for bit those of the reference built with -ffp-contract=off; the image differs by the colour sums' fused multiply-adds (~2e-7).
 * 0: exp2 of a pre-scaled fused form (rounds 1-3: about 2 pixels per million land on the other side of a threshold); 5 % faster.
 * Must not change between a frame's forward and its backward call. */
/* "roctx" (0/1, default 0; WG_ROCTX=1 in the environment switches it on from the first call): a roctx range around every stage
 * ("wg:K1 preprocess" ... "wg:K10-K11 preprocess_backward"), for `rocprofv3 --marker-trace --kernel-trace`.  The marker library is
 * dlopen()ed on demand; WG_ERR_INVALID_ARGUMENT if none is found. */
int wg_set_option(const char* name, int value);
/* Current value of an option (every name above except the two test-only caps); -1 = unknown name.
 * "grad_record" (1/0, default 1): see wg_rasterize_backward.
 * Thread safety: options are process-wide; wg_set_option may be called from any host thread at any time -- every forward /
 * backward call copies the whole set once at its start and works from that copy (a call never sees half of an update, and the
 * scratch layout never depends on an option read twice).  The per-stage profiler keeps its events per device. */
int wg_get_option(const char* name);

const char* wg_status_string(int status);
const char* wg_last_hip_error(void);
const char* wg_version(void);

#ifdef __cplusplus
}
#endif
#endif /* WG_RASTERIZER_H_INCLUDED */
