/*
 * wg_rasterizer.h -- C-ABI of the MI355X-native differentiable Gaussian-splat rasterizer (libwg_rasterizer.so).
 *
 * The drop-in boundary for the hot path of jkulhanek/wild-gaussians: it replaces the static C++ interface
 * CudaRasterizer::Rasterizer::{forward,backward,markVisible} (submodules/diff-gaussian-rasterization/cuda_rasterizer/
 * rasterizer.h:24-88), i.e. what the reference's torch binding (rasterize_points.cu:35-225) calls.  Differences, all forced by
 * "plain C, no C++ / torch types in the signature": std::function<char*(size_t)> allocators become (function pointer, void* user)
 * pairs; every call takes the HIP stream to launch on (NULL = the default stream; the reference uses the legacy default stream,
 * rasterizer_impl.cu:148,292); errors are negative wg_status codes instead of C++ exceptions (rasterizer_impl.cu:244-247,
 * auxiliary.h:166-173); bool becomes int / unsigned char.
 *
 * All pointers are DEVICE pointers to contiguous float32 / int32 data unless stated otherwise.  NULL means "not provided" exactly as
 * in the reference (shs vs colors_precomp, scales + rotations vs cov3D_precomp; forward.cu:217,253, backward.cu:426,430).  Matrices
 * have the reference's layout: viewmatrix = W2C transposed, projmatrix = (P W2C) transposed, row-major (auxiliary.h:58-77).
 *
 * The surface is TWO pairs of entry points: wg_rasterize_forward / _backward, argument for argument the reference's interface, and
 * wg_rasterize_forward_ex / _backward_ex, which take ONE struct: the same arguments plus optional blocks for everything beyond the
 * reference (NULL = absent) and the PER-CALL options that affect results.  The first pair is the second with no block and default
 * options.  The library keeps no result-affecting process-wide state: wg_set_option only holds tuning switches whose every setting
 * gives bit-identical results (docs/OPTIONS.md).
 *
 * The three scratch buffers (geometry / binning / image state) are opaque (the layout differs from the reference's GeometryState /
 * BinningState / ImageState).  Documented: the image-state buffer starts, at its first 256-byte-aligned address, with final_T as
 * float[H*W], and holds at wg_image_accumulation_offset() behind it accumulation = 1 - final_T as float[H*W], written by the forward
 * pass itself (the reference's Python wrapper computes it from final_T, diff_gaussian_rasterization/__init__.py:101-113).
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
    WG_ERR_INVALID_ARGUMENT = -1, /* bad sizes / missing mandatory pointer / both-or-neither optional inputs / an inconsistent block */
    WG_ERR_ALLOC = -2,            /* an allocator callback returned NULL */
    WG_ERR_HIP = -3,              /* a HIP runtime call or kernel launch failed (wg_last_hip_error() has the text) */
    WG_ERR_OVERFLOW = -4,         /* more than 2^31-1 (tile, Gaussian) instances */
    WG_ERR_SPECULATION = -5       /* "speculative_forward" = 2 only: the calling thread's PREVIOUS forward call did not fit the binning buffer
                                     it had predicted; that call's image is NaN and its gradients are zero -- repeat the step */
} wg_status;

/* Replaces std::function<char*(size_t N)> (rasterizer.h:34-36, rasterize_points.cu:27-33): returns a device pointer to at least
 * `bytes` bytes that stays valid until the matching backward call.  binning_alloc may be called TWICE in one forward call (the
 * speculative forward re-issues a frame that did not fit its predicted buffer); the buffer returned last is the one in use. */
typedef char* (*wg_alloc_fn)(size_t bytes, void* user);

/* Scratch sizes (bytes), for callers that preallocate (the binning size is an upper bound). */
size_t wg_geometry_buffer_size(int P);
size_t wg_image_buffer_size(int width, int height);
size_t wg_binning_buffer_size(int num_rendered);
size_t wg_image_accumulation_offset(int width, int height); /* bytes from final_T to accumulation[H*W] */

/* Rasterizer::forward (rasterizer.h:33-59, rasterizer_impl.cu:198-340).  Returns num_rendered (>= 0) or a negative wg_status.
 * out_color: float[3*H*W] planar CHW, fully written.  radii: int[P] or NULL.  subpixel_offset: float[H*W*2], or NULL (beyond the
 * reference: all zero, nothing read).  The reference's one host<->device rendezvous (num_rendered sizes the binning buffer,
 * rasterizer_impl.cu:284) sits behind the call's LAST launch, not in its middle (docs/OPTIONS.md: "speculative_forward"). */
int wg_rasterize_forward(wg_alloc_fn geometry_alloc, void* geometry_user, wg_alloc_fn binning_alloc, void* binning_user,
                         wg_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background, int width, int height,
                         const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                         const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                         const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                         float kernel_size, const float* subpixel_offset, int prefiltered, float* out_color, int* radii, int debug,
                         void* stream);

/* Rasterizer::backward (rasterizer.h:61-88, rasterizer_impl.cu:344-443).  The reference wants all nine gradient outputs zero-filled
 * (rasterize_points.cu:157-165); here (options->grad_record = 1, the default) all nine are fully OVERWRITTEN (zeros for culled
 * Gaussians) -- a caller following the reference's protocol gets the same values -- and dL_dconic (an intermediate of the
 * reference) and, with SH colours, dL_dcolor may be NULL.  dL_dconic: float[P*4], 16-byte aligned ([0],[1],[3] used);
 * dL_dmean2D: float[P*3] (x, y NDC-scaled, z = abs-gradient, backward.cu:590-595); dL_dsh may be NULL when M == 0. */
int wg_rasterize_backward(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                          const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                          const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                          const float* campos, float tan_fovx, float tan_fovy, float kernel_size, const float* subpixel_offset,
                          const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer, const float* dL_dpix,
                          float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
                          float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int debug, void* stream);

/* Rasterizer::markVisible (rasterizer.h:26-31, rasterizer_impl.cu:141-153). present: unsigned char[P]. */
int wg_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, unsigned char* present, void* stream);

/* ---- optional blocks of the _ex calls (everything beyond the reference; SURVEY.md 8f N3 and the caller's two renders per step) ---- */

/* Result-affecting switches, PER CALL (NULL = WG_CALL_OPTIONS_DEFAULT).  A frame's backward call must carry its forward call's
 * exact_compositing: the library remembers it per image buffer and returns WG_ERR_INVALID_ARGUMENT on a mismatch.
 *   exact_compositing (1): every skip / stop decision of forward.cu:356-372 / backward.cu:536-546 on values computed with the
 *     reference's own float32 operations and hipcc's float32 `exp` expansion: n_contrib, final_T and the blended set equal the
 *     reference's -ffp-contract=off build bit for bit (image within ~2e-7).  0: exp2 of a fused form, ~2 pixels per million flip, 5 % faster.
 *   deterministic_backward (0): 1 = every (tile, Gaussian) instance stores its wave-reduced sums in a slot of its own and a
 *     per-Gaussian pass adds them in a fixed order instead of float atomics: bit-identical gradients run to run.  Its scratch is a block
 *     of the library's own, leased for the call: not inside a stream capture (WG_ERR_INVALID_ARGUMENT -- a replay would write a block others hold).
 *   grad_record (1): the per-tile pass accumulates into a 48-byte record per Gaussian inside geom_buffer (see wg_rasterize_backward);
 *     0 = into dL_dmean2D / dL_dconic / dL_dopacity / dL_dcolor themselves, which must then be non-NULL and zero on entry. */
typedef struct wg_call_options {
    int exact_compositing, deterministic_backward, grad_record;
} wg_call_options;
#define WG_CALL_OPTIONS_DEFAULT {1, 0, 1}

/* Per-Gaussian affine on the SH coefficients inside K1 / K11 ("the appearance MLP output feeds SH eval directly"):
 *   x = min(shs[i][k][c], pre_clamp_max);  t = x * mul[i][c] + (k == 0 ? offset[i][c] : 0);  used = min(t, post_clamp_max)
 * (separate multiply and add: the values of `(features.clamp_max(1) * mul + cat(offset / C0, 0)).clamp_max(1)`,
 * wildgaussians/method.py:890-900, 1590-1595).  INFINITY switches a clamp off; mul / offset may be NULL (1 / 0).  Backward: dL_dsh is
 * then the gradient of the RAW coefficients; dL_dmul / dL_doffset ([P,3], overwritten) are required where mul / offset are given. */
typedef struct wg_sh_tone {
    const float* mul;      /* [P,3] or NULL */
    const float* offset;   /* [P,3] or NULL */
    float pre_clamp_max, post_clamp_max;
    float* dL_dmul;        /* backward only */
    float* dL_doffset;     /* backward only */
} wg_sh_tone;

/* The caller's get_gaussians() (wildgaussians/method.py:1060-1086) inside K1 / K11: `opacities`, `scales`, `rotations` are the RAW
 * parameters (logit, log-scale, unnormalised quaternion); q = r / max(|r|, 1e-12), s = sqrt(exp(ls)^2 + f^2), o = sigmoid(lo) *
 * sqrt(prod exp(ls)^2 / prod s^2).  Backward: dL_dopacity / dL_dscale / dL_drot are the RAW parameters' gradients (needs grad_record).
 * Scale / rotation pairs only.  Same device functions and bits as wg_activations_forward / _backward (include/wg_activations.h). */
typedef struct wg_raw_gaussians {
    const float* filter_3D;       /* [P] */
    const float* raw_opacities;   /* [P], backward only: the forward call's `opacities` */
} wg_raw_gaussians;

/* A SECOND image composited in the same walk (WildGaussians renders raw and toned colours over identical geometry every step,
 * method.py:1573-1611; the decisions do not depend on the colours).  Either a second set of precomputed colours (colors_precomp2, with
 * colors_precomp) or -- sh_second = 1 in the args -- the SAME SH coefficients through `tone2`.  Both images share the background.
 * Backward: dL_dpix2 = out_color2's cotangent (zeros if none); dL_dcolor2 [P,3] out (may be NULL with sh_second); the geometry
 * gradients are those of BOTH images' losses (what autograd adds over the reference's two calls).  Needs grad_record or
 * deterministic_backward. */
typedef struct wg_second_image {
    const float* colors_precomp2;   /* forward; NULL with sh_second */
    float* out_color2;              /* forward: float[3*H*W], fully written */
    const float* dL_dpix2;          /* backward */
    float* dL_dcolor2;              /* backward */
} wg_second_image;

/* Recolouring: a further image of the SAME Gaussians through the SAME camera with other precomputed colours, along the parent
 * call's sorted lists.  parent_*: the three buffers a forward call returned (R = its return value), alive and unmodified.  The call
 * allocates ONE new geometry buffer (geometry_alloc), reads only colors_precomp / background / subpixel_offset / out_color / radii
 * of the args, and returns R; its backward call takes the NEW geometry buffer and the parent's binning and image buffers. */
typedef struct wg_recolor_parent {
    char *geom_buffer, *binning_buffer, *image_buffer;
    int R;
} wg_recolor_parent;

typedef struct wg_forward_args {
    size_t struct_size;   /* the CALLER's sizeof(wg_forward_args): fields are only ever appended; a shorter (older) struct is accepted
                             from version 0.5's 288 bytes on, its missing tail reads as absent; a longer (newer) one is refused */
    wg_alloc_fn geometry_alloc; void* geometry_user;
    wg_alloc_fn binning_alloc;  void* binning_user;
    wg_alloc_fn image_alloc;    void* image_user;
    int P, D, M, width, height, prefiltered, debug;
    float scale_modifier, tan_fovx, tan_fovy, kernel_size;
    const float *background, *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
    const float *viewmatrix, *projmatrix, *cam_pos, *subpixel_offset;
    float* out_color;
    int* radii;
    void* stream;
    /* beyond the reference: NULL / 0 = absent */
    const wg_sh_tone* tone;            /* with shs */
    const wg_sh_tone* tone2;           /* with sh_second: the second image's tone (NULL = no affine, no clamp) */
    int sh_second;                     /* second image = the same SH coefficients through tone2 (needs `second`->out_color2) */
    const wg_second_image* second;
    const wg_raw_gaussians* raw;
    const wg_recolor_parent* recolor;
    int binning_capacity;              /* > 0: NO host rendezvous at all (hipGraph capture): the caller's capacity in instances; returns it
                                          (hand it to backward as R); a frame that does not fit gives a NaN image and zero gradients,
                                          wg_forward_status() tells afterwards.  <= 36864 tiles, no debug, no deterministic backward. */
    const wg_call_options* options;
} wg_forward_args;

typedef struct wg_backward_args {
    size_t struct_size;
    int P, D, M, R, width, height, debug;
    float scale_modifier, tan_fovx, tan_fovy, kernel_size;
    const float *background, *means3D, *shs, *colors_precomp, *scales, *rotations, *cov3D_precomp;
    const float *viewmatrix, *projmatrix, *campos, *subpixel_offset;
    const int* radii;
    char *geom_buffer, *binning_buffer, *image_buffer;
    const float* dL_dpix;
    float *dL_dmean2D, *dL_dconic, *dL_dopacity, *dL_dcolor, *dL_dmean3D, *dL_dcov3D, *dL_dsh, *dL_dscale, *dL_drot;
    void* stream;
    const wg_sh_tone* tone;
    const wg_sh_tone* tone2;
    int sh_second;
    const wg_second_image* second;
    const wg_raw_gaussians* raw;
    const wg_call_options* options;
} wg_backward_args;

int wg_rasterize_forward_ex(const wg_forward_args* args);
int wg_rasterize_backward_ex(const wg_backward_args* args);
/* num_rendered and fits (1 / 0) of the forward call that produced image_buffer; synchronises the stream. */
int wg_forward_status(char* image_buffer, int width, int height, int* num_rendered, int* fits, void* stream);

/* ---- introspection used by the parity tests (device pointers into the opaque buffers) ---- */
typedef struct wg_geometry_view {
    const float* depths;          /* [P]   view-space z (forward.cu:262) */
    const int* radii;             /* [P]   internal copy */
    const float* splats;          /* [P*12] 48-byte records: mx,my,conic.x,conic.y | conic.z,opacity*coef,r2,r | g,b,g2,b2 (r2,g2,b2: the second image's colours) */
    const float* cov3D;           /* [P*6] */
    const unsigned char* clamped; /* [P]   bit c: SH colour channel c was clamped at 0 (forward.cu:67-69); bits 3-5: the second image's */
    const uint32_t* tiles_touched;/* [P] */
    const uint32_t* point_offsets;/* [P]   inclusive prefix sum (global-sort and deterministic paths) */
} wg_geometry_view;
typedef struct wg_binning_view {
    const uint32_t* point_list;   /* [R] Gaussian ids sorted by (tile | depth), stable */
} wg_binning_view;
typedef struct wg_image_view {
    const float* final_T;       /* [H*W] */
    const float* accumulation;  /* [H*W] 1 - final_T */
    const uint32_t* n_contrib;  /* [H*W] */
    const uint32_t* ranges;     /* [tiles*2] (start,end) */
    const uint32_t* tile_last;  /* [tiles] max n_contrib over the tile's pixels */
    const uint32_t* tile_near;  /* [tiles] near / far split: near instances of the tile */
    const uint32_t* split;      /* [2] {depth-code threshold of the split (0xffffffff = off), bands that needed their far instances} */
    const uint32_t* order_fwd;  /* [tiles] launch order of the forward render kernel ("forward_order"; valid when order_key[0] != 0xffffffff) */
    const uint32_t* order_key;  /* [4] {the camera's row in the launch-order table or 0xffffffff = none, tag lo, tag hi, 1 = the row held this camera} */
    const uint32_t* order_bwd;  /* [tiles] launch order of the backward render kernel (written by the frame's backward call) */
} wg_image_view;
int wg_view_geometry(char* geom_buffer, int P, wg_geometry_view* out);
int wg_view_binning(char* binning_buffer, int R, wg_binning_view* out);
int wg_view_image(char* image_buffer, int width, int height, wg_image_view* out);

/* ---- per-stage timing with HIP events on the caller's stream (bench.py's roofline): enable, run, read (synchronises) ---- */
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

/* Tuning / test switches, process-wide, copied once per call; EVERY setting gives bit-identical results (binning strategies, host
 * flow, profiling).  Names, defaults and measurements: docs/OPTIONS.md.  -1 / WG_ERR_INVALID_ARGUMENT for an unknown name --
 * including the three result-affecting switches, which are per call (wg_call_options). */
int wg_set_option(const char* name, int value);
int wg_get_option(const char* name);

const char* wg_status_string(int status);
const char* wg_last_hip_error(void);
const char* wg_version(void);

#ifdef __cplusplus
}
#endif
#endif /* WG_RASTERIZER_H_INCLUDED */
