"""Drop-in ``diff_gaussian_rasterization`` for WildGaussians on AMD MI355X (gfx950).

Public surface = what ``wildgaussians/method.py:26,1529-1631`` imports and calls, i.e. the surface of the reference's
``submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py``:

* ``GaussianRasterizationSettings`` -- 15-field NamedTuple, same field names and order (reference :175-190)
* ``GaussianRasterizer(raster_settings)`` -- ``nn.Module``; ``forward(means3D, means2D, opacities, shs=None,
  colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None) -> (color, radii, accumulation)`` and
  ``markVisible(positions)`` (reference :192-241)
* ``rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
  raster_settings)`` (reference :21-42)

Semantics kept from the reference: exactly-one-of validation raising ``Exception``; "not provided" inputs are the
zero-sized ``torch.Tensor([])`` sentinel; gradients come back for (means3D, means2D, sh, colors_precomp, opacities,
scales, rotations, cov3Ds_precomp); ``means2D`` is a zero [P,3] carrier whose gradient holds (d/dx_ndc, d/dy_ndc,
sum|.|) (reference backward.cu:590-595, read at method.py:1471-1476); ``radii`` and ``accumulation`` are
non-differentiable; ``debug=True`` dumps the CPU copy of the arguments to snapshot_{fw,bw}.dump when the native call
raises.  The native side is ``_C`` -> ``libwg_rasterizer.so`` (hand-written HIP); there is no CPU/PyTorch fallback.
"""
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from . import _C

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
call_options = _C.call_options   # `with call_options(deterministic_backward=1): ...` (thread-local, scoped; see _C.py)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    kernel_size: float
    subpixel_offset: torch.Tensor
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    return_accumulation: bool


def _absent() -> torch.Tensor:
    return torch.Tensor([])


def _call_native(fn, args, debug: bool, dump_path: str, what: str):
    """Run a native entry point; in debug mode keep a CPU copy of the inputs and dump it if the call raises."""
    if not debug:
        return fn(*args)
    saved = tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)
    try:
        return fn(*args)
    except Exception:
        torch.save(saved, dump_path)
        print(f"\nAn error occured in {what}. Writing {dump_path} for debugging.\n")
        raise


def _accumulation_from_image_state(img_buffer: torch.Tensor, height: int, width: int) -> torch.Tensor:
    """accumulation = 1 - final_T (reference __init__.py:101-113).  The forward kernels write it themselves, as the second array of
    the image-state buffer (wg_rasterizer.h): a view, no elementwise kernel per call.  The memory belongs to this call's scratch
    (kept alive by the view); calls that share an image state through the binding's geometry reuse share the values too."""
    align = _C.IMAGE_STATE_ALIGNMENT
    start = (-img_buffer.data_ptr()) % align + _C.accumulation_offset(height, width)
    return img_buffer[start:start + 4 * height * width].view(torch.float32).view(height, width)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                sh_mul=None, sh_offset=None, sh_pre_clamp_max=None, sh_post_clamp_max=None, binning_capacity=None, colors_precomp2=None,
                filter_3D=None, sh_second=False, sh_mul2=None, sh_offset2=None, sh_pre_clamp_max2=None, sh_post_clamp_max2=None, call_options=None):
        rs = raster_settings
        native_args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                       rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.kernel_size, rs.subpixel_offset,
                       rs.image_height, rs.image_width, sh, rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)
        # Everything below is beyond the reference (opt-in by keyword, see GaussianRasterizer.forward) and goes to the native module as the
        # optional blocks of ONE call (include/wg_rasterizer.h: wg_forward_args).
        # per-call options: resolved ONCE here (keyword > the calling thread's defaults) and carried to the backward call on ctx
        ctx.call_options = _C.resolve_call_options(call_options)
        extra = dict(options=ctx.call_options)
        # per-Gaussian affine + clamps on the SH coefficients in-kernel (wg_sh_tone)
        ctx.sh_tone = None
        if sh_mul is not None or sh_offset is not None or sh_pre_clamp_max is not None or sh_post_clamp_max is not None:
            if sh.numel() == 0:
                raise Exception("sh_mul / sh_offset / sh_*_clamp_max act on SH coefficients: provide shs, not colors_precomp")
            ctx.sh_tone = (None if sh_mul is None else sh_mul.detach(), None if sh_offset is None else sh_offset.detach(),
                           sh_pre_clamp_max, sh_post_clamp_max)
            extra["sh_tone"] = ctx.sh_tone
        if binning_capacity is not None:   # no host rendezvous (wg_forward_args::binning_capacity), capturable in a hipGraph
            extra["binning_capacity"] = int(binning_capacity)
        # a second set of precomputed colours composited in the same walk (wg_second_image)
        ctx.dual = colors_precomp2 is not None
        # opacities / scales / rotations are the RAW parameters, get_gaussians() runs in-kernel (wg_raw_gaussians)
        ctx.raw = filter_3D is not None
        if ctx.raw and (ctx.dual or binning_capacity is not None):
            raise Exception("filter_3D (raw-parameter mode) cannot be combined with colors_precomp2 / binning_capacity")
        if ctx.raw:
            extra["filter_3D"] = filter_3D
        # both colour sets from the same SH block, each through its own tone (wg_forward_args::sh_second + tone2)
        ctx.sh_second = None
        if sh_second:
            if sh.numel() == 0 or ctx.dual or binning_capacity is not None:
                raise Exception("sh_second renders a second tone of the SH coefficients: provide shs, and neither colors_precomp2 nor binning_capacity")
            ctx.sh_second = (None if sh_mul2 is None else sh_mul2.detach(), None if sh_offset2 is None else sh_offset2.detach(),
                             sh_pre_clamp_max2, sh_post_clamp_max2)
            extra["sh_second"] = ctx.sh_second
        elif ctx.dual:
            if ctx.sh_tone is not None or binning_capacity is not None:
                raise Exception("colors_precomp2 cannot be combined with sh_mul / sh_offset / binning_capacity")
            extra["colors2"] = colors_precomp2
        res = _call_native(lambda *a: _C.rasterize_gaussians(*a, **extra), native_args, rs.debug, "snapshot_fw.dump", "forward")
        num_rendered, color, radii, geom_buf, binning_buf, img_buf = res[:6]
        color2 = res[6] if len(res) > 6 else None

        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        # radii and accumulation take no gradient (reference backward signature (ctx, grad_out_color, _, _), :117): do not let
        # autograd materialise P- and H*W-sized zero tensors for them on every backward pass
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom_buf, binning_buf, img_buf,
                              *((filter_3D, opacities) if ctx.raw else ()))

        accumulation = None
        if rs.return_accumulation:
            if means3D.shape[0] == 0:
                accumulation = torch.zeros((rs.image_height, rs.image_width), dtype=torch.float32, device=color.device)
            else:
                accumulation = _accumulation_from_image_state(img_buf, rs.image_height, rs.image_width)
        if ctx.dual or ctx.sh_second is not None:
            return color, radii, accumulation, color2
        return color, radii, accumulation

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii, _grad_accumulation, grad_out_color2=None):
        rs = ctx.raster_settings
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom_buf, binning_buf, img_buf = ctx.saved_tensors[:10]
        zeros = lambda: torch.zeros((3, rs.image_height, rs.image_width), dtype=torch.float32, device=means3D.device)
        if grad_out_color is None:  # the image itself took no gradient (only radii / accumulation were used downstream)
            grad_out_color = zeros()
        native_args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
                       rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.kernel_size, rs.subpixel_offset, grad_out_color, sh,
                       rs.sh_degree, rs.campos, geom_buf, ctx.num_rendered, binning_buf, img_buf, rs.debug)
        extra = dict(options=ctx.call_options)   # the frame's forward call's
        if ctx.sh_tone is not None:
            extra["sh_tone"] = ctx.sh_tone
        if ctx.raw:
            extra["raw"] = tuple(ctx.saved_tensors[10:12])
        if ctx.sh_second is not None or ctx.dual:
            extra["dL_dout_color2"] = zeros() if grad_out_color2 is None else grad_out_color2   # (None: the second image took no gradient)
        if ctx.sh_second is not None:
            extra["sh_second"] = ctx.sh_second
        res = _call_native(lambda *a: _C.rasterize_gaussians_backward(*a, **extra), native_args, rs.debug, "snapshot_bw.dump", "backward")
        (g_means2D, g_colors, g_opacities, g_means3D, g_cov3Ds, g_sh, g_scales, g_rotations) = res[:8]
        g_mul = g_offset = g_colors2 = g_mul2 = g_offset2 = None
        shaped = lambda g, like: None if (g is None or like is None) else g.view(like.shape)
        if ctx.sh_second is not None:
            g_mul, g_offset, g_mul2, g_offset2 = res[8:12]
            if ctx.sh_tone is not None:
                g_mul, g_offset = shaped(g_mul, ctx.sh_tone[0]), shaped(g_offset, ctx.sh_tone[1])
            g_mul2, g_offset2 = shaped(g_mul2, ctx.sh_second[0]), shaped(g_offset2, ctx.sh_second[1])
        elif ctx.dual:
            g_colors2 = res[8]
        elif ctx.sh_tone is not None:
            g_mul, g_offset = shaped(res[8], ctx.sh_tone[0]), shaped(res[9], ctx.sh_tone[1])
        # order of forward()'s inputs; None for raster_settings, the clamp constants and the options
        return (g_means3D, g_means2D, g_sh, g_colors, g_opacities, g_scales, g_rotations, g_cov3Ds, None, g_mul, g_offset, None, None, None, g_colors2, None,
                None, g_mul2, g_offset2, None, None, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                        sh_mul=None, sh_offset=None, sh_pre_clamp_max=None, sh_post_clamp_max=None, binning_capacity=None, colors_precomp2=None,
                        filter_3D=None, sh_second=False, sh_mul2=None, sh_offset2=None, sh_pre_clamp_max2=None, sh_post_clamp_max2=None, call_options=None):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings, sh_mul, sh_offset, sh_pre_clamp_max, sh_post_clamp_max, binning_capacity, colors_precomp2,
                                     filter_3D, sh_second, sh_mul2, sh_offset2, sh_pre_clamp_max2, sh_post_clamp_max2, call_options)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """Boolean mask of points that pass the near-plane test of the preprocess stage (view-space z > 0.2)."""
        rs = self.raster_settings
        with torch.no_grad():
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs: Optional[torch.Tensor] = None,
                colors_precomp: Optional[torch.Tensor] = None, scales: Optional[torch.Tensor] = None,
                rotations: Optional[torch.Tensor] = None, cov3D_precomp: Optional[torch.Tensor] = None, *,
                sh_mul: Optional[torch.Tensor] = None, sh_offset: Optional[torch.Tensor] = None,
                sh_pre_clamp_max: Optional[float] = None, sh_post_clamp_max: Optional[float] = None,
                binning_capacity: Optional[int] = None, colors_precomp2: Optional[torch.Tensor] = None,
                filter_3D: Optional[torch.Tensor] = None, sh_second: bool = False, sh_mul2: Optional[torch.Tensor] = None,
                sh_offset2: Optional[torch.Tensor] = None, sh_pre_clamp_max2: Optional[float] = None,
                sh_post_clamp_max2: Optional[float] = None, exact_compositing: Optional[bool] = None,
                deterministic_backward: Optional[bool] = None, grad_record: Optional[bool] = None):
        """`exact_compositing=` / `deterministic_backward=` / `grad_record=` (keyword-only, beyond the reference): the three switches that affect
        results, PER CALL (wg_call_options, include/wg_rasterizer.h); None = the calling thread's default (`_C.call_options(...)` sets it for a
        `with` block -- for callers that cannot pass keywords); the frame's backward pass runs with its forward call's values.

        `sh_second=True` (keyword-only, beyond the reference; with `shs`): a SECOND image from the same SH coefficients through a tone
        of its own (`sh_mul2` / `sh_offset2` / `sh_*_clamp_max2`, each optional), composited in the SAME call as the first -- WildGaussians'
        step (method.py:1573-1611) renders the raw and the toned colours of one SH block: `rast(shs=f, sh_mul=mul, sh_offset=offset / C0,
        sh_pre_clamp_max=1, sh_post_clamp_max=1, sh_second=True, sh_pre_clamp_max2=1)` returns `(toned, radii, accumulation, raw)` from one
        projection, one binning, one forward and one backward walk.  Gradients flow to `shs` (both images' losses), and to each tone's
        `mul` / `offset`.  Composes with `filter_3D`.

        `filter_3D=` (keyword-only, beyond the reference): `opacities`, `scales`, `rotations` are then the caller's RAW
        parameters (logit, log-scale, unnormalised quaternion) and `get_gaussians()` (method.py:1060-1086: normalise, exp, sigmoid, 3-D
        filter with this [P,1] tensor) runs inside the preprocess kernels, forward and backward: the gradients returned for those three
        inputs are the raw parameters'.  Composes with `shs` + `sh_mul` / `sh_offset` (the whole step before the operator in-kernel).

        `colors_precomp2=` (keyword-only, beyond the reference; with `colors_precomp`): a second [P,3] colour set composited in the SAME
        call -- one projection, one binning, one forward and one backward walk for both (WildGaussians' raw and toned colours,
        method.py:1573-1611).  Returns `(color, radii, accumulation, color2)`; gradients flow to both colour
        tensors, the geometry gradients are those of both images' losses together.

        `binning_capacity=` (keyword-only, beyond the reference): the forward pass without any host rendezvous
        (wg_forward_args::binning_capacity: the caller supplies the number of (tile, Gaussian) instances the binning buffer holds), for steps
        captured in a hipGraph; a frame that does not fit comes back as NaN, `_C.forward_status` tells.  Otherwise:

        The reference's signature (diff_gaussian_rasterization/__init__.py:208-241) plus four keyword-only opt-ins (SURVEY.md 8f
        N3): with `shs`, the kernels evaluate `min(min(shs, sh_pre_clamp_max) * sh_mul[:, None, :] + [k == 0] * sh_offset[:, None, :],
        sh_post_clamp_max)` instead of `shs` -- WildGaussians' appearance toning (method.py:890-900, 1590-1595: pass the clamped
        features' raw tensor, `mul`, `offset / C0`, and 1.0 for both clamps) without the P x 48 intermediate tensors -- and return
        gradients for `shs` (raw), `sh_mul` and `sh_offset` ([P, 3] each)."""
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if colors_precomp2 is not None and colors_precomp is None:
            raise Exception('colors_precomp2 is a second set of precomputed colours: provide colors_precomp too')
        has_scale_rot = scales is not None or rotations is not None
        complete_scale_rot = scales is not None and rotations is not None
        if (cov3D_precomp is None and not complete_scale_rot) or (cov3D_precomp is not None and has_scale_rot):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')

        return rasterize_gaussians(
            means3D, means2D,
            _absent() if shs is None else shs,
            _absent() if colors_precomp is None else colors_precomp,
            opacities,
            _absent() if scales is None else scales,
            _absent() if rotations is None else rotations,
            _absent() if cov3D_precomp is None else cov3D_precomp,
            self.raster_settings, sh_mul, sh_offset, sh_pre_clamp_max, sh_post_clamp_max, binning_capacity, colors_precomp2, filter_3D,
            sh_second, sh_mul2, sh_offset2, sh_pre_clamp_max2, sh_post_clamp_max2,
            None if exact_compositing is None and deterministic_backward is None and grad_record is None else
            dict(exact_compositing=exact_compositing, deterministic_backward=deterministic_backward, grad_record=grad_record))
