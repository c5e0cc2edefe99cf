"""Run-time opt-ins for the reference's caller, WITHOUT editing its source.

    import wildgaussians.method as method          # the reference's module, unchanged on disk
    import wg_integration
    undo = wg_integration.apply_optins(method)     # before or after the model is built
    ...
    undo()                                          # puts the original attributes back

Only pieces that are module- or class-level names can be swapped this way; what is written inline in `_render_internal` /
`train_iteration` (SH evaluation and appearance toning in the operator, the fused L1 + DSSIM loss, `subpixel_offset=None`) still
needs the few-line edits INTEGRATION.md lists.  What IS swapped, each with the results its tests pin:

  ssim                  `method.ssim` (method.py:644-673, called at :1949)           -> wg_fused_ssim.ssim (same signature)
  adam                  `GaussianModel._setup_optimizers` (method.py:1029-1054)      -> the same, then FusedAdam.adopt(self.optimizer);
                        an optimizer that already exists on `model` is adopted at once (pass `model=`)
  densification_stats   `GaussianModel.add_densification_stats` (method.py:1470-1477) -> wg_fused_gaussians.add_densification_stats
  activations           `GaussianModel.get_gaussians` (method.py:1060-1086)          -> wg_fused_gaussians.activate (same dict)
  eval_sh               `method.eval_sh` (method.py:493-548, called at :1564, :1597) -> wg_fused_gaussians.eval_sh; calls it does not cover
                        (degree 4, a channel count other than 3, CPU tensors) go to the original function
  edited_module         (off by default) `GaussianModel._render_internal` -> the one of a module the INTEGRATOR supplies: a copy of the caller with
                        INTEGRATION.md section 5's "two_colour" or "two_tone" edit applied (the documented diff is the deliverable; this package
                        does not rewrite anybody's source -- tests/real_caller/render_edits.py is the test tool that builds such a module in memory)
  geometry_reuse        library option "geometry_reuse" = 1 (opt-in since round 4): the toned and depth calls of `_render_internal`
                        (method.py:1573-1631) ride on the raw call's projection and binning.  The caller's training loop qualifies
                        (it writes geometry only between a backward pass and the next forward pass); undo() switches it off again
"""
from __future__ import annotations

import torch


def apply_optins(method_module, model=None, ssim: bool = True, adam: bool = True, densification_stats: bool = True, activations: bool = True,
                 eval_sh: bool = True, geometry_reuse: bool = True, edited_module=None):
    """-> a function that restores everything that was replaced.  `model`: an already constructed GaussianModel (e.g.
    `WildGaussians(...).model`) whose existing optimizer should be adopted too.
    edited_module (default None): a module object holding a copy of the caller with INTEGRATION.md section 5's edit of `_render_internal`
    applied (the raw and the toned render in ONE rasterizer call); its `GaussianModel._render_internal` ALSO replaces the caller's, and it is
    handed the swapped `ssim` / `eval_sh`."""
    import wg_fused_gaussians as FG
    import wg_fused_ssim
    saved = []

    def swap(obj, name, new):
        saved.append((obj, name, getattr(obj, name)))
        setattr(obj, name, new)

    GM = method_module.GaussianModel
    if ssim:
        swap(method_module, "ssim", wg_fused_ssim.ssim)
    if adam:
        orig_setup = GM._setup_optimizers

        def _setup_optimizers(self):
            orig_setup(self)
            self.optimizer = FG.FusedAdam.adopt(self.optimizer)
        swap(GM, "_setup_optimizers", _setup_optimizers)
        if model is not None and getattr(model, "optimizer", None) is not None and not isinstance(model.optimizer, FG.FusedAdam):
            saved.append((model, "optimizer", model.optimizer))
            model.optimizer = FG.FusedAdam.adopt(model.optimizer)
    if densification_stats:
        def add_densification_stats(self, viewspace_point_tensor, update_filter):
            # the kernel's visibility test is radii > 0: the boolean filter IS that test's result (method.py:1622), as 0 / 1
            gof = self.config.use_gof_abs_gradient
            FG.add_densification_stats(update_filter.to(torch.int32), viewspace_point_tensor.grad, self.xyz_grad, self.denom,
                                       xyz_gradient_accum_abs=self.xyz_gradient_accum_abs if gof else None,
                                       xyz_gradient_accum_abs_max=self.xyz_gradient_accum_abs_max if gof else None)
        swap(GM, "add_densification_stats", add_densification_stats)
    if activations:
        def get_gaussians(self):
            features = self.features_dc
            if self.features_rest is not None:
                features = torch.cat((features, self.features_rest), dim=-1)
            opacities, scales, rotations = FG.activate(self.opacities, self.scales, self.rotations, self.filter_3D)
            return {"xyz": self.xyz, "opacities": opacities, "scales": scales, "rotations": rotations, "features": features}
        swap(GM, "get_gaussians", get_gaussians)

    if eval_sh:
        orig_eval_sh = method_module.eval_sh

        def fused_eval_sh(deg, sh, dirs):
            d = int(deg)
            if (d > 3 or not torch.is_tensor(sh) or not torch.is_tensor(dirs) or sh.dim() < 2 or sh.shape[-2] != 3 or not sh.is_cuda
                    or sh.dtype != torch.float32 or dirs.dtype != torch.float32):
                return orig_eval_sh(deg, sh, dirs)   # the caller's own code, not a fallback of this library
            return FG.eval_sh(d, sh, dirs)
        swap(method_module, "eval_sh", fused_eval_sh)

    if edited_module is not None:
        edited = edited_module
        for name in ("ssim", "eval_sh"):   # the edited function looks module-level names up in ITS module: hand it the swapped ones
            swap(edited, name, getattr(method_module, name))   # (undo() puts its own back: the module object is cached in sys.modules)
        swap(GM, "_render_internal", edited.GaussianModel._render_internal)

    reuse_before = None
    if geometry_reuse:
        from diff_gaussian_rasterization import _C
        reuse_before = _C.get_option("geometry_reuse")
        _C.set_option("geometry_reuse", 1)

    def undo():
        while saved:
            obj, name, old = saved.pop()
            setattr(obj, name, old)
        if reuse_before is not None:
            from diff_gaussian_rasterization import _C
            _C.set_option("geometry_reuse", reuse_before)
            _C.forget_geometry()
    return undo
