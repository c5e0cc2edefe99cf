"""TEST TOOL (moved out of the product package in round 5: the documented diff of INTEGRATION.md section 5 is the deliverable, a source
rewriter is not).  The two caller edits INTEGRATION.md section 5 documents for `_render_internal` (wildgaussians/method.py:1479-1632), as text replacements
applied IN MEMORY to the caller's own `method.py` -- nothing is written to disk; a replacement whose original text is not found exactly once
raises (an edit stays one that applies to the reference as it is):

  EDITS           "two_colour": `_render_internal` rasterizes raw and toned colours over identical geometry with two calls
                  (method.py:1573-1611); with the edit the first call is skipped when both are wanted and the second hands the raw colours
                  over as `colors_precomp2=`.
  EDITS_TWO_TONE  "two_tone": the one call takes the SH features themselves and the appearance MLP's affine (`shs=`, `sh_mul=`,
                  `sh_offset=`, `sh_second=True`): no eval_sh, no P x 48 toned tensor in torch.

`wg_integration.apply_optins(method, edited_module=import_edited_method(method, "two_tone"))` takes the edited function at run time;
tests/test_real_caller.py holds both edits to the unedited function's results.  The `old` halves are the reference's own lines, quoted as anchors -- the only way to say where an edit goes."""
from __future__ import annotations

import importlib.util
import os
import sys

EDITS = [
    # 1. the raw call happens on its own only when there is no toned call to ride on
    ("        if not self.config.appearance_enabled or (self.config.appearance_separate_tuned_color and return_raw):\n",
     "        two_colour = self.config.appearance_enabled and self.config.appearance_separate_tuned_color and return_raw\n"
     "        if not self.config.appearance_enabled:\n"),
    # 2. the toned call carries the raw colours as the second set and returns the raw image as a fourth output
    ("            rendered_image, _radii, _accumulation = rasterizer(\n"
     "                means3D=means3D,\n"
     "                means2D=means2D,\n"
     "                colors_precomp=colors_toned,\n",
     "            rendered_image, _radii, _accumulation, *_raw = rasterizer(\n"
     "                means3D=means3D,\n"
     "                means2D=means2D,\n"
     "                colors_precomp=colors_toned,\n"
     "                **({\"colors_precomp2\": colors} if two_colour else {}),\n"),
    # 3. ... which is the raw render
    ("            raw_rendered_image = rendered_image if not self.config.appearance_separate_tuned_color else raw_rendered_image\n",
     "            raw_rendered_image = _raw[0] if _raw else raw_rendered_image\n"
     "            raw_rendered_image = rendered_image if not self.config.appearance_separate_tuned_color else raw_rendered_image\n"),
]


# The further-reaching edit ("two tones of one SH block"): the toned call takes the SH features themselves
# (`shs=`), the MLP's affine as `sh_mul=` / `sh_offset=`, and asks for the untoned render of the same coefficients as its second image
# (`sh_second=True`).  Neither eval_sh nor the P x 48 toned tensor is evaluated in torch; one rasterizer call per step.  Applies to the
# default appearance model (appearance_model_sh = False: a 3-wide affine) with SH features.
EDITS_TWO_TONE = [
    # 1. torch's eval_sh of the raw colours is only needed without the two-tone call
    ("        dir_pp_normalized = F.normalize(means3D - camera_center.repeat(features.shape[0], 1), dim=1)\n"
     "        if features.shape[-1] == 3:\n",
     "        dir_pp_normalized = F.normalize(means3D - camera_center.repeat(features.shape[0], 1), dim=1)\n"
     "        two_tone = self.config.appearance_enabled and not self.config.appearance_model_sh and features.shape[-1] != 3\n"
     "        if two_tone:\n"
     "            colors = None\n"
     "        elif features.shape[-1] == 3:\n"),
    # 2. no raw call of its own
    ("        if not self.config.appearance_enabled or (self.config.appearance_separate_tuned_color and return_raw):\n",
     "        if not two_tone and (not self.config.appearance_enabled or (self.config.appearance_separate_tuned_color and return_raw)):\n"),
    # 3. the one call: toned image first, the untoned render of the same coefficients second
    ("        if self.config.appearance_enabled:\n"
     "            assert self.appearance_mlp is not None\n",
     "        if two_tone:\n"
     "            shdim = (self.config.sh_degree + 1) ** 2\n"
     "            inp = torch.cat((features[..., :3], self.embeddings, embedding_expanded), dim=-1)\n"
     "            offset, mul = torch.split(self.appearance_mlp.mlp(inp) * 0.01, [3, 3], dim=-1)\n"
     "            rendered_image, radii, accumulation, raw_rendered_image = rasterizer(\n"
     "                means3D=means3D, means2D=means2D, shs=gaussians[\"features\"].view(-1, shdim, 3), colors_precomp=None,\n"
     "                opacities=opacity, scales=scales, rotations=rotations, cov3D_precomp=None,\n"
     "                sh_mul=mul, sh_offset=offset / C0, sh_pre_clamp_max=1.0, sh_post_clamp_max=1.0, sh_second=True, sh_pre_clamp_max2=1.0)\n"
     "            raw_rendered_image = raw_rendered_image if self.config.appearance_separate_tuned_color else rendered_image\n"
     "        elif self.config.appearance_enabled:\n"
     "            assert self.appearance_mlp is not None\n"),
]


def edited_source(path: str, edits=None) -> str:
    src = open(path).read()
    for old, new in (EDITS if edits is None else edits):
        if src.count(old) != 1:
            raise RuntimeError(f"the documented edit no longer applies to {path}: {old.strip().splitlines()[0]!r} found {src.count(old)} times")
        src = src.replace(old, new)
    return src


EDIT_SETS = {"two_colour": EDITS, "two_tone": EDITS_TWO_TONE}


def import_edited_method(method_module, edits=None, name=None, which=None):
    """A second module object compiled from `method_module`'s own source file with one edit set applied, in the same package (its
    relative imports resolve to the same modules; the operator packages are the ones `method_module` bound).  which = "two_colour" /
    "two_tone" (default name `<module>_<which>`), or an explicit list of (old, new) replacements and a module name."""
    if which is not None:
        edits = EDIT_SETS[which]
    elif edits is None:
        which, edits = "two_colour", EDITS
    if name is None:
        name = f"{method_module.__name__}_{which or 'edited'}"
    if name in sys.modules:
        return sys.modules[name]
    path = method_module.__file__
    spec = importlib.util.spec_from_loader(name, loader=None, origin=path)
    mod = importlib.util.module_from_spec(spec)
    mod.__file__ = path
    mod.__package__ = method_module.__package__
    sys.modules[name] = mod
    try:
        exec(compile(edited_source(path, edits), os.path.join(os.path.dirname(path), name.rsplit(".", 1)[-1] + ".py"), "exec"), mod.__dict__)
    except BaseException:
        sys.modules.pop(name, None)
        raise
    return mod
