"""The caller edit INTEGRATION.md section 5 documents for the two-colour call, as text replacements applied IN MEMORY to the staged
(byte-identical) `wildgaussians/method.py`: `_render_internal` (method.py:1573-1611) rasterizes raw and toned colours over identical
geometry with two calls; with the edit the first call is skipped when both are wanted and the second hands the raw colours over as
`colors_precomp2=`.  TEST INFRASTRUCTURE: nothing is written to disk; a replacement whose original text is not found exactly once raises
(the edit documented in INTEGRATION.md stays one that applies to the reference as it is)."""
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


def edited_source(path: str) -> str:
    src = open(path).read()
    for old, new in EDITS:
        if src.count(old) != 1:
            raise RuntimeError(f"the documented edit no longer applies to {path}: {old.strip().splitlines()[0]!r} found {src.count(old)} times")
        src = src.replace(old, new)
    return src


def import_edited_method(method_module):
    """A second module object, `wildgaussians.method_two_colour`, compiled from the staged method.py with EDITS applied (same package:
    its relative imports resolve to the same staged modules; the operator packages are the ones `method_module` bound)."""
    name = "wildgaussians.method_two_colour"
    if name in sys.modules:
        return sys.modules[name]
    path = method_module.__file__
    spec = importlib.util.spec_from_loader(name, loader=None, origin=path)
    mod = importlib.util.module_from_spec(spec)
    mod.__file__ = path
    mod.__package__ = "wildgaussians"
    sys.modules[name] = mod
    exec(compile(edited_source(path), os.path.join(os.path.dirname(path), "method_two_colour.py"), "exec"), mod.__dict__)
    return mod
