"""``diff_gaussian_rasterization._C`` -- binding of the MI355X rasterizer's C-ABI library.

Stands where the reference's pybind11 module of the same name stands
(submodules/diff-gaussian-rasterization/ext.cpp:15-19, rasterize_points.{h,cu}) and exports the same
three functions with the same argument order and return tuples:

    rasterize_gaussians(...)          -> (num_rendered, out_color, radii, geomBuffer, binningBuffer, imgBuffer)
    rasterize_gaussians_backward(...) -> (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh,
                                          dL_dscales, dL_drotations)
    mark_visible(means3D, viewmatrix, projmatrix) -> bool[P]

It is a thin ctypes layer over ``libwg_rasterizer.so`` (include/wg_rasterizer.h): torch is used only to
allocate device memory and to supply the current HIP stream.  There is NO fallback: if the HIP library has
not been built, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import weakref

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("WG_RASTERIZER_LIB") or os.path.join(_HERE, "libwg_rasterizer.so")  # override: another build of the same C-ABI

if not os.path.exists(_LIB_PATH):
    raise ImportError(
        f"{_LIB_PATH} is missing: build the HIP library first "
        "(python wild-gaussians_amd/build.py, or __graft_entry__.build()).  There is no CPU fallback.")

_lib = C.CDLL(_LIB_PATH)

# The compiled torch binding of the same C-ABI (csrc/torch_binding.cpp -> _C_torch*.so, built by build.py --torch-binding; INTEGRATION.md
# section 2): the reference's three functions with the reference's signatures + the `_ex` pair carrying every optional block.  It is the
# DEFAULT whenever it has been built (end of this file); WG_BINDING=ctypes (or use_binding("ctypes")) selects the ctypes code below, which
# covers the same surface and is what serves a variant library (WG_RASTERIZER_LIB).
_torch_ext = None


def use_binding(name: str) -> str:
    """"torch" (the default when built) or "ctypes": which binding serves the calls.  Returns the previous name."""
    global _torch_ext
    prev = "ctypes" if _torch_ext is None else "torch"
    if name == "torch":
        if _LIB_PATH != os.path.join(_HERE, "libwg_rasterizer.so"):
            raise ImportError("the compiled binding links the in-tree libwg_rasterizer.so; WG_RASTERIZER_LIB points elsewhere")
        from . import _C_torch  # noqa: F401  (ImportError when it has not been built: python wild-gaussians_amd/build.py --torch-binding)
        _torch_ext = _C_torch
    elif name == "ctypes":
        _torch_ext = None
    else:
        raise ValueError(name)
    return prev


def binding_name() -> str:
    return "ctypes" if _torch_ext is None else "torch"


from ._abi import _ALLOC_FN, _vp, _i, _f  # noqa: E402

_lib.wg_rasterize_forward.restype = _i
_lib.wg_rasterize_forward.argtypes = [_ALLOC_FN, _vp, _ALLOC_FN, _vp, _ALLOC_FN, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp,
                                      _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _vp, _i, _vp, _vp, _i, _vp]
_lib.wg_rasterize_backward.restype = _i
_lib.wg_rasterize_backward.argtypes = [_i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _f,
                                       _vp, _vp, _vp, _vp, _vp, _vp] + [_vp] * 9 + [_i, _vp]


from ._abi import (_ShTone, _SecondImage, _RawGaussians, _RecolorParent, _CallOptions, _ForwardArgs, _BackwardArgs)  # noqa: E402  (pure ctypes)

_lib.wg_rasterize_forward_ex.restype = _i
_lib.wg_rasterize_forward_ex.argtypes = [C.POINTER(_ForwardArgs)]
_lib.wg_rasterize_backward_ex.restype = _i
_lib.wg_rasterize_backward_ex.argtypes = [C.POINTER(_BackwardArgs)]
_lib.wg_forward_status.restype = _i
_lib.wg_forward_status.argtypes = [_vp, _i, _i, C.POINTER(_i), C.POINTER(_i), _vp]
_lib.wg_mark_visible.restype = _i
_lib.wg_mark_visible.argtypes = [_i, _vp, _vp, _vp, _vp, _vp]
for _name in ("wg_geometry_buffer_size", "wg_binning_buffer_size"):
    getattr(_lib, _name).restype = C.c_size_t
    getattr(_lib, _name).argtypes = [_i]
_lib.wg_image_buffer_size.restype = C.c_size_t
_lib.wg_image_buffer_size.argtypes = [_i, _i]
_lib.wg_image_accumulation_offset.restype = C.c_size_t
_lib.wg_image_accumulation_offset.argtypes = [_i, _i]
for _name in ("wg_status_string",):
    getattr(_lib, _name).restype = C.c_char_p
    getattr(_lib, _name).argtypes = [_i]
_lib.wg_get_option.restype = _i
_lib.wg_get_option.argtypes = [C.c_char_p]
for _name in ("wg_last_hip_error", "wg_version"):
    getattr(_lib, _name).restype = C.c_char_p
    getattr(_lib, _name).argtypes = []


class _GeometryView(C.Structure):
    _fields_ = [(n, _vp) for n in ("depths", "radii", "splats", "cov3D", "clamped", "tiles_touched", "point_offsets")]


class _BinningView(C.Structure):
    _fields_ = [(n, _vp) for n in ("point_list",)]


class _ImageView(C.Structure):
    _fields_ = [(n, _vp) for n in ("final_T", "accumulation", "n_contrib", "ranges", "tile_last", "tile_near", "split", "order_fwd", "order_key", "order_bwd")]


_lib.wg_view_geometry.restype = _i
_lib.wg_view_geometry.argtypes = [_vp, _i, C.POINTER(_GeometryView)]
_lib.wg_view_binning.restype = _i
_lib.wg_view_binning.argtypes = [_vp, _i, C.POINTER(_BinningView)]
_lib.wg_view_image.restype = _i
_lib.wg_view_image.argtypes = [_vp, _i, _i, C.POINTER(_ImageView)]

IMAGE_STATE_ALIGNMENT = 256  # final_T sits at the first 256-byte aligned address of imgBuffer (wg_rasterizer.h)
_acc_offsets = {}


def accumulation_offset(height: int, width: int) -> int:
    """Bytes from final_T to the accumulation array inside imgBuffer (wg_image_accumulation_offset; cached per frame size)."""
    k = (height, width)
    if k not in _acc_offsets:
        _acc_offsets[k] = int(_lib.wg_image_accumulation_offset(int(width), int(height)))
    return _acc_offsets[k]



def _check(status: int, what: str) -> int:
    if status < 0:
        msg = _lib.wg_status_string(status).decode()
        if status == -3:
            msg += ": " + _lib.wg_last_hip_error().decode()
        raise RuntimeError(f"{what} failed: {msg}")
    return status


def _f32(t: torch.Tensor, device) -> torch.Tensor:
    """float32, contiguous, on `device`; zero-sized tensors are the reference's "absent" sentinel."""
    if t.numel() == 0:
        return t
    if t.device != device or t.dtype != torch.float32:
        t = t.to(device=device, dtype=torch.float32)
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t: torch.Tensor):
    return t.data_ptr() if t.numel() != 0 else None


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class _Scratch:
    """One resizable byte tensor per scratch buffer: the role of resizeFunctional (rasterize_points.cu:27-33)."""

    def __init__(self, device):
        self.device = device
        self.tensor = torch.empty(0, dtype=torch.uint8, device=device)
        self.callback = _ALLOC_FN(self._alloc)

    def _alloc(self, nbytes, _user):
        self.tensor = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self.tensor.data_ptr()

    def take(self) -> torch.Tensor:
        """Hand the tensor over and break the self -> callback -> bound method -> self cycle: with the cycle left in place the
        buffer would stay allocated until Python's cyclic collector runs, some tens of steps later -- hundreds of MB of dead
        scratch per step for the caching allocator to cover with fresh (slow, synchronising) device allocations."""
        t, self.tensor, self.callback = self.tensor, None, None
        return t


def _tone_block(sh_tone, device, P, grads=None):
    """sh_tone = (mul [P,3] or None, offset [P,3] or None, pre_clamp_max or None, post_clamp_max or None) -> (_ShTone, keep-alive).
    Beyond the reference: see wg_sh_tone in include/wg_rasterizer.h."""
    mul, offset, pre, post = sh_tone
    keep = []
    t = _ShTone()
    for name, v in (("mul", mul), ("offset", offset)):
        if v is not None:
            v = _f32(v, device)
            if v.numel() != 3 * P:
                raise RuntimeError(f"sh_{name} must have 3 * P elements")
            keep.append(v)
            setattr(t, name, v.data_ptr())
    t.pre_clamp_max = float("inf") if pre is None else float(pre)
    t.post_clamp_max = float("inf") if post is None else float(post)
    if grads is not None:
        t.dL_dmul, t.dL_doffset = (None if g is None else g.data_ptr() for g in grads)
    return t, keep


# ----------------------------------------------------------------------------------------------------------
# Geometry reuse across consecutive calls (option "geometry_reuse", OPT-IN since round 4: default off).  WildGaussians rasterizes the
# same Gaussians through the same camera twice per step -- raw colours, then toned colours (method.py:1573-1611), a third time for
# depth -- and the reference projects, bins and sorts each time.  With the option on the binding remembers the LAST full forward call
# of the calling thread: when the next call hands over the very same geometry tensors (same Python objects, same autograd versions,
# same data pointers: nothing wrote to them that autograd knows of), the same camera tensors and scalars and only other precomputed
# colours, it takes wg_forward_args::recolor -- a copy of the projected state with the new colours and the compositing along the
# parent's sorted lists; binning and image-state buffers are shared with the parent call's (the per-pixel values are identical).
# Identity is by object (weak references), never by address alone: a freed tensor's address can be handed to a new one.
# WHY OPT-IN: a write through `tensor.data` (or `set_()` to equal-shaped storage at the same address) moves no version counter, and the
# binding cannot see it without reading the data back.  The remembered call ends with the first BACKWARD call of the process after it
# (`_reuse_epoch`; on whatever thread autograd runs it), so a training step -- forward calls, backward, optimiser -- is safe; a
# forward-only loop that edits geometry through `.data` between two renders of one camera is not, and a drop-in must not depend on its
# caller never doing that.  Callers that switch it on (wg_integration.apply_optins does) make that promise, or call forget_geometry().
# A parent call whose frame may turn out not to have fit (binning_capacity=, speculative_forward = 2) is never remembered: its child
# would composite along empty lists and return a clean background image where the contract says NaN.
GEOMETRY_REUSE_DEFAULT = 0


class _ReuseState:
    def __init__(self):
        self.last = None        # the remembered forward call (strong references to its scratch buffers)
        self.hits = 0
        self.last_fixed = None


class _PerThread:
    """`_reuse.<attr>` = the calling thread's _ReuseState.  A plain registry instead of threading.local, so that a backward call
    (usually on autograd's thread) can drop EVERY thread's stale entry -- and with it the references that keep a frame's scratch
    buffers alive until that thread's next forward call."""
    _states = {}

    @classmethod
    def state(cls):
        ident = threading.get_ident()
        s = cls._states.get(ident)
        if s is None:
            s = cls._states[ident] = _ReuseState()
        return s

    def __getattr__(self, name):
        return getattr(self.state(), name)

    def __setattr__(self, name, value):
        setattr(self.state(), name, value)


_reuse = _PerThread()
_reuse_epoch = 0   # bumped by every backward call (any thread; the GIL orders it)


def _tensor_token(t):
    """What must be unchanged for a tensor argument to count as 'the same': the object and its version counter (zero-sized
    'absent' sentinels are fresh objects on every call: they match each other)."""
    if t is None or t.numel() == 0:
        return None
    try:
        version = t._version
    except RuntimeError:   # inference-mode tensors track no version: nothing can vouch for "unchanged", so they never match
        return (lambda: None, object(), tuple(t.shape), t.dtype)
    return (weakref.ref(t), version, tuple(t.shape), t.dtype, t.data_ptr(), tuple(t.stride()))


def _same_token(a, b):
    if a is None or b is None:
        return a is None and b is None
    return a[0]() is not None and a[0]() is b[0]() and a[1:] == b[1:]


def _reuse_key(background, means3D, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy,
               kernel_size, subpixel_offset, H, W, prefiltered, device):
    tensors = tuple(_tensor_token(t) for t in (means3D, opacity, scales, rotations, cov3D_precomp, viewmatrix, projmatrix, background,
                                               subpixel_offset))
    scalars = (float(scale_modifier), float(tan_fovx), float(tan_fovy), float(kernel_size), H, W, bool(prefiltered), str(device),
               _stream(device))
    return tensors, scalars


def _reuse_lookup(key):
    last = _reuse.last
    if last is None or last.get("epoch") != _reuse_epoch or last["scalars"] != key[1] or last.get("opts") != key[2] or len(last["tensors"]) != len(key[0]):
        return None
    if not all(_same_token(a, b) for a, b in zip(last["tensors"], key[0])):
        return None
    return last


def forget_geometry():
    """Drop the calling thread's remembered forward call (and the scratch buffers it keeps alive)."""
    _reuse.last = None


# ----------------------------------------------------------------------------------------------------------
# Per-call options (wg_call_options in include/wg_rasterizer.h): the three switches that AFFECT RESULTS travel with every call; the
# library has no process-wide state for them.  A call's values: the `options=` argument (a dict, or the resolved 3-tuple the autograd
# layer keeps on its ctx so that a frame's backward call carries its forward call's values), else the CALLING THREAD's defaults --
# which `call_options(...)` overrides for a `with` block and `set_option(name, value)` sets for the thread (what the tests' and
# bench.py's `--option deterministic_backward=1` use; other threads, and other callers in the process, are not affected).
CALL_OPTION_DEFAULTS = dict(exact_compositing=1, deterministic_backward=0, grad_record=1)
_tls = threading.local()


def _thread_call_options() -> dict:
    d = getattr(_tls, "call_options", None)
    if d is None:
        d = _tls.call_options = dict(CALL_OPTION_DEFAULTS)
    return d


def resolve_call_options(options=None):
    """-> (exact_compositing, deterministic_backward, grad_record) as ints: `options` (dict with any of the three names, None values = not
    given; or an already resolved 3-tuple) over the calling thread's defaults."""
    if isinstance(options, tuple):
        return options
    d = _thread_call_options()
    if options:
        unknown = set(options) - set(CALL_OPTION_DEFAULTS)
        if unknown:
            raise ValueError(f"unknown call option(s) {sorted(unknown)}")
        d = dict(d, **{k: int(v) for k, v in options.items() if v is not None})
    return (int(bool(d["exact_compositing"])), int(bool(d["deterministic_backward"])), int(bool(d["grad_record"])))


import contextlib  # noqa: E402


@contextlib.contextmanager
def call_options(**kw):
    """`with _C.call_options(deterministic_backward=1): ...` -- the calling thread's defaults for the block (for callers that cannot pass
    keywords, e.g. the reference's unedited method.py)."""
    unknown = set(kw) - set(CALL_OPTION_DEFAULTS)
    if unknown:
        raise ValueError(f"unknown call option(s) {sorted(unknown)}")
    d = _thread_call_options()
    saved = dict(d)
    d.update({k: int(v) for k, v in kw.items()})
    try:
        yield
    finally:
        d.clear()
        d.update(saved)


def _options_block(opts):
    return _CallOptions(*opts)


def _forward_args(geom, binning, img, P, degree, M, H, W, background, means3D, sh, colors, opacity, scales, scale_modifier, rotations, cov3D_precomp,
                  viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, kernel_size, subpixel_offset, prefiltered, debug, out_color, radii, device):
    a = _ForwardArgs()
    a.struct_size = C.sizeof(_ForwardArgs)
    a.geometry_alloc = geom.callback
    if binning is not None:
        a.binning_alloc, a.image_alloc = binning.callback, img.callback
    a.P, a.D, a.M, a.width, a.height, a.prefiltered, a.debug = P, int(degree), M, W, H, int(bool(prefiltered)), int(bool(debug))
    a.scale_modifier, a.tan_fovx, a.tan_fovy, a.kernel_size = float(scale_modifier), float(tan_fovx), float(tan_fovy), float(kernel_size)
    for name, t in (("background", background), ("means3D", means3D), ("shs", sh), ("colors_precomp", colors), ("opacities", opacity), ("scales", scales),
                    ("rotations", rotations), ("cov3D_precomp", cov3D_precomp), ("viewmatrix", viewmatrix), ("projmatrix", projmatrix), ("cam_pos", campos),
                    ("subpixel_offset", subpixel_offset)):
        setattr(a, name, None if t is None else _ptr(t))
    a.out_color = out_color.data_ptr()
    a.radii = None if radii is None else radii.data_ptr()
    a.stream = _stream(device)
    return a


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                        projmatrix, tan_fovx, tan_fovy, kernel_size, subpixel_offset, image_height, image_width, sh, degree,
                        campos, prefiltered, debug, sh_tone=None, binning_capacity=None, colors2=None, filter_3D=None, sh_second=None, options=None):
    """The reference's `rasterize_gaussians` (rasterize_points.h:18-40) plus, by keyword, everything beyond it (include/wg_rasterizer.h:
    the optional blocks of wg_rasterize_forward_ex):
    sh_tone: (mul [P,3] | None, offset [P,3] | None, pre_clamp_max | None, post_clamp_max | None) on the SH coefficients (wg_sh_tone).
    sh_second: a 4-tuple like sh_tone (all four may be None): a SECOND image from the same SH coefficients through this tone, composited in
      the same walk; the result then ends with a seventh element, the second image.  Composes with sh_tone and filter_3D.
    colors2: a second [P,3] set of precomputed colours composited in the same walk (wg_second_image); seventh element likewise.
    filter_3D: opacity / scales / rotations are the caller's RAW parameters, get_gaussians() (method.py:1060-1086) runs in-kernel.
    binning_capacity: an int = no host rendezvous at all (capturable in a hipGraph); `rendered` is then the capacity, and
      forward_status(imgBuffer, H, W) tells the real count and whether the frame fit.
    options: per-call options (see resolve_call_options)."""
    opts = resolve_call_options(options)
    if _torch_ext is not None and not _geometry_reuse_on:
        # the usual path: the compiled binding (csrc/torch_binding.cpp) checks its arguments itself (same messages), nothing is remembered
        _reuse.last = None
        res = _torch_ext.rasterize_gaussians_ex(background, means3D, colors, opacity, scales, rotations, float(scale_modifier), cov3D_precomp, viewmatrix,
                                                projmatrix, float(tan_fovx), float(tan_fovy), float(kernel_size), subpixel_offset, int(image_height),
                                                int(image_width), sh, int(degree), campos, bool(prefiltered), bool(debug), sh_tone, binning_capacity, colors2,
                                                filter_3D, sh_second, opts)
        if binning_capacity is not None:
            _reuse.last_fixed = (res[5], int(image_height), int(image_width))
        return res
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:59-61
    if not means3D.is_cuda:
        raise RuntimeError("means3D must live on a HIP device: this rasterizer has no CPU path")
    device = means3D.device
    P, H, W = means3D.size(0), int(image_height), int(image_width)

    if filter_3D is not None and (colors2 is not None or binning_capacity is not None or scales.numel() == 0 or filter_3D.numel() != P):
        raise RuntimeError("filter_3D (raw-parameter mode) needs scales and rotations, P filter values, and neither colors2 nor binning_capacity")
    if sh_second is not None and (colors2 is not None or binning_capacity is not None or sh.numel() == 0 or colors.numel() != 0):
        raise RuntimeError("sh_second (two tones of one SH block) needs SH colours and neither colors2 nor binning_capacity")
    if colors2 is not None:
        if sh_tone is not None or binning_capacity is not None or sh.numel() != 0 or colors.numel() != 3 * P or colors2.numel() != 3 * P:
            raise RuntimeError("colors2 needs precomputed colours of P x 3 in both sets (no SH, no sh_tone, no binning_capacity)")
    if binning_capacity is not None and debug:
        raise RuntimeError("binning_capacity (the fixed-capacity forward) has no debug mode")
    # precomputed colours over remembered geometry: no projection, no binning
    reusable = (P != 0 and sh_tone is None and not debug and colors.numel() == 3 * P and sh.numel() == 0 and colors2 is None
                and filter_3D is None and sh_second is None and _geometry_reuse_on)
    key = None
    if reusable:
        key = _reuse_key(background, means3D, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx,
                         tan_fovy, kernel_size, subpixel_offset, H, W, prefiltered, device) + (opts,)
        last = _reuse_lookup(key)
        if last is not None:
            geom = _Scratch(device)
            out_color = torch.empty((3, H, W), dtype=torch.float32, device=device)
            colors_f, bg_f = _f32(colors, device), _f32(background, device)
            so = None if subpixel_offset is None else _f32(subpixel_offset, device)
            a = _forward_args(geom, None, None, P, 0, 0, H, W, bg_f, None, None, colors_f, None, None, 1.0, None, None, None, None, None, 1.0, 1.0, 0.0, so,
                              False, False, out_color, None, device)
            parent = _RecolorParent(last["geom"].data_ptr(), last["binning"].data_ptr(), last["img"].data_ptr(), int(last["R"]))
            ob = _options_block(opts)
            a.recolor, a.options = C.pointer(parent), C.pointer(ob)
            try:
                with torch.cuda.device(device):
                    rendered = _lib.wg_rasterize_forward_ex(C.byref(a))
            finally:
                child = geom.take()
            _check(rendered, "wg_rasterize_forward_ex (recolor)")
            _reuse.hits += 1
            # (a fresh radii tensor, as every call of the reference returns: 4 bytes per Gaussian)
            return (rendered, out_color, last["radii"].clone(), child, last["binning"], last["img"])
    else:
        _reuse.last = None   # any other kind of call ends the remembered one's reach

    if _torch_ext is not None:   # the compiled binding (csrc/torch_binding.cpp), when it is loaded: the whole surface but the recolouring above
        res = _torch_ext.rasterize_gaussians_ex(background, means3D, colors, opacity, scales, rotations, float(scale_modifier), cov3D_precomp, viewmatrix,
                                                projmatrix, float(tan_fovx), float(tan_fovy), float(kernel_size), subpixel_offset, H, W, sh, int(degree),
                                                campos, bool(prefiltered), bool(debug), sh_tone, binning_capacity, colors2, filter_3D, sh_second, opts)
    else:
        res = _forward_ctypes(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx,
                              tan_fovy, kernel_size, subpixel_offset, H, W, sh, degree, campos, prefiltered, debug, sh_tone, binning_capacity, colors2,
                              filter_3D, sh_second, opts, device, P)
    rendered, radii, buffers = res[0], res[2], res[3:6]
    if binning_capacity is not None:
        _reuse.last_fixed = (buffers[2], H, W)
    # (never a parent whose verdict is not in yet: see "Geometry reuse" above)
    if key is not None and P != 0 and binning_capacity is None and _lib.wg_get_option(b"speculative_forward") != 2:
        _reuse.last = dict(tensors=key[0], scalars=key[1], opts=key[2], R=rendered, radii=radii, geom=buffers[0], binning=buffers[1], img=buffers[2],
                           epoch=_reuse_epoch)
    return res


def _forward_ctypes(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy,
                    kernel_size, subpixel_offset, H, W, sh, degree, campos, prefiltered, debug, sh_tone, binning_capacity, colors2, filter_3D, sh_second,
                    opts, device, P):
    geom, binning, img = _Scratch(device), _Scratch(device), _Scratch(device)
    two = colors2 is not None or sh_second is not None
    if P == 0:  # rasterize_points.cu:83: nothing is launched, the image stays zero
        return (0, torch.zeros((3, H, W), dtype=torch.float32, device=device), torch.zeros((0,), dtype=torch.int32, device=device),
                geom.take(), binning.take(), img.take()) + ((torch.zeros((3, H, W), dtype=torch.float32, device=device),) if two else ())
    # both outputs are fully written by the kernels (every pixel, every Gaussian): no need for the reference's zero fill
    out_color = torch.empty((3, H, W), dtype=torch.float32, device=device)
    radii = torch.empty((P,), dtype=torch.int32, device=device)
    means3D = _f32(means3D, device)
    background, colors, opacity = _f32(background, device), _f32(colors, device), _f32(opacity, device)
    scales, rotations, cov3D_precomp = _f32(scales, device), _f32(rotations, device), _f32(cov3D_precomp, device)
    viewmatrix, projmatrix, campos = _f32(viewmatrix, device), _f32(projmatrix, device), _f32(campos, device)
    subpixel_offset = None if subpixel_offset is None else _f32(subpixel_offset, device)   # None: no offsets, nothing allocated or read
    sh = _f32(sh, device)
    M = sh.size(1) if sh.numel() != 0 else 0  # rasterize_points.cu:85-89
    a = _forward_args(geom, binning, img, P, degree, M, H, W, background, means3D, sh, colors, opacity, scales, scale_modifier, rotations, cov3D_precomp,
                      viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, kernel_size, subpixel_offset, prefiltered, debug, out_color, radii, device)
    keep = []
    if sh_tone is not None:
        tone, k = _tone_block(sh_tone, device, P)
        keep += [tone, k]
        a.tone = C.pointer(tone)
    out_color2 = torch.empty((3, H, W), dtype=torch.float32, device=device) if two else None
    if sh_second is not None:
        tone2, k = _tone_block(sh_second, device, P)
        second = _SecondImage(None, out_color2.data_ptr(), None, None)
        keep += [tone2, k, second]
        a.tone2, a.sh_second, a.second = C.pointer(tone2), 1, C.pointer(second)
    elif colors2 is not None:
        colors2 = _f32(colors2, device)
        second = _SecondImage(colors2.data_ptr(), out_color2.data_ptr(), None, None)
        keep += [colors2, second]
        a.second = C.pointer(second)
    if filter_3D is not None:
        filter_3D = _f32(filter_3D, device)
        rawg = _RawGaussians(filter_3D.data_ptr(), None)
        keep += [filter_3D, rawg]
        a.raw = C.pointer(rawg)
    if binning_capacity is not None:
        if int(binning_capacity) <= 0:
            raise RuntimeError("binning_capacity must be positive")
        a.binning_capacity = int(binning_capacity)
    ob = _options_block(opts)
    a.options = C.pointer(ob)
    try:
        with torch.cuda.device(device):
            rendered = _lib.wg_rasterize_forward_ex(C.byref(a))
    finally:
        buffers = (geom.take(), binning.take(), img.take())
    _check(rendered, "wg_rasterize_forward")
    return (rendered, out_color, radii) + buffers + (() if out_color2 is None else (out_color2,))


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, kernel_size, subpixel_offset, dL_dout_color, sh,
                                 degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug, sh_tone=None, dL_dout_color2=None, raw=None,
                                 sh_second=None, options=None):
    """The reference's `rasterize_gaussians_backward` (rasterize_points.h:42-66) -> the eight tensors, plus by keyword:
    sh_tone: two more tensors are appended, dL_dsh_mul, dL_dsh_offset (None where the input was None); dL_dsh is then the gradient of the raw
      coefficients.
    sh_second (a frame rasterized with sh_second; needs dL_dout_color2): the eight + (dL_dsh_mul, dL_dsh_offset, dL_dsh_mul2, dL_dsh_offset2).
    dL_dout_color2 (a frame rasterized with colors2): the second image's cotangent; the result ends with dL_dcolors2 [P,3].
    raw = (filter_3D, raw_opacities) of a raw-parameter forward call: dL_dopacity / dL_dscales / dL_drotations are the RAW parameters' gradients.
    options: the frame's forward call's per-call options (resolve_call_options)."""
    global _reuse_epoch
    opts = resolve_call_options(options)
    _reuse_epoch += 1   # what the forward calls remembered ends here (see "Geometry reuse" above) ...
    states = _PerThread._states
    for st in list(states.values()):                     # ... and so do the references that kept those frames' scratch alive (any thread's)
        st.last = None
    if len(states) > 8:                                  # (threads that have ended: their states go with them)
        alive = {t.ident for t in threading.enumerate()}
        for ident in [i for i in states if i not in alive]:
            states.pop(ident, None)
    if sh_second is not None and dL_dout_color2 is None:
        raise RuntimeError("sh_second needs the second image's cotangent (dL_dout_color2)")
    record = bool(opts[2] or opts[1])
    dual = dL_dout_color2 is not None and sh_second is None
    P = means3D.size(0)
    if raw is not None and (len(raw) != 2 or raw[0].numel() != P or raw[1].numel() != P):
        raise RuntimeError("raw = (filter_3D, raw_opacities) must have P elements each")
    if raw is not None and (not record or dual):
        raise RuntimeError("the raw-parameter backward pass needs grad_record = 1 and cannot be combined with the two-colour call")
    if (dual or sh_second is not None) and ((dual and sh_tone is not None) or (not record and P != 0)):
        raise RuntimeError("the two-colour backward pass needs the gradient record (grad_record = 1 or deterministic_backward = 1) and no sh_tone")
    if _torch_ext is not None:
        return _torch_ext.rasterize_gaussians_backward_ex(background, means3D, radii, colors, scales, rotations, float(scale_modifier), cov3D_precomp,
                                                          viewmatrix, projmatrix, float(tan_fovx), float(tan_fovy), float(kernel_size), subpixel_offset,
                                                          dL_dout_color, sh, int(degree), campos, geomBuffer, int(R), binningBuffer, imageBuffer,
                                                          bool(debug), sh_tone, dL_dout_color2, raw, sh_second, opts)
    device = means3D.device
    H, W = dL_dout_color.size(1), dL_dout_color.size(2)
    sh = _f32(sh, device)
    M = sh.size(1) if sh.numel() != 0 else 0

    # The reference zero-fills nine gradient tensors per call (rasterize_points.cu:157-165).  With the gradient record (the default,
    # include/wg_rasterizer.h) the library clears its own accumulator and every output is fully written by the per-Gaussian kernel
    # (zeros for culled Gaussians); without it the four accumulation targets of the per-tile pass share one zeroed allocation.
    record = P != 0 and record
    if record:
        dL_dconic = None   # the reference's intermediate: not returned (rasterize_points.cu:201), so not requested
        dL_dmeans2D = torch.empty((P, 3), dtype=torch.float32, device=device)
        dL_dopacity = torch.empty((P, 1), dtype=torch.float32, device=device)
        dL_dcolors = torch.empty((P, 3), dtype=torch.float32, device=device)
    else:
        acc = torch.zeros((P * 11,), dtype=torch.float32, device=device)
        dL_dconic = acc[0:4 * P].view(P, 2, 2)
        dL_dmeans2D = acc[4 * P:7 * P].view(P, 3)
        dL_dcolors = acc[7 * P:10 * P].view(P, 3)
        dL_dopacity = acc[10 * P:11 * P].view(P, 1)
    have_scales = scales.numel() != 0
    alloc = torch.empty if P != 0 else torch.zeros
    dL_dmeans3D = alloc((P, 3), dtype=torch.float32, device=device)
    dL_dcov3D = alloc((P, 6), dtype=torch.float32, device=device)
    dL_dsh = alloc((P, M, 3), dtype=torch.float32, device=device)
    dL_dscales = (alloc if have_scales else torch.zeros)((P, 3), dtype=torch.float32, device=device)
    dL_drotations = (alloc if have_scales else torch.zeros)((P, 4), dtype=torch.float32, device=device)
    dL_dcolors2 = alloc((P, 3), dtype=torch.float32, device=device) if dual else None
    tone_grads = None
    if sh_tone is not None:
        tone_grads = tuple(None if v is None else alloc((P, 3), dtype=torch.float32, device=device) for v in sh_tone[:2])
    tone2_grads = None
    if sh_second is not None:
        tone2_grads = tuple(None if v is None else alloc((P, 3), dtype=torch.float32, device=device) for v in sh_second[:2])

    if P != 0:
        means3D = _f32(means3D, device)
        background, colors = _f32(background, device), _f32(colors, device)
        scales, rotations, cov3D_precomp = _f32(scales, device), _f32(rotations, device), _f32(cov3D_precomp, device)
        viewmatrix, projmatrix, campos = _f32(viewmatrix, device), _f32(projmatrix, device), _f32(campos, device)
        subpixel_offset = None if subpixel_offset is None else _f32(subpixel_offset, device)
        dL_dout_color = _f32(dL_dout_color, device)
        radii = radii if radii.is_contiguous() else radii.contiguous()
        a = _BackwardArgs()
        a.struct_size = C.sizeof(_BackwardArgs)
        a.P, a.D, a.M, a.R, a.width, a.height, a.debug = P, int(degree), M, int(R), W, H, int(bool(debug))
        a.scale_modifier, a.tan_fovx, a.tan_fovy, a.kernel_size = float(scale_modifier), float(tan_fovx), float(tan_fovy), float(kernel_size)
        for name, t in (("background", background), ("means3D", means3D), ("shs", sh), ("colors_precomp", colors), ("scales", scales),
                        ("rotations", rotations), ("cov3D_precomp", cov3D_precomp), ("viewmatrix", viewmatrix), ("projmatrix", projmatrix),
                        ("campos", campos), ("subpixel_offset", subpixel_offset), ("radii", radii), ("dL_dpix", dL_dout_color), ("dL_dsh", dL_dsh)):
            setattr(a, name, None if t is None else _ptr(t))
        a.geom_buffer, a.binning_buffer, a.image_buffer = geomBuffer.data_ptr(), binningBuffer.data_ptr(), imageBuffer.data_ptr()
        a.dL_dmean2D, a.dL_dconic = dL_dmeans2D.data_ptr(), None if dL_dconic is None else dL_dconic.data_ptr()
        a.dL_dopacity, a.dL_dcolor, a.dL_dmean3D, a.dL_dcov3D = dL_dopacity.data_ptr(), dL_dcolors.data_ptr(), dL_dmeans3D.data_ptr(), dL_dcov3D.data_ptr()
        a.dL_dscale, a.dL_drot, a.stream = dL_dscales.data_ptr(), dL_drotations.data_ptr(), _stream(device)
        keep = []
        if sh_tone is not None:
            tone, k = _tone_block(sh_tone, device, P, tone_grads)
            keep += [tone, k]
            a.tone = C.pointer(tone)
        if sh_second is not None:
            dL2 = _f32(dL_dout_color2, device)
            tone2, k = _tone_block(sh_second, device, P, tone2_grads)
            second = _SecondImage(None, None, dL2.data_ptr(), None)
            keep += [dL2, tone2, k, second]
            a.tone2, a.sh_second, a.second = C.pointer(tone2), 1, C.pointer(second)
        elif dual:
            dL2 = _f32(dL_dout_color2, device)
            second = _SecondImage(None, None, dL2.data_ptr(), dL_dcolors2.data_ptr())
            keep += [dL2, second]
            a.second = C.pointer(second)
        if raw is not None:
            f3d, rop = _f32(raw[0], device), _f32(raw[1], device)
            rawg = _RawGaussians(f3d.data_ptr(), rop.data_ptr())
            keep += [f3d, rop, rawg]
            a.raw = C.pointer(rawg)
        ob = _options_block(opts)
        a.options = C.pointer(ob)
        with torch.cuda.device(device):
            status = _lib.wg_rasterize_backward_ex(C.byref(a))
        _check(status, "wg_rasterize_backward")
    out = (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)
    if sh_second is not None:
        return out + (tone_grads or (None, None)) + tone2_grads
    if dual:
        return out + (dL_dcolors2,)
    return out if sh_tone is None else out + tone_grads   # (raw-parameter mode: same tuple, slots 2, 6, 7 are gradients of the raw parameters)


def forward_status(imageBuffer, H, W):
    """(num_rendered, fits) of the forward call that produced imageBuffer (wg_forward_status: synchronises the current stream)."""
    n, f = _i(0), _i(0)
    with torch.cuda.device(imageBuffer.device):
        _check(_lib.wg_forward_status(imageBuffer.data_ptr(), int(W), int(H), C.byref(n), C.byref(f), _stream(imageBuffer.device)), "wg_forward_status")
    return int(n.value), bool(f.value)


def last_forward_status():
    """forward_status of the calling thread's latest binning_capacity= (fixed-capacity) forward call; None if there was none."""
    last = _reuse.last_fixed
    return None if last is None else forward_status(*last)


def mark_visible(means3D, viewmatrix, projmatrix):
    if _torch_ext is not None:
        return _torch_ext.mark_visible(means3D, viewmatrix, projmatrix)
    device = means3D.device
    P = means3D.size(0)
    present = torch.zeros((P,), dtype=torch.bool, device=device)
    if P != 0:
        means3D, viewmatrix, projmatrix = _f32(means3D, device), _f32(viewmatrix, device), _f32(projmatrix, device)
        with torch.cuda.device(device):
            _check(_lib.wg_mark_visible(P, _ptr(means3D), _ptr(viewmatrix), _ptr(projmatrix), present.data_ptr(),
                                        _stream(device)), "wg_mark_visible")
    return present


# ----------------------------------------------------------------------------------------------------------
# Introspection for the parity tests: typed views into the opaque scratch buffers (device tensors, no copy).
def _from_ptr(ptr, shape, dtype, owner):
    # torch.frombuffer cannot wrap device memory; the CUDA array interface can
    class _Holder:
        pass

    h = _Holder()
    n = 1
    for s in shape:
        n *= s
    typestr = {torch.float32: "<f4", torch.int32: "<i4", torch.uint8: "|u1", torch.int64: "<i8"}[dtype]
    h.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}
    t = torch.as_tensor(h, device=owner.device) if n else torch.empty(shape, dtype=dtype, device=owner.device)
    t._wg_owner = owner  # keep the scratch buffer alive
    return t


def view_geometry(geomBuffer, P):
    v = _GeometryView()
    _check(_lib.wg_view_geometry(geomBuffer.data_ptr(), int(P), C.byref(v)), "wg_view_geometry")
    return dict(depths=_from_ptr(v.depths, (P,), torch.float32, geomBuffer), radii=_from_ptr(v.radii, (P,), torch.int32, geomBuffer),
                splats=_from_ptr(v.splats, (P, 12), torch.float32, geomBuffer), cov3D=_from_ptr(v.cov3D, (P, 6), torch.float32, geomBuffer),
                clamped=_from_ptr(v.clamped, (P,), torch.uint8, geomBuffer),
                tiles_touched=_from_ptr(v.tiles_touched, (P,), torch.int32, geomBuffer),
                point_offsets=_from_ptr(v.point_offsets, (P,), torch.int32, geomBuffer))


def view_binning(binningBuffer, R):
    v = _BinningView()
    _check(_lib.wg_view_binning(binningBuffer.data_ptr(), int(R), C.byref(v)), "wg_view_binning")
    return dict(point_list=_from_ptr(v.point_list, (R,), torch.int32, binningBuffer))


def view_image(imageBuffer, H, W):
    v = _ImageView()
    _check(_lib.wg_view_image(imageBuffer.data_ptr(), int(W), int(H), C.byref(v)), "wg_view_image")
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    return dict(final_T=_from_ptr(v.final_T, (H, W), torch.float32, imageBuffer),
                accumulation=_from_ptr(v.accumulation, (H, W), torch.float32, imageBuffer),
                n_contrib=_from_ptr(v.n_contrib, (H, W), torch.int32, imageBuffer),
                ranges=_from_ptr(v.ranges, (tiles, 2), torch.int32, imageBuffer),
                tile_last=_from_ptr(v.tile_last, (tiles,), torch.int32, imageBuffer),
                tile_near=_from_ptr(v.tile_near, (tiles,), torch.int32, imageBuffer),
                split=_from_ptr(v.split, (2,), torch.int32, imageBuffer),
                order_fwd=_from_ptr(v.order_fwd, (tiles,), torch.int32, imageBuffer),
                order_key=_from_ptr(v.order_key, (4,), torch.int32, imageBuffer),
                order_bwd=_from_ptr(v.order_bwd, (tiles,), torch.int32, imageBuffer))


_lib.wg_set_option.restype = _i
_lib.wg_set_option.argtypes = [C.c_char_p, _i]


_geometry_reuse_on = False   # the option "geometry_reuse" is read by this binding only: kept here too, so that no call has to ask the library


def set_option(name: str, value: int) -> None:
    """wg_set_option: a tuning switch of the library (process-wide; every setting gives identical results: docs/OPTIONS.md), e.g.
    set_option("force_global_sort", 1).  For the three RESULT-AFFECTING names (exact_compositing, deterministic_backward, grad_record), which
    the library takes per call, it sets the CALLING THREAD's default instead (see call_options)."""
    global _geometry_reuse_on
    if name in CALL_OPTION_DEFAULTS:
        _thread_call_options()[name] = int(bool(int(value)))
    else:
        _check(_lib.wg_set_option(name.encode(), int(value)), f"wg_set_option({name})")
        if name == "geometry_reuse":
            _geometry_reuse_on = bool(int(value))
    forget_geometry()   # a remembered forward call was made under the old options


def geometry_reuse_hits() -> int:
    """How many calls of this thread took the geometry-reuse path so far."""
    return _reuse.hits


def get_option(name: str) -> int:
    """wg_get_option: current value of a library option (-1: unknown name); for the three per-call names, the calling thread's default."""
    if name in CALL_OPTION_DEFAULTS:
        return int(_thread_call_options()[name])
    return int(_lib.wg_get_option(name.encode()))


def version() -> str:
    return _lib.wg_version().decode()


# ----------------------------------------------------------------------------------------------------------
# Per-stage HIP-event timing (wg_profile_* in wg_rasterizer.h); used by bench.py for the roofline object.
STAGE_COUNT = 9  # WG_STAGE_COUNT (include/wg_rasterizer.h)


class _StageTimes(C.Structure):
    _fields_ = [("total_ms", C.c_double * STAGE_COUNT), ("launches", C.c_longlong * STAGE_COUNT)]


_lib.wg_profile_enable.restype = _i
_lib.wg_profile_enable.argtypes = [_i]
_lib.wg_profile_reset.restype = _i
_lib.wg_profile_reset.argtypes = []
_lib.wg_profile_read.restype = _i
_lib.wg_profile_read.argtypes = [C.POINTER(_StageTimes)]
_lib.wg_stage_name.restype = C.c_char_p
_lib.wg_stage_name.argtypes = [_i]


def profile_enable(on: bool) -> None:
    _check(_lib.wg_profile_enable(int(bool(on))), "wg_profile_enable")


def profile_reset() -> None:
    _check(_lib.wg_profile_reset(), "wg_profile_reset")


def profile_read() -> dict:
    """{stage name: (total_ms, launches)} accumulated since the last profile_reset()."""
    st = _StageTimes()
    _check(_lib.wg_profile_read(C.byref(st)), "wg_profile_read")
    return {_lib.wg_stage_name(i).decode(): (float(st.total_ms[i]), int(st.launches[i])) for i in range(STAGE_COUNT)}

# WG_OPTIONS="name=value,name=value": library options applied at import (e.g. WG_OPTIONS=force_global_sort=1).  One of the three per-call
# names sets the PROCESS default (CALL_OPTION_DEFAULTS: what every thread starts from, also threads created later), not only the importing
# thread's -- WG_OPTIONS=deterministic_backward=1 has to reach worker and view-parallel threads too.
for _kv in filter(None, os.environ.get("WG_OPTIONS", "").split(",")):
    _k, _, _v = _kv.partition("=")
    _k = _k.strip()
    if _k in CALL_OPTION_DEFAULTS:
        CALL_OPTION_DEFAULTS[_k] = int(bool(int(_v or "1")))
        _thread_call_options()[_k] = CALL_OPTION_DEFAULTS[_k]
    else:
        set_option(_k, int(_v or "1"))


def set_default_call_option(name: str, value: int) -> None:
    """The PROCESS-wide default of a per-call option: what threads that have not set their own start from (set_option / call_options stay
    per thread).  Threads that already resolved their defaults keep them; the calling thread takes the new value."""
    if name not in CALL_OPTION_DEFAULTS:
        raise ValueError(name)
    CALL_OPTION_DEFAULTS[name] = int(bool(int(value)))
    _thread_call_options()[name] = CALL_OPTION_DEFAULTS[name]


# Which binding serves the calls: the compiled module (csrc/torch_binding.cpp -> _C_torch*.so) when it has been built and links this very
# library -- it carries the whole surface and costs less host time per call --, the ctypes code above otherwise.  WG_BINDING=ctypes|torch
# forces one (torch: ImportError if it has not been built).
_want = os.environ.get("WG_BINDING", "")
if _want:
    use_binding(_want)
elif _LIB_PATH == os.path.join(_HERE, "libwg_rasterizer.so"):
    try:
        use_binding("torch")
    except ImportError:
        pass
