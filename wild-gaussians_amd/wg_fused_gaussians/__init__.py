"""Fused Gaussian activations + 3-D filter -- opt-in replacement of GaussianModel.get_gaussians's arithmetic
(wildgaussians/method.py:1060-1086).

    from wg_fused_gaussians import activate
    opacities, scales, rotations = activate(raw_opacities, raw_scales, raw_rotations, filter_3D)

with `raw_opacities` [P,1] logits, `raw_scales` [P,3] log-scales, `raw_rotations` [P,4] unnormalised quaternions and the
`filter_3D` [P,1] buffer; the results equal

    rotations = F.normalize(raw_rotations);  s = exp(raw_scales);  scales = sqrt(s^2 + filter_3D^2)
    opacities = sigmoid(raw_opacities) * sqrt(prod(s^2) / prod(s^2 + filter_3D^2))[:, None]

One HIP kernel forward, one backward (include/wg_activations.h, csrc/activations.hip); gradients flow to the three raw
parameters.  No CPU path: float32 tensors on a HIP device.

SURVEY.md 8f N4, the per-Gaussian bookkeeping after the backward pass (method.py:1995-1998, :1470-1477), in place, one kernel,
no host synchronisation (the torch code runs six boolean-mask index operations, each with a nonzero()):

    from wg_fused_gaussians import add_densification_stats
    add_densification_stats(radii, viewspace_points.grad, model.xyz_grad, model.denom, max_radii2D=model.max_radii2D,
                            xyz_gradient_accum_abs=model.xyz_gradient_accum_abs,
                            xyz_gradient_accum_abs_max=model.xyz_gradient_accum_abs_max)
"""
from __future__ import annotations

import ctypes as C

import torch

from diff_gaussian_rasterization import _C as _native

_lib = _native._lib
_vp, _i = C.c_void_p, C.c_int
_lib.wg_activations_forward.restype = _i
_lib.wg_activations_forward.argtypes = [_i] + [_vp] * 8
_lib.wg_activations_backward.restype = _i
_lib.wg_activations_backward.argtypes = [_i] + [_vp] * 11
_lib.wg_densification_stats.restype = _i
_lib.wg_densification_stats.argtypes = [_i] + [_vp] * 8


def _prep(t, cols, name):
    if not (t.is_cuda and t.dtype == torch.float32):
        raise RuntimeError(f"wg_fused_gaussians: {name} must be a float32 tensor on a HIP device (there is no CPU path)")
    if t.numel() % cols:
        raise RuntimeError(f"wg_fused_gaussians: {name} has {t.numel()} elements, not a multiple of {cols}")
    return t.contiguous()


class _Activate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw_opacities, raw_scales, raw_rotations, filter_3D):
        o, s, r, f = _prep(raw_opacities, 1, "raw_opacities"), _prep(raw_scales, 3, "raw_scales"), \
            _prep(raw_rotations, 4, "raw_rotations"), _prep(filter_3D, 1, "filter_3D")
        P = o.numel()
        if s.numel() != 3 * P or r.numel() != 4 * P or f.numel() != P:
            raise RuntimeError("wg_fused_gaussians: inconsistent numbers of Gaussians")
        rot, sc, op = torch.empty_like(r), torch.empty_like(s), torch.empty_like(o)
        stream = torch.cuda.current_stream(o.device).cuda_stream
        with torch.cuda.device(o.device):
            _native._check(_lib.wg_activations_forward(P, r.data_ptr(), s.data_ptr(), o.data_ptr(), f.data_ptr(), rot.data_ptr(),
                                                       sc.data_ptr(), op.data_ptr(), stream), "wg_activations_forward")
        ctx.save_for_backward(o, s, r, f)
        return op, sc, rot

    @staticmethod
    def backward(ctx, g_op, g_sc, g_rot):
        o, s, r, f = ctx.saved_tensors
        P = o.numel()
        go, gs, gr = torch.empty_like(o), torch.empty_like(s), torch.empty_like(r)
        ptr = lambda t: t.contiguous().data_ptr() if t is not None else None
        g_op, g_sc, g_rot = [None if t is None else t.contiguous() for t in (g_op, g_sc, g_rot)]
        stream = torch.cuda.current_stream(o.device).cuda_stream
        with torch.cuda.device(o.device):
            _native._check(_lib.wg_activations_backward(P, r.data_ptr(), s.data_ptr(), o.data_ptr(), f.data_ptr(), ptr(g_rot), ptr(g_sc),
                                                        ptr(g_op), gr.data_ptr(), gs.data_ptr(), go.data_ptr(), stream),
                           "wg_activations_backward")
        return go, gs, gr, None


def activate(raw_opacities, raw_scales, raw_rotations, filter_3D):
    """-> (opacities [like raw_opacities], scales [like raw_scales], rotations [like raw_rotations])."""
    return _Activate.apply(raw_opacities, raw_scales, raw_rotations, filter_3D)


def add_densification_stats(radii, viewspace_grad, xyz_grad, denom, max_radii2D=None, xyz_gradient_accum_abs=None,
                            xyz_gradient_accum_abs_max=None):
    """In place, for every Gaussian with radii > 0 (the loop's visibility_filter): xyz_grad += |grad[:, :2]|, denom += 1,
    max_radii2D = max(max_radii2D, radii) and, when the two GOF buffers are given, xyz_gradient_accum_abs += |grad[:, 2]|,
    xyz_gradient_accum_abs_max = max(., |grad[:, 2]|).  `radii` int32 [P], `viewspace_grad` float32 [P, 3], the rest float32 with
    P elements ([P] or [P, 1]), all contiguous on one HIP device.  (A NaN gradient does not propagate into the two max buffers.)"""
    P = radii.numel()
    if not (radii.is_cuda and radii.dtype == torch.int32 and radii.is_contiguous()):
        raise RuntimeError("wg_fused_gaussians: radii must be a contiguous int32 tensor on a HIP device (there is no CPU path)")
    if (xyz_gradient_accum_abs is None) != (xyz_gradient_accum_abs_max is None):
        raise RuntimeError("wg_fused_gaussians: pass both GOF buffers or neither")
    g = _prep(viewspace_grad, 3, "viewspace_grad")
    if g.numel() != 3 * P:
        raise RuntimeError("wg_fused_gaussians: viewspace_grad must be [P, 3]")
    bufs = []
    for name, t in (("xyz_grad", xyz_grad), ("xyz_gradient_accum_abs", xyz_gradient_accum_abs),
                    ("xyz_gradient_accum_abs_max", xyz_gradient_accum_abs_max), ("denom", denom), ("max_radii2D", max_radii2D)):
        if t is None:
            bufs.append(None)
            continue
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == P):
            raise RuntimeError(f"wg_fused_gaussians: {name} must be a contiguous float32 tensor of {P} elements on the HIP device")
        bufs.append(t.data_ptr())
    stream = torch.cuda.current_stream(radii.device).cuda_stream
    with torch.cuda.device(radii.device):
        _native._check(_lib.wg_densification_stats(P, radii.data_ptr(), g.data_ptr(), bufs[0], bufs[1], bufs[2], bufs[3], bufs[4], stream),
                       "wg_densification_stats")


# ---- fused Adam (include/wg_adam.h, csrc/adam.hip) -------------------------------------------------------------
class _AdamTensor(C.Structure):
    _fields_ = [("param", _vp), ("grad", _vp), ("exp_avg", _vp), ("exp_avg_sq", _vp), ("numel", C.c_size_t),
                ("beta2", C.c_float), ("one_minus_beta1", C.c_float), ("one_minus_beta2", C.c_float), ("step_size", C.c_float),
                ("bias_correction2_sqrt", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float)]


_lib.wg_fused_adam.restype = _i
_lib.wg_fused_adam.argtypes = [_i, C.POINTER(_AdamTensor), _vp]


class FusedAdam(torch.optim.Adam):
    """``torch.optim.Adam`` as the reference builds it (wildgaussians/method.py:1030-1049: one parameter group per Gaussian attribute,
    per-group ``lr`` / ``weight_decay``, ``eps=1e-15``) with ``step()`` (method.py:2019) as ONE kernel launch over all parameters
    instead of torch's ~ten elementwise passes per tensor:

        torch.optim.Adam = wg_fused_gaussians.FusedAdam      # before the model builds its optimizer, or edit that one line

    It IS a ``torch.optim.Adam``: ``param_groups``, ``state`` (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter, the layout the
    reference's densification code edits in place: method.py:1094-1102, 1268-1278, 1284-1297, 1312-1328), ``state_dict`` /
    ``load_state_dict``, ``zero_grad`` and the learning-rate schedule (``param_group['lr'] = ...``, method.py:1206-1210) are
    inherited; a state dict moves freely between the two classes.  Parameters must be float32 on a HIP device (others, and any
    group with ``amsgrad`` / ``maximize``, raise: there is no silent fallback); parameters without a gradient are skipped, as torch
    does.  The update is torch's, operation for operation in float32 (include/wg_adam.h)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, **kw):
        for k in ("amsgrad", "maximize", "capturable", "differentiable"):
            if kw.get(k):
                raise NotImplementedError(f"wg_fused_gaussians.FusedAdam: {k}=True is not implemented")
        kw.pop("foreach", None)
        kw.pop("fused", None)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, foreach=False, fused=False,
                         **{k: v for k, v in kw.items() if k in ("amsgrad", "maximize", "capturable", "differentiable")})

    @classmethod
    def adopt(cls, optimizer: torch.optim.Adam) -> "FusedAdam":
        """A FusedAdam over the SAME parameters, group options (``lr``, ``name``, ``weight_decay`` ...) and state tensors as an existing
        ``torch.optim.Adam`` -- for callers whose code constructs the optimizer itself (wg_integration.apply_optins)."""
        groups = [{k: v for k, v in g.items() if k not in ("foreach", "fused")} for g in optimizer.param_groups]
        new = cls(groups, **{k: optimizer.defaults[k] for k in ("lr", "betas", "eps", "weight_decay")})
        for p, st in optimizer.state.items():
            new.state[p] = st
        return new

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        by_device = {}
        for group in self.param_groups:
            if group.get("amsgrad") or group.get("maximize") or group.get("decoupled_weight_decay"):
                raise NotImplementedError("wg_fused_gaussians.FusedAdam: amsgrad / maximize / decoupled_weight_decay are not implemented")
            beta1, beta2 = group["betas"]
            if not (0.5 < beta1 < 1.0 and 0.0 <= beta2 < 1.0):
                raise NotImplementedError("wg_fused_gaussians.FusedAdam: betas outside (0.5, 1) x [0, 1) are not implemented")
            lr = float(group["lr"])
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                if g.is_sparse:
                    raise RuntimeError("wg_fused_gaussians.FusedAdam does not support sparse gradients")
                if not (p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and p.is_contiguous()):
                    raise RuntimeError("wg_fused_gaussians.FusedAdam: parameters must be contiguous float32 tensors on a HIP device "
                                       "(there is no CPU path)")
                st = self.state[p]
                if len(st) == 0:   # torch's lazy state initialisation (torch/optim/adam.py: _init_group)
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if torch.is_tensor(st["step"]):
                    st["step"] += 1   # in place: a host scalar (a state dict loaded from torch.optim.Adam keeps it that way)
                    step = float(st["step"])
                else:
                    st["step"] = step = st["step"] + 1
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if not (m.is_contiguous() and v.is_contiguous() and m.shape == p.shape and v.shape == p.shape
                        and m.dtype == torch.float32 and v.dtype == torch.float32 and m.device == p.device and v.device == p.device):
                    raise RuntimeError("wg_fused_gaussians.FusedAdam: exp_avg / exp_avg_sq must be contiguous float32 tensors shaped "
                                       "and placed like their parameter")
                g = g if g.is_contiguous() else g.contiguous()
                # the step's scalars in double precision, rounded once when they enter the struct -- as torch's Python computes them
                d = _AdamTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), beta2, 1.0 - beta1, 1.0 - beta2,
                                lr / (1.0 - beta1 ** step), (1.0 - beta2 ** step) ** 0.5, float(group["eps"]), float(group["weight_decay"]))
                by_device.setdefault(p.device, []).append((d, g))   # g: keeps a contiguous copy alive until the launch is queued
                # the kernel writes p, m and v behind torch's back: say so, as an in-place torch op would (autograd's saved-tensor
                # checks and the rasterizer binding's geometry reuse both read the version counter)
                torch.autograd.graph.increment_version(p)
                torch.autograd.graph.increment_version(m)
                torch.autograd.graph.increment_version(v)
        for dev, items in by_device.items():
            arr = (_AdamTensor * len(items))(*[d for d, _ in items])
            with torch.cuda.device(dev):
                _native._check(_lib.wg_fused_adam(len(items), arr, torch.cuda.current_stream(dev).cuda_stream), "wg_fused_adam")
        return loss


# ---- fused eval_sh (include/wg_sh_eval.h, csrc/sh_eval.hip) ------------------------------------------------------
_lib.wg_eval_sh_forward.restype = _i
_lib.wg_eval_sh_forward.argtypes = [_i, _i, _i, _vp, _vp, _vp, _vp]
_lib.wg_eval_sh_backward.restype = _i
_lib.wg_eval_sh_backward.argtypes = [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]


class _EvalSH(torch.autograd.Function):
    @staticmethod
    def forward(ctx, deg, sh, dirs):
        P, K = sh.shape[0], sh.shape[2]
        out = torch.empty((P, 3), device=sh.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(sh.device).cuda_stream
        with torch.cuda.device(sh.device):
            _native._check(_lib.wg_eval_sh_forward(P, deg, K, sh.data_ptr(), dirs.data_ptr(), out.data_ptr(), stream), "wg_eval_sh_forward")
        ctx.deg = deg
        ctx.save_for_backward(sh, dirs)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        sh, dirs = ctx.saved_tensors
        P, K = sh.shape[0], sh.shape[2]
        g = grad_out.contiguous()
        need_sh, need_dirs = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        if not (need_sh or need_dirs):
            return None, None, None
        grad_sh = torch.empty_like(sh)   # written whole by the kernel (it is the cheaper of the two outputs to always produce)
        grad_dirs = torch.empty_like(dirs) if need_dirs else None
        stream = torch.cuda.current_stream(sh.device).cuda_stream
        with torch.cuda.device(sh.device):
            _native._check(_lib.wg_eval_sh_backward(P, ctx.deg, K, sh.data_ptr(), dirs.data_ptr(), g.data_ptr(), grad_sh.data_ptr(),
                                                    None if grad_dirs is None else grad_dirs.data_ptr(), stream), "wg_eval_sh_backward")
        return None, grad_sh if need_sh else None, grad_dirs


def eval_sh(deg, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """The reference's ``eval_sh(deg, sh, dirs)`` (wildgaussians/method.py:493-548) for ``sh`` [..., 3, K] and ``dirs`` [..., 3], degrees
    0..3, as one kernel forward and one backward (gradients to ``sh`` and ``dirs``) instead of ~60 elementwise kernels over strided
    slices and, backward, a zero-fill + slice-add of a [..., 3, K] tensor per coefficient.  float32 on a HIP device; no CPU path."""
    deg = int(deg)
    if not 0 <= deg <= 3:
        raise NotImplementedError("wg_fused_gaussians.eval_sh: degrees 0..3 are implemented")
    if sh.shape[-2] != 3 or dirs.shape[-1] != 3 or sh.shape[:-2] != dirs.shape[:-1] or sh.shape[-1] < (deg + 1) ** 2:
        raise RuntimeError("wg_fused_gaussians.eval_sh: expected sh [..., 3, K >= (deg + 1)^2] and dirs [..., 3] with equal leading dimensions")
    for name, t in (("sh", sh), ("dirs", dirs)):
        if not (t.is_cuda and t.dtype == torch.float32):
            raise RuntimeError(f"wg_fused_gaussians.eval_sh: {name} must be a float32 tensor on a HIP device (there is no CPU path)")
    lead = sh.shape[:-2]
    out = _EvalSH.apply(deg, sh.reshape(-1, 3, sh.shape[-1]).contiguous(), dirs.reshape(-1, 3).contiguous())
    return out.reshape(*lead, 3)
