"""Fused Adam (include/wg_adam.h, wg_fused_gaussians.FusedAdam) against torch.optim.Adam -- the optimizer the reference
builds at wildgaussians/method.py:1030-1049 and steps at :2019 -- in lockstep on identical parameters and gradients, including the
in-place state surgery of the reference's densification code (method.py:1094-1102, 1284-1297, 1312-1328)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _groups(dev, seed=0, n=1000):
    g = torch.Generator(device="cpu").manual_seed(seed)
    mk = lambda *shape: torch.nn.Parameter(torch.randn(*shape, generator=g).to(dev))
    return [
        {"params": [mk(n, 3)], "lr": 1.6e-4, "name": "xyz"},
        {"params": [mk(n, 1, 3)], "lr": 2.5e-3, "name": "features_dc"},
        {"params": [mk(n, 1)], "lr": 5e-2, "name": "opacities"},
        {"params": [mk(n, 3)], "lr": 5e-3, "name": "scales"},
        {"params": [mk(n, 4)], "lr": 1e-3, "name": "rotations"},
        {"params": [mk(7, 32)], "lr": 1e-3, "name": "appearance_embeddings", "weight_decay": 0.01},
        {"params": [mk(n, 24)], "lr": 5e-3, "name": "embeddings"},
        {"params": [mk(n, 15, 3)], "lr": 2.5e-3 / 20, "name": "features_rest"},
        {"params": [mk(64, 67), mk(64), mk(13), mk(1)], "lr": 5e-4, "name": "appearance_mlp"},   # odd sizes: the ragged tails
    ]


def _pair(dev, **kw):
    from wg_fused_gaussians import FusedAdam
    ga = _groups(dev, **kw)
    gb = copy.deepcopy(ga)
    for a, b in zip(ga, gb):
        b["params"] = [torch.nn.Parameter(p.detach().clone()) for p in a["params"]]
    return torch.optim.Adam(ga, lr=1.0, eps=1e-15), FusedAdam(gb, lr=1.0, eps=1e-15)


def _set_grads(ref, fused, step, skip=()):
    g = torch.Generator(device="cpu").manual_seed(1000 + step)
    for ga, gb in zip(ref.param_groups, fused.param_groups):
        for pa, pb in zip(ga["params"], gb["params"]):
            if ga["name"] in skip:
                pa.grad = pb.grad = None
                continue
            gr = (torch.randn(pa.shape, generator=g) * (10.0 ** float(torch.randint(-6, 1, (1,), generator=g)))).to(pa.device)
            pa.grad, pb.grad = gr.clone(), gr.clone()


def _assert_same(ref, fused, rtol=2e-6, what=""):
    for ga, gb in zip(ref.param_groups, fused.param_groups):
        for pa, pb in zip(ga["params"], gb["params"]):
            scale = float(pa.detach().abs().max()) + 1e-30
            assert float((pa.detach() - pb.detach()).abs().max()) <= rtol * scale, (what, ga["name"], "param")
            sa, sb = ref.state.get(pa, {}), fused.state.get(pb, {})
            assert set(sa) == set(sb), (what, ga["name"], set(sa), set(sb))
            for k in ("exp_avg", "exp_avg_sq"):
                if k in sa:
                    s = float(sa[k].abs().max()) + 1e-30
                    assert float((sa[k] - sb[k]).abs().max()) <= rtol * s, (what, ga["name"], k)
            if "step" in sa:
                assert float(sa["step"]) == float(sb["step"])


def test_fused_adam_follows_torch_adam_step_for_step():
    dev = torch.device("cuda")
    ref, fused = _pair(dev)
    for step in range(30):
        if step == 10:   # the learning-rate schedule of method.py:1206-1210 writes the group's lr
            for o in (ref, fused):
                o.param_groups[0]["lr"] = 3.1e-5
        _set_grads(ref, fused, step, skip=("embeddings",) if step % 7 == 3 else ())   # a parameter without a gradient is left alone
        ref.step()
        fused.step()
        ref.zero_grad(set_to_none=True)
        fused.zero_grad(set_to_none=True)
        if step in (0, 1, 9, 29):
            _assert_same(ref, fused, what=f"step {step}")


def test_fused_adam_survives_the_reference_style_state_surgery_and_state_dict_round_trips():
    """prune (boolean mask), append (cat with zeros) and replace (zeros) of a group's tensor and its exp_avg / exp_avg_sq, exactly as
    method.py:1284-1297, :1312-1328 and :1094-1102 do it; then a state dict crosses over between the two classes."""
    from wg_fused_gaussians import FusedAdam
    dev = torch.device("cuda")
    ref, fused = _pair(dev, seed=3, n=500)
    for step in range(5):
        _set_grads(ref, fused, step)
        ref.step(), fused.step()
    mask = (torch.arange(500, device=dev) % 3) != 0
    per_gaussian = {"xyz", "features_dc", "opacities", "scales", "rotations", "embeddings", "features_rest"}
    for o in (ref, fused):
        for group in o.param_groups:
            if group["name"] not in per_gaussian:
                continue
            old = group["params"][0]
            st = o.state.get(old)
            ext = torch.full_like(old[:17], 0.25)
            st["exp_avg"] = torch.cat((st["exp_avg"][mask], torch.zeros_like(ext)), dim=0)
            st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"][mask], torch.zeros_like(ext)), dim=0)
            del o.state[old]
            group["params"][0] = torch.nn.Parameter(torch.cat((old.detach()[mask], ext), dim=0).requires_grad_(True))
            o.state[group["params"][0]] = st
            if group["name"] == "opacities":   # reset_opacity: fresh moments for a replaced tensor
                st["exp_avg"] = torch.zeros_like(group["params"][0])
                st["exp_avg_sq"] = torch.zeros_like(group["params"][0])
    for step in range(5, 12):
        _set_grads(ref, fused, step)
        ref.step(), fused.step()
    _assert_same(ref, fused, what="after surgery")
    # state dicts cross over: torch's into a FusedAdam and back
    sd_ref, sd_fused = copy.deepcopy(ref.state_dict()), copy.deepcopy(fused.state_dict())
    ref.load_state_dict(sd_fused)
    fused.load_state_dict(sd_ref)
    for step in range(12, 16):
        _set_grads(ref, fused, step)
        ref.step(), fused.step()
    _assert_same(ref, fused, what="after the state dicts crossed over")
    assert isinstance(fused, torch.optim.Adam) and isinstance(fused, FusedAdam)


def test_fused_adam_takes_more_tensors_than_one_launch_holds_and_fails_loudly_where_it_must():
    from wg_fused_gaussians import FusedAdam
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    shapes = [(int(torch.randint(1, 3000, (1,), generator=g)),) for _ in range(61)]   # three launches of <= 24
    pa = [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    ref, fused = torch.optim.Adam(pa, lr=1e-2, weight_decay=0.1), FusedAdam(pb, lr=1e-2, weight_decay=0.1)
    for step in range(3):
        for a, b in zip(pa, pb):
            a.grad = torch.randn(a.shape, generator=g).to(dev)
            b.grad = a.grad.clone()
        ref.step(), fused.step()
    for a, b in zip(pa, pb):
        assert float((a.detach() - b.detach()).abs().max()) <= 2e-6 * (float(a.detach().abs().max()) + 1e-30)
    with pytest.raises(NotImplementedError):
        FusedAdam(pb, amsgrad=True)
    cpu = torch.nn.Parameter(torch.zeros(4))
    o = FusedAdam([cpu])
    cpu.grad = torch.ones(4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        o.step()
    FusedAdam([torch.nn.Parameter(torch.zeros(4, device=dev))]).step()   # nothing has a gradient: a no-op, as in torch


def test_real_training_loop_with_the_fused_optimizer(monkeypatch):
    """The reference's real train_iteration (tests/real_caller) with its optimizer class swapped for FusedAdam -- the one-line
    opt-in of INTEGRATION.md -- through densification, pruning and the opacity reset, i.e. through every piece of the caller's code
    that edits the optimizer's state."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "real_caller"))
    import harness
    if not harness.staged_available():
        pytest.skip("no reference checkout and nothing staged")
    from wg_fused_gaussians import FusedAdam
    monkeypatch.setattr(torch.optim, "Adam", FusedAdam)
    m, wg = harness.make_method(30_000, 480, 320, n_cams=3, overrides={
        "densify_from_iter": 10, "densification_interval": 15, "opacity_reset_interval": 40, "densify_until_iter": 55,
        "densify_grad_threshold": 0.00002})
    assert type(wg.model.optimizer) is FusedAdam
    counts, losses = [], []
    for i in range(60):
        out = wg.train_iteration(i)
        counts.append(out["num_gaussians"])
        losses.append(out["loss"])
    assert np.isfinite(losses).all() and len(set(counts)) > 2
    for p in (wg.model.xyz, wg.model.scales, wg.model.rotations, wg.model.opacities, wg.model.features_dc):
        assert torch.isfinite(p).all() and p.shape[0] == counts[-1]
