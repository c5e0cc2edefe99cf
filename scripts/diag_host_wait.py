#!/usr/bin/env python3
"""Where the HOST is while a forward call runs: classic flow (wait for the instance count in the middle of the call), speculative flow
(the count is looked at behind the call's last launch), deferred speculation (looked at by the thread's next call) and fixed-capacity
flow (never looked at).

For each flow: N forward calls, each issued on an idle GPU (device synchronised before it); host-side duration of the call, GPU-side
duration of its kernels (HIP events around the call), whether the call's poll found the count already there.  With the GPU idle at
the start of a call the host is never "ahead" of anything but this call's own kernels: what it waits for is exactly the rendezvous.
usage: python scripts/diag_host_wait.py [gaussians width height calls]    -> one JSON line"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import wg_scenes as S  # noqa: E402
from diff_gaussian_rasterization import GaussianRasterizer, _C  # noqa: E402
from tests.wg_testlib import make_settings, to_dev  # noqa: E402

P, W, H, N = (int(a) for a in (sys.argv[1:5] + ["1000000", "1920", "1080", "50"][len(sys.argv) - 1:]))
dev = torch.device("cuda", 0)
cloud = S.make_cloud(P, W, H, sh_degree=3, seed=0)
rs = make_settings(S.make_camera(W, H), 3, device=dev)
rast = GaussianRasterizer(rs)
t = {k: to_dev(v, dev) for k, v in cloud.items()}
m2 = torch.zeros_like(t["means3D"])


def call(capacity=None):
    with torch.no_grad():
        return rast(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"],
                    binning_capacity=capacity)


def measure(label, capacity=None):
    for _ in range(5):
        call(capacity)
    torch.cuda.synchronize()
    polls0, waited0 = _C.get_option("forward_polls"), _C.get_option("forward_polls_waited")
    host, gpu = [], []
    for _ in range(N):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        t0 = time.perf_counter()
        call(capacity)
        t1 = time.perf_counter()
        b.record()
        torch.cuda.synchronize()
        host.append(1e3 * (t1 - t0))
        gpu.append(a.elapsed_time(b))
    q = lambda v, f: round(float(np.quantile(v, f)), 4)  # noqa: E731
    return {"flow": label, "host_call_ms_p50": q(host, 0.5), "host_call_ms_p90": q(host, 0.9), "gpu_ms_p50": q(gpu, 0.5),
            "host_returned_before_gpu_finished_ms_p50": q(np.array(gpu) - np.array(host), 0.5),
            "polls": _C.get_option("forward_polls") - polls0, "polls_that_waited": _C.get_option("forward_polls_waited") - waited0}


out = {"workload": f"{P} Gaussians, {W}x{H}, SH 3, forward only, every call issued on an idle GPU", "calls": N, "flows": []}
_C.set_option("speculative_forward", 0)
out["flows"].append(measure("classic (rendezvous in the middle of the call, rasterizer_impl.cu:284)"))
_C.set_option("speculative_forward", 1)
out["flows"].append(measure("speculative (rendezvous behind the call's last launch)"))
_C.set_option("speculative_forward", 2)
_C.set_option("spec_margin_pct", 50)
out["flows"].append(measure("deferred speculation (the verdict is read by the thread's next call)"))
_C.set_option("spec_margin_pct", 25)
_C.set_option("speculative_forward", 1)
call(8 * 1024 * 1024)
n, _ = _C.last_forward_status()
out["flows"].append(measure("fixed capacity (no rendezvous)", capacity=int(1.25 * n) + 4096))
# the compiled torch binding (csrc/torch_binding.cpp, WG_BINDING=torch) against the ctypes one, same flows: what the marshalling costs
try:
    _C.use_binding("torch")
    _C.set_option("speculative_forward", 1)
    out["flows"].append(measure("speculative, compiled torch binding"))
    _C.set_option("speculative_forward", 2)
    _C.set_option("spec_margin_pct", 50)
    out["flows"].append(measure("deferred speculation, compiled torch binding"))
    _C.set_option("spec_margin_pct", 25)
    _C.set_option("speculative_forward", 1)
except ImportError as ex:
    out["compiled_binding"] = f"not built: {ex}"
finally:
    _C.use_binding("ctypes")
print(json.dumps(out))
