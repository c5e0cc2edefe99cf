#!/usr/bin/env python3
"""Where does a slow step of the fwd+bwd loop lose its time: the host (thread descheduled / blocked in a runtime call, the GPU starved) or the
GPU (the work itself took longer)?  Headline scene, N steps in the default flow (the host in step with the GPU).  Per step: host clock, the
thread's CPU time and involuntary context switches (getrusage(RUSAGE_THREAD)), the wall time of the forward call and of the backward call,
and two device events (in front of the forward's first launch, behind the backward's last): `gpu_busy` = end - start of a step on the GPU's
clock, `gpu_gap` = start of step i - end of step i-1 (the GPU idle between steps: ~0 while the host keeps ahead).
usage: python scripts/diag_stall_source.py [steps] [--deferred]   (--deferred: speculative_forward = 2, the host does not wait for the frame)"""
import gc, json, os, resource, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd")); sys.path.insert(0, ROOT)
import torch
import wg_scenes as S
from diff_gaussian_rasterization import GaussianRasterizer, _C
from tests.wg_testlib import make_settings, to_dev

args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if args else 3000
deferred = "--deferred" in sys.argv
if deferred:
    _C.set_option("speculative_forward", 2)
W, H, P = 1920, 1080, 1_000_000
dev = torch.device("cuda", 0)
cloud = S.make_cloud(P, W, H, sh_degree=3, seed=0)
rast = GaussianRasterizer(make_settings(S.make_camera(W, H), 3, device=dev))
t = {k: to_dev(v, dev).requires_grad_(True) for k, v in cloud.items()}
m2d = torch.zeros((P, 3), device=dev, requires_grad=True)
cot = to_dev(S.make_cotangent(W, H), dev)
ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(N)]
ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(N)]
rows = []


def step(i=None):
    for v in t.values():
        v.grad = None
    m2d.grad = None
    if i is not None:
        ev0[i].record()
    a = time.perf_counter()
    img = rast(means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])[0]
    b = time.perf_counter()
    img.backward(cot)
    c = time.perf_counter()
    if i is not None:
        ev1[i].record()
        ru = resource.getrusage(resource.RUSAGE_THREAD)
        rows.append((c, b - a, c - b, ru.ru_utime + ru.ru_stime, ru.ru_nivcsw, ru.ru_nvcsw))


def box_state():
    """what the box says about the usual suspects for periodic process-wide GPU queue evictions (automatic NUMA balancing, huge-page
    compaction: MMU-notifier invalidations) -- readable without privileges"""
    out = {}
    for name, path in (("numa_balancing", "/proc/sys/kernel/numa_balancing"), ("thp", "/sys/kernel/mm/transparent_hugepage/enabled"),
                       ("thp_defrag", "/sys/kernel/mm/transparent_hugepage/defrag"), ("loadavg", "/proc/loadavg")):
        try:
            out[name] = open(path).read().strip()
        except OSError as e:
            out[name] = f"unreadable ({e.errno})"
    try:
        keys = ("numa_pte_updates", "numa_hint_faults", "pgmigrate_success", "compact_stall", "thp_fault_alloc", "thp_collapse_alloc")
        out["vmstat"] = {k: int(v) for k, v in (l.split() for l in open("/proc/vmstat")) if k in keys}
    except OSError:
        out["vmstat"] = None
    return out


gc.collect(); gc.disable()
box0 = box_state()
for _ in range(100):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
ru = resource.getrusage(resource.RUSAGE_THREAD)
prev = (t0, 0, 0, ru.ru_utime + ru.ru_stime, ru.ru_nivcsw, ru.ru_nvcsw)
for i in range(N):
    step(i)
torch.cuda.synchronize()
busy = [ev0[i].elapsed_time(ev1[i]) for i in range(N)]
gap = [0.0] + [ev1[i - 1].elapsed_time(ev0[i]) for i in range(1, N)]
out, wall = [], []
for i, r in enumerate(rows):
    w = 1e3 * (r[0] - prev[0])
    wall.append(w)
    out.append(dict(step=i, wall_ms=round(w, 3), fwd_call_ms=round(1e3 * r[1], 3), bwd_call_ms=round(1e3 * r[2], 3), cpu_ms=round(1e3 * (r[3] - prev[3]), 3),
                    invol_switches=r[4] - prev[4], vol_switches=r[5] - prev[5], gpu_busy_ms=round(busy[i], 3), gpu_gap_ms=round(gap[i], 3)))
    prev = r
med = sorted(wall)[N // 2]
mb = sorted(busy)[N // 2]
slow = [o for o in out if o["wall_ms"] > 1.25 * med or o["gpu_busy_ms"] > 1.25 * mb or o["gpu_gap_ms"] > 0.25]
box1 = box_state()
if box0.get("vmstat") and box1.get("vmstat"):
    box1["vmstat_delta_over_the_run"] = {k: box1["vmstat"][k] - box0["vmstat"][k] for k in box1["vmstat"]}
print(json.dumps({"box": box1, "flow": "deferred (speculative_forward = 2)" if deferred else "default (host waits for the frame's verdict)", "steps": N,
                  "median_wall_ms": round(med, 4), "mean_wall_ms": round(sum(wall) / N, 4), "median_gpu_busy_ms": round(mb, 4),
                  "total_invol_switches": sum(o["invol_switches"] for o in out), "total_vol_switches": sum(o["vol_switches"] for o in out),
                  "n_slow": len(slow), "slow": slow[:80]}))
