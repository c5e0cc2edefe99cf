#!/usr/bin/env python3
"""Diagnostics of slow steps in a long fwd+bwd loop (round 4's investigation of 2-5 ms stalls, which turned out to be the shared box: EXPERIMENTS R4.4),
folded into ONE script with sub-commands (round 5; they were six files):
  source   Where does a slow step of the fwd+bwd loop lose its time: the host (thread descheduled / blocked in a runtime call, the GPU starved) or the
  threads  Do the sporadic multi-millisecond steps of the fwd+bwd loop need autograd's worker thread?  Alternating blocks of the SAME step made two
  trace    Post-processor of a rocprofv3 --kernel-trace of a long fwd+bwd loop (scripts/diag_step_blips.py): is a slow step a GAP on the GPU's
  blips    Which steps of a long fwd+bwd loop are slow, and how regularly?  Headline scene, N steps, host clock after every step (the host is in
  jitter   Wall-clock duration of every train step of bench.py's workload (host side; every forward waits for the mailbox, so a step's
  host     Host-side duration of each forward call vs the GPU time (diagnostic for allocation / driver stalls).

usage: python scripts/diag_stall.py <source|threads|trace|blips|jitter|host> [that command's arguments]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd"))
sys.path.insert(0, ROOT)


def cmd_source(argv):
    """Where does a slow step of the fwd+bwd loop lose its time: the host (thread descheduled / blocked in a runtime call, the GPU starved) or the
GPU (the work itself took longer)?  Headline scene, N steps in the default flow (the host in step with the GPU).  Per step: host clock, the
thread's CPU time and involuntary context switches (getrusage(RUSAGE_THREAD)), the wall time of the forward call and of the backward call,
and two device events (in front of the forward's first launch, behind the backward's last): `gpu_busy` = end - start of a step on the GPU's
clock, `gpu_gap` = start of step i - end of step i-1 (the GPU idle between steps: ~0 while the host keeps ahead).
usage: python scripts/diag_stall_source.py [steps] [--deferred]   (--deferred: speculative_forward = 2, the host does not wait for the frame)"""
    import gc, json, os, resource, sys, time
    import torch
    import wg_scenes as S
    from diff_gaussian_rasterization import GaussianRasterizer, _C
    from tests.wg_testlib import make_settings, to_dev

    args = [a for a in argv[1:] if not a.startswith("--")]
    N = int(args[0]) if args else 3000
    deferred = "--deferred" in argv
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


def cmd_threads(argv):
    """Do the sporadic multi-millisecond steps of the fwd+bwd loop need autograd's worker thread?  Alternating blocks of the SAME step made two
ways on the same box: (A) as a training loop makes it -- GaussianRasterizer(...) + image.backward(cotangent): the backward launches come
from autograd's device thread, the main thread waits for it -- and (B) the binding's two entry points called directly from the main thread
(_C.rasterize_gaussians / _C.rasterize_gaussians_backward: the reference's own tests call its extension this way), no second thread.
Per mode: steps, median, mean, and every step above 1.5 x the median.   usage: diag_stall_threads.py [blocks=10] [steps_per_block=2000]"""
    import gc, json, os, sys, time
    import torch
    import wg_scenes as S
    from diff_gaussian_rasterization import GaussianRasterizer, _C
    from tests.wg_testlib import make_settings, to_dev

    blocks = int(argv[1]) if len(argv) > 1 else 10
    per = int(argv[2]) if len(argv) > 2 else 2000
    W, H, P = 1920, 1080, 1_000_000
    dev = torch.device("cuda", 0)
    cloud = S.make_cloud(P, W, H, sh_degree=3, seed=0)
    rs = make_settings(S.make_camera(W, H), 3, device=dev)
    rast = GaussianRasterizer(rs)
    t = {k: to_dev(v, dev).requires_grad_(True) for k, v in cloud.items()}
    m2d = torch.zeros((P, 3), device=dev, requires_grad=True)
    cot = to_dev(S.make_cotangent(W, H), dev)
    E = torch.Tensor([])
    d = {k: v.detach() for k, v in t.items()}


    def step_autograd():
        for v in t.values():
            v.grad = None
        m2d.grad = None
        rast(means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])[0].backward(cot)


    def step_direct():
        R, color, radii, gb, bb, ib = _C.rasterize_gaussians(rs.bg, d["means3D"], E, d["opacities"], d["scales"], d["rotations"], rs.scale_modifier, E, rs.viewmatrix,
                                                             rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.kernel_size, rs.subpixel_offset, rs.image_height,
                                                             rs.image_width, d["shs"], rs.sh_degree, rs.campos, rs.prefiltered, False)
        return _C.rasterize_gaussians_backward(rs.bg, d["means3D"], radii, E, d["scales"], d["rotations"], rs.scale_modifier, E, rs.viewmatrix, rs.projmatrix,
                                               rs.tanfovx, rs.tanfovy, rs.kernel_size, rs.subpixel_offset, cot, d["shs"], rs.sh_degree, rs.campos, gb, R, bb, ib, False)


    gc.collect(); gc.disable()
    for fn in (step_autograd, step_direct):
        for _ in range(100):
            fn()
    torch.cuda.synchronize()
    rec = {"autograd": [], "direct": []}
    for b in range(blocks):
        for name, fn in (("autograd", step_autograd), ("direct", step_direct)):
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            prev = time.perf_counter()
            for _ in range(per):
                fn()
                now = time.perf_counter()
                rec[name].append(now - prev)
                prev = now
            torch.cuda.synchronize()
    out = {"box_loadavg": open("/proc/loadavg").read().strip()}
    for name, v in rec.items():
        ms = [1e3 * x for x in v]
        med = sorted(ms)[len(ms) // 2]
        slow = [(i, round(x, 3)) for i, x in enumerate(ms) if x > 1.5 * med]
        out[name] = {"steps": len(ms), "median_ms": round(med, 4), "mean_ms": round(sum(ms) / len(ms), 4), "n_over_1.5x_median": len(slow),
                     "excess_ms_total": round(sum(x - med for _, x in slow), 2), "slow": slow[:40]}
    print(json.dumps(out))


def cmd_trace(argv):
    """Post-processor of a rocprofv3 --kernel-trace of a long fwd+bwd loop (scripts/diag_step_blips.py): is a slow step a GAP on the GPU's
timeline (the GPU starved: nothing to run for milliseconds -- the host side was late) or a STRETCHED kernel (the GPU paused or slowed with
work in hand)?  Lists every idle gap between consecutive kernels above `gap_ms` with the kernels either side, and every kernel that ran
longer than `stretch` x its own median.    usage: diag_stall_trace.py results.db [gap_ms=0.5] [stretch=2.5]"""
    import json, sqlite3, sys

    db = sqlite3.connect(argv[1])
    gap_ms = float(argv[2]) if len(argv) > 2 else 0.5
    stretch = float(argv[3]) if len(argv) > 3 else 2.5
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    s, e = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp") if "start_timestamp" in cols else (None, None)
    if s is None:
        print(json.dumps({"error": "no start / end columns", "columns": cols}))
        sys.exit(1)
    rows = list(db.execute(f"select name, {s}, {e} from kernels order by {s}"))
    short = lambda n: n.split("(")[0].replace("void ", "")[:60]
    by = {}
    for n, a, b in rows:
        by.setdefault(n, []).append(b - a)
    med = {n: sorted(v)[len(v) // 2] for n, v in by.items()}
    gaps, stretched = [], []
    end_so_far = rows[0][2]
    for i, (n, a, b) in enumerate(rows):
        if i and a - end_so_far > gap_ms * 1e6:
            gaps.append(dict(at_kernel=i, idle_ms=round((a - end_so_far) / 1e6, 3), before=short(rows[i - 1][0]), after=short(n)))
        if b - a > stretch * med[n] and b - a > 0.2e6:
            stretched.append(dict(at_kernel=i, kernel=short(n), ms=round((b - a) / 1e6, 3), median_ms=round(med[n] / 1e6, 3)))
        end_so_far = max(end_so_far, b)
    span = (rows[-1][2] - rows[0][1]) / 1e6
    busy = sum(b - a for _, a, b in rows) / 1e6
    print(json.dumps({"kernels": len(rows), "span_ms": round(span, 1), "sum_of_kernel_ms": round(busy, 1), "idle_gaps_over_%.2f_ms" % gap_ms: gaps[:60],
                      "n_gaps": len(gaps), "kernels_over_%.1fx_their_median" % stretch: stretched[:60], "n_stretched": len(stretched)}))


def cmd_blips(argv):
    """Which steps of a long fwd+bwd loop are slow, and how regularly?  Headline scene, N steps, host clock after every step (the host is in
step with the GPU); prints the indices and durations of the steps above 1.15 x the median and the gaps between them.
Run it twice: as it is, and with HSA_KERNARG_POOL_SIZE=<bytes> in the environment (ROCclr's kernel-argument ring: when it wraps the
runtime waits for every launch in flight -- a queue drain once per ring).  usage: python scripts/diag_step_blips.py [steps]"""
    import json, os, sys, time
    import torch
    import wg_scenes as S
    from diff_gaussian_rasterization import GaussianRasterizer
    from tests.wg_testlib import make_settings, to_dev
    import gc

    N = int(argv[1]) if len(argv) > 1 else 3000
    W, H, P = 1920, 1080, 1_000_000
    dev = torch.device("cuda", 0)
    cloud = S.make_cloud(P, W, H, sh_degree=3, seed=0)
    rast = GaussianRasterizer(make_settings(S.make_camera(W, H), 3, device=dev))
    t = {k: to_dev(v, dev).requires_grad_(True) for k, v in cloud.items()}
    m2d = torch.zeros((P, 3), device=dev, requires_grad=True)
    cot = to_dev(S.make_cotangent(W, H), dev)


    def step():
        for v in t.values():
            v.grad = None
        m2d.grad = None
        rast(means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])[0].backward(cot)


    gc.collect(); gc.disable()
    for _ in range(100):
        step()
    torch.cuda.synchronize()
    st = [time.perf_counter()]
    for _ in range(N):
        step()
        st.append(time.perf_counter())
    torch.cuda.synchronize()
    d = [1e3 * (b - a) for a, b in zip(st[:-1], st[1:])]
    med = sorted(d)[len(d) // 2]
    slow = [(i, round(x, 3)) for i, x in enumerate(d) if x > 1.15 * med]
    idx = [i for i, _ in slow]
    print(json.dumps({"HSA_KERNARG_POOL_SIZE": os.environ.get("HSA_KERNARG_POOL_SIZE"), "steps": N, "median_ms": round(med, 4), "mean_ms": round(sum(d) / len(d), 4),
                      "slow_steps": slow[:60], "gaps_between_slow_steps": [b - a for a, b in zip(idx[:-1], idx[1:])][:60]}))


def cmd_jitter(argv):
    """Wall-clock duration of every train step of bench.py's workload (host side; every forward waits for the mailbox, so a step's
host time tracks its GPU time): where do the occasional slow timed passes come from?  usage: diag_step_jitter.py [steps] [gc]"""
    import gc, os, sys, time
    import numpy as np, torch, wg_scenes as S
    from diff_gaussian_rasterization import GaussianRasterizer
    from tests.wg_testlib import make_settings, to_dev
    steps = int(argv[1]) if len(argv) > 1 else 600
    if len(argv) > 2 and argv[2] == "nogc":
        gc.disable()
    dev = torch.device("cuda", 0)
    P, W, H = 1_000_000, 1920, 1080
    cloud = S.make_cloud(P, W, H, sh_degree=3, seed=0); cam = S.make_camera(W, H)
    rast = GaussianRasterizer(make_settings(cam, 3, device=dev))
    t = {k: to_dev(v, dev).requires_grad_(True) for k, v in cloud.items()}
    m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
    cot = to_dev(S.make_cotangent(W, H), dev); cf = cot.reshape(-1)
    def step():
        for v in t.values(): v.grad = None
        m2.grad = None
        c, r, a = rast(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
        c.backward(cot)
        return torch.dot(c.detach().reshape(-1), cf).reshape(1)
    for _ in range(20): step()
    torch.cuda.synchronize()
    print("loadavg", os.getloadavg(), "cpus", len(os.sched_getaffinity(0)), "gc", gc.isenabled(), gc.get_threshold())
    gcs = []
    gc.callbacks.append(lambda ph, info: gcs.append((time.perf_counter(), ph, info.get("generation"))))
    for rep in range(3):
        ts = np.empty(steps + 1); a0 = torch.cuda.memory_stats()["num_device_alloc"]
        torch.cuda.synchronize(); ts[0] = time.perf_counter()
        for i in range(steps):
            step(); ts[i + 1] = time.perf_counter()
        torch.cuda.synchronize(); tend = time.perf_counter()
        d = np.diff(ts) * 1e3
        med = np.median(d)
        spikes = [(int(i), round(float(x), 2)) for i, x in enumerate(d) if x > 2 * med]
        print(f"pass {rep}: total {1e3*(tend-ts[0])/steps:.4f} ms/step, median {med:.4f}, p90 {np.quantile(d,0.9):.4f}, p99 {np.quantile(d,0.99):.4f}, max {d.max():.2f}, "
              f"sum of excess over median {float((d-med).clip(0).sum()):.1f} ms, device_allocs +{torch.cuda.memory_stats()['num_device_alloc']-a0}")
        print("   spikes (>2x median):", spikes[:30], "gc events in pass:", len([g for g in gcs if g[0] >= ts[0] and g[1] == 'start']), [g[2] for g in gcs if g[0] >= ts[0] and g[1]=='start'][:20])


def cmd_host(argv):
    """Host-side duration of each forward call vs the GPU time (diagnostic for allocation / driver stalls)."""
    import os, sys, time
    import torch, wg_scenes as S
    from diff_gaussian_rasterization import GaussianRasterizer
    from tests.wg_testlib import make_settings, to_dev
    dev = torch.device("cuda", 0)
    P, W, H = int(argv[1]), int(argv[2]), int(argv[3])
    cloud = S.make_cloud(P, W, H, sh_degree=3, seed=0); cam = S.make_camera(W, H)
    rs = make_settings(cam, 3, device=dev); rast = GaussianRasterizer(rs)
    t = {k: to_dev(v, dev) for k, v in cloud.items()}
    m2 = torch.zeros_like(t["means3D"])
    def f():
        with torch.no_grad():
            return rast(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    for _ in range(3): f()
    torch.cuda.synchronize()
    for i in range(12):
        st = torch.cuda.memory_stats()
        a0, f0 = st["num_device_alloc"], st["num_device_free"]
        t0 = time.perf_counter(); f(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        st = torch.cuda.memory_stats()
        print(f"iter {i}: call {1e3*(t1-t0):.2f} ms, sync {1e3*(t2-t1):.2f} ms, device_alloc +{st['num_device_alloc']-a0} free +{st['num_device_free']-f0}, reserved {st['reserved_bytes.all.current']/2**30:.2f} GiB")


if __name__ == "__main__":
    cmds = {k[4:]: v for k, v in globals().items() if k.startswith("cmd_")}
    if len(sys.argv) < 2 or sys.argv[1] not in cmds:
        raise SystemExit(__doc__)
    cmds[sys.argv[1]]([sys.argv[0] + " " + sys.argv[1]] + sys.argv[2:])
