"""View-parallel harness: one process per GPU, one camera per rank, Gaussians replicated, and the ONLY collective is
an all-reduce of the scalar loss (BASELINE.json north_star).  The rasterizer itself has no cross-view
state, so there is no data-path collective.  Backend: ``nccl`` (= RCCL over xGMI on ROCm) on GPUs, ``gloo`` in the CPU
tests.  The reference has no distributed code at all; this is new host logic.
"""
from __future__ import annotations

import os
from typing import List

import torch
import torch.distributed as dist

import wg_scenes as S


def env_world():
    """(rank, local_rank, world_size) from the torchrun / torch.distributed.run environment (1-process default)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def device_for_rank(local_rank: int, device_count: int, backend: str | None) -> int:
    """One process per GPU means one GPU per process: a local rank without a device of its own is an ERROR (RCCL refuses two ranks on
    one device, and silently wrapping around would report a 'scaling' number measured on shared hardware) -- unless the job was
    EXPLICITLY put on the host-side transport (WG_DIST_BACKEND=gloo / backend="gloo": the 1-GPU test box exercising the N > 1 flow),
    where ranks share devices round-robin."""
    if device_count <= 0:
        raise RuntimeError("no HIP device visible")
    if local_rank < device_count:
        return local_rank
    if backend == "gloo":
        return local_rank % device_count
    raise RuntimeError(f"local rank {local_rank} has no GPU of its own: {device_count} device(s) visible.  Launch at most one rank per "
                       f"visible GPU, or set WG_DIST_BACKEND=gloo to let ranks share devices over the host-side transport (tests only)")


def cores_for_rank(local_rank: int, local_world: int, cores: List[int]) -> List[int]:
    """The host cores local rank r of `local_world` may run on: an equal, contiguous slice of the allowed set (every rank has a thread
    that polls the rasterizer's mailbox once per forward pass; eight of them must not share cores).  Fewer cores than ranks: all."""
    cores = sorted(cores)
    per = len(cores) // max(local_world, 1)
    if per < 1:
        return cores
    return cores[local_rank * per:(local_rank + 1) * per]


def pin_rank(local_rank: int, local_world: int) -> List[int]:
    """Apply cores_for_rank to this process (WG_NO_AFFINITY=1 leaves the affinity alone) and size the OpenMP / torch intra-op pools
    to it.  Returns the core list in force."""
    if not hasattr(os, "sched_getaffinity"):
        return []
    allowed = sorted(os.sched_getaffinity(0))
    if local_world <= 1 or os.environ.get("WG_NO_AFFINITY") == "1":
        return allowed
    mine = cores_for_rank(local_rank, local_world, allowed)
    try:
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(len(mine), int(os.environ.get("OMP_NUM_THREADS", len(mine))))))
    except OSError:
        return allowed
    return mine


def init(backend: str | None = None) -> tuple[int, int, int]:
    rank, local_rank, world = env_world()
    if backend is None:  # WG_DIST_BACKEND=gloo: exercise the N > 1 flow of bench.py where RCCL cannot run (ranks sharing one GPU)
        backend = os.environ.get("WG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        local_rank = device_for_rank(local_rank, torch.cuda.device_count(), backend if world > 1 else "gloo")
        torch.cuda.set_device(local_rank)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        pin_rank(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def _host_side() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo"


def views_for_rank(num_views: int, rank: int, world: int) -> List[int]:
    """rank r renders views {r, r+G, r+2G, ...}."""
    return list(range(rank, num_views, world))


def view_cameras(num_views: int, width: int, height: int, yaw_step_deg: float = 5.0):
    """BASELINE.json configs[3]: the base camera yawed by k * 5 degrees."""
    return [S.make_camera(width, height, yaw_deg=yaw_step_deg * k) for k in range(num_views)]


def allreduce_loss(loss: torch.Tensor) -> torch.Tensor:
    """SUM all-reduce of a 1-element fp32 tensor: 4 bytes on the wire per step, latency-bound."""
    if dist.is_available() and dist.is_initialized():
        if _host_side() and loss.is_cuda:  # gloo (tests): reduce a host copy
            host = loss.detach().cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            loss.copy_(host)
        else:
            dist.all_reduce(loss, op=dist.ReduceOp.SUM)
    return loss


class LossStream:
    """The step's scalar loss and its all-reduce, without stalling the rasterizer's stream on the collective.

    ``submit(image, cotangent_flat)`` computes loss = <image, cotangent> on the caller's stream and queues the 4-byte SUM all-reduce
    ASYNCHRONOUSLY (RCCL runs it on its own stream behind the dot product): the loss value gates nothing in the view-parallel step --
    the image's cotangent is known before it, as a training loop's logged loss does not gate its next step -- so the backward pass
    is not made to wait for a latency-bound collective.  Nothing is skipped: a device-wide synchronize (the end of bench.py's timed
    region) covers RCCL's stream, and ``last()`` returns the newest all-reduced value.  With the host-side gloo transport of the
    tests it is the synchronous ``allreduce_loss``.  (Also moving the dot product to a side stream, beside the backward pass, was
    measured at N = 1: 1.038 -> 1.055 ms per step -- the cross-stream event traffic costs more than the 20 us dot it hides.)"""

    def __init__(self, device=None):
        self.keep = []   # the last few (loss, work) pairs: keeps the tensors alive until their collective has run

    def submit(self, image: torch.Tensor, cotangent_flat: torch.Tensor) -> torch.Tensor:
        loss = torch.dot(image.detach().reshape(-1), cotangent_flat).reshape(1)
        work = None
        if dist.is_available() and dist.is_initialized():
            if _host_side():
                allreduce_loss(loss)
            else:
                work = dist.all_reduce(loss, op=dist.ReduceOp.SUM, async_op=True)
        self.keep.append((loss, work))
        if len(self.keep) > 4:
            self.keep.pop(0)
        return loss

    def last(self) -> float:
        """The newest all-reduced loss (waits for it)."""
        if not self.keep:
            return float("nan")
        loss, work = self.keep[-1]
        if work is not None:
            work.wait()
        return float(loss.item())


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if _host_side() else device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device) -> float:
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if _host_side() else device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(values, device):
    """All-gather of a short float vector: one row per rank (bench.py's "who took part" record)."""
    t = torch.tensor(values, dtype=torch.float64, device="cpu" if _host_side() else device)
    if dist.is_available() and dist.is_initialized():
        out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(out, t)
        return [o.cpu().tolist() for o in out]
    return [t.cpu().tolist()]


def backend_name() -> str:
    """"nccl" is RCCL on ROCm; "none" for a single process."""
    return dist.get_backend() if dist.is_available() and dist.is_initialized() else "none"


def timed_region(fn, steps: int, device, keep=None, stamps=None, warmup: int = 0) -> float:
    """bench.py's timing contract: barrier + device synchronize on both sides of exactly `steps` calls of fn, MAX over ranks.
    keep (a one-element list) receives this rank's own time, taken before it waits for the others.  stamps (a list) receives the host
    clock after every call (steps + 1 values with the start; ~0.1 us each): the host is in step with the GPU here, so their differences
    locate a slow step inside the region without putting a device event into it.  warmup: that many untimed calls of fn run HERE, after
    the collector pass and directly in front of the region's opening barrier + synchronize, so that the GPU is not left idle between its
    warm-up and its timed steps (a cycle collection over a large heap takes tens of milliseconds; an idle MI355X drops its clocks and needs
    ~20 ms of load to regain them: the opening steps of a region measured right after one ran 10 - 15 % slow)."""
    import gc
    import time
    sync = (lambda: torch.cuda.synchronize(device)) if torch.cuda.is_available() and device is not None and torch.device(device).type == "cuda" else (lambda: None)
    # The host is in step with the GPU here (a forward call returns the frame's exact instance count, i.e. waits for its scan), so a
    # pause of the interpreter's cycle collector is a pause of the GPU, and with N ranks the slowest rank's pauses are everybody's:
    # collect before the region, keep the collector off inside it (no work of a step is skipped; reference counting still frees).
    # A caller that has switched the collector off already (bench.py: once, in front of all its passes) has collected too: a cycle
    # collection takes tens of milliseconds in a process holding a scene, the GPU idles meanwhile, an idle MI355X drops its clocks and
    # needs ~20-30 ms of load to regain them (scripts/diag_idle_ramp.py) -- the opening steps of the region would measure that ramp.
    gc_was_on = gc.isenabled()
    if gc_was_on:
        gc.collect()
        gc.disable()
    try:
        for _ in range(warmup):
            fn()
        barrier()
        sync()
        t0 = time.perf_counter()
        if stamps is None:
            for _ in range(steps):
                fn()
        else:
            stamps.append(t0)
            for _ in range(steps):
                fn()
                stamps.append(time.perf_counter())
        sync()
        if keep is not None:
            keep[0] = time.perf_counter() - t0
        barrier()
        return max_over_ranks(time.perf_counter() - t0, device)
    finally:
        if gc_was_on:
            gc.enable()


def job_fields(world: int, steps: int, t_region: float, ranks_seen, baseline_iters_per_s: float | None = None, views_total: int | None = None) -> dict:
    """The whole-job part of bench.py's JSON line from the timed region and the all-gathered (rank, device, own ms/step[, instances of the
    rank's views, number of views]) rows.  `value` = iterations (one view forward + backward each) of ALL ranks per second of the slowest
    rank's region: `views_total` views per step over the job (default one per rank: weak scaling; a fixed batch, e.g. BASELINE config 4's
    eight views, is strong scaling).  With the 1-GPU rate supplied the line also states efficiency = T(N) / (N T(1)) -- the driver computes
    its own from the per-N lines."""
    rows = sorted(ranks_seen, key=lambda r: r[0])
    views = world if views_total is None else int(views_total)
    out = {"value": round(views * steps / t_region, 3), "n_gpus": world, "ms_per_step": round(1000.0 * t_region / steps, 4),
           "rccl_ranks_seen": [int(r[0]) for r in rows], "per_rank_ms_per_step": {str(int(r[0])): round(r[2], 4) for r in rows},
           "per_rank_device": {str(int(r[0])): int(r[1]) for r in rows},
           "per_rank_num_rendered": {str(int(r[0])): {"instances": int(r[3]), "views": int(r[4])} for r in rows if len(r) >= 5}}
    if len(rows) != world or out["rccl_ranks_seen"] != list(range(world)):
        raise RuntimeError(f"expected ranks 0..{world - 1}, saw {out['rccl_ranks_seen']}")
    if baseline_iters_per_s:
        out["scaling_efficiency"] = {"baseline_iters_per_s_1gpu": baseline_iters_per_s,
                                     "efficiency": round(out["value"] / (world * baseline_iters_per_s), 4),
                                     "definition": "T(N) / (N * T(1)), T = whole-job iterations per second"}
    return out
