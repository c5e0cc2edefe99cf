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


def init(backend: str | None = None) -> tuple[int, int, int]:
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:  # WG_DIST_BACKEND=gloo: exercise the N > 1 flow of bench.py where RCCL cannot run (ranks sharing one GPU)
            backend = os.environ.get("WG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            local_rank %= torch.cuda.device_count()
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        local_rank %= torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
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
