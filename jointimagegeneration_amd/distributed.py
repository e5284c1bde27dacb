"""One process per GPU, volumes sharded over ranks, NO collective on the data path.

The reference samples single-process (ccdm/ddpm_eval.py:52 runs `run_eval(0, ...)`; latentdiffusion/sample_diffusion.py
never creates a process group).  Units of work are whole volumes (a CCDM chain + its own autoregressive slice loop):
they never exchange data, so the only communication is the launcher's barrier and a MAX-reduce of the elapsed time.
"""
from __future__ import annotations

import os
import time
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: Optional[str] = None, device: Optional[torch.device] = None) -> Tuple[int, int]:
    """backend 'nccl' (= RCCL on ROCm) for GPU ranks, 'gloo' for CPU rehearsals; MASTER_ADDR/PORT come from the launcher."""
    rank, _, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        backend = os.environ.get("GG_DIST_BACKEND") or backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return rank, world


def shard(n_units: int, rank: int, world: int) -> List[int]:
    """volume_id -> rank = volume_id mod world_size."""
    return [i for i in range(n_units) if i % world == rank]


def barrier(device: Optional[torch.device] = None) -> None:
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def timed_region(fn: Callable[[], None], device: Optional[torch.device] = None) -> float:
    """barrier + sync on both sides of `fn`; returns the MAX elapsed seconds over ranks."""
    barrier(device)
    t0 = time.time()
    fn()
    barrier(device)
    elapsed = time.time() - t0
    if dist.is_available() and dist.is_initialized():
        on_gpu = device is not None and device.type == "cuda" and dist.get_backend() == "nccl"
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def finalize() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
