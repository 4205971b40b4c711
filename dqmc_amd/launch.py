"""One-process-per-GPU plumbing shared by bench.py and the replica-exchange
driver: torch.distributed is used for rendezvous, barriers, the max-over-ranks
timing and the point-to-point replica swaps (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" on CPU for tests).  Chains never communicate during sweeps
(source/main.cpp:128-171), so there is no data-path collective here."""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class Dist:
    rank: int = 0
    world: int = 1
    local_rank: int = 0
    backend: Optional[str] = None
    device: Optional[torch.device] = None

    @property
    def active(self) -> bool:
        return self.world > 1


def dist_init(backend: Optional[str] = None) -> Dist:
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run) and
    initialises the process group when WORLD_SIZE > 1."""
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    if backend is None:
        backend = "nccl" if use_cuda else "gloo"
    dev = torch.device("cuda", local_rank) if (use_cuda and backend == "nccl") else torch.device("cpu")
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
    d = Dist(rank, world, local_rank, backend if world > 1 else None, dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": dev} if dev.type == "cuda" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return d


def barrier(d: Dist):
    """barrier + device synchronize on both sides (bench.py timing contract)."""
    if d.device is not None and d.device.type == "cuda":
        torch.cuda.synchronize()
    if d.active:
        import torch.distributed as dist
        dist.barrier()
    if d.device is not None and d.device.type == "cuda":
        torch.cuda.synchronize()


def max_over_ranks(d: Dist, x: float) -> float:
    if not d.active:
        return float(x)
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=d.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(d: Dist, x: float) -> float:
    """The two MPI_Reduce(SUM) of source/main.cpp:186-187 (time, acceptance)."""
    if not d.active:
        return float(x)
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=d.device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def finalize(d: Dist):
    if d.active:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
