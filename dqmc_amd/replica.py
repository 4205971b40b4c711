"""update::replica_exchange (source/update.cpp:47-117) between neighbouring
ranks, one Markov chain (one inverse temperature) per rank / GPU.

The reference's four MPI messages per round become
torch.distributed point-to-point operations on the default process group:
backend "nccl" is RCCL on ROCm, so on the 8 x MI355X node each pair's swap is
one grouped send/recv over a single xGMI link (4 disjoint pairs per round on
8 GPUs; no ring, no all-reduce).  With backend "gloo" the same code runs on
CPU tensors (tests).  The payload is the HS field array exactly as the
reference ships it: nt x nv int64 (source/update.cpp:60-69).
"""
from __future__ import annotations

import math
from typing import Callable, Tuple

import numpy as np
import torch
import torch.distributed as dist


def partner_rank(rank: int, world_size: int, exchange_attempt: int) -> int:
    """source/update.cpp:34-45 (even attempt: even rank <-> rank+1; odd: even rank <-> rank-1; wraps)."""
    even = exchange_attempt % 2 == 0
    off = (1 if rank % 2 == 0 else -1) if even else (-1 if rank % 2 == 0 else 1)
    return (rank + off + world_size) % world_size


def _sendrecv(send: torch.Tensor, partner: int) -> torch.Tensor:
    """MPI_Sendrecv: one grouped isend/irecv pair (ncclGroupStart/End under RCCL)."""
    recv = torch.empty_like(send)
    ops = [dist.P2POp(dist.isend, send, partner), dist.P2POp(dist.irecv, recv, partner)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    return recv


def replica_exchange(rank: int, world_size: int, bernoulli: Callable[[float], bool], exchange_attempt: int,
                     exchange_accepted: int, engine, device: torch.device) -> Tuple[int, int, bool]:
    """One exchange round.  `engine` is a dqmc_amd.abi.Engine (single chain); `bernoulli(p)` plays
    utility::random::bernoulli.  Returns (exchange_attempt, exchange_accepted, accepted)."""
    exchange_attempt += 1
    partner = partner_rank(rank, world_size, exchange_attempt)
    if partner < 0 or partner >= world_size:
        return exchange_attempt, exchange_accepted, False
    # --- field exchange (MPI_Sendrecv tag 0) ---
    my_fields = np.ascontiguousarray(engine.get_fields(), dtype=np.int64)
    partner_fields = _sendrecv(torch.from_numpy(my_fields).to(device), partner).cpu().numpy()
    SC = engine.global_action()                                   # S_r({s}_r)
    engine.set_fields(partner_fields); engine.init()              # sim.init_stacks + init_greenfunctions
    SC_prime = engine.global_action()                             # S_r({s}_partner)
    # tags 1 and 2: the two cross actions, shipped together
    got = _sendrecv(torch.tensor([SC_prime, SC], dtype=torch.float64, device=device), partner).cpu()
    SC_prime_partner, SC_partner = float(got[0]), float(got[1])
    flag = torch.zeros(1, dtype=torch.int32, device=device)
    if rank < partner:                                            # the lower rank decides (source/update.cpp:93-102)
        deltaS = (SC_prime + SC_prime_partner) - (SC + SC_partner)
        accept = bool(bernoulli(min(1.0, math.exp(-deltaS)) if deltaS > -700 else 1.0))
        if rank == 0:
            exchange_accepted += int(accept)
        flag[0] = int(accept)
        dist.send(flag, partner)                                  # MPI_Send tag 3
    else:
        dist.recv(flag, partner)                                  # MPI_Recv tag 3
        accept = bool(int(flag.item()))
    if not accept:                                                # restore (source/update.cpp:109-115)
        engine.set_fields(my_fields); engine.init()
    return exchange_attempt, exchange_accepted, accept
