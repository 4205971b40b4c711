#!/usr/bin/env python3
"""Parallel-tempering run (BASELINE.json configs[3]): one inverse temperature per rank / GPU, replica exchange every
`sweep_steps` sweeps (source/main.cpp:39-67,146-153) through the library's own transport -- dqmc_replica_exchange_round
over an RCCL communicator (include/dqmc_hip.h), the field arrays going HBM to HBM over xGMI.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        dqmc_amd/pt_run.py --sweeps 40 --sweep-steps 5

torch.distributed only carries the rendezvous (the 128-byte RCCL id) here.  The number of betas must equal the world
size and the world size must be even, the checks the reference makes before MPI_Abort (source/main.cpp:52-63).  The C++
driver (dqmc_amd/host/main.cpp) is the same loop without Python."""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def gloo_sendrecv(dist):
    """An MPI_Sendrecv-shaped callback over torch.distributed point-to-point (any backend that does CPU tensors, i.e. gloo): the
    callback transport of dqmc_comm_create_callbacks between real processes -- ranks that share a GPU (RCCL refuses two ranks on
    one device), or a host without RCCL."""
    import torch

    def sendrecv(send: bytes, partner: int, tag: int) -> bytes:
        t_send = torch.frombuffer(bytearray(send), dtype=torch.uint8); t_recv = torch.empty_like(t_send)
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, t_send, partner), dist.P2POp(dist.irecv, t_recv, partner)]):
            w.wait()
        return t_recv.numpy().tobytes()
    return sendrecv


def run_pt(d, lib, betas, L, U, nt, n_stab, therm, sweeps, sweep_steps, seed=1234, log=print, transport="rccl", device=None):
    """Returns (sweeps/s of the slowest rank, exchange_attempt, exchange_accepted of rank 0's pairs).  transport: "rccl" (one GPU per
    rank, fields HBM to HBM) or "callbacks" (dqmc_comm_create_callbacks over torch.distributed point-to-point)."""
    import torch.distributed as dist
    from dqmc_amd import HubbardModel
    if len(betas) != d.world:
        raise SystemExit(f"ERROR: The number of betas ({len(betas)}) must match the number of processes ({d.world}).")
    if d.world % 2 != 0:
        raise SystemExit(f"ERROR: currently number of processor ( nprocs = {d.world}) need to be even for replica exchange")
    dev = d.local_rank if device is None else device
    if transport == "rccl":
        ids = [lib.comm_unique_id() if d.rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        comm = lib.comm_rccl(ids[0], d.world, d.rank, dev)
    else:
        comm = lib.comm_callbacks(d.world, d.rank, gloo_sendrecv(dist))
    model = HubbardModel(L1=L, L2=L, U=U, beta=float(betas[d.rank]), nt=nt, n_stab=n_stab)
    eng = model.engine(lib, device=dev)
    eng.set_fields(model.random_fields(seed + d.rank)); eng.init()
    rng = np.random.default_rng(seed + 1000 + d.rank)

    def sweep():
        eng.sweep_0_to_beta(*model.random_stream(rng)); eng.sweep_beta_to_0(*model.random_stream(rng))

    for _ in range(therm):
        sweep()
    eng.sync(); comm.barrier()
    attempt = accepted = 0
    t_ex = 0.0
    t0 = time.perf_counter()
    for isweep in range(1, sweeps + 1):
        if isweep % sweep_steps == 0:
            comm.barrier()                                            # MPI_Barrier, source/main.cpp:148
            attempt += 1
            te = time.perf_counter()
            res = comm.exchange_round(eng, attempt, float(rng.random()))
            t_ex += time.perf_counter() - te
            if d.rank == 0:
                accepted += res.accepted                               # source/update.cpp:99-101
        sweep()
    eng.sync(); comm.barrier()
    dt = time.perf_counter() - t0
    st = eng.stats()
    sums = comm.allreduce_sum([dt, st.n_accepted / max(1, st.n_proposed)])        # the two MPI_Reduce(SUM), source/main.cpp:186-187
    slow = [None] * d.world
    dist.all_gather_object(slow, dt)
    dt = max(slow)                                                                 # the job is as fast as its slowest rank
    if d.rank == 0:
        log(f"PT: {d.world} replicas over {comm.transport}, {sweeps} sweeps in {dt:.2f} s (mean over ranks {sums[0] / d.world:.2f} s) = {d.world * sweeps / dt:.2f} sweeps/s total; "
            f"acceptance {sums[1] / d.world:.4f}; exchange rate {accepted / max(1, attempt):.4f} ({accepted}/{attempt}), "
            f"{1e3 * t_ex / max(1, attempt):.1f} ms per round; max wrap err {st.max_err:.3e}")
    comm.close(); eng.close()
    return sweeps / dt, attempt, accepted


def main():
    from dqmc_amd import CFG4_BETAS, CONFIGS
    cfg = CONFIGS["cfg4"]
    ap = argparse.ArgumentParser()
    ap.add_argument("--betas", default=",".join(str(b) for b in CFG4_BETAS))
    ap.add_argument("--L", type=int, default=cfg["L1"])
    ap.add_argument("--U", type=float, default=cfg["U"])
    ap.add_argument("--nt", type=int, default=cfg["nt"])
    ap.add_argument("--n-stab", type=int, default=cfg["n_stab"])
    ap.add_argument("--therm", type=int, default=5)
    ap.add_argument("--sweeps", type=int, default=20)
    ap.add_argument("--sweep-steps", type=int, default=5)
    args = ap.parse_args()
    import dqmc_amd
    from dqmc_amd.launch import dist_init, finalize
    d = dist_init()
    betas = [float(b) for b in args.betas.split(",") if b.strip()]
    run_pt(d, dqmc_amd.lib(), betas, args.L, args.U, args.nt, args.n_stab, args.therm, args.sweeps, args.sweep_steps)
    finalize(d)


if __name__ == "__main__":
    main()
