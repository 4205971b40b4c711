#!/usr/bin/env python3
"""Parallel-tempering driver (BASELINE.json configs[3]): one inverse temperature per rank / GPU,
replica exchange every `sweep_steps` sweeps over RCCL (source/main.cpp:39-67,146-153).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        dqmc_amd/pt_run.py --betas 8,7,6,5,4,3.5,3,2.5 --sweeps 40 --sweep-steps 5

The number of betas must equal the world size and the world size must be even, exactly the checks
the reference makes before MPI_Abort (source/main.cpp:52-63)."""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run_pt(d, engine_factory, betas, L, U, nt, n_stab, therm, sweeps, sweep_steps, seed=1234, log=print):
    """`engine_factory(model) -> Engine`; returns (sweeps/s of this rank, exchange_attempt, exchange_accepted)."""
    from dqmc_amd import HubbardModel
    from dqmc_amd.launch import barrier, max_over_ranks, sum_over_ranks
    from dqmc_amd.replica import replica_exchange
    if len(betas) != d.world:
        raise SystemExit(f"ERROR: The number of betas ({len(betas)}) must match the number of processes ({d.world}).")
    if d.world % 2 != 0:
        raise SystemExit(f"ERROR: currently number of processor ( nprocs = {d.world}) need to be even for replica exchange")
    model = HubbardModel(L1=L, L2=L, U=U, beta=float(betas[d.rank]), nt=nt, n_stab=n_stab)
    eng = engine_factory(model)
    eng.set_fields(model.random_fields(seed + d.rank)); eng.init()
    rng = np.random.default_rng(seed + 1000 + d.rank)
    bern = lambda p: bool(rng.random() < p)

    def sweep():
        eng.sweep_0_to_beta(*model.random_stream(rng)); eng.sweep_beta_to_0(*model.random_stream(rng))

    for _ in range(therm):
        sweep()
    eng.sync(); barrier(d)
    attempt = accepted = 0
    t0 = time.perf_counter()
    for isweep in range(1, sweeps + 1):
        if isweep % sweep_steps == 0:
            barrier(d)                                               # MPI_Barrier, source/main.cpp:148
            attempt, accepted, _ = replica_exchange(d.rank, d.world, bern, attempt, accepted, eng, d.device)
        sweep()
    eng.sync(); barrier(d)
    dt = max_over_ranks(d, time.perf_counter() - t0)
    st = eng.stats()
    acc = sum_over_ranks(d, st.n_accepted / max(1, st.n_proposed)) / d.world
    if d.rank == 0:
        log(f"PT: {d.world} replicas, {sweeps} sweeps in {dt:.2f} s = {d.world * sweeps / dt:.2f} sweeps/s total; "
            f"acceptance {acc:.4f}; exchange rate {accepted / max(1, attempt):.4f} ({accepted}/{attempt}); "
            f"max wrap err {st.max_err:.3e}")
    return sweeps / dt, attempt, accepted


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--betas", default="8,7,6,5,4,3.5,3,2.5")
    ap.add_argument("--L", type=int, default=16)
    ap.add_argument("--U", type=float, default=8.0)
    ap.add_argument("--nt", type=int, default=200)
    ap.add_argument("--n-stab", type=int, default=10)
    ap.add_argument("--therm", type=int, default=5)
    ap.add_argument("--sweeps", type=int, default=20)
    ap.add_argument("--sweep-steps", type=int, default=5)
    args = ap.parse_args()
    import dqmc_amd
    from dqmc_amd.launch import dist_init, finalize
    d = dist_init()
    lib = dqmc_amd.lib()
    betas = [float(b) for b in args.betas.split(",") if b.strip()]
    run_pt(d, lambda m: m.engine(lib, device=d.local_rank), betas, args.L, args.U, args.nt, args.n_stab, args.therm, args.sweeps, args.sweep_steps)
    finalize(d)


if __name__ == "__main__":
    main()
