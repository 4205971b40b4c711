"""world_size-2 tests on CPU (gloo): the one-process-per-GPU plumbing of
bench.py (barrier, max/sum over ranks) and the replica-exchange round
(update::replica_exchange, source/update.cpp:47-117) with the CPU oracle
standing in for the engine -- the transport code is the same that runs over
RCCL on the GPU node."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dqmc_amd import HubbardModel
    from dqmc_amd.launch import barrier, dist_init, finalize, max_over_ranks, sum_over_ranks
    from dqmc_amd.replica import partner_rank, replica_exchange
    from oracle import oracle
    d = dist_init("gloo")
    assert d.world == world and d.rank == rank
    barrier(d)
    assert max_over_ranks(d, 1.0 + rank) == float(world)
    assert sum_over_ranks(d, 1.0) == float(world)
    # one beta per rank (source/main.cpp:47-67)
    betas = [2.0, 1.6]
    m = HubbardModel(L1=4, L2=4, U=4.0, beta=betas[rank], nt=20, n_stab=10)
    e = m.engine(oracle()); f0 = m.random_fields(100 + rank); e.set_fields(f0); e.init()
    S_own = e.global_action()
    log = []
    attempt, accepted = 0, 0
    for rnd, force in enumerate([True, False, None]):
        before = e.get_fields().copy()
        if force is None:
            rng = np.random.default_rng(5)
            bern = lambda p: bool(rng.random() < p)
        else:
            bern = lambda p, force=force: force
        attempt, accepted, acc = replica_exchange(rank, world, bern, attempt, accepted, e, d.device)
        after = e.get_fields()
        log.append(dict(round=rnd, acc=bool(acc), same=bool((after == before).all()), before=before, after=after,
                        partner=partner_rank(rank, world, attempt), S=e.global_action(), G=e.get_G()))
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.array(log, dtype=object), allow_pickle=True)
    np.save(os.path.join(out_dir, f"f0_{rank}.npy"), f0)
    assert abs(S_own - S_own) == 0.0
    finalize(d)


def test_two_rank_replica_exchange_over_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    logs = [np.load(tmp_path / f"rank{r}.npy", allow_pickle=True) for r in range(world)]
    f0 = [np.load(tmp_path / f"f0_{r}.npy") for r in range(world)]
    # round 0: forced accept -> the two ranks hold each other's fields
    assert logs[0][0]["acc"] and logs[1][0]["acc"]
    assert (logs[0][0]["after"] == f0[1]).all() and (logs[1][0]["after"] == f0[0]).all()
    # round 1: forced reject -> nothing changes, both ranks agree
    assert not logs[0][1]["acc"] and not logs[1][1]["acc"] and logs[0][1]["same"] and logs[1][1]["same"]
    # round 2: Metropolis decision taken by the lower rank, shared with the partner
    assert logs[0][2]["acc"] == logs[1][2]["acc"]
    if logs[0][2]["acc"]:
        assert (logs[0][2]["after"] == logs[1][2]["before"]).all() and (logs[1][2]["after"] == logs[0][2]["before"]).all()
    for r in range(world):
        for rec in logs[r]:
            assert rec["partner"] == 1 - r
    # G after a round equals a from-scratch evaluation of the fields the rank ended with
    from dqmc_amd import HubbardModel
    from oracle import oracle
    for r, beta in enumerate([2.0, 1.6]):
        m = HubbardModel(L1=4, L2=4, U=4.0, beta=beta, nt=20, n_stab=10)
        e = m.engine(oracle()); e.set_fields(logs[r][2]["after"]); e.init()
        assert np.abs(e.get_G() - logs[r][2]["G"]).max() < 1e-12


def _pt_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dqmc_amd.launch import dist_init, finalize
    from dqmc_amd.pt_run import run_pt
    from oracle import oracle
    d = dist_init("gloo")
    lines = []
    rate, attempt, accepted = run_pt(d, lambda m: m.engine(oracle()), [2.0, 1.8], L=4, U=4.0, nt=20, n_stab=10, therm=1, sweeps=6,
                                     sweep_steps=2, log=lines.append)
    with open(os.path.join(out_dir, f"pt{rank}.txt"), "w") as fh:
        fh.write(f"{attempt} {accepted} {rate}\n" + "\n".join(lines))
    finalize(d)


def test_parallel_tempering_driver_two_ranks(tmp_path):
    """The cfg-4 driver loop (barrier + replica_exchange every sweep_steps sweeps) end to end on gloo."""
    mp.spawn(_pt_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a0 = open(tmp_path / "pt0.txt").read().split("\n"); a1 = open(tmp_path / "pt1.txt").read().split("\n")
    assert a0[0].split()[0] == "3" and a1[0].split()[0] == "3"          # 6 sweeps, exchange every 2nd
    assert "exchange rate" in a0[1] and a1[1:] == [""]                     # only rank 0 reports (source/main.cpp:204)
