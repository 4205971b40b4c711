"""world_size > 1 tests on CPU (gloo): the one-process-per-GPU plumbing of bench.py (rendezvous, barrier, max / sum over
ranks, the self-launch of `bench.py --gpus N`) and the library's communicator (dqmc_comm_*: barrier, allreduce) across real
processes with gloo point-to-point as the transport callback.  The exchange round itself needs engines, i.e. a GPU:
tests/test_gpu_replica.py."""
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
    import torch.distributed as dist
    import dqmc_amd
    from dqmc_amd.launch import barrier, dist_init, finalize, max_over_ranks, sum_over_ranks
    d = dist_init("gloo")
    assert d.world == world and d.rank == rank
    barrier(d)
    assert max_over_ranks(d, 1.0 + rank) == float(world)
    assert sum_over_ranks(d, 1.0) == float(world)
    # the library's communicator (include/dqmc_hip.h) with gloo point-to-point as the MPI_Sendrecv callback: its barrier and
    # allreduce are what the C++ driver calls at source/main.cpp:148,186-187
    lib = dqmc_amd.lib()

    def sendrecv(send: bytes, partner: int, tag: int) -> bytes:
        t_send = torch.frombuffer(bytearray(send), dtype=torch.uint8); t_recv = torch.empty_like(t_send)
        ops = [dist.P2POp(dist.isend, t_send, partner), dist.P2POp(dist.irecv, t_recv, partner)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        return t_recv.numpy().tobytes()
    comm = lib.comm_callbacks(world, rank, sendrecv)
    comm.barrier()
    got = comm.allreduce_sum([1.0 + rank, 0.5])
    assert got.tolist() == [world * (world + 1) / 2.0, 0.5 * world]
    assert lib.partner_rank(rank, world, 1) == (rank + (-1 if rank % 2 == 0 else 1) + world) % world
    comm.barrier(); comm.close()
    np.save(os.path.join(out_dir, f"ok{rank}.npy"), got)
    finalize(d)


@pytest.mark.parametrize("world", [2, 4])
def test_library_communicator_across_processes_over_gloo(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert os.path.exists(tmp_path / f"ok{r}.npy")


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` without a launcher spawns N fresh rank processes itself (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set), rank 0 prints the one JSON line with n_gpus = N; a launcher whose WORLD_SIZE disagrees with --gpus, or too
    few visible devices, is an error -- never a silent N = 1 run."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec == {"launch_check": True, "n_gpus": 2, "max_over_ranks": 2.0, "backend": "gloo"}
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True, timeout=600,
                         env=dict(env, WORLD_SIZE="4", RANK="0"))
    assert bad.returncode == 2 and "does not match WORLD_SIZE" in bad.stderr
    if torch.cuda.device_count() < 2:
        few = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=600, env=env)
        assert few.returncode == 2 and "GPU(s) visible" in few.stderr and not few.stdout.strip()
