"""Child process of tests/test_gpu_replica.py::test_exchange_rounds_between_two_processes: ONE rank of a two-rank world, one HIP
engine on device 0, the library's communicator over the callback transport with torch.distributed (gloo) point-to-point as the
MPI_Sendrecv -- what the reference's ranks are (processes, source/main.cpp:20-37).  Two exchange rounds (forced accept, forced
reject), then the cfg-4 driver loop of dqmc_amd/pt_run.py over the same transport.  usage: pt_two_proc.py <out_dir>
(RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dqmc_amd                                                     # noqa: E402
from dqmc_amd import HubbardModel                                   # noqa: E402
from dqmc_amd.launch import dist_init, finalize                     # noqa: E402
from dqmc_amd.pt_run import gloo_sendrecv, run_pt                   # noqa: E402

BETAS = [2.0, 1.6]


def main():
    out_dir = sys.argv[1]
    import torch.distributed as dist
    d = dist_init("gloo")
    assert d.world == 2
    lib = dqmc_amd.lib()
    m = HubbardModel(L1=4, L2=4, U=4.0, beta=BETAS[d.rank], nt=20, n_stab=10)
    e = m.engine(lib, device=0); e.set_fields(m.random_fields(100 + d.rank)); e.init()
    comm = lib.comm_callbacks(2, d.rank, gloo_sendrecv(dist))
    comm.barrier()
    rec = {}
    for attempt, u in ((1, 0.0), (2, 2.0)):                         # u = 0 < p: accepted; u = 2 > p: rejected whatever deltaS is
        r = comm.exchange_round(e, attempt, u)
        rec[f"res{attempt}"] = np.array([r.partner, r.decider, r.accepted, r.S, r.S_prime, r.S_partner, r.S_prime_partner, r.deltaS])
        rec[f"fields{attempt}"] = e.get_fields(); rec[f"G{attempt}"] = e.get_G(); rec[f"logdet{attempt}"] = np.array(e.get_logdet())
    comm.barrier(); comm.close(); e.close()
    # the cfg-4 loop of pt_run.py (barrier, exchange every sweep_steps sweeps, the two reductions) over the same transport
    lines = []
    rate, attempts, accepted = run_pt(d, lib, BETAS, L=4, U=4.0, nt=20, n_stab=10, therm=1, sweeps=6, sweep_steps=2, log=lines.append,
                                      transport="callbacks", device=0)
    rec["pt"] = np.array([rate, attempts, accepted]); rec["pt_log"] = np.array(lines[0] if lines else "")
    np.savez(os.path.join(out_dir, f"rank{d.rank}.npz"), **rec)
    finalize(d)


if __name__ == "__main__":
    main()
