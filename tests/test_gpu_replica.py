"""GPU tests (-m gpu) of replica exchange, BASELINE.json configs[3] (cfg 4: 16x16 Hubbard, 8 inverse temperatures):
HIP engines driven through update::replica_exchange of the host facade (dqmc_amd/host/dqmc_host.hpp) and through the
C ABI (dqmc_replica_exchange_round) with in-process transports, against CPU-oracle engines fed with the same random
streams (tests/pt_twin.py restates source/update.cpp:34-117).  Decisions, partners, counters and exchanged fields must
match exactly; the actions S, S' to 1e-8 relative; G after acceptance / after the restoring re-initialisation to
1e-10 * max(1, max|G|).  The RCCL transport itself needs one GPU per rank (RCCL refuses two ranks on one device), so on
the one-GPU box the wire is the callback transport; bench.py exercises the RCCL path when it runs on N > 1 GPUs."""
import os
import re
import subprocess
import threading

import numpy as np
import pytest

import dqmc_amd
from dqmc_amd import CONFIGS, HubbardModel

from pt_twin import HostPT, OracleTwin, ini_text, load_host
from test_replica import PyHub

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _compare_round(tag, res, ref, pt, tw, n, nt):
    W = len(ref)
    for r in range(W):
        a, b = res[r], ref[r]
        assert a.partner == b["partner"], (tag, r)
        assert bool(a.accepted) == bool(b["accepted"]), (tag, r, a.deltaS, b["deltaS"])
        assert a.decider == b["decider"]
        for got, want in ((a.S, b["S"]), (a.S_prime, b["S_prime"]), (a.S_partner, b["S_partner"]), (a.S_prime_partner, b["S_prime_partner"])):
            assert abs(got - want) <= 1e-8 * max(1.0, abs(want)), (tag, r, got, want)
        assert abs(a.deltaS - b["deltaS"]) <= 1e-6 * max(1.0, abs(b["deltaS"])) + 1e-8 * max(abs(b["S"]), abs(b["S_prime"]))
        g, o = pt.get(r, n, nt), tw.get(r)
        assert np.array_equal(g["fields"], o["fields"]), (tag, r)
        scale = max(1.0, np.abs(o["G"]).max())
        err = np.abs(g["G"] - o["G"]).max()
        assert err <= TOL * scale, (tag, r, err, scale)
        assert abs(g["logdet"] - o["logdet"]) <= 1e-9 * max(1.0, abs(o["logdet"]))
        assert (g["attempt"], g["accepted"]) == (o["attempt"], o["accepted"])


def test_cfg4_replica_exchange_on_hip_engines(hip, orc):
    """8 HIP engines at 8 betas (16x16, U = 8, Ltau = 200) on one GPU, 4 exchange rounds (odd attempt with the 0 <-> 7 wrap,
    even attempt, odd again, one more after a full sweep of every replica) through update::replica_exchange."""
    cfg = CONFIGS["cfg4"]; L, U, nt, n_stab = cfg["L1"], cfg["U"], cfg["nt"], cfg["n_stab"]; n = L * L
    betas = [8.0, 7.9, 7.0, 6.9, 6.0, 5.9, 5.0, 4.9]          # near pairs: probabilistic decisions; far pairs: forced rejections
    seeds = [1000 + r for r in range(8)]
    ini = ini_text(L, U, nt, n_stab)
    h = load_host()
    orc.set_backend("lapack")                                 # MKL dgeqp3 / dgetrf when present (what the reference links); built-in otherwise
    pt = HostPT(h, ini, betas, seeds); tw = OracleTwin(orc, h, ini, betas, seeds, n, nt, n_stab)
    try:
        for r in range(8):                                    # same initial fields on both sides (GHQField's by-value generator copy)
            assert np.array_equal(pt.get(r, n, nt)["fields"], tw.get(r)["fields"])
        seen = set()
        for rnd in range(3):
            res, ref = pt.exchange(), tw.exchange()
            _compare_round(f"round {rnd}", res, ref, pt, tw, n, nt)
            seen |= {(rnd % 2, bool(x["accepted"])) for x in ref}
            if rnd == 0:
                assert res[0].partner == 7 and res[7].partner == 0 and res[0].decider == 1        # the wrap pair, rank 0 decides
        assert {True, False} <= {a for _, a in seen}, "both an accepted and a rejected swap must occur"
        # one full sweep of every replica between rounds, all 8 engines AT ONCE on one device (three of them on the persistent slice
        # kernel, five on the kernel pairs: slice_reserve), compared with the oracle entry by entry
        pt.sweeps(1, concurrently=True); tw.sweeps(1)
        for r in range(8):
            g, o = pt.get(r, n, nt), tw.get(r)
            if not np.array_equal(g["fields"], o["fields"]):
                bad = np.nonzero((g["fields"] != o["fields"]).any(axis=1))[0]
                print(f"replica {r}: {int((g['fields'] != o['fields']).sum())} field entries in {len(bad)} slices differ, slices {bad[:12]}, sites of the first: "
                      f"{np.nonzero(g['fields'][bad[0]] != o['fields'][bad[0]])[0][:12]}, gpu {g['fields'][bad[0]][:8]} cpu {o['fields'][bad[0]][:8]}, "
                      f"max|dG| {np.abs(g['G'] - o['G']).max():.3e}, accepted gpu/cpu {g['accepted']}/{o['accepted']}")
            assert np.array_equal(g["fields"], o["fields"]), r
            assert np.abs(g["G"] - o["G"]).max() <= TOL * max(1.0, np.abs(o["G"]).max())
        res, ref = pt.exchange(), tw.exchange()
        _compare_round("round 3 (after a sweep)", res, ref, pt, tw, n, nt)
        for r in range(8):                                    # generators advanced identically (only deciders drew, two words each)
            assert pt.rng_peek(r) == h.dqmc_host_rng_next(tw.rng[r]), r
    finally:
        pt.close(); tw.close(); orc.set_backend("builtin")


def test_many_engines_sweeping_at_once_on_one_device(hip):
    """8 single-chain engines (cfg-4 size) driven by 8 host threads on ONE device, two sweeps each: the first three hold CU
    reservations for the persistent slice kernel, the others take the scan / flush kernel pairs (slice_reserve, update.hip).  No
    hand-off may fail (a failure is an error return), and every engine must end self-consistent: its G equals the from-scratch
    evaluation of its own final fields."""
    cfg = CONFIGS["cfg4"]; L, U, nt, n_stab = cfg["L1"], cfg["U"], cfg["nt"], cfg["n_stab"]; n = L * L
    betas = [8.0 - 0.5 * r for r in range(8)]; seeds = [500 + r for r in range(8)]
    h = load_host()
    pt = HostPT(h, ini_text(L, U, nt, n_stab), betas, seeds)
    try:
        pt.sweeps(2, concurrently=True)
        for r in range(8):
            g = pt.get(r, n, nt)
            m = HubbardModel(L1=L, L2=L, U=U, beta=betas[r], nt=nt, n_stab=n_stab)
            e = m.engine(hip); e.set_fields(g["fields"]); e.init()
            scale = max(1.0, np.abs(g["G"]).max())
            assert np.abs(e.get_G() - g["G"]).max() <= 1e-8 * scale, r
            assert abs(e.get_logdet() - g["logdet"]) <= 1e-8 * max(1.0, abs(g["logdet"]))
            e.close()
    finally:
        pt.close()


def test_exchange_round_through_the_c_abi(hip, orc):
    """dqmc_replica_exchange_round called directly (ctypes) by two ranks = two Python threads with a Python
    MPI_Sendrecv callback; a forced accept (u = 0) and a forced reject (u = 1 - eps with deltaS > 0)."""
    betas = [2.0, 1.6]
    models = [HubbardModel(L1=4, L2=4, U=4.0, beta=b, nt=20, n_stab=10) for b in betas]
    f0 = [models[r].random_fields(100 + r) for r in range(2)]
    hub = PyHub(); results = {}; errs = []
    eng = [models[r].engine(hip) for r in range(2)]
    ref = [models[r].engine(orc) for r in range(2)]
    for r in range(2):
        for e in (eng[r], ref[r]):
            e.set_fields(f0[r]); e.init()

    def run(r, attempt, u):
        try:
            c = hip.comm_callbacks(2, r, hub.endpoint(r))
            results[(r, attempt)] = c.exchange_round(eng[r], attempt, u)
            c.close()
        except Exception as e:                      # noqa: BLE001
            errs.append((r, repr(e)))

    def both(attempt, u):
        th = [threading.Thread(target=run, args=(r, attempt, u)) for r in range(2)]
        [t.start() for t in th]; [t.join(300) for t in th]
        assert not errs, errs
        return results[(0, attempt)], results[(1, attempt)]

    # reference values of the four actions
    S = [ref[r].global_action() for r in range(2)]
    Sp = []
    for r in range(2):
        ref[r].set_fields(f0[1 - r]); ref[r].init(); Sp.append(ref[r].global_action())
    dS = (Sp[0] + Sp[1]) - (S[0] + S[1])
    a0, a1 = both(1, 0.0)                                     # u = 0 < p: accepted whatever deltaS is
    assert (a0.partner, a1.partner, a0.decider, a1.decider) == (1, 0, 1, 0)
    assert a0.accepted == 1 and a1.accepted == 1
    for r, a in enumerate((a0, a1)):
        assert abs(a.S - S[r]) < 1e-9 * abs(S[r]) and abs(a.S_prime - Sp[r]) < 1e-9 * abs(Sp[r])
        assert abs(a.S_partner - S[1 - r]) < 1e-9 * abs(S[r]) and abs(a.deltaS - dS) < 1e-7
        assert np.array_equal(eng[r].get_fields(), f0[1 - r])                    # swapped
        assert np.abs(eng[r].get_G() - ref[r].get_G()).max() < TOL               # ref[r] holds the partner's fields too
    # now swap back with a decision that must fail: fields are exchanged again, deltaS' = -dS; pick u to reject
    b0, b1 = both(2, 1.0 - 1e-12 if -dS > 0 else 2.0)
    assert b0.accepted == 0 and b1.accepted == 0
    for r in range(2):
        assert np.array_equal(eng[r].get_fields(), f0[1 - r])                    # restored = still the swapped configuration
        assert np.abs(eng[r].get_G() - ref[r].get_G()).max() < TOL
    for e in eng + ref:
        e.close()


def test_exchange_rounds_between_two_processes(hip, orc, tmp_path):
    """The reference's ranks are PROCESSES (source/main.cpp:20-37): two child processes, one HIP engine each on device 0, the
    library's communicator over the callback transport with gloo point-to-point as the MPI_Sendrecv (tests/pt_two_proc.py).  Round 1
    is a forced accept, round 2 a forced reject; partners, decider, the four actions, deltaS, the exchanged fields and G are
    compared with oracle engines here.  The children then run dqmc_amd/pt_run.py's cfg-4 loop over the same transport."""
    import socket, sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    here = os.path.dirname(os.path.abspath(__file__))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(here, "pt_two_proc.py"), str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, (r, outs[r][0][-1500:], outs[r][1][-3000:])
    z = [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]
    betas = [2.0, 1.6]
    models = [HubbardModel(L1=4, L2=4, U=4.0, beta=b, nt=20, n_stab=10) for b in betas]
    f0 = [models[r].random_fields(100 + r) for r in range(2)]
    ref = [models[r].engine(orc) for r in range(2)]
    S, Sp = [], []
    for r in range(2):
        ref[r].set_fields(f0[r]); ref[r].init(); S.append(ref[r].global_action())
    for r in range(2):
        ref[r].set_fields(f0[1 - r]); ref[r].init(); Sp.append(ref[r].global_action())       # ref[r] now holds the partner's fields
    dS = (Sp[0] + Sp[1]) - (S[0] + S[1])
    for r in range(2):
        partner, decider, accepted, s_, sp_, s_part, sp_part, ds_ = z[r]["res1"]
        assert (int(partner), int(decider), int(accepted)) == (1 - r, 1 if r == 0 else 0, 1)
        assert abs(s_ - S[r]) < 1e-9 * abs(S[r]) and abs(sp_ - Sp[r]) < 1e-9 * abs(Sp[r]) and abs(s_part - S[1 - r]) < 1e-9 * abs(S[1 - r])
        assert abs(sp_part - Sp[1 - r]) < 1e-9 * abs(Sp[1 - r]) and abs(ds_ - dS) < 1e-7
        assert np.array_equal(z[r]["fields1"], f0[1 - r])                                    # accepted: the partner's fields, across the process boundary
        assert np.abs(z[r]["G1"] - ref[r].get_G()).max() < TOL
        assert abs(float(z[r]["logdet1"]) - ref[r].get_logdet()) < 1e-9 * max(1.0, abs(ref[r].get_logdet()))
        partner, decider, accepted, s_, sp_, *_ = z[r]["res2"]
        assert (int(partner), int(accepted)) == (1 - r, 0)                                   # attempt 2 of a world of two pairs the same ranks
        assert abs(s_ - Sp[r]) < 1e-9 * abs(Sp[r]) and abs(sp_ - S[r]) < 1e-9 * abs(S[r])    # own action is now the swapped one, the trial the original
        assert np.array_equal(z[r]["fields2"], f0[1 - r])                                    # rejected: restored = still the swapped configuration
        assert np.abs(z[r]["G2"] - ref[r].get_G()).max() < TOL
        rate, attempts, acc = z[r]["pt"]
        assert rate > 0 and int(attempts) == 3 and 0 <= int(acc) <= 3                        # 6 sweeps, exchange every 2
    assert "2 replicas over callbacks" in str(z[0]["pt_log"])
    for e in ref:
        e.close()


def test_driver_parallel_tempering_in_process(hip, tmp_path):
    """dqmc_driver with [ParallelTempering] enabled = true and no launcher: one thread per beta on the visible GPU,
    swaps through update::InProcessHub (source/main.cpp:39-67,146-153,203-208)."""
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    driver = os.path.join(here, "dqmc_amd", "dqmc_driver")
    ini = ("[Lattice]\nL1 = 4\nL2 = 4\n[hubbard]\nU = 4.0\nt = 1.0\nmu = -0.1\n[simulation]\nbeta = 2.0\nnt = 20\nn_therms = 3\nn_sweeps = 4\n"
           "n_bins = 2\nn_stab = 10\nsymmetric = false\nisMeasureUnequalTime = false\n[ParallelTempering]\nenabled = true\nsweep_steps = 2\n"
           "betas = 2.0, 1.9, 1.8, 1.7\n")
    (tmp_path / "parameters.in").write_text(ini)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([driver, "parameters.in", "0", "4242"], cwd=tmp_path, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]
    assert "Parallel Tempering enabled, 4 rank(s), transport callbacks" in out.stdout, out.stdout
    m = re.search(r"Parallel tempering exchange rate = ([\d.]+) \((\d+)/(\d+)\)", out.stdout)
    assert m and int(m.group(3)) == 4, out.stdout                 # 8 sweeps, exchange every 2
    assert 0 <= int(m.group(2)) <= 4
    for r in range(4):
        assert re.search(rf"rank {r}: device 0, beta {[2, 1.9, 1.8, 1.7][r]}", out.stdout), out.stdout
        assert len(re.findall(rf"rank {r} bin \d+ \(4 sweeps\)", out.stdout)) == 2
    # the reference's two MPI_Abort checks (source/main.cpp:52-62)
    (tmp_path / "parameters.in").write_text(ini.replace("betas = 2.0, 1.9, 1.8, 1.7", "betas = 2.0, 1.9, 1.8"))
    out = subprocess.run([driver, "parameters.in", "0", "4242"], cwd=tmp_path, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 1 and "need to be even" in out.stderr
    env2 = dict(env, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    out = subprocess.run([driver, "parameters.in", "0", "4242"], cwd=tmp_path, capture_output=True, text=True, timeout=300, env=env2)
    assert out.returncode == 1 and "must match the number of MPI processes" in out.stderr


@pytest.mark.parametrize("with_torch_nccl", [False, True])
def test_rccl_transport_loopback_on_one_gpu(hip, with_torch_nccl):
    """The RCCL transport on hardware as far as one GPU allows (RCCL refuses two ranks on one device): a world-1 communicator
    from dqmc_comm_unique_id / dqmc_comm_create_rccl, then dqmc_comm_selftest = grouped ncclSend + ncclRecv to the own rank through
    the same helper a round uses, and one ncclAllReduce.  Second case: in a process that has already brought up torch.distributed's
    own "nccl" (= RCCL) process group, as bench.py --gpus N has when it creates the communicator.  Runs in a child process: a crash
    inside the collective library must fail this test, not end the session."""
    import sys
    code = (
        "import os, sys\n"
        f"sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})\n"
        "import numpy as np\n"
        + ("import torch, torch.distributed as dist\n"
           "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29577', RANK='0', WORLD_SIZE='1')\n"
           "torch.cuda.set_device(0); dist.init_process_group('nccl', rank=0, world_size=1)\n"
           "t = torch.ones(4, device='cuda'); dist.all_reduce(t); torch.cuda.synchronize(); assert float(t.sum()) == 4.0\n" if with_torch_nccl else "")
        + "import dqmc_amd\n"
        "lib = dqmc_amd.lib()\n"
        "uid = lib.comm_unique_id(); assert len(uid) == 128 and any(uid)\n"
        "c = lib.comm_rccl(uid, 1, 0, 0)\n"
        "assert c.transport == 'rccl' and c.rank == 0 and c.world == 1\n"
        "c.selftest(); c.barrier(); assert float(c.allreduce_sum([2.5])[0]) == 2.5\n"
        "m = dqmc_amd.HubbardModel(**dqmc_amd.CONFIGS['cfg1']); e = m.engine(lib); e.set_fields(m.random_fields(1)); e.init()\n"
        "r = c.exchange_round(e, 1, 0.5); assert r.partner == -1 and r.accepted == 0      # a single replica has nobody to swap with\n"
        "c.selftest(); c.close(); e.close()\n"
        + ("dist.destroy_process_group()\n" if with_torch_nccl else "")
        + "print('LOOPBACK OK')\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "LOOPBACK OK" in out.stdout, (out.returncode, out.stdout[-2000:], out.stderr[-4000:])
