"""CPU tests of the replica-exchange boundary (include/dqmc_hip.h: dqmc_comm_*, dqmc_partner_rank,
dqmc_replica_exchange_round) and of the host facade's in-process hub.  No compute call needs a GPU here: the callback
transport's collectives run on host buffers; the RCCL entry points must fail cleanly without a device."""
import threading

import numpy as np
import pytest

import dqmc_amd
from dqmc_amd import DqmcError

from pt_twin import OracleTwin, load_host


@pytest.fixture(scope="module")
def host():
    return load_host()


def test_partner_rank_abi_matches_reference_rule():
    lib = dqmc_amd.lib()
    for world in (2, 4, 8):
        for attempt in range(1, 6):
            got = [lib.partner_rank(r, world, attempt) for r in range(world)]
            assert got == [OracleTwin.partner_rank(r, world, attempt) for r in range(world)]
            assert all(got[got[r]] == r and got[r] != r for r in range(world))
    assert [lib.partner_rank(r, 8, 1) for r in range(8)] == [7, 2, 1, 4, 3, 6, 5, 0]      # odd attempt: 0 <-> W-1 (source/update.cpp:44)


def test_rccl_entry_points_fail_cleanly_without_device():
    lib = dqmc_amd.lib()
    if lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(DqmcError) as ei:
        lib.comm_unique_id()
    assert ei.value.code == -2
    with pytest.raises(DqmcError) as ei:
        lib.comm_rccl(b"\0" * 128, 2, 0, 0)
    assert ei.value.code == -2


def test_exchange_round_rejects_null_arguments():
    import ctypes as C
    lib = dqmc_amd.lib()
    assert lib._sym("replica_exchange_round")(None, None, 1, 0.5, None) == -1
    assert "null" in lib._sym("last_error")().decode()
    h = C.c_void_p()
    assert lib._sym("comm_create_callbacks")(C.byref(h), 2, 5, dqmc_amd.abi.SENDRECV_FN(lambda *a: 0), None) == -1    # rank >= world


class PyHub:
    """MPI_Sendrecv between Python threads."""

    def __init__(self):
        self.cv = threading.Condition(); self.box = {}

    def endpoint(self, rank):
        def sendrecv(send: bytes, partner: int, tag: int) -> bytes:
            with self.cv:
                self.box.setdefault((rank, partner, tag), []).append(send)
                self.cv.notify_all()
                q = self.box.setdefault((partner, rank, tag), [])
                assert self.cv.wait_for(lambda: len(q) > 0, timeout=60)
                return q.pop(0)
        return sendrecv


@pytest.mark.parametrize("world", [2, 8])
def test_callback_transport_collectives(world):
    lib = dqmc_amd.lib(); hub = PyHub(); out = [None] * world; errs = []

    def rank_main(r):
        try:
            c = lib.comm_callbacks(world, r, hub.endpoint(r))
            assert (c.rank, c.world, c.transport) == (r, world, "callbacks")
            c.barrier()
            out[r] = c.allreduce_sum([r + 1.0, 10.0 * r])
            c.barrier(); c.close()
        except Exception as e:                      # noqa: BLE001
            errs.append((r, repr(e)))
    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join(120) for t in th]
    assert not errs, errs
    want = np.array([world * (world + 1) / 2.0, 10.0 * world * (world - 1) / 2.0])
    for r in range(world):
        assert np.array_equal(out[r], want)


def test_in_process_hub_delivers_every_message(host):
    assert host.dqmc_host_hub_selftest(2, 5) == 0
    assert host.dqmc_host_hub_selftest(8, 9) == 0


def test_decider_uniform_is_the_bernoulli_draw(host):
    """update::draw_bernoulli_uniform: u < p  <=>  rng.bernoulli(p) on twin generators, and both advance by two words."""
    a = host.dqmc_host_rng_create(31337); b = host.dqmc_host_rng_create(31337)
    try:
        ps = np.random.default_rng(3).random(4000); ps[:4] = [0.0, 1.0, 1e-300, 1.0 - 1e-16]
        for p in ps:
            assert bool(host.dqmc_host_rng_bernoulli(a, float(p))) == (host.dqmc_host_rng_bernoulli_uniform(b) < p)
        assert host.dqmc_host_rng_next(a) == host.dqmc_host_rng_next(b)
    finally:
        host.dqmc_host_rng_destroy(a); host.dqmc_host_rng_destroy(b)
