"""CPU tests of the C++ host facade (dqmc_amd/host/dqmc_host.hpp) through
libdqmc_host.so: RNG stream semantics, INI reader, model
constants, replica pairing."""
import ctypes as C
import os

import numpy as np
import pytest

import dqmc_amd
from dqmc_amd import CONFIGS, HubbardModel
from dqmc_amd.replica import partner_rank


@pytest.fixture(scope="module")
def host():
    if not os.path.exists(dqmc_amd.HOST_LIB_PATH):
        pytest.fail("libdqmc_host.so missing: run `make` / __graft_entry__.build()")
    h = C.CDLL(dqmc_amd.HOST_LIB_PATH)
    h.dqmc_host_rng_create.restype = C.c_void_p; h.dqmc_host_rng_create.argtypes = [C.c_uint]
    h.dqmc_host_rng_destroy.argtypes = [C.c_void_p]
    h.dqmc_host_rng_next.restype = C.c_uint; h.dqmc_host_rng_next.argtypes = [C.c_void_p]
    for f in (h.dqmc_host_draw_slice, h.dqmc_host_draw_slice_literal):
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    h.dqmc_host_bernoulli_check.argtypes = [C.c_uint, C.c_int, C.c_void_p]
    h.dqmc_host_expm.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    h.dqmc_host_model.argtypes = [C.c_char_p, C.c_double, C.c_uint] + [C.c_void_p] * 8 + [C.c_char_p, C.c_int]
    h.dqmc_host_param.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_char_p, C.c_int]
    return h


def _draw(h, fn, rng, nv):
    perm = np.empty(nv, np.int32); k = np.empty(nv, np.uint8); u = np.empty(nv, np.float64)
    fn(rng, nv, perm.ctypes.data, k.ctypes.data, u.ctypes.data)
    return perm, k, u


@pytest.mark.parametrize("nv", [16, 64, 256, 37])
def test_slice_stream_matches_literal_reference_semantics(host, nv):
    a = host.dqmc_host_rng_create(4242 + nv); b = host.dqmc_host_rng_create(4242 + nv)
    try:
        for _ in range(25):
            pa, ka, ua = _draw(host, host.dqmc_host_draw_slice, a, nv)
            pb, kb, ub = _draw(host, host.dqmc_host_draw_slice_literal, b, nv)
            assert (pa == pb).all() and (ka == kb).all() and (ua == ub).all()
            assert sorted(pa.tolist()) == list(range(nv)) and ka.max() <= 2 and 0.0 <= ua.min() and ua.max() < 1.0
        # both generators advanced identically (by-value proposal copy does not advance the shared one)
        assert host.dqmc_host_rng_next(a) == host.dqmc_host_rng_next(b)
    finally:
        host.dqmc_host_rng_destroy(a); host.dqmc_host_rng_destroy(b)


def test_bernoulli_is_u_less_than_p_and_consumes_two_words(host):
    p = np.random.default_rng(0).random(5000); p[:3] = [0.0, 1.0, 0.5]
    assert host.dqmc_host_bernoulli_check(99, len(p), p.ctypes.data) == 0


def test_partner_rank_pairing():
    # source/update.cpp:34-45
    for world in (2, 4, 8):
        for attempt in range(1, 5):
            for r in range(world):
                p = partner_rank(r, world, attempt)
                assert 0 <= p < world and p != r and partner_rank(p, world, attempt) == r
    assert [partner_rank(r, 8, 2) for r in range(8)] == [1, 0, 3, 2, 5, 4, 7, 6]
    assert [partner_rank(r, 8, 1) for r in range(8)] == [7, 2, 1, 4, 3, 6, 5, 0]     # odd attempt wraps 0 <-> W-1


def test_host_capi_partner_rank(host):
    for world in (2, 8):
        for attempt in range(1, 4):
            assert [host.dqmc_host_partner_rank(r, world, attempt) for r in range(world)] == [partner_rank(r, world, attempt) for r in range(world)]


def test_expm_and_model_constants(host):
    ini = b"[Lattice]\nL1 = 8\nL2 = 8\n[hubbard]\nU = 4.0\nt = 1.0\nmu = -0.1\n[simulation]\nbeta = 4.0\nnt = 80\nn_stab = 10\n"
    ns = C.c_int(); nt = C.c_int(); g = C.c_double()
    eK = np.empty((64, 64), order="F"); iK = np.empty((64, 64), order="F"); f = np.empty((80, 64), dtype=np.int64, order="F")
    gam = np.empty(4); eta = np.empty(4); err = C.create_string_buffer(256)
    rc = host.dqmc_host_model(ini, 4.0, 5, C.byref(ns), C.byref(nt), C.byref(g), eK.ctypes.data, iK.ctypes.data, f.ctypes.data,
                              gam.ctypes.data, eta.ctypes.data, err, 256)
    assert rc == 0, err.value
    m = HubbardModel(**CONFIGS["cfg2"])
    assert ns.value == 64 and nt.value == 80 and abs(g.value - m.g) < 1e-15
    assert np.abs(eK - m.expK).max() < 1e-14 and np.abs(iK - m.invexpK).max() < 1e-14
    assert np.allclose(gam, m.gamma, rtol=0, atol=1e-16) and np.allclose(eta, m.eta, rtol=0, atol=1e-15)
    assert f.min() >= 0 and f.max() <= 3 and len(np.unique(f)) == 4
    # bad lattice -> exception text comes back
    rc = host.dqmc_host_model(b"[Lattice]\nL1 = 0\nL2 = 4\n[hubbard]\nU=1\nt=1\nmu=0\n[simulation]\nnt=10\n", 1.0, 1, C.byref(ns), C.byref(nt),
                              C.byref(g), None, None, None, None, None, err, 256)
    assert rc == -1 and b"Bad lattice" in err.value


def test_ini_reader(host):
    ini = (b"# comment\n; another\nglobalkey = 3\n[simulation]\nnt = 1_000   # inline\nbeta = 4.5 ; c\nsymmetric = Yes\n"
           b"name = \"quoted\"\n[ParallelTempering]\nenabled = off\nbetas = 5.0, 4.5 , 4.0,3.5\n")
    out = C.c_double(); err = C.create_string_buffer(256)
    def get(sec, key, kind):
        rc = host.dqmc_host_param(ini, sec, key, kind, C.byref(out), err, 256)
        return rc, out.value, err.value.decode()
    assert get(b"simulation", b"nt", 0)[:2] == (0, 1000.0)
    assert get(b"simulation", b"beta", 1)[:2] == (0, 4.5)
    assert get(b"simulation", b"symmetric", 2)[:2] == (0, 1.0)
    assert get(b"ParallelTempering", b"enabled", 2)[:2] == (0, 0.0)
    assert get(b"ParallelTempering", b"betas", 3)[:2] == (0, 4.0)
    assert get(b"global", b"globalkey", 0)[:2] == (0, 3.0)
    rc, _, msg = get(b"simulation", b"missing", 0); assert rc == -1 and "not found" in msg
    rc, _, msg = get(b"nosuch", b"nt", 0); assert rc == -1 and "Section" in msg
    rc, _, msg = get(b"simulation", b"name", 0); assert rc == -1 and "Cannot convert" in msg
