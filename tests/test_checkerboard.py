"""Checkerboard break-up of exp(-dtau K) (the reference's README.md:40 lists it as future work).

CPU side: the bond groups, the oracle's pair-by-pair restatement against plain numpy products of dense 2x2-block
matrices, the Trotter order of the break-up, and the oracle's checkerboard sweep against its own dense-GEMM sweep run
on the same E / E^-1 (the two differ only in summation order)."""
import numpy as np
import pytest

from dqmc_amd import DqmcError
from dqmc_amd.model import CONFIGS, HubbardModel, build_K, checkerboard_groups, lattice_bonds


@pytest.mark.parametrize("L1,L2", [(4, 4), (6, 4), (3, 3), (5, 4), (2, 2), (2, 4), (16, 16), (24, 24)])
def test_groups_partition_the_bonds_of_K(L1, L2):
    groups = checkerboard_groups(L1, L2)
    K = np.zeros((L1 * L2,) * 2)
    for g in groups:
        sites = [x for b in g for x in b]
        assert len(sites) == len(set(sites)), "bonds of one group must be disjoint"
        for i, j in g:
            K[i, j] -= 1.0; K[j, i] -= 1.0
    assert np.array_equal(K, build_K(L1, L2, 1.0, 0.0))             # every bond of source/model.cpp:39-60 exactly once
    assert sum(len(g) for g in groups) == len(lattice_bonds(L1, L2))
    if L1 % 2 == 0 and L2 % 2 == 0 and L1 >= 4 and L2 >= 4:
        assert len(groups) == 4 and all(len(g) == L1 * L2 // 2 for g in groups)


@pytest.mark.parametrize("L1,L2", [(4, 4), (6, 4), (3, 3), (5, 4), (2, 2), (2, 4), (16, 16)])
def test_host_facade_builds_the_same_groups(L1, L2):
    """AttractiveHubbard::checkerboard_groups (dqmc_amd/host/dqmc_host.hpp) against dqmc_amd.model.checkerboard_groups."""
    import ctypes as C
    import dqmc_amd
    h = C.CDLL(dqmc_amd.HOST_LIB_PATH)
    h.dqmc_host_checkerboard_groups.argtypes = [C.c_char_p, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_char_p, C.c_int]
    ini = f"[Lattice]\nL1 = {L1}\nL2 = {L2}\n[hubbard]\nU = 4.0\nt = 1.3\nmu = -0.2\n[simulation]\nbeta = 2.0\nnt = 20\nn_stab = 10\n".encode()
    bonds = np.zeros((4 * L1 * L2, 2), np.int32); sizes = np.zeros(16, np.int32); par = np.zeros(3); err = C.create_string_buffer(256)
    G = h.dqmc_host_checkerboard_groups(ini, 2.0, bonds.ctypes.data, len(bonds), sizes.ctypes.data, 16, par.ctypes.data, err, 256)
    assert G > 0, err.value
    m = HubbardModel(L1, L2, 4.0, 2.0, 20, t=1.3, mu=-0.2)
    groups, c, s, f = m.checkerboard()
    assert G == len(groups) and list(sizes[:G]) == [len(g) for g in groups]
    assert [tuple(b) for b in bonds[:sizes[:G].sum()]] == [tuple(b) for g in groups for b in g]
    assert np.allclose(par, [c, s, f], rtol=1e-15, atol=0)


def test_single_row_lattice_is_refused():
    with pytest.raises(ValueError):
        checkerboard_groups(1, 4)


def test_break_up_is_second_order_in_dtau():
    """|E - exp(-dtau K)| shrinks 4x when dtau halves (first-order break-up: local error O(dtau^2)); E E^-1 = 1 exactly."""
    errs = []
    for nt in (20, 40, 80):
        m = HubbardModel(6, 6, 4.0, 2.0, nt)
        E, Ei = m.checkerboard_expK(), m.checkerboard_expK(inverse=True)
        assert np.abs(E @ Ei - np.eye(m.n)).max() < 1e-14
        errs.append(np.abs(E - m.expK).max())
    assert 3.5 < errs[0] / errs[1] < 4.5 and 3.5 < errs[1] / errs[2] < 4.5
    assert errs[0] < 0.02


def _cb_engine(lib, m):
    e = m.engine(lib)
    e.set_checkerboard(*m.checkerboard())
    return e


@pytest.mark.parametrize("shape", [(4, 4), (6, 4), (3, 3), (5, 4)])
def test_oracle_pairwise_products_match_dense_numpy(orc, shape):
    m = HubbardModel(shape[0], shape[1], 4.0, 2.0, 20)
    E, Ei = m.checkerboard_expK(), m.checkerboard_expK(inverse=True)
    f = m.random_fields(3); rng = np.random.default_rng(4)
    e = _cb_engine(orc, m); e.set_fields(f)
    ev = lambda l, s=1.0: np.exp(s * m.g * m.eta[f[l]])
    ref = np.eye(m.n)
    for l in range(10):
        ref = ev(l)[:, None] * (E @ ref)                              # source/dqmc.cpp:88-105 with expK -> E
    assert np.abs(e.calculate_Bbar(0) - ref).max() < 1e-12 * np.abs(ref).max()
    G = rng.standard_normal((m.n, m.n))
    for l in (0, 7):
        e.set_G(G); e.wrap_forward(l)
        want = (ev(l)[:, None] * E) @ G @ (Ei * ev(l, -1.0)[None, :])     # source/dqmc.cpp:113-132
        assert np.abs(e.get_G() - want).max() < 1e-12 * np.abs(want).max()
        e.set_G(G); e.wrap_backward(l)
        want = (Ei * ev(l, -1.0)[None, :]) @ G @ (ev(l)[:, None] * E)     # source/dqmc.cpp:169-187
        assert np.abs(e.get_G() - want).max() < 1e-12 * np.abs(want).max()


@pytest.mark.parametrize("cfg", ["cfg1", "cfg2"])
def test_oracle_checkerboard_sweep_equals_dense_sweep_on_the_same_E(orc, cfg):
    """The pair-by-pair path and the reference's dense-GEMM path (expK := E, invexpK := E^-1) are the same Markov chain."""
    m = HubbardModel(**CONFIGS[cfg]); rng = np.random.default_rng(12)
    f0 = m.random_fields(5); sf, sb = m.random_stream(rng), m.random_stream(rng)
    a = _cb_engine(orc, m)
    b = orc.engine(m.n, m.nt, m.n_stab, m.g, m.gamma, m.eta, m.checkerboard_expK(), m.checkerboard_expK(inverse=True))
    for e in (a, b):
        e.set_fields(f0); e.init()
    assert np.abs(a.get_G() - b.get_G()).max() < 1e-11
    for e in (a, b):
        e.sweep_0_to_beta(*sf); e.sweep_beta_to_0(*sb)
    assert np.array_equal(a.get_fields(), b.get_fields())
    assert np.abs(a.get_G() - b.get_G()).max() < 1e-10
    assert a.stats().n_accepted == b.stats().n_accepted and a.stats().max_err < 1e-6
    # and it is a different discretisation from the dense exponential (|E - expK| ~ 1e-3 at cfg 2): same fields in, another G
    c = m.engine(orc); c.set_fields(f0); c.init()
    a.set_fields(f0); a.init()
    d = np.abs(a.get_G() - c.get_G()).max()
    assert d > 1e-6 if cfg == "cfg2" else d < 1e-10                  # 4x4: the four groups commute, the break-up is exact


def test_oracle_rejects_malformed_bond_lists(orc):
    m = HubbardModel(**CONFIGS["cfg1"]); e = m.engine(orc)
    groups, c, s, f = m.checkerboard()
    for bad in ([[(0, 1), (1, 2)]], [[(0, 16)]], [[(3, 3)]], [[(-1, 2)]]):
        with pytest.raises(DqmcError):
            e.set_checkerboard(bad, c, s, f)
    with pytest.raises(DqmcError):
        e.set_checkerboard(groups, c, s, 0.0)
    e.set_checkerboard(groups, c, s, f)
