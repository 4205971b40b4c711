"""CPU tests: the oracle against analytic answers, the independent numpy/scipy
evaluation, its LAPACK back end and the committed golden vectors.  The
reference ships no tests or fixtures, so these are what
pins the oracle ("parity unpinned" by the reference itself)."""
import os

import numpy as np
import pytest

import golden_util

from dqmc_amd import CONFIGS, HubbardModel
from oracle.numpy_ref import NumpyChain, free_fermion_G
import oracle.numpy_ref as nr

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _ldr_to_mat(F):
    L, d, R = F
    return (L * d[None, :]) @ R


@pytest.mark.parametrize("L,beta,nt", [(4, 2.0, 20), (6, 4.0, 40), (8, 4.0, 80)])
def test_free_fermions_known_answer(orc, L, beta, nt):
    # U = 0 => B_l = exp(-dtau K) for any field => G(0,0) = (I + exp(-beta K))^-1 analytically
    m = HubbardModel(L1=L, L2=L, U=0.0, beta=beta, nt=nt)
    e = m.engine(orc); e.set_fields(m.random_fields(3)); e.init()
    G, ld = free_fermion_G(m)
    assert np.abs(e.get_G() - G).max() < 1e-13
    assert abs(e.get_logdet() - ld) < 1e-10


@pytest.mark.parametrize("n", [16, 36, 64])
def test_ldr_algebra_vs_numpy(orc, n):
    rng = np.random.default_rng(n)
    M1 = rng.standard_normal((n, n)) * np.exp(rng.uniform(-6, 6, n))[None, :]
    M2 = rng.standard_normal((n, n)) * np.exp(rng.uniform(-6, 6, n))[:, None]
    F1 = orc.to_ldr(M1); F1n = nr.to_ldr(M1)
    assert np.allclose(_ldr_to_mat(F1), M1, rtol=1e-12, atol=1e-12 * np.abs(M1).max())
    assert np.allclose(F1[1], F1n[1], rtol=1e-11)                       # d (pivot order identical)
    assert np.abs(F1[0].T @ F1[0] - np.eye(n)).max() < 1e-13            # L orthogonal
    F2 = orc.to_ldr(M2)
    for got, ref, exact in [(orc.ldr_mul_mat(F1, M2), nr.ldr_mul_mat(F1, M2), M1 @ M2),
                            (orc.mat_mul_ldr(M2, F1), nr.mat_mul_ldr(M2, F1), M2 @ M1),
                            (orc.ldr_mul_ldr(F1, F2), nr.ldr_mul_ldr(F1, F2), M1 @ M2)]:
        scale = np.abs(exact).max()
        assert np.abs(_ldr_to_mat(got) - exact).max() < 1e-11 * scale
        assert np.abs(_ldr_to_mat(got) - _ldr_to_mat(ref)).max() < 1e-11 * scale
    G, ld = orc.inv_I_plus_ldr(F1); Gn, ldn = nr.inv_I_plus_ldr(F1)
    assert np.abs(G - Gn).max() < 1e-10 * max(1.0, np.abs(Gn).max()) and abs(ld - ldn) < 1e-9
    G2 = orc.inv_I_plus_ldr_mul_ldr(F1, F2); G2n = nr.inv_I_plus_ldr_mul_ldr(F1, F2)
    assert np.abs(G2 - G2n).max() < 1e-9 * max(1.0, np.abs(G2n).max())


@pytest.mark.parametrize("cfg", ["cfg1", "cfg2"])
def test_sweep_vs_numpy_chain(orc, cfg):
    m = HubbardModel(**CONFIGS[cfg]); f = m.random_fields(7)
    e = m.engine(orc); e.set_fields(f); e.init()
    c = NumpyChain(m, f); c.init()
    assert np.abs(e.get_G() - c.G).max() < 1e-11 and abs(e.get_logdet() - c.logdet) < 1e-10
    rng = np.random.default_rng(3)
    for _ in range(2):
        s1, s2 = m.random_stream(rng), m.random_stream(rng)
        e.sweep_0_to_beta(*s1); e.sweep_beta_to_0(*s2)
        c.sweep_fwd(*s1); c.sweep_bwd(*s2)
        assert (e.get_fields() == c.f).all()
        assert np.abs(e.get_G() - c.G).max() < 1e-11
    st = e.stats()
    assert st.n_accepted == c.n_acc and st.n_proposed == 4 * m.nt * m.n
    assert abs(st.max_err - max(c.errs)) < 0.5 * max(c.errs) + 1e-12
    assert abs(e.global_action() - c.global_action()) < 1e-8


def test_lapack_backend_agrees(orc):
    if not orc.set_backend("lapack"):
        pytest.skip("no LAPACK library on this host")
    try:
        m = HubbardModel(**CONFIGS["cfg2"]); f = m.random_fields(11)
        e = m.engine(orc); e.set_fields(f); e.init(); Gl = e.get_G(); ldl = e.get_logdet()
    finally:
        orc.set_backend("builtin")
    e2 = m.engine(orc); e2.set_fields(f); e2.init()
    assert np.abs(Gl - e2.get_G()).max() < 1e-11 and abs(ldl - e2.get_logdet()) < 1e-10


def test_identities(orc):
    # SM vs recompute, ratio vs delta-logdet, G(beta,beta)=G(0,0)
    m = HubbardModel(**CONFIGS["cfg2"]); f = m.random_fields(5)
    e = m.engine(orc); e.set_fields(f); e.init(); G0 = e.get_G(); ld0 = e.get_logdet()
    # one flip at slice 0 site i: update acts on Gtt[1] = B_0 G B_0^-1
    e.wrap_forward(0); G1 = e.get_G()
    i = 5; old = int(f[0, i]); new = (old + 1) % 4
    delta = np.exp(m.g * (m.eta[new] - m.eta[old])) - 1.0
    Gsm = orc.rank1_update(G1, i, delta)
    f2 = f.copy(); f2[0, i] = new
    e2 = m.engine(orc); e2.set_fields(f2); e2.init(); e2.wrap_forward(0)
    assert np.abs(Gsm - e2.get_G()).max() < 1e-9 * max(1.0, np.abs(Gsm).max())
    assert abs(np.log(abs(1.0 + (1.0 - G1[i, i]) * delta)) - (e2.get_logdet() - ld0)) < 1e-9
    # a full forward pass with every proposal rejected (u = 1) returns to G(0,0)
    rng = np.random.default_rng(0); perm, k, u = m.random_stream(rng); u[:] = 1.0
    e.set_G(G0); e.sweep_0_to_beta(perm, k, u)
    assert np.abs(e.get_G() - G0).max() < 1e-9 * max(1.0, np.abs(G0).max())
    assert e.stats().n_accepted == 0 and e.stats().max_err < 1e-6


def test_stack_index_errors(orc):
    from dqmc_amd import DqmcError
    m = HubbardModel(**CONFIGS["cfg1"]); e = m.engine(orc); e.set_fields(m.random_fields(1)); e.init()
    assert e.n_stack() == 2
    with pytest.raises(DqmcError):
        e.get_stack(2)                      # LDRStack::operator[] std::out_of_range (include/stackngf.h:61)
    with pytest.raises(DqmcError):
        e.wrap_forward(m.nt)


def test_short_last_block(orc):
    # nt % n_stab != 0 (source/dqmc.cpp:13-18)
    m = HubbardModel(L1=4, L2=4, U=4.0, beta=2.3, nt=23, n_stab=10); f = m.random_fields(2)
    e = m.engine(orc); e.set_fields(f); e.init(); c = NumpyChain(m, f); c.init()
    assert e.n_stack() == 3
    rng = np.random.default_rng(1); s1, s2 = m.random_stream(rng), m.random_stream(rng)
    e.sweep_0_to_beta(*s1); e.sweep_beta_to_0(*s2); c.sweep_fwd(*s1); c.sweep_bwd(*s2)
    assert (e.get_fields() == c.f).all() and np.abs(e.get_G() - c.G).max() < 1e-10


@pytest.mark.parametrize("name", golden_util.NAMES)
def test_golden_vectors(orc, name):
    """Committed fixtures (tests/golden/make_golden.py): inputs + outputs of the
    numpy/scipy evaluation; the oracle must reproduce them (both dense back ends)."""
    z, m, streams = golden_util.load(name)
    backends = ["builtin"] + (["lapack"] if orc.set_backend("lapack") else [])
    try:
        for be in backends:
            orc.set_backend(be)
            e = m.engine(orc); e.set_fields(z["fields"]); e.init()
            err, scale = golden_util.g0_error(z, e.get_G())
            assert err < float(z["tol"]) * scale, (be, err, scale)
            assert abs(e.get_logdet() - float(z["logdet"])) < 1e-8 * max(1.0, abs(float(z["logdet"])))
            if streams is not None and (m.n <= 64 or be == "lapack" or len(backends) == 1):      # the built-in kernels take ~25 s for a cfg-3 sweep
                e.sweep_0_to_beta(*streams[0]); e.sweep_beta_to_0(*streams[1])
                assert (e.get_fields() == z["fields_after"]).all()
                errA = np.abs(e.get_G() - z["G_after"]).max(); scaleA = max(1.0, np.abs(z["G_after"]).max())
                print(f"{name} [{be}]: init {err:.2e} / {scale:.2e}, after sweep {errA:.2e} / {scaleA:.2e}")
                assert errA < float(z["tol"]) * scaleA, (be, errA, scaleA)
                assert e.stats().n_accepted == int(z["n_accepted"])
            e.close()
    finally:
        orc.set_backend("builtin")


# ---- equal-time observables -------------------------------------------------------------
def test_observables_free_fermions_analytic(orc):
    """U = 0: density, double occupancy and s-wave pairing have closed forms in k space
    (n_k = 1/(e^{beta eps_k}+1), eps_k = -2t(cos kx + cos ky) - mu): density = (2/N) sum n_k, doubleOcc = (density/2)^2,
    swave = (1/N) sum n_k^2 -- independent of how G was built."""
    L1, L2, beta = 6, 4, 3.0
    m = HubbardModel(L1=L1, L2=L2, U=0.0, beta=beta, nt=30, n_stab=10)
    e = m.engine(orc); e.set_fields(m.random_fields(1)); e.init()
    sc, chi = e.measure_equal_time(L1, L2)
    kx = 2 * np.pi * np.arange(L1) / L1; ky = 2 * np.pi * np.arange(L2) / L2
    eps = -2.0 * m.t * (np.cos(kx)[:, None] + np.cos(ky)[None, :]) - m.mu
    nk = 1.0 / (np.exp(beta * eps) + 1.0)
    dens = 2.0 * nk.mean()
    assert abs(sc[0] - dens) < 1e-12 and abs(sc[1] - (dens / 2) ** 2) < 1e-12 and abs(sc[2] - (nk ** 2).mean()) < 1e-12
    # translation invariance: the displacement-space sum reproduces any single row of the site matrix
    sc2, chi2 = nr.equal_time_observables(e.get_G(), L1, L2)
    assert np.abs(chi - chi2).max() < 1e-13 and chi.shape == (L1, L2)


@pytest.mark.parametrize("cfg", ["cfg1", "cfg2"])
def test_observables_vs_numpy(orc, cfg):
    """The oracle keeps the reference's loops (source/model.cpp:167-288, include/measurementh5.h:13-66); numpy_ref evaluates the same
    formulas vectorised.  Also the bin accumulation (sum of per-sweep values, count, reset)."""
    m = HubbardModel(**CONFIGS[cfg]); rng = np.random.default_rng(3)
    e = m.engine(orc); e.set_fields(m.random_fields(9)); e.init()
    tot_s = np.zeros(3); tot_c = np.zeros((m.L1, m.L2))
    for _ in range(2):
        e.sweep_0_to_beta(*m.random_stream(rng)); e.sweep_beta_to_0(*m.random_stream(rng))
        sc, chi = e.measure_equal_time(m.L1, m.L2)
        sc2, chi2 = nr.equal_time_observables(e.get_G(), m.L1, m.L2)
        assert np.abs(sc - sc2).max() < 1e-12 * max(1.0, np.abs(sc2).max()) and np.abs(chi - chi2).max() < 1e-12 * max(1.0, np.abs(chi2).max())
        e.measure_accumulate(m.L1, m.L2); tot_s += sc; tot_c += chi
    s_sum, c_sum, cnt = e.measure_fetch(m.L1, m.L2, reset=True)
    assert cnt == 2 and np.abs(s_sum - tot_s).max() < 1e-12 and np.abs(c_sum - tot_c).max() < 1e-12
    assert e.measure_fetch(m.L1, m.L2)[2] == 0
    from dqmc_amd import DqmcError
    with pytest.raises(DqmcError):
        e.measure_equal_time(m.L1 + 1, m.L2)


# ---- unequal-time path --------------------------------------------------------------------
@pytest.mark.parametrize("cfg", ["cfg1", "cfg2"])
def test_unequal_time_oracle(orc, cfg):
    """sweep_unequalTime (source/dqmc.cpp:458-515): boundary identities G(beta,beta) = G(0,0), G(beta,0) = I - G(0,0),
    G(0,beta) = -G(0,0); inside the first block the series equal the short direct products B_{l-1}..B_0 G(0,0) etc.; at the first
    stabilisation point the stabilised values equal an independent numpy evaluation of stablelinalg::inv_invldr_plus_ldr /
    inv_I_plus_ldr_mul_ldr on numpy-built factors; the wrap errors land in the statistics."""
    m = HubbardModel(**CONFIGS[cfg]); rng = np.random.default_rng(2)
    e = m.engine(orc); e.set_fields(m.random_fields(4)); e.init()
    e.sweep_0_to_beta(*m.random_stream(rng)); e.sweep_beta_to_0(*m.random_stream(rng))
    G0 = e.get_G(); f = e.get_fields(); n_err0 = e.stats().n_err
    e.sweep_unequal_time()
    st = e.stats()
    assert st.n_err == n_err0 + 3 * e.n_stack() and st.max_err < 1e-6
    n = m.n; I = np.eye(n)
    assert np.abs(e.get_G_tau("tt", 0) - G0).max() == 0 and np.abs(e.get_G_tau("0t", 0) - (G0 - I)).max() == 0
    assert np.abs(e.get_G_tau("tt", m.nt) - G0).max() < 1e-9 and np.abs(e.get_G_tau("t0", m.nt) - (I - G0)).max() < 1e-9
    assert np.abs(e.get_G_tau("0t", m.nt) + e.get_G_tau("tt", m.nt)).max() == 0
    Bs = [np.diag(np.exp(m.g * m.eta[f[l]])) @ m.expK for l in range(m.nt)]
    P = I.copy(); Pinv = I.copy()
    for l in range(1, m.n_stab):                     # un-stabilised part of the first block: short, well-conditioned products
        P = Bs[l - 1] @ P; Pinv = Pinv @ np.linalg.inv(Bs[l - 1])
        assert np.abs(e.get_G_tau("t0", l) - P @ G0).max() < 1e-10
        assert np.abs(e.get_G_tau("0t", l) - (G0 - I) @ Pinv).max() < 1e-10 * max(1.0, np.abs(Pinv).max())
    Bbar0 = I.copy()
    for l in range(m.n_stab): Bbar0 = Bs[l] @ Bbar0
    Bt0 = nr.to_ldr(Bbar0); Bbt = e.get_stack(1)
    l1 = m.n_stab
    assert np.abs(e.get_G_tau("t0", l1) - nr.inv_invldr_plus_ldr(Bt0, Bbt)).max() < 1e-10
    assert np.abs(e.get_G_tau("0t", l1) + nr.inv_invldr_plus_ldr(Bbt, Bt0)).max() < 1e-10
    assert np.abs(e.get_G_tau("tt", l1) - nr.inv_I_plus_ldr_mul_ldr(Bt0, Bbt)).max() < 1e-10
    from dqmc_amd import DqmcError
    with pytest.raises(DqmcError):
        e.get_G_tau("tt", m.nt + 1)


def test_dynamical_observables_vs_numpy(orc):
    """greenTau / doublonTau / currxxTau (source/model.cpp:290-394) per slice in displacement space: the oracle's loops against the
    vectorised numpy evaluation on a non-square lattice; bin accumulation."""
    L1, L2 = 6, 4
    m = HubbardModel(L1=L1, L2=L2, U=4.0, beta=2.0, nt=20, n_stab=10); rng = np.random.default_rng(1)
    e = m.engine(orc); e.set_fields(m.random_fields(5)); e.init()
    e.sweep_0_to_beta(*m.random_stream(rng)); e.sweep_beta_to_0(*m.random_stream(rng)); e.sweep_unequal_time()
    cube = e.measure_unequal_time(L1, L2)
    G = [np.stack([e.get_G_tau(w, l) for l in range(m.nt + 1)]) for w in ("tt", "t0", "0t")]
    ref = nr.unequal_time_observables(G[0], G[1], G[2], L1, L2)
    assert cube.shape == (3, m.nt + 1, L1, L2) and np.abs(cube - ref).max() < 1e-13 * max(1.0, np.abs(ref).max())
    # tau = 0: greenTau(r) = 2 G(0,0) averaged over translations
    assert abs(cube[0, 0, L1 // 2 - 1, L2 // 2 - 1] - 2.0 * np.trace(G[0][0]) / m.n) < 1e-13
    e.measure_unequal_time(L1, L2, accumulate=True); e.measure_unequal_time(L1, L2, accumulate=True)
    tot, cnt = e.measure_unequal_fetch(L1, L2)
    assert cnt == 2 and np.abs(tot - 2 * cube).max() < 1e-12 and e.measure_unequal_fetch(L1, L2)[1] == 0


def test_half_warp_oracle(orc):
    """DQMC::half_warp (source/dqmc.cpp:288-315): invexpK_half @ M @ expK_half for the equal-time G and for slices of the unequal-time
    series, against plain numpy; exp(-dtau K / 2) squared is exp(-dtau K), so a half warp applied twice is a whole one."""
    from dqmc_amd.model import expm_sym
    m = HubbardModel(**CONFIGS["cfg1"]); rng = np.random.default_rng(8)
    eh, ieh = expm_sym(-0.5 * m.dtau * m.K), expm_sym(0.5 * m.dtau * m.K)
    assert np.abs(eh @ eh - m.expK).max() < 1e-14
    e = m.engine(orc); e.set_fields(m.random_fields(3)); e.init()
    e.sweep_0_to_beta(*m.random_stream(rng)); e.sweep_beta_to_0(*m.random_stream(rng))
    from dqmc_amd import DqmcError
    with pytest.raises(DqmcError):
        e.half_warp(eh, ieh, "tt", 0)                      # before sweep_unequal_time
    G0 = e.get_G()
    assert np.abs(e.half_warp(eh, ieh) - ieh @ G0 @ eh).max() < 1e-13
    e.sweep_unequal_time()
    for which in ("tt", "t0", "0t"):
        for l in (0, 7, m.nt):
            assert np.abs(e.half_warp(None, None, which, l) - ieh @ e.get_G_tau(which, l) @ eh).max() < 1e-13      # matrices kept from the first call
    assert np.abs(e.get_G() - G0).max() == 0               # the engine's own G is untouched
    with pytest.raises(DqmcError):
        e.half_warp(eh, None)

