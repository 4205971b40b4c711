"""Checkerboard break-up on the device (dqmc_set_checkerboard, checkerboard.hip) against the oracle's pair-by-pair
restatement, through the C ABI.  Tolerance as everywhere: max|dG| <= 1e-10 * max(1, max|G|); fields and accepted counts
bit-exact."""
import numpy as np
import pytest

from dqmc_amd import DqmcError
from dqmc_amd.model import CONFIGS, HubbardModel

pytestmark = pytest.mark.gpu

TOL = 1e-10


def close(a, b, tol=TOL):
    return np.abs(np.asarray(a) - np.asarray(b)).max() <= tol * max(1.0, np.abs(np.asarray(b)).max())


def cb_pair(hip, orc, m, fields):
    out = []
    for lib in (hip, orc):
        e = m.engine(lib); e.set_checkerboard(*m.checkerboard()); e.set_fields(fields); out.append(e)
    return out


@pytest.mark.parametrize("shape", [(4, 4), (6, 4), (3, 3), (5, 4), (16, 16), (20, 16), (24, 24), (32, 32)])
def test_wraps_and_bbar(hip, orc, shape):
    """One wrap each way and a B-bar product: odd rings (groups that leave sites out), N = 256 (transposed copy written by the
    wrap), N = 320 / 576 / 1024 (the narrower column strips)."""
    m = HubbardModel(shape[0], shape[1], 4.0, 2.0, 20)
    f = m.random_fields(3); rng = np.random.default_rng(5)
    e, o = cb_pair(hip, orc, m, f)
    G = rng.standard_normal((m.n, m.n))
    for l in (0, 13):
        for op in ("wrap_forward", "wrap_backward"):
            e.set_G(G); o.set_G(G)
            getattr(e, op)(l); getattr(o, op)(l)
            assert close(e.get_G(), o.get_G(), 1e-12), (op, l)
    for i in range(2):
        a, b = e.calculate_Bbar(i), o.calculate_Bbar(i)
        assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max()


@pytest.mark.parametrize("cfg", ["cfg1", "cfg2", "cfg3"])
def test_init_and_sweep_parity(hip, orc, cfg):
    m = HubbardModel(**CONFIGS[cfg]); rng = np.random.default_rng(21)
    f0 = m.random_fields(6); sf, sb = m.random_stream(rng), m.random_stream(rng)
    e, o = cb_pair(hip, orc, m, f0)
    e.init(); o.init()
    assert close(e.get_G(), o.get_G()) and abs(e.get_logdet() - o.get_logdet()) < 1e-8 * max(1.0, abs(o.get_logdet()))
    for i in (0, m.n_stack - 1):
        if m.n >= 64:                                      # panel-pivoted to_LDR (qr_panel.hip): another pivot order, the same product
            (L, d, R), (Lo, do, Ro) = e.get_stack(i), o.get_stack(i)
            assert close((L * d[None, :]) @ R, (Lo * do[None, :]) @ Ro, 1e-9)
        else:
            for a, b in zip(e.get_stack(i), o.get_stack(i)):
                assert close(a, b, 1e-9)
    e.sweep_0_to_beta(*sf); o.sweep_0_to_beta(*sf)
    assert np.array_equal(e.get_fields(), o.get_fields()) and close(e.get_G(), o.get_G())
    e.sweep_beta_to_0(*sb); o.sweep_beta_to_0(*sb)
    assert np.array_equal(e.get_fields(), o.get_fields()) and close(e.get_G(), o.get_G())
    se, so = e.stats(), o.stats()
    # wrap error between stabilisations (the reference's 1e-6 alarm, source/dqmc.cpp:390): i.i.d. fields at cfg 3 sit right at it in any
    # evaluation (two numpy runs of the dgeqp3 route: 4.9e-7 and 8.9e-7), and it moves by tens of percent with the pivot order of the
    # factorisations in between (panel-pivoted to_LDR on the device) -- hold the device to the oracle's own figure, not to the constant
    assert se.n_accepted == so.n_accepted and so.max_err < 1e-6 and se.max_err < max(1e-6, 3 * so.max_err)
    assert abs(e.global_action() - o.global_action()) < 1e-7 * max(1.0, abs(o.global_action()))


def test_device_path_equals_dense_gemm_path_on_the_same_E(hip):
    """Pair kernels vs the fp64-MFMA GEMMs given E / E^-1 as dense matrices: the same chain (fields, accepted counts), G to 1e-10."""
    m = HubbardModel(**CONFIGS["cfg2"]); rng = np.random.default_rng(22)
    f0 = m.random_fields(7); sf, sb = m.random_stream(rng), m.random_stream(rng)
    a = m.engine(hip); a.set_checkerboard(*m.checkerboard())
    b = hip.engine(m.n, m.nt, m.n_stab, m.g, m.gamma, m.eta, m.checkerboard_expK(), m.checkerboard_expK(inverse=True))
    for e in (a, b):
        e.set_fields(f0); e.init(); e.sweep_0_to_beta(*sf); e.sweep_beta_to_0(*sb)
    assert np.array_equal(a.get_fields(), b.get_fields()) and a.stats().n_accepted == b.stats().n_accepted
    assert close(a.get_G(), b.get_G())


def test_large_lattice_sweep(hip, orc):
    """24x24 (sub-matrix slice kernel, cooperative QRCP, no register walk) with the pair kernels in the wraps and B-bar chains."""
    m = HubbardModel(24, 24, 4.0, 2.0, 20); rng = np.random.default_rng(23)
    f0 = m.random_fields(8); sf = m.random_stream(rng)
    e, o = cb_pair(hip, orc, m, f0)
    e.init(); o.init()
    assert close(e.get_G(), o.get_G())
    e.sweep_0_to_beta(*sf); o.sweep_0_to_beta(*sf)
    assert np.array_equal(e.get_fields(), o.get_fields()) and close(e.get_G(), o.get_G())


def test_batched_engine_with_one_dtau_per_chain(hip, orc):
    """Three chains at three inverse temperatures in one engine: cosh / sinh / diag_factor per chain."""
    betas = [2.0, 3.0, 4.0]; C = len(betas)
    ms = [HubbardModel(8, 8, 4.0, b, 40) for b in betas]
    m0 = ms[0]; rng = np.random.default_rng(24)
    fs = np.stack([m.random_fields(30 + c) for c, m in enumerate(ms)])
    eb = hip.engine(m0.n, m0.nt, m0.n_stab, [m.g for m in ms], m0.gamma, m0.eta, np.stack([m.expK for m in ms]),
                    np.stack([m.invexpK for m in ms]), n_chains=C)
    groups = m0.checkerboard()[0]
    eb.set_checkerboard(groups, *[[m.checkerboard()[k] for m in ms] for k in (1, 2, 3)])
    eb.set_fields(fs); eb.init()
    streams = [m0.random_stream(rng) for _ in range(C)]
    s = tuple(np.stack([st[k] for st in streams]) for k in range(3))
    G0 = eb.get_G(); eb.sweep_0_to_beta(*s); G1 = eb.get_G(); f1 = eb.get_fields()
    for c, m in enumerate(ms):
        o = m.engine(orc); o.set_checkerboard(*m.checkerboard()); o.set_fields(fs[c]); o.init()
        assert close(G0[c], o.get_G())
        o.sweep_0_to_beta(*streams[c])
        assert np.array_equal(f1[c], o.get_fields()) and close(G1[c], o.get_G())


def test_unequal_time_series_uses_the_same_E(hip, orc):
    m = HubbardModel(**CONFIGS["cfg1"]); rng = np.random.default_rng(25)
    f0 = m.random_fields(9); sf, sb = m.random_stream(rng), m.random_stream(rng)
    e, o = cb_pair(hip, orc, m, f0)
    for x in (e, o):
        x.init(); x.sweep_0_to_beta(*sf); x.sweep_beta_to_0(*sb); x.sweep_unequal_time()
    for which in ("tt", "t0", "0t"):
        for l in (0, 7, m.nt):
            assert close(e.get_G_tau(which, l), o.get_G_tau(which, l), 1e-9), (which, l)


def test_switching_after_init_needs_a_new_init_and_bad_bonds_are_refused(hip):
    m = HubbardModel(**CONFIGS["cfg1"]); e = m.engine(hip); e.set_fields(m.random_fields(1)); e.init()
    groups, c, s, f = m.checkerboard()
    for bad in ([[(0, 1), (1, 2)]], [[(0, 16)]], [[(3, 3)]], [[(-1, 2)]]):
        with pytest.raises(DqmcError) as ei:
            e.set_checkerboard(bad, c, s, f)
        assert ei.value.code == -1                                    # DQMC_EINVAL
    with pytest.raises(DqmcError):
        e.set_checkerboard(groups, c, s, -1.0)
    e.get_stack(0)                                                    # refused calls left the engine as it was
    e.set_checkerboard(groups, c, s, f)
    rng = np.random.default_rng(2)
    with pytest.raises(DqmcError):
        e.sweep_0_to_beta(*m.random_stream(rng))                      # the stack belongs to the old propagator
    e.init(); e.sweep_0_to_beta(*m.random_stream(rng))


def test_host_facade_switch_reaches_the_engine(hip, orc):
    """[simulation] checkerboard = true in the INI: the facade's DQMC calls dqmc_set_checkerboard with the model's groups; G after
    init_stacks on the facade's own initial fields equals the oracle's checkerboard engine on the same fields (and not the dense one)."""
    from pt_twin import HostPT, ini_text, load_host
    L, nt = 6, 20
    ini = ini_text(L, 4.0, nt, 10, beta=2.0).replace("symmetric = false", "symmetric = false\ncheckerboard = true")
    h = load_host(); pt = HostPT(h, ini, [2.0], [5])
    try:
        got = pt.get(0, L * L, nt)
        m = HubbardModel(L, L, 4.0, 2.0, nt)
        o = m.engine(orc); o.set_checkerboard(*m.checkerboard()); o.set_fields(got["fields"]); o.init()
        assert close(got["G"], o.get_G())
        d = m.engine(orc); d.set_fields(got["fields"]); d.init()
        assert np.abs(got["G"] - d.get_G()).max() > 1e-6
        pt.sweeps(1)
        assert pt.max_err(0) < 1e-6
    finally:
        pt.close()
