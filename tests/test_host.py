"""CPU tests of the C++ host facade (dqmc_amd/host/dqmc_host.hpp) through
libdqmc_host.so: RNG stream semantics, INI reader, model
constants, replica pairing."""
import ctypes as C
import os

import numpy as np
import pytest

import dqmc_amd
from dqmc_amd import CONFIGS, HubbardModel


def partner_rank(rank, world, attempt):
    return dqmc_amd.lib().partner_rank(rank, world, attempt)


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


# ---- on-disk results in the reference's HDF5 layout (dqmc_amd/host/results_h5.hpp) ------------------
def _h5_read(h, path, dataset):
    h.dqmc_host_results_read.restype = C.c_longlong
    h.dqmc_host_results_read.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_char_p, C.c_int]
    nd = C.c_int(0); dims = (C.c_ulonglong * 8)(); err = C.create_string_buffer(256)
    cnt = h.dqmc_host_results_read(path.encode(), dataset.encode(), C.byref(nd), dims, None, 0, err, 256)
    assert cnt >= 0, err.value
    data = np.empty(cnt, np.float64)
    assert h.dqmc_host_results_read(path.encode(), dataset.encode(), None, None, data.ctypes.data, cnt, err, 256) == cnt
    return data.reshape([dims[k] for k in range(nd.value)])


def _chi_k_numpy(chi_r, L1, L2):
    """transform::chi_r_to_chi_k (include/measurementh5.h:77-117) for the square lattice a1 = x, a2 = y:
    chi_r[t, dx_idx + L1 * dy_idx] -> chi_k[kx, ky, t], k = (qx * 2pi / L1, qy * 2pi / L2), q = idx - L/2 + 1."""
    n_tau = chi_r.shape[0]
    r = chi_r.reshape(n_tau, L2, L1)                       # [t, y_idx, x_idx]
    dx = np.arange(L1) - (L1 // 2 - 1); dy = np.arange(L2) - (L2 // 2 - 1)
    kx = (np.arange(L1) - L1 // 2 + 1) * 2 * np.pi / L1; ky = (np.arange(L2) - L2 // 2 + 1) * 2 * np.pi / L2
    wx = np.exp(-1j * np.outer(kx, dx)); wy = np.exp(-1j * np.outer(ky, dy))          # [kx, x], [ky, y]
    return np.einsum("ax,by,tyx->abt", wx, wy, r)


@pytest.mark.parametrize("L1,L2,nt", [(4, 4, 6), (6, 4, 5)])
def test_results_file_has_the_reference_layout(host, tmp_path, L1, L2, nt):
    """/bin_k/{scalar,equaltime,unequaltime} and /binK_k/{equaltime,unequaltime} with the reference's dataset names, dims and
    element order (include/measurementh5.h:277-362, include/h5utils.h:9-119): C-order [dx][dy][tau] cubes, complex as a
    trailing dimension of 2; the k-space data against an independent numpy evaluation of the transform."""
    h = host
    h.dqmc_host_results_write.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_int]
    rng = np.random.default_rng(L1 * 100 + L2)
    n_bins, n_tau, nb = 3, nt + 1, L1 * L2
    scalars = rng.normal(size=(n_bins, 3)); chi = rng.normal(size=(n_bins, nb)); ut = rng.normal(size=(n_bins, 3, n_tau, nb))
    err = C.create_string_buffer(256)
    rc = h.dqmc_host_results_write(str(tmp_path).encode(), 5, L1, L2, n_bins, scalars.ctypes.data, chi.ctypes.data, ut.ctypes.data, n_tau, err, 256)
    if rc != 0 and b"libhdf5 not found" in err.value:
        pytest.skip("no libhdf5 on this machine")
    assert rc == 0, err.value
    path = str(tmp_path / "data_5.h5")
    assert os.path.exists(path)
    for b in range(n_bins):
        for k, name in enumerate(("density", "doubleOcc", "swave")):
            v = _h5_read(h, path, f"/bin_{b}/scalar/{name}")
            assert v.shape == (1,) and v[0] == scalars[b, k]
        d = _h5_read(h, path, f"/bin_{b}/equaltime/densityCorr")
        assert d.shape == (L1, L2, 1)
        np.testing.assert_array_equal(d[:, :, 0], chi[b].reshape(L2, L1).T)              # [dx_idx][dy_idx]
        dk = _h5_read(h, path, f"/binK_{b}/equaltime/densityCorr")
        assert dk.shape == (L1, L2, 1, 2)
        ref = _chi_k_numpy(chi[b][None, :], L1, L2)
        np.testing.assert_allclose(dk[..., 0] + 1j * dk[..., 1], ref, atol=1e-12)
        for o, name in enumerate(("greenTau", "doublonTau", "currxxTau")):
            d = _h5_read(h, path, f"/bin_{b}/unequaltime/{name}")
            assert d.shape == (L1, L2, n_tau)
            np.testing.assert_array_equal(d, ut[b, o].reshape(n_tau, L2, L1).transpose(2, 1, 0))
            dk = _h5_read(h, path, f"/binK_{b}/unequaltime/{name}")
            assert dk.shape == (L1, L2, n_tau, 2)
            np.testing.assert_allclose(dk[..., 0] + 1j * dk[..., 1], _chi_k_numpy(ut[b, o], L1, L2), atol=1e-12)
    # the file is a regular HDF5 file other tools can open: h5dump (shipped with the library) lists the same groups
    import shutil, subprocess
    h5dump = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if os.path.exists("/opt/conda/bin/h5dump") else None)
    if h5dump:
        out = subprocess.run([h5dump, "-n", path], capture_output=True, text=True, timeout=60)
        assert out.returncode == 0, out.stderr
        for g in ("/bin_0/scalar/density", "/bin_2/unequaltime/currxxTau", "/binK_1/equaltime/densityCorr", "/binK_2/unequaltime/greenTau"):
            assert g in out.stdout


def test_results_k_space_of_a_delta_is_flat(host, tmp_path):
    """chi_r = delta at r = 0 transforms to chi_k = 1 for every k (known answer)."""
    h = host
    h.dqmc_host_results_write.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_int]
    L1 = L2 = 6
    chi = np.zeros((1, L1 * L2)); chi[0, (L1 // 2 - 1) + L1 * (L2 // 2 - 1)] = 1.0
    scalars = np.zeros((1, 3)); err = C.create_string_buffer(256)
    rc = h.dqmc_host_results_write(str(tmp_path).encode(), 0, L1, L2, 1, scalars.ctypes.data, chi.ctypes.data, None, 0, err, 256)
    if rc != 0 and b"libhdf5 not found" in err.value:
        pytest.skip("no libhdf5 on this machine")
    assert rc == 0, err.value
    dk = _h5_read(h, str(tmp_path / "data_0.h5"), "/binK_0/equaltime/densityCorr")
    np.testing.assert_allclose(dk[..., 0], 1.0, atol=1e-14); np.testing.assert_allclose(dk[..., 1], 0.0, atol=1e-14)
