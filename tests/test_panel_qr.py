"""The numpy statements of the panel-pivoted factorisations (oracle/panel_qr.py: test infrastructure, the algorithm
dqmc_amd/csrc/qr_panel.hip implements is `qr_sketch(b=16, p=16, sign=True, local_pivot=False)`): every variant is a valid
factorisation M[:, P] = Q R0, the schemes that see the dependencies between the columns are rank-revealing on graded matrices,
the one that does not (largest norms) is not, and the sketching matrix is the fixed function of the row index the device
generates in registers."""
import numpy as np
import pytest

from oracle import panel_qr as pq


def graded(n, kind, seed, span=28.0):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, n)); s1 = np.exp(rng.uniform(-span, span, n)); s2 = np.exp(rng.uniform(-span, span, n))
    return X * s1[None, :] if kind == "columns" else (s1[:, None] * X if kind == "rows" else s1[:, None] * X * s2[None, :])


VARIANTS = {
    "tournament": lambda M: pq.qr_tournament(M, 16),
    "sketch_gauss_local": lambda M: pq.qr_sketch(M, 16, 8),
    "sketch_device": lambda M: pq.qr_sketch(M, 16, 16, local_pivot=False, sign=True),
    "lookahead_guarded": lambda M: pq.qr_sketch_lookahead(M, 16, 48, guard=1e-10),
}


@pytest.mark.parametrize("kind", ["columns", "rows", "both"])
@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_panel_pivoted_factorisations_are_rank_revealing(name, kind):
    n = 96
    M = graded(n, kind, 11)
    Q, R0, P = VARIANTS[name](M)
    assert sorted(P.tolist()) == list(range(n))
    assert np.abs(Q.T @ Q - np.eye(n)).max() < 1e-13
    assert (np.abs(Q @ R0 - M[:, P]).max(axis=0) / np.abs(M[:, P]).max(axis=0)).max() < 1e-12      # column by column
    assert np.abs(np.tril(R0, -1)).max() == 0.0
    up, grade = pq.quality(R0)
    assert up <= 8.0 and grade <= 8.0, (name, kind, up, grade)


def test_largest_norm_panels_are_not_rank_revealing():
    """The control: choosing the b columns of largest norm per panel does not see that they may depend on each other.  On a matrix whose
    16 largest columns span only 4 directions the factorisation is still valid, but a tiny diagonal entry precedes large ones."""
    rng = np.random.default_rng(3)
    n = 64
    M = 1e-3 * rng.standard_normal((n, n))
    M[:, :16] = rng.standard_normal((n, 4)) @ rng.standard_normal((4, 16)) + 1e-9 * rng.standard_normal((n, 16))
    Q, R0, P = pq.qr_normpanel(M, 16)
    assert np.abs(Q @ R0 - M[:, P]).max() < 1e-13
    assert pq.quality(R0)[1] > 1e3
    assert max(pq.quality(pq.qr_sketch(M, 16, 16, local_pivot=False, sign=True)[1])) <= 8.0


def test_sketching_matrix_is_the_device_hash():
    """Omega[i, r] = -1 when bit i of fmix32(r * 0x9E3779B1 + 0x85EBCA77) is set: qp_row_bits / qp_sign of qr_panel.hip, restated with
    Python integers here."""
    def bits(r):
        h = (r * 0x9E3779B1 + 0x85EBCA77) & 0xFFFFFFFF
        h ^= h >> 16; h = (h * 0x85EBCA6B) & 0xFFFFFFFF
        h ^= h >> 13; h = (h * 0xC2B2AE35) & 0xFFFFFFFF
        h ^= h >> 16
        return h
    Om = pq.omega_sign(32, 300)
    for r in (0, 1, 17, 255, 299):
        h = bits(r)
        assert [(-1.0 if (h >> i) & 1 else 1.0) for i in range(32)] == Om[:, r].tolist()
    assert abs(Om.mean()) < 0.05                                  # balanced signs
    assert np.abs(Om @ Om.T / 300 - np.eye(32)).max() < 0.25      # rows nearly orthogonal: a Johnson-Lindenstrauss sketch


def test_lookahead_guard_falls_back_instead_of_selecting_noise():
    """A matrix whose trailing part is 1e-18 of its first 16 columns: after eliminating the first panel in SKETCH space the residual sketch
    of every other column is rounding noise.  Without the guard the look-ahead picks by that noise; with it the next panel selects from
    its own fresh sketch and the factorisation keeps its grading."""
    rng = np.random.default_rng(8)
    n = 64
    U = np.linalg.qr(rng.standard_normal((n, n)))[0]
    M = U[:, :16] @ rng.standard_normal((16, n)) + 1e-18 * (U[:, 16:] * np.exp(rng.uniform(-3, 3, n - 16))[None, :]) @ rng.standard_normal((n - 16, n))
    stats = []
    Q, R0, P = pq.qr_sketch_lookahead(M, 16, 48, guard=1e-10, stats=stats)
    assert stats[0] == 0                                        # the first look-ahead was abandoned
    assert (np.abs(Q @ R0 - M[:, P]).max(axis=0) / np.abs(M[:, P]).max(axis=0)).max() < 1e-12
    d = np.abs(np.diag(R0))
    assert d[:16].min() > 1e10 * d[16:].max()                   # the numerical rank shows
    assert pq.quality(R0[16:, 16:])[1] <= 8.0                   # and the trailing block is graded by the fresh selections
