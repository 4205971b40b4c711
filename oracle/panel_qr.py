"""Panel-pivoted (blocked) QR variants evaluated against dgeqp3 -- TEST INFRASTRUCTURE ONLY.

`stablelinalg::to_LDR` (source/stablelinalg.cpp:35-55) calls `arma::qr(Q, R, P, M, "vector")` = LAPACK dgeqp3: one
global pivot decision per column, i.e. N dependent chip-wide reductions.  The functions here restate factorisations
that take ONE global decision per PANEL of b columns and are used (a) by `scripts/eval_panel_qr.py` to decide whether a
panel-pivoted to_LDR holds the 1e-10 parity bar on the DQMC matrices, (b) as the numpy statement of the algorithm the
HIP kernels in dqmc_amd/csrc/qr_panel.hip implement (`qr_sketch`), for the unit tests.

Every function returns (Q, R0, P) with M[:, P] = Q @ R0 like scipy.linalg.qr(M, pivoting=True).

* qr_tournament  -- communication-avoiding RRQR (Demmel, Grigori, Gu, Xiang): the b pivots of a panel are the winners of
                    a binary play-off of local dgeqp3 runs on groups of 2b full-height columns.
* qr_sketch      -- randomised panel pivoting (Duersch & Gu; Martinsson et al., HQRRP): the b pivots of a panel are the
                    first b pivots of dgeqp3 on the (b + p) x n_c sketch  Y = Omega . A_trailing  with a FIXED Gaussian
                    Omega (deterministic), the sketch re-formed from the updated trailing matrix for every panel (the
                    down-dated sketch of HQRRP loses the small columns of a graded DQMC matrix to cancellation).
* qr_normpanel   -- the b columns of largest current norm, exact dgeqp3 inside the panel (control: no view of the
                    dependencies between candidates and the rest).
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla


def _finish(A, k, perm, Qs):
    """Assemble (Q, R0, P) from the in-place reduced matrix."""
    n = A.shape[0]
    Q = np.eye(n)
    for (k0, Qp) in reversed(Qs):
        Q[k0:, k0:] = Qp @ Q[k0:, k0:]
    return Q, np.triu(A), np.array(perm)


def _apply_panel(A, k, b, perm, cols, Qs, local_pivot):
    """Move trailing columns `cols` (indices relative to k, in pivot order) to the front of the trailing matrix, factor
    the panel by Householder QR (exact dgeqp3 inside the panel when local_pivot) and update the rest."""
    n = A.shape[0]
    nc = n - k
    rest = [c for c in range(nc) if c not in set(cols)]
    order = list(cols) + rest
    A[:, k:] = A[:, k + np.array(order)]
    perm[k:] = [perm[k + c] for c in order]
    b = len(cols)
    if local_pivot:
        Qp, Rp, pp = sla.qr(A[k:, k:k + b], pivoting=True)
        A[:k, k:k + b] = A[:k, k + pp]
        perm[k:k + b] = [perm[k + c] for c in pp]
    else:
        Qp, Rp = sla.qr(A[k:, k:k + b])
    A[k:, k:k + b] = Rp
    A[k:, k + b:] = Qp.T @ A[k:, k + b:]
    Qs.append((k, Qp))


def _qr_sign(Q, R0):
    return Q, R0


def qr_tournament(M, b=32, local_pivot=False):
    A = np.array(M, dtype=np.float64, copy=True)
    n = A.shape[0]
    perm = list(range(n)); Qs = []
    k = 0
    while k < n:
        nc = n - k
        bb = min(b, nc)
        cand = list(range(nc))
        while len(cand) > bb:
            nxt = []
            for g in range(0, len(cand), 2 * bb):
                grp = cand[g:g + 2 * bb]
                if len(grp) <= bb:
                    nxt += grp
                    continue
                _, _, pg = sla.qr(A[k:, k + np.array(grp)], pivoting=True, mode="economic")
                nxt += [grp[j] for j in pg[:bb]]
            cand = nxt
        # a final ordering of the winners among themselves
        if len(cand) > 1:
            _, _, pg = sla.qr(A[k:, k + np.array(cand)], pivoting=True, mode="economic")
            cand = [cand[j] for j in pg]
        _apply_panel(A, k, bb, perm, cand, Qs, local_pivot)
        k += bb
    return _finish(A, k, perm, Qs)


_OMEGA = {}


def omega(rows, n, seed=20240229):
    """The fixed sketching matrix: rows x n i.i.d. standard normal from a seeded generator (the HIP library carries the same
    numbers: a counter-based generator evaluated on the device would do as well, the values only need to be fixed)."""
    key = (rows, n, seed)
    if key not in _OMEGA:
        _OMEGA[key] = np.random.default_rng(seed).standard_normal((rows, n))
    return _OMEGA[key]


def omega_sign(rows, n):
    """Rademacher sketching matrix from an integer hash of the ROW index r of the matrix: Omega[i, r] = -1 when bit i of
    fmix32(r * 0x9E3779B1 + 0x85EBCA77) is set, +1 otherwise (rows <= 32).  The same function as `qp_row_bits` / `qp_sign`
    of dqmc_amd/csrc/qr_panel.hip, which generate the matrix in registers: the products Omega . A are sums of +-a_rc."""
    assert rows <= 32
    key = (rows, n, "sign")
    if key not in _OMEGA:
        M32 = np.uint64(0xFFFFFFFF)
        h = (np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B1) + np.uint64(0x85EBCA77)) & M32
        h ^= h >> np.uint64(16); h = (h * np.uint64(0x85EBCA6B)) & M32
        h ^= h >> np.uint64(13); h = (h * np.uint64(0xC2B2AE35)) & M32
        h ^= h >> np.uint64(16)
        bits = (h[None, :] >> np.arange(rows, dtype=np.uint64)[:, None]) & np.uint64(1)
        _OMEGA[key] = np.where(bits == 1, -1.0, 1.0)
    return _OMEGA[key]


def qr_sketch(M, b=32, p=8, local_pivot=True, sign=False, cand=0):
    A = np.array(M, dtype=np.float64, copy=True)
    n = A.shape[0]
    perm = list(range(n)); Qs = []
    k = 0
    while k < n:
        nc = n - k
        bb = min(b, nc)
        if nc <= bb:
            cand_cols = list(range(nc))
            if not local_pivot and nc > 1:
                _, _, pg = sla.qr(A[k:, k:], pivoting=True, mode="economic")
                cand_cols = list(pg)
        else:
            Om = (omega_sign(b + p, n) if sign else omega(b + p, n))[:, k:]
            Y = Om @ A[k:, k:]
            if cand and nc > cand:
                # the greedy selection restricted to the `cand` columns of largest sketch norm (what ONE wave can hold, a column per lane:
                # no cross-wave exchange per step); the others are left for later panels
                top = np.argsort(-np.einsum("ij,ij->j", Y, Y), kind="stable")[:cand]
                _, _, pg = sla.qr(Y[:, top], pivoting=True, mode="economic")
                pg = top[pg]
            else:
                _, _, pg = sla.qr(Y, pivoting=True, mode="economic")
            cand_cols = list(pg[:bb])
        _apply_panel(A, k, bb, perm, cand_cols, Qs, local_pivot)
        k += bb
    return _finish(A, k, perm, Qs)


def _house_apply(Yw, j, p):
    """One Householder step of the sketch QR: pivot column p of the working sketch Yw (rows j..), reflector applied to all columns."""
    x = Yw[j:, p].copy()
    alpha = x[0]; t2 = float(x[1:] @ x[1:])
    if t2 == 0.0:
        return
    beta = -np.copysign(np.sqrt(alpha * alpha + t2), alpha)
    v = x / (alpha - beta); v[0] = 1.0
    tau = (beta - alpha) / beta
    Yw[j:, :] -= tau * np.outer(v, v @ Yw[j:, :])


def qr_sketch_lookahead(M, b=16, sr=48, stats=None, guard=0.0):
    """qr_sketch with the selection taken OFF the critical path: the pivots of panel k + 1 are chosen from the sketch of the matrix
    BEFORE panel k's update -- Y_k = Omega . A_k is fresh, the b known pivots of panel k are eliminated from it first (forced steps),
    and b greedy steps on the remaining sr - b rows pick panel k + 1 (Duersch & Gu: the residual block of the sketch after b
    Householder steps is itself a sketch of the updated trailing matrix).  On the device the selection of panel k + 1 then runs beside
    panel k's Householder factorisation instead of in front of it (dqmc_amd/csrc/qr_panel.hip)."""
    A = np.array(M, dtype=np.float64, copy=True)
    n = A.shape[0]
    perm = list(range(n)); Qs = []
    Om = omega_sign(min(sr, 32), n) if sr <= 32 else np.vstack([omega_sign(32, n), omega_sign2(sr - 32, n)])
    k = 0
    nxt = None                      # original column indices chosen for the coming panel
    while k < n:
        nc = n - k
        bb = min(b, nc)
        Y = Om[:, k:] @ A[k:, k:]
        cols = perm[k:]
        if nxt is None:             # first panel: greedy on the fresh sketch
            Yw = Y.copy(); live = np.ones(nc, bool); cur = []
            for j in range(bb):
                nr = np.where(live, np.einsum("ij,ij->j", Yw[j:], Yw[j:]), -1.0)
                p = int(np.argmax(nr)); cur.append(p); live[p] = False
                _house_apply(Yw, j, p)
        else:
            cur = [cols.index(c) for c in nxt]
            Yw = Y.copy(); live = np.ones(nc, bool)
            for j, p in enumerate(cur):
                live[p] = False
                _house_apply(Yw, j, p)
        # greedy continuation: the pivots of the NEXT panel.  guard > 0: a winner whose sketch residual has fallen below guard x its
        # sketch norm before the elimination carries no digits any more (the elimination happened in sketch space, in floating point):
        # the look-ahead is abandoned and the next panel selects from ITS fresh sketch (nxt = None)
        nb2 = min(b, nc - bb)
        nx = []
        n0 = np.einsum("ij,ij->j", Y, Y)
        ok = True
        for j in range(bb, bb + nb2):
            nr = np.where(live, np.einsum("ij,ij->j", Yw[j:], Yw[j:]), -1.0)
            p = int(np.argmax(nr))
            if guard > 0 and nr[p] < (guard * guard) * n0[p]:          # the best candidate's residual is below the noise of its own elimination
                ok = False; break
            nx.append(p); live[p] = False
            _house_apply(Yw, j, p)
        if stats is not None:
            stats.append(1 if (ok and nx) else 0)
        nxt = [cols[p] for p in nx] if (nx and ok) else None
        _apply_panel(A, k, bb, perm, cur, Qs, False)
        k += bb
    return _finish(A, k, perm, Qs)


def omega_sign2(rows, n):
    """Sketch rows 32 .. 63: the bits of a second mix of the row index (qp_row_bits2 of qr_panel.hip)."""
    assert rows <= 32
    key = (rows, n, "sign2")
    if key not in _OMEGA:
        M32 = np.uint64(0xFFFFFFFF)
        h = (np.arange(n, dtype=np.uint64) * np.uint64(0x7FEB352D) + np.uint64(0x846CA68B)) & M32
        h ^= h >> np.uint64(16); h = (h * np.uint64(0x85EBCA6B)) & M32
        h ^= h >> np.uint64(13); h = (h * np.uint64(0xC2B2AE35)) & M32
        h ^= h >> np.uint64(16)
        bits = (h[None, :] >> np.arange(rows, dtype=np.uint64)[:, None]) & np.uint64(1)
        _OMEGA[key] = np.where(bits == 1, -1.0, 1.0)
    return _OMEGA[key]


def qr_normpanel(M, b=32):
    A = np.array(M, dtype=np.float64, copy=True)
    n = A.shape[0]
    perm = list(range(n)); Qs = []
    k = 0
    while k < n:
        nc = n - k
        bb = min(b, nc)
        nrm = np.linalg.norm(A[k:, k:], axis=0)
        cand = list(np.argsort(-nrm, kind="stable")[:bb])
        _apply_panel(A, k, bb, perm, cand, Qs, True)
        k += bb
    return _finish(A, k, perm, Qs)


def quality(R0):
    """(max |r_ij| / |r_ii| over j > i, max d_j / d_i over j > i): both are <= 1 for dgeqp3."""
    d = np.abs(np.diag(R0))
    Rn = np.abs(np.triu(R0, 1)) / d[:, None]
    up = float(Rn.max()) if Rn.size else 0.0
    suf = np.maximum.accumulate(d[::-1])[::-1]
    grade = float(np.max(suf[1:] / d[:-1])) if len(d) > 1 else 0.0
    return up, grade
