"""Independent numpy/scipy evaluation of the kfkq/DQMC equal-time path.

TEST INFRASTRUCTURE ONLY (see oracle/dqmc_oracle.cpp header).  This is the
second, independently written statement of the algorithm used to pin the C++
oracle: scipy.linalg.qr(pivoting=True) is LAPACK dgeqp3 -- the routine
Armadillo's arma::qr(Q,R,P,M,"vector") forwards to (source/stablelinalg.cpp:41)
-- and numpy.linalg.solve is dgesv (arma::solve, :112,123,147,155).

Pure-Python loops over sites: use at cfg 1-2 sizes, or a few slices of cfg 3.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla

from dqmc_amd.model import PROPOSAL, HubbardModel


_QR = None          # None = dgeqp3 (the reference's); oracle/panel_qr.py variants are plugged in by scripts/eval_panel_qr.py


def set_qr(fn):
    """Replace the pivoted QR behind to_ldr by fn(M) -> (Q, R0, P) (evaluation of panel-pivoted variants); None restores dgeqp3."""
    global _QR
    _QR = fn


def to_ldr(M):
    """stablelinalg::to_LDR, source/stablelinalg.cpp:35-55."""
    Q, R0, P = sla.qr(M, pivoting=True) if _QR is None else _QR(M)
    d = np.abs(np.diag(R0))
    Rn = R0 / d[:, None]
    R = np.empty_like(Rn)
    R[:, P] = Rn                       # R_normalized.cols(sort_index(P))
    return Q, d, R


def ldr_mul_mat(F, M):                 # source/stablelinalg.cpp:57-67
    L, d, R = F
    q, dd, r = to_ldr(d[:, None] * (R @ M))
    return L @ q, dd, r


def mat_mul_ldr(M, F):                 # source/stablelinalg.cpp:69-79
    L, d, R = F
    q, dd, r = to_ldr((M @ L) * d[None, :])
    return q, dd, r @ R


def ldr_mul_ldr(F1, F2):               # source/stablelinalg.cpp:81-92
    L1, d1, R1 = F1; L2, d2, R2 = F2
    q, dd, r = to_ldr(d1[:, None] * (R1 @ L2) * d2[None, :])
    return L1 @ q, dd, r @ R2


def _split(d):                         # source/stablelinalg.cpp:100-108
    return np.where(d >= 1.0, d, 1.0), np.where(d >= 1.0, 1.0, d)


def inv_I_plus_ldr(F):                 # source/stablelinalg.cpp:94-126
    L, d, R = F
    Dl, Ds = _split(d)
    X = np.linalg.solve(R, np.diag(1.0 / Dl))
    M = X + L * Ds[None, :]
    logdet = np.sum(np.log(Dl)) + np.linalg.slogdet(M)[1]
    G = np.linalg.solve(M.T, X.T).T
    return G, logdet


def inv_I_plus_ldr_mul_ldr(F1, F2):    # source/stablelinalg.cpp:128-158
    L1, d1, R1 = F1; L2, d2, R2 = F2
    D1l, D1s = _split(d1); D2l, D2s = _split(d2)
    X = np.linalg.solve(R2, np.diag(1.0 / D2l))
    TA = (1.0 / D1l)[:, None] * (L1.T @ X)
    TB = D1s[:, None] * (R1 @ (L2 * D2s[None, :]))
    Y = np.linalg.solve(TA + TB, (1.0 / D1l)[:, None] * L1.T)
    return X @ Y


class NumpyChain:
    """DQMC + model state of one chain, numpy only (source/dqmc.cpp)."""

    def __init__(self, model: HubbardModel, fields):
        self.m = model
        self.f = np.array(fields, dtype=np.int64).reshape(model.nt, model.n).copy()
        self.stack = [None] * model.n_stack
        self.G = None; self.logdet = 0.0
        self.errs = []; self.n_acc = 0
        nt, ns = model.nt, model.n_stab
        self.loc_end = [ns - 1] * model.n_stack
        if nt % ns:
            self.loc_end[-1] = nt % ns - 1

    def B(self, l): return self.m.B(self.f[l])
    def invB(self, l): return self.m.invexpK * np.exp(-self.m.g * self.m.eta[self.f[l]])[None, :]

    def Bbar(self, i):                 # source/dqmc.cpp:88-105
        out = np.eye(self.m.n)
        for loc in range(self.loc_end[i] + 1):
            out = self.B(i * self.m.n_stab + loc) @ out
        return out

    def init(self):                    # source/dqmc.cpp:43-72
        n_stack = self.m.n_stack
        for i in range(n_stack - 1, -1, -1):
            F = to_ldr(self.Bbar(i))
            self.stack[i] = F if i == n_stack - 1 else ldr_mul_ldr(self.stack[i + 1], F)
        self.G, self.logdet = inv_I_plus_ldr(self.stack[0])

    def local_update(self, l, perm, kprop, u):    # source/update.cpp:5-32, source/model.cpp:90-138
        m = self.m; G = self.G; acc = 0
        for idx in range(m.n):
            i = int(perm[idx]); old = int(self.f[l, i]); new = int(PROPOSAL[old, int(kprop[idx])])
            gammaR = m.gamma[new] / m.gamma[old]
            bosonR = np.exp(m.alpha * m.g * (m.eta[new] - m.eta[old]))
            delta = 1.0 / bosonR - 1.0
            R = gammaR * bosonR * (1.0 + (1.0 - G[i, i]) * delta) ** 2
            if u[idx] < min(1.0, abs(R)):
                acc += 1
                pref = delta / (1.0 + (1.0 - G[i, i]) * delta)
                U = G[:, i].copy(); V = G[i, :].copy(); V[i] -= 1.0
                G += pref * np.outer(U, V)
                self.f[l, i] = new
        self.n_acc += acc
        return acc

    def sweep_fwd(self, perm, kprop, u, record=None):      # source/dqmc.cpp:337-396
        m = self.m
        for l in range(m.nt):
            self.G = self.B(l) @ self.G @ self.invB(l)
            self.local_update(l, perm[l], kprop[l], u[l])
            i = l // m.n_stab
            if l % m.n_stab == self.loc_end[i]:
                Gtmp = self.G.copy(); bb = self.Bbar(i)
                self.stack[i] = to_ldr(bb) if i == 0 else mat_mul_ldr(bb, self.stack[i - 1])
                if l == m.nt - 1:
                    self.G, self.logdet = inv_I_plus_ldr(self.stack[i])
                else:
                    self.G = inv_I_plus_ldr_mul_ldr(self.stack[i], self.stack[i + 1])
                self.errs.append(np.abs(Gtmp - self.G).max())
                if record is not None: record.append(self.G.copy())

    def sweep_bwd(self, perm, kprop, u, record=None):      # source/dqmc.cpp:398-456
        m = self.m
        for l in range(m.nt - 1, -1, -1):
            self.local_update(l, perm[l], kprop[l], u[l])
            self.G = self.invB(l) @ self.G @ self.B(l)
            i = l // m.n_stab
            if l % m.n_stab == 0:
                Gtmp = self.G.copy(); bb = self.Bbar(i)
                self.stack[i] = to_ldr(bb) if i == m.n_stack - 1 else ldr_mul_mat(self.stack[i + 1], bb)
                if l == 0:
                    self.G, self.logdet = inv_I_plus_ldr(self.stack[i])
                else:
                    self.G = inv_I_plus_ldr_mul_ldr(self.stack[i - 1], self.stack[i])
                self.errs.append(np.abs(Gtmp - self.G).max())
                if record is not None: record.append(self.G.copy())

    def global_action(self):           # source/model.cpp:140-159
        m = self.m
        return -2.0 * self.logdet - (np.sum(m.alpha * m.g * m.eta[self.f]) + np.sum(np.log(m.gamma[self.f])))


def free_fermion_G(model: HubbardModel):
    """Analytic G(0,0) = (I + exp(-beta K))^-1 and log det for U = 0, from the
    PBC dispersion eps_k = -2t(cos kx + cos ky) - mu (valid for L1, L2 > 2;
    source/model.cpp:39-60)."""
    L1, L2, n = model.L1, model.L2, model.n
    x = np.arange(n) % L1; y = np.arange(n) // L1
    G = np.zeros((n, n)); logdet = 0.0
    for qx in range(L1):
        for qy in range(L2):
            kx, ky = 2 * np.pi * qx / L1, 2 * np.pi * qy / L2
            eps = -2 * model.t * (np.cos(kx) + np.cos(ky)) - model.mu
            ph = np.exp(1j * (kx * x + ky * y)) / np.sqrt(n)
            G += (np.outer(ph, ph.conj()) / (1.0 + np.exp(-model.beta * eps))).real
            logdet += np.log1p(np.exp(-model.beta * eps))
    return G, logdet


def equal_time_observables(G, L1, L2):
    """Independent numpy evaluation of the reference's equal-time observables (formulas of source/model.cpp:167-288, displacement
    binning of include/measurementh5.h:13-66, n_orb = 1) -- vectorised, so the summation order differs from the loops the oracle keeps.
    Returns (scalars[3] = density, doubleOcc, swave; chi_r[L1, L2] indexed [dx_idx, dy_idx])."""
    G = np.asarray(G, dtype=np.float64); n = G.shape[0]
    assert n == L1 * L2
    Gc = np.eye(n) - G
    dens = 2.0 * np.trace(Gc) / n
    docc = np.sum(np.diag(Gc) ** 2) / n
    swave = np.sum(Gc * Gc) / n
    ni = 2.0 * (1.0 - np.diag(G))
    ninj = np.outer(ni, ni) + 2.0 * (1.0 - G.T) * G - (ni.mean()) ** 2          # ninj[i, j]; (1 - G(j,i)) G(i,j)
    x = np.arange(n) % L1; y = np.arange(n) // L1
    def pbc(d, L):
        d = np.where(d > L // 2, d - L, d); return np.where(d <= -(L // 2), d + L, d)
    dx = pbc(x[None, :] - x[:, None], L1) + L1 // 2 - 1          # [i, j]
    dy = pbc(y[None, :] - y[:, None], L2) + L2 // 2 - 1
    chi = np.zeros((L1, L2))
    np.add.at(chi, (dx, dy), ninj / n)
    return np.array([dens, docc, swave]), chi


def inv_invldr_plus_ldr(F1, F2):          # source/stablelinalg.cpp:160-190: [F1^-1 + F2]^-1
    (L1, d1, R1), (L2, d2, R2) = F1, F2
    D1l, D1s = _split(d1); D2l, D2s = _split(d2)
    X = np.linalg.solve(R2, np.diag(1.0 / D2l))
    M = (1.0 / D1l)[:, None] * (L1.T @ X) + D1s[:, None] * (R1 @ (L2 * D2s[None, :]))
    Y = np.linalg.solve(M, D1s[:, None] * R1)
    return X @ Y


def unequal_time_observables(Gtt, Gt0, G0t, L1, L2):
    """Vectorised numpy evaluation of greenTau / doublonTau / currxxTau (source/model.cpp:290-394) in displacement space
    (include/measurementh5.h:20-66).  Gtt, Gt0, G0t: arrays [nt + 1][n][n].  Returns [3][nt + 1][L1][L2]."""
    Gtt, Gt0, G0t = (np.asarray(a, dtype=np.float64) for a in (Gtt, Gt0, G0t))
    ntau, n, _ = Gtt.shape
    idx = np.arange(n); x = idx % L1; y = idx // L1
    nbx = y * L1 + (x + 1) % L1                                   # site_neighbors(i, {1,0}, 0)
    def pbc(d, L):
        d = np.where(d > L // 2, d - L, d); return np.where(d <= -(L // 2), d + L, d)
    dx = pbc(x[None, :] - x[:, None], L1) + L1 // 2 - 1; dy = pbc(y[None, :] - y[:, None], L2) + L2 // 2 - 1
    G00 = Gtt[0]
    out = np.zeros((3, ntau, L1, L2))
    dc1j = 2.0 * G00[nbx, idx]; dc2j = 2.0 * G00[idx, nbx]            # [j]
    for tau in range(ntau):
        A, T0, Z = Gtt[tau], Gt0[tau], G0t[tau]
        green = 2.0 * T0; doublon = T0 * T0
        dc1i = 2.0 * A[nbx, idx]; dc2i = 2.0 * A[idx, nbx]              # [i]
        # c_k[i, j]
        c1 = 2.0 * Z[nbx][:, idx].T * T0[nbx][:, idx]                    # G0t(jx,i) Gt0(ix,j)
        c2 = 2.0 * Z.T * T0[nbx][:, nbx]                                 # G0t(j,i) Gt0(ix,jx)
        c3 = 2.0 * Z[nbx][:, nbx].T * T0                                 # G0t(jx,ix) Gt0(i,j)
        c4 = 2.0 * Z[:, nbx].T * T0[:, nbx]                              # G0t(j,ix) Gt0(i,jx)
        t1 = dc1i[:, None] * dc1j[None, :] - c1; t2 = dc1i[:, None] * dc2j[None, :] - c2
        t3 = dc2i[:, None] * dc1j[None, :] - c3; t4 = dc2i[:, None] * dc2j[None, :] - c4
        curr = -(t1 - t2 - t3 + t4)
        for ob, M in enumerate((green, doublon, curr)):
            np.add.at(out[ob, tau], (dx, dy), M / n)
    return out
