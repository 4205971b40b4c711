"""Host-side model constants for the sweep engine (test/bench plumbing).

Mirrors what AttractiveHubbard's constructor derives before the hot path
starts (source/model.cpp:3-60), GHQField's tables (include/field.h:29-48) and
Lattice's neighbour rule (include/lattice.h:100-107).  The C++17 facade in
dqmc_amd/host carries the same logic for C++ callers.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict

import numpy as np


def ghq_tables():
    """gamma, eta of the 4-state Gauss-Hermite HS field (include/field.h:32-43)."""
    s6 = np.sqrt(6.0)
    gamma = np.array([1.0 - s6 / 3.0, 1.0 + s6 / 3.0, 1.0 + s6 / 3.0, 1.0 - s6 / 3.0])
    eta = np.array([-np.sqrt(2.0 * (3.0 + s6)), -np.sqrt(2.0 * (3.0 - s6)),
                    np.sqrt(2.0 * (3.0 - s6)), np.sqrt(2.0 * (3.0 + s6))])
    return gamma, eta


PROPOSAL = np.array([[1, 2, 3], [0, 2, 3], [0, 1, 3], [0, 1, 2]])   # include/field.h:45-48


def build_K(L1: int, L2: int, t: float, mu: float) -> np.ndarray:
    """AttractiveHubbard::build_K_matrix (source/model.cpp:39-60): assignment
    (not +=) of -t on +x and +y bonds, site = y*L1 + x, PBC."""
    n = L1 * L2
    K = np.zeros((n, n))
    for i in range(n):
        K[i, i] = -mu
        ux, uy = i % L1, i // L1
        nx = uy * L1 + (ux + 1) % L1
        ny = ((uy + 1) % L2) * L1 + ux
        K[i, nx] = -t; K[nx, i] = -t
        K[i, ny] = -t; K[ny, i] = -t
    return K


def lattice_bonds(L1: int, L2: int):
    """The distinct nearest-neighbour bonds (i, j) build_K sets to -t (include/lattice.h:100-107, PBC; a bond reached
    twice -- L = 2 -- counts once, as the assignment above does).  L = 1 is refused: there the reference's assignment
    overwrites the diagonal of K with -t, which no bond list can express."""
    if L1 < 2 or L2 < 2:
        raise ValueError("checkerboard break-up needs L1, L2 >= 2")
    n = L1 * L2
    seen, out = set(), []
    for i in range(n):
        ux, uy = i % L1, i // L1
        for j in (uy * L1 + (ux + 1) % L1, ((uy + 1) % L2) * L1 + ux):
            key = (min(i, j), max(i, j))
            if i != j and key not in seen:
                seen.add(key); out.append((i, j))
    return out


def checkerboard_groups(L1: int, L2: int):
    """Bond groups of the checkerboard break-up (README.md:40 of the reference lists it as future work): every group is
    a set of disjoint bonds, so exp(dtau t sum_group(c_i^+ c_j + h.c.)) is a product of independent 2x2 blocks.  Even L:
    the four classic groups (x bonds from even / odd columns, y bonds from even / odd rows); otherwise a greedy edge
    colouring of the same bond list (an odd ring needs a third colour per direction)."""
    groups = []
    for b in lattice_bonds(L1, L2):
        for g in groups:
            if all(b[0] not in q and b[1] not in q for q in g):
                g.append(b); break
        else:
            groups.append([b])
    if L1 % 2 == 0 and L2 % 2 == 0 and L1 >= 4 and L2 >= 4:
        site = lambda x, y: (y % L2) * L1 + (x % L1)
        groups = [[(site(x, y), site(x + 1, y)) for y in range(L2) for x in range(p, L1, 2)] for p in (0, 1)] + \
                 [[(site(x, y), site(x, y + 1)) for x in range(L1) for y in range(p, L2, 2)] for p in (0, 1)]
    return groups


def expm_sym(A: np.ndarray) -> np.ndarray:
    """exp of a real symmetric matrix by eigendecomposition (the reference
    calls arma::expmat, source/model.cpp:32-35; setup only, not on the path)."""
    w, V = np.linalg.eigh(A)
    return (V * np.exp(w)) @ V.T


@dataclass
class HubbardModel:
    L1: int
    L2: int
    U: float
    beta: float
    nt: int
    n_stab: int = 10
    t: float = 1.0
    mu: float = -0.1
    expK: np.ndarray = field(init=False, repr=False)
    invexpK: np.ndarray = field(init=False, repr=False)

    def __post_init__(self):
        self.n = self.L1 * self.L2
        self.dtau = self.beta / self.nt                          # source/model.cpp:23
        self.g = float(np.sqrt(0.5 * abs(self.U) * self.dtau))   # source/model.cpp:27
        self.alpha = -1.0
        self.gamma, self.eta = ghq_tables()
        self.K = build_K(self.L1, self.L2, self.t, self.mu)
        self.expK = np.asfortranarray(expm_sym(-self.dtau * self.K))
        self.invexpK = np.asfortranarray(expm_sym(self.dtau * self.K))
        self.n_stack = -(-self.nt // self.n_stab)

    def engine(self, lib, device: int = 0, n_chains=None):
        return lib.engine(self.n, self.nt, self.n_stab, self.g, self.gamma, self.eta, self.expK, self.invexpK,
                          device=device, n_chains=n_chains)

    def checkerboard(self):
        """(groups, cosh(dtau t), sinh(dtau t), exp(dtau mu)): the arguments of Engine.set_checkerboard for this model."""
        return (checkerboard_groups(self.L1, self.L2), float(np.cosh(self.dtau * self.t)), float(np.sinh(self.dtau * self.t)),
                float(np.exp(self.dtau * self.mu)))

    def checkerboard_expK(self, inverse: bool = False) -> np.ndarray:
        """Dense E = f E_{G-1} ... E_0 (or its inverse) from 2x2 blocks, plain numpy: what the engines must hold after
        set_checkerboard."""
        groups, c, s, f = self.checkerboard()
        E = np.eye(self.n)
        for g in groups:
            Eg = np.eye(self.n)
            for i, j in g:
                Eg[i, i] = Eg[j, j] = c; Eg[i, j] = Eg[j, i] = -s if inverse else s
            E = (E @ Eg) if inverse else (Eg @ E)
        return np.asfortranarray(E / f if inverse else E * f)

    def random_fields(self, seed: int) -> np.ndarray:
        """i.i.d. uniform {0,1,2,3} fields, shape (nt, n) (include/field.h:54-57)."""
        return np.random.default_rng(seed).integers(0, 4, size=(self.nt, self.n), dtype=np.int64)

    def random_stream(self, rng: np.random.Generator, rows: int = None):
        """(perm, kprop, u) for `rows` slices (default nt): what one half-sweep of
        update::local_update consumes (source/update.cpp:10-25)."""
        rows = self.nt if rows is None else rows
        perm = np.stack([rng.permutation(self.n) for _ in range(rows)]).astype(np.int32)
        kprop = rng.integers(0, 3, size=(rows, self.n), dtype=np.uint8)
        u = rng.random((rows, self.n))
        return perm, kprop, u

    # plain numpy evaluation of B_l (source/dqmc.cpp:78-80, source/model.cpp:62-72)
    def B(self, fields_l: np.ndarray) -> np.ndarray:
        return np.exp(self.g * self.eta[fields_l])[:, None] * self.expK


# BASELINE.json configs
CONFIGS: Dict[str, dict] = {
    "cfg1": dict(L1=4, L2=4, U=4.0, beta=2.0, nt=20, n_stab=10),
    "cfg2": dict(L1=8, L2=8, U=4.0, beta=4.0, nt=80, n_stab=10),
    "cfg3": dict(L1=16, L2=16, U=8.0, beta=8.0, nt=200, n_stab=10),
    # cfg 4 = replica exchange: one engine per inverse temperature (CFG4_BETAS), swaps over RCCL; `beta` here is rank 0's
    "cfg4": dict(L1=16, L2=16, U=8.0, beta=8.0, nt=200, n_stab=10),
    "cfg5": dict(L1=24, L2=24, U=4.0, beta=10.0, nt=400, n_stab=10),
}
# BASELINE.json configs[3]: "16x16 2D Hubbard replica exchange across 8 inverse temperatures" (the list is not given there;
# a geometric-ish ladder below the headline beta = 8, the shape of examples/parameters.in's [ParallelTempering] betas)
CFG4_BETAS = [8.0, 7.0, 6.0, 5.0, 4.0, 3.5, 3.0, 2.5]
