"""Generates the golden vectors in this directory.

The reference ships no fixtures and cannot be built here, so
these are produced by the independent numpy/scipy evaluation
(oracle/numpy_ref.py: scipy pivoted QR = LAPACK dgeqp3, numpy solve = dgesv),
NOT by the C++ oracle or the HIP library they are used to check.

    python tests/golden/make_golden.py

Each .npz holds inputs (model parameters, HS fields, the random stream of one
forward + one backward sweep) and outputs (G(0,0), log det, the fields and G
after the sweep).  `tol` is the tolerance a checker should use relative to
max(1, max|G|): 1e-10 on thermalised fields.  cfg 3 keeps its file small by
storing the SEED of the sweep's random stream (model.random_stream of
numpy's default_rng; `stream_sha256` guards against a numpy whose generator
drifted) instead of the 1.2 MB stream; cfg 5 (N = 576, i.i.d. fields, no CPU
can thermalise it in reasonable time) stores every 9th row of G(0,0).
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dqmc_amd import CONFIGS, HubbardModel      # noqa: E402
from oracle.numpy_ref import NumpyChain         # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def stream_digest(*streams):
    h = hashlib.sha256()
    for st in streams:
        for a in st:
            h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def make(name, cfg, seed, therm_sweeps, with_sweep, tol, stream_seed=None, keep_fields=None, row_step=None):
    m = HubbardModel(**cfg)
    rng = np.random.default_rng(seed)
    c = NumpyChain(m, m.random_fields(seed)); c.init()
    for _ in range(therm_sweeps):
        c.sweep_fwd(*m.random_stream(rng)); c.sweep_bwd(*m.random_stream(rng))
    fields = c.f.copy()
    if keep_fields is not None:
        assert np.array_equal(fields, keep_fields), "regenerated thermalised fields differ from the committed fixture"
    c2 = NumpyChain(m, fields); c2.init()             # from-scratch evaluation on the saved fields
    out = dict(L1=m.L1, L2=m.L2, U=m.U, beta=m.beta, nt=m.nt, n_stab=m.n_stab, t=m.t, mu=m.mu,
               fields=fields.astype(np.int8), G0=c2.G.copy(), logdet=c2.logdet, tol=tol, therm_sweeps=therm_sweeps)
    if row_step:
        out["G0_rows"] = np.arange(0, m.n, row_step); out["G0"] = out["G0"][::row_step].copy()
    if with_sweep:
        srng = rng if stream_seed is None else np.random.default_rng(stream_seed)
        sf, sb = m.random_stream(srng), m.random_stream(srng)
        c2.sweep_fwd(*sf); c2.sweep_bwd(*sb)
        if stream_seed is None:
            out.update(perm_f=sf[0].astype(np.int16), k_f=sf[1], u_f=sf[2], perm_b=sb[0].astype(np.int16), k_b=sb[1], u_b=sb[2])
        else:
            out.update(stream_seed=stream_seed, stream_sha256=stream_digest(sf, sb))
        out.update(fields_after=c2.f.astype(np.int8), G_after=c2.G.copy(), max_wrap_err=max(c2.errs), n_accepted=c2.n_acc)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "max|G0| = %.3e" % np.abs(out["G0"]).max(), "logdet = %.6f" % out["logdet"])


if __name__ == "__main__":
    make("cfg1_therm", CONFIGS["cfg1"], 101, 5, True, 1e-10)
    make("cfg2_therm", CONFIGS["cfg2"], 102, 5, True, 1e-10)
    make("cfg2_random", CONFIGS["cfg2"], 103, 0, True, 1e-10)
    make("ex6x6_therm", dict(L1=6, L2=6, U=4.0, beta=4.0, nt=40, n_stab=10), 104, 5, True, 1e-10)   # examples/parameters.in
    old = os.path.join(HERE, "cfg3_therm.npz")
    make("cfg3_therm", CONFIGS["cfg3"], 105, 3, True, 1e-10, stream_seed=1053, keep_fields=np.load(old)["fields"] if os.path.exists(old) else None)
    make("cfg5_random_init", CONFIGS["cfg5"], 107, 0, False, 1e-10, row_step=9)
