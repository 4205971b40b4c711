"""Step 1 of the panel-pivoted to_LDR: does a factorisation with one global pivot decision per PANEL hold the parity bar?

Runs the numpy chain (oracle/numpy_ref.py) with dgeqp3 replaced by each variant of oracle/panel_qr.py on
  (A) the cfg-3 thermalised fixture: G(0,0) after init, then one full sweep (forward + backward) on the fixture's stream,
  (B) cfg 3 from i.i.d. fields (seed 12): init + forward sweep, against the same evaluation with dgeqp3,
  (C) cfg5_random_init (N = 576): G(0,0) against the fixture's rows,
and prints max|dG|, wrap errors, whether the Markov trajectory (fields) is the fixture's, and the quality numbers of the
factorisations (max |r_ij|/|r_ii|, worst grading violation d_j/d_i, j > i).     python scripts/eval_panel_qr.py [variant ...]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dqmc_amd import CONFIGS, HubbardModel, fixtures      # noqa: E402
from oracle import numpy_ref as nr                          # noqa: E402
from oracle import panel_qr as pq                           # noqa: E402

QUAL = []
LA_STATS = []


def wrap(fn):
    def f(M):
        Q, R0, P = fn(M)
        QUAL.append(pq.quality(R0))
        return Q, R0, P
    return f


VARIANTS = {
    "dgeqp3": None,
    "tournament_b16": lambda M: pq.qr_tournament(M, 16),
    "tournament_b32": lambda M: pq.qr_tournament(M, 32),
    "sketch_b16_p8": lambda M: pq.qr_sketch(M, 16, 8),
    "sketch_b32_p8": lambda M: pq.qr_sketch(M, 32, 8),
    "sketch_b32_p0": lambda M: pq.qr_sketch(M, 32, 0),
    "sketch_b32_p8_nolocal": lambda M: pq.qr_sketch(M, 32, 8, local_pivot=False),
    "sketch_b64_p8": lambda M: pq.qr_sketch(M, 64, 8),
    "sketch_b16_p16_nolocal": lambda M: pq.qr_sketch(M, 16, 16, local_pivot=False),
    "sign_b16_p8_nolocal": lambda M: pq.qr_sketch(M, 16, 8, local_pivot=False, sign=True),
    "sign_b16_p16_nolocal": lambda M: pq.qr_sketch(M, 16, 16, local_pivot=False, sign=True),
    "sign_b32_p16_nolocal": lambda M: pq.qr_sketch(M, 32, 16, local_pivot=False, sign=True),
    "sign_b32_p32_nolocal": lambda M: pq.qr_sketch(M, 32, 32, local_pivot=False, sign=True),
    "lookahead_b16_sr48": lambda M: pq.qr_sketch_lookahead(M, 16, 48),
    "lookahead_b16_sr40": lambda M: pq.qr_sketch_lookahead(M, 16, 40),
    "lookahead_b16_sr64": lambda M: pq.qr_sketch_lookahead(M, 16, 64),
    "lookahead_b16_sr48_g1e-10": lambda M: pq.qr_sketch_lookahead(M, 16, 48, guard=1e-10, stats=LA_STATS),
    "lookahead_b16_sr48_g1e-7": lambda M: pq.qr_sketch_lookahead(M, 16, 48, guard=1e-7, stats=LA_STATS),
    "sign_b16_p16_cand64": lambda M: pq.qr_sketch(M, 16, 16, local_pivot=False, sign=True, cand=64),
    "sign_b16_p16_cand32": lambda M: pq.qr_sketch(M, 16, 16, local_pivot=False, sign=True, cand=32),
    "normpanel_b32": lambda M: pq.qr_normpanel(M, 32),
    "normpanel_b16": lambda M: pq.qr_normpanel(M, 16),
}


def qual_line():
    if LA_STATS:
        print("      look-ahead selections usable: %d of %d panels" % (sum(LA_STATS), len(LA_STATS))); LA_STATS.clear()
    if not QUAL:
        return ""
    q = np.array(QUAL); QUAL.clear()
    return "max|r_ij/r_ii| %.2f  grading d_j/d_i %.2f  (%d factorisations)" % (q[:, 0].max(), q[:, 1].max(), len(q))


def run(name):
    fn = VARIANTS[name]
    nr.set_qr(None if fn is None else wrap(fn))
    t0 = time.time()
    print("== %s" % name, flush=True)
    # (A)
    z, m, streams = fixtures.load("cfg3_therm")
    c = nr.NumpyChain(m, z["fields"]); c.init()
    print("  A cfg3 therm  init: max|dG0| %.2e (max|G| %.1f)  %s" % (np.abs(c.G - z["G0"]).max(), np.abs(z["G0"]).max(), qual_line()), flush=True)
    c.sweep_fwd(*streams[0]); c.sweep_bwd(*streams[1])
    same = np.array_equal(c.f, z["fields_after"])
    print("  A cfg3 therm sweep: fields %s  n_acc %d (fixture %d)  max|dG_after| %.2e  max wrap err %.2e (fixture %.2e)  %s"
          % ("identical" if same else "DIFFER (%d sites)" % int((c.f != z["fields_after"]).sum()), c.n_acc, int(z["n_accepted"]),
             np.abs(c.G - z["G_after"]).max(), max(c.errs), float(z["max_wrap_err"]), qual_line()), flush=True)
    # (B)
    m3 = HubbardModel(**CONFIGS["cfg3"])
    f = m3.random_fields(12); rng = np.random.default_rng(1200); sf = m3.random_stream(rng)
    nr.set_qr(None)
    r = nr.NumpyChain(m3, f); r.init(); G0r = r.G.copy(); r.sweep_fwd(*sf)
    nr.set_qr(None if fn is None else wrap(fn))
    c = nr.NumpyChain(m3, f); c.init(); d0 = np.abs(c.G - G0r).max(); c.sweep_fwd(*sf)
    print("  B cfg3 iid    init: max|dG0| %.2e (max|G| %.2e, rel %.2e)   fwd sweep: fields %s  max|dG| %.2e (max|G| %.1f)  wrap err %.2e (dgeqp3 %.2e)  %s"
          % (d0, np.abs(G0r).max(), d0 / np.abs(G0r).max(), "identical" if np.array_equal(c.f, r.f) else "DIFFER",
             np.abs(c.G - r.G).max(), np.abs(r.G).max(), max(c.errs), max(r.errs), qual_line()), flush=True)
    if os.environ.get("EVAL_SKIP_C"):
        nr.set_qr(None); return
    # (C)
    z, m, _ = fixtures.load("cfg5_random_init")
    c = nr.NumpyChain(m, z["fields"]); c.init()
    e, sc = fixtures.g0_error(z, c.G)
    print("  C cfg5 iid    init: max|dG0| %.2e (max|G| %.1f, rel %.2e)  %s   [%.0f s]" % (e, sc, e / sc, qual_line(), time.time() - t0), flush=True)
    nr.set_qr(None)


if __name__ == "__main__":
    for v in (sys.argv[1:] or list(VARIANTS)):
        run(v)
