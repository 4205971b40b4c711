"""Loading of the committed golden fixtures (tests/golden/make_golden.py): shared by the tests and by bench.py's parity leg."""
import hashlib
import os

import numpy as np

from .model import HubbardModel

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
NAMES = sorted(fn[:-4] for fn in os.listdir(GOLD) if fn.endswith(".npz")) if os.path.isdir(GOLD) else []


def load(name):
    """-> (z, model, streams): streams = ((perm, k, u) forward, (perm, k, u) backward) or None when the fixture has no sweep."""
    z = np.load(os.path.join(GOLD, name + ".npz"))
    m = HubbardModel(L1=int(z["L1"]), L2=int(z["L2"]), U=float(z["U"]), beta=float(z["beta"]), nt=int(z["nt"]), n_stab=int(z["n_stab"]))
    streams = None
    if "perm_f" in z.files:
        streams = ((z["perm_f"], z["k_f"], z["u_f"]), (z["perm_b"], z["k_b"], z["u_b"]))
    elif "stream_seed" in z.files:                       # the stream is regenerated from its seed; the digest pins it
        rng = np.random.default_rng(int(z["stream_seed"]))
        streams = (m.random_stream(rng), m.random_stream(rng))
        h = hashlib.sha256()
        for st in streams:
            for a in st:
                h.update(np.ascontiguousarray(a).tobytes())
        assert h.hexdigest() == str(z["stream_sha256"]), "numpy's generator produced a different stream than the fixture was made with"
    return z, m, streams


def g0_error(z, G):
    """max|G - G0| over the rows the fixture stores, and the scale max(1, max|G0|)."""
    ref = z["G0"]
    got = G[z["G0_rows"]] if "G0_rows" in z.files else G
    return float(np.abs(got - ref).max()), max(1.0, float(np.abs(ref).max()))
