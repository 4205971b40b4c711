"""CPU oracle binding -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package (see oracle/dqmc_oracle.cpp header).  PARITY UNPINNED by
the reference itself (it has no tests and cannot be built here); pinned by
analytic cases + oracle/numpy_ref.py + built-in-vs-LAPACK agreement.
"""
from __future__ import annotations

import os
import subprocess
from functools import lru_cache

from dqmc_amd.abi import DqmcLib

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIB_PATH = os.path.join(_HERE, "libdqmc_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "dqmc_oracle.cpp")
    if force or not os.path.exists(ORACLE_LIB_PATH) or os.path.getmtime(ORACLE_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libdqmc_oracle.so"])
    return ORACLE_LIB_PATH


class OracleLib(DqmcLib):
    def set_backend(self, name: str) -> bool:
        """'builtin' (self-contained kernels) or 'lapack' (dlopen'd MKL/LAPACK:
        dgeqp3/dorgqr/dgetrf/dgetrs/dgemm, what Armadillo forwards to)."""
        f = self._sym("set_backend"); f.argtypes = [__import__("ctypes").c_char_p]
        return f(name.encode()) == 0


@lru_cache(maxsize=None)
def oracle() -> OracleLib:
    build()
    return OracleLib(ORACLE_LIB_PATH, "orc_")
