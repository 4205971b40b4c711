"""dqmc_amd -- MI355X-native equal-time sweep engine for determinant QMC.

Python here is test/bench plumbing over the C ABI (include/dqmc_hip.h); the
product is the HIP library ``dqmc_amd/libdqmc_hip.so`` (dqmc_amd/csrc) and
the C++17 host facade (dqmc_amd/host).  There is no CPU fallback: ``lib()``
raises if the HIP library has not been built.
"""
from __future__ import annotations

import os
from functools import lru_cache

from .abi import ABI_SYMBOLS, Comm, DqmcError, DqmcLib, Engine, ExchangeResult, Stats  # noqa: F401
from .model import HubbardModel, ghq_tables, CONFIGS, CFG4_BETAS  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(_HERE, "libdqmc_hip.so")
HOST_LIB_PATH = os.path.join(_HERE, "libdqmc_host.so")


@lru_cache(maxsize=None)
def lib() -> DqmcLib:
    """The product library (HIP, gfx950).  Fails loudly when missing."""
    return DqmcLib(HIP_LIB_PATH, "dqmc_")
