"""CPU tests of the boundary: the HIP library loads without a GPU, exports
every symbol include/dqmc_hip.h declares, and fails loudly (no fallback)."""
import os
import re

import numpy as np
import pytest

import dqmc_amd
from dqmc_amd import ABI_SYMBOLS, DqmcError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "dqmc_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dqmc_[a-zA-Z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = dqmc_amd.lib()                       # raises if libdqmc_hip.so was not built
    declared = _header_symbols()
    assert len(declared) >= 30
    missing = [s for s in declared if not hasattr(lib._dll, s)]
    assert not missing, f"declared in dqmc_hip.h but not exported: {missing}"
    assert sorted("dqmc_" + s for s in ABI_SYMBOLS) == declared
    assert lib.backend() == "hip:gfx950"


def test_no_cpu_fallback_without_device():
    lib = dqmc_amd.lib()
    if lib.device_count() > 0:
        pytest.skip("a GPU is visible; the no-device behaviour is checked on CPU-only hosts")
    m = dqmc_amd.HubbardModel(**dqmc_amd.CONFIGS["cfg1"])
    with pytest.raises(DqmcError) as ei:
        m.engine(lib)
    assert ei.value.code == -2                 # DQMC_ENODEVICE
    with pytest.raises(DqmcError):
        lib.to_ldr(np.eye(4))
    with pytest.raises(DqmcError):
        lib.gemm(np.eye(4), np.eye(4))


def test_product_does_not_reference_oracle():
    # nothing under dqmc_amd/ may import, link or call the oracle
    bad = []
    for dp, _, fns in os.walk(os.path.join(ROOT, "dqmc_amd")):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(dp, fn)).read()
                if re.search(r"(import\s+oracle|from\s+oracle|libdqmc_oracle|orc_[a-z_]+\s*\()", txt):
                    bad.append(fn)
    assert not bad, bad


def test_walk_kernels_leave_the_hand_managed_registers_alone():
    """The delayed-update walk keeps its pending pairs and prefetched columns in registers named literally in inline asm
    (scripts/gen_walk_bodies.py); the allocator is kept out of them by the kernels' register budget, which only the ISA can confirm."""
    import shutil
    import subprocess
    import sys
    hipcc = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc")
    if not (os.path.exists(hipcc) or shutil.which("hipcc")):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "check_walk_regs.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
