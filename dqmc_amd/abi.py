"""ctypes binding of the C ABI declared in include/dqmc_hip.h.

The binder is generic over (shared-library path, symbol prefix) because the
CPU oracle exports the same function set with prefix ``orc_`` (see
oracle/__init__.py -- test infrastructure only).  The product instance, bound
to ``dqmc_amd/libdqmc_hip.so`` with prefix ``dqmc_``, is created by
``dqmc_amd.lib()`` and fails loudly when the HIP library is missing or no GPU
is present: there is no fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

c_double_p = C.POINTER(C.c_double)
c_int64_p = C.POINTER(C.c_int64)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)


class Stats(C.Structure):
    """dqmc_stats of include/dqmc_hip.h."""
    _fields_ = [("acc_rate", C.c_double), ("max_err", C.c_double), ("sum_err", C.c_double),
                ("n_err", C.c_double), ("n_accepted", C.c_int64), ("n_proposed", C.c_int64)]

    @property
    def mean_err(self) -> float:            # DQMC::mean_err(), include/dqmc.h:80
        return self.sum_err / self.n_err if self.n_err else 0.0


class DqmcError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[{code}] {msg}")
        self.code = code


# every symbol include/dqmc_hip.h declares (without prefix): the CPU test
# `test_abi_exports` checks the built library exports each of them.
ABI_SYMBOLS = [
    "last_error", "backend", "device_count",
    "to_ldr", "ldr_mul_mat", "mat_mul_ldr", "ldr_mul_ldr", "inv_I_plus_ldr", "inv_I_plus_ldr_mul_ldr",
    "gemm", "rank1_update",
    "create", "create_batch", "n_chains", "destroy", "set_fields", "get_fields", "init", "get_G", "set_G", "get_logdet",
    "n_stack", "get_stack", "sweep_0_to_beta", "sweep_beta_to_0", "sync", "get_stats",
    "wrap_forward", "wrap_backward", "local_update_slice", "calculate_Bbar", "global_action", "set_checkerboard",
    "update_kernel_time", "set_profiling", "slice_path", "debug_snapshot",
    "measure_equal_time", "measure_accumulate", "measure_fetch",
    "sweep_unequal_time", "get_G_tau", "half_warp", "measure_unequal_time", "measure_unequal_fetch",
    "comm_unique_id", "comm_create_rccl", "comm_create_callbacks", "comm_destroy", "comm_rank", "comm_world_size",
    "comm_transport", "comm_barrier", "comm_allreduce_sum", "comm_selftest", "partner_rank", "replica_exchange_round",
]

UNIQUE_ID_BYTES = 128
# dqmc_sendrecv_fn: int (*)(void* user, const void* send, void* recv, size_t bytes, int partner, int tag)
SENDRECV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int)


class ExchangeResult(C.Structure):
    """dqmc_exchange_result of include/dqmc_hip.h."""
    _fields_ = [("partner", C.c_int), ("decider", C.c_int), ("accepted", C.c_int), ("pad", C.c_int),
                ("S", C.c_double), ("S_prime", C.c_double), ("S_partner", C.c_double), ("S_prime_partner", C.c_double),
                ("deltaS", C.c_double)]


def _f64(a) -> np.ndarray:
    """Column-major (Fortran) contiguous float64 copy/view."""
    return np.asfortranarray(a, dtype=np.float64)


def _p(a: np.ndarray):
    return a.ctypes.data_as(c_double_p)


class DqmcLib:
    """One loaded shared library exposing <prefix>* of include/dqmc_hip.h."""

    def __init__(self, path: str, prefix: str):
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} not found: build it first (python -c 'import __graft_entry__ as g; g.build()' or make)")
        self.path, self.prefix = path, prefix
        self._dll = C.CDLL(path, mode=C.RTLD_GLOBAL)
        g = self._sym
        g("last_error").restype = C.c_char_p
        g("backend").restype = C.c_char_p
        g("destroy").restype = None
        g("destroy").argtypes = [C.c_void_p]
        for name in ("set_fields", "get_fields"):
            g(name).argtypes = [C.c_void_p, c_int64_p]
        for name in ("init", "sync", "n_stack"):
            g(name).argtypes = [C.c_void_p]
        g("create").argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                c_double_p, c_double_p, c_double_p, c_double_p]
        if self.has_symbol("create_batch"):
            g("create_batch").argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_double_p,
                                          c_double_p, c_double_p, c_double_p, c_double_p]
            g("n_chains").argtypes = [C.c_void_p]
        for name in ("get_G", "set_G", "get_logdet", "global_action"):
            g(name).argtypes = [C.c_void_p, c_double_p]
        g("get_stack").argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p]
        for name in ("sweep_0_to_beta", "sweep_beta_to_0"):
            g(name).argtypes = [C.c_void_p, c_int32_p, c_uint8_p, c_double_p]
        g("get_stats").argtypes = [C.c_void_p, C.c_void_p]
        for name in ("wrap_forward", "wrap_backward"):
            g(name).argtypes = [C.c_void_p, C.c_int]
        g("local_update_slice").argtypes = [C.c_void_p, C.c_int, c_int32_p, c_uint8_p, c_double_p, C.c_void_p]
        g("calculate_Bbar").argtypes = [C.c_void_p, C.c_int, c_double_p]
        g("set_checkerboard").argtypes = [C.c_void_p, C.c_int, c_int32_p, c_int32_p, c_double_p, c_double_p, c_double_p]
        g("update_kernel_time").argtypes = [C.c_void_p, c_double_p, c_int64_p, c_int64_p]
        g("set_profiling").argtypes = [C.c_void_p, C.c_int]
        if self.has_symbol("slice_path"):
            g("slice_path").argtypes = [C.c_void_p]
        if self.has_symbol("measure_equal_time"):
            g("measure_equal_time").argtypes = [C.c_void_p, C.c_int, C.c_int, c_double_p, c_double_p]
            g("measure_accumulate").argtypes = [C.c_void_p, C.c_int, C.c_int]
            g("measure_fetch").argtypes = [C.c_void_p, c_double_p, c_double_p, c_int64_p, C.c_int]
        if self.has_symbol("sweep_unequal_time"):
            g("sweep_unequal_time").argtypes = [C.c_void_p]
            g("get_G_tau").argtypes = [C.c_void_p, C.c_int, C.c_int, c_double_p]
            g("measure_unequal_time").argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, c_double_p]
            g("measure_unequal_fetch").argtypes = [C.c_void_p, c_double_p, c_int64_p, C.c_int]
        if self.has_symbol("half_warp"):
            g("half_warp").argtypes = [C.c_void_p, c_double_p, c_double_p, C.c_int, C.c_int, c_double_p]
        if self.has_symbol("debug_snapshot"):
            g("debug_snapshot").argtypes = [C.c_void_p, c_double_p, C.c_void_p, C.c_void_p, C.c_void_p]
        if self.has_symbol("replica_exchange_round"):
            g("comm_unique_id").argtypes = [C.c_void_p]
            g("comm_create_rccl").argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_int]
            g("comm_create_callbacks").argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, SENDRECV_FN, C.c_void_p]
            g("comm_destroy").argtypes = [C.c_void_p]; g("comm_destroy").restype = None
            g("comm_rank").argtypes = [C.c_void_p]; g("comm_world_size").argtypes = [C.c_void_p]
            g("comm_transport").argtypes = [C.c_void_p]; g("comm_transport").restype = C.c_char_p
            g("comm_barrier").argtypes = [C.c_void_p]; g("comm_selftest").argtypes = [C.c_void_p]
            g("comm_allreduce_sum").argtypes = [C.c_void_p, c_double_p, C.c_int]
            g("partner_rank").argtypes = [C.c_int, C.c_int, C.c_int]
            g("replica_exchange_round").argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.POINTER(ExchangeResult)]
        g("to_ldr").argtypes = [C.c_int] + [c_double_p] * 4
        g("ldr_mul_mat").argtypes = [C.c_int] + [c_double_p] * 7
        g("mat_mul_ldr").argtypes = [C.c_int] + [c_double_p] * 7
        g("ldr_mul_ldr").argtypes = [C.c_int] + [c_double_p] * 9
        g("inv_I_plus_ldr").argtypes = [C.c_int] + [c_double_p] * 5
        g("inv_I_plus_ldr_mul_ldr").argtypes = [C.c_int] + [c_double_p] * 7
        g("gemm").argtypes = [C.c_int, c_double_p, C.c_int, c_double_p, C.c_int, c_double_p]
        g("rank1_update").argtypes = [C.c_int, c_double_p, C.c_int, C.c_double]

    def _sym(self, name: str):
        return getattr(self._dll, self.prefix + name)

    def has_symbol(self, name: str) -> bool:
        return hasattr(self._dll, self.prefix + name)

    def check(self, rc: int):
        if rc != 0:
            raise DqmcError(rc, (self._sym("last_error")() or b"").decode())

    # -- misc -----------------------------------------------------------
    def backend(self) -> str:
        return self._sym("backend")().decode()

    def device_count(self) -> int:
        return int(self._sym("device_count")())

    # -- stateless stable linear algebra (include/stablelinalg.h:36-46) --
    def to_ldr(self, M) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        M = _f64(M); n = M.shape[0]
        L, d, R = np.empty((n, n), order="F"), np.empty(n), np.empty((n, n), order="F")
        self.check(self._sym("to_ldr")(n, _p(M), _p(L), _p(d), _p(R)))
        return L, d, R

    def ldr_mul_mat(self, F, M):
        L, d, R = (_f64(x) for x in F); M = _f64(M); n = M.shape[0]
        Lo, do, Ro = np.empty((n, n), order="F"), np.empty(n), np.empty((n, n), order="F")
        self.check(self._sym("ldr_mul_mat")(n, _p(L), _p(d), _p(R), _p(M), _p(Lo), _p(do), _p(Ro)))
        return Lo, do, Ro

    def mat_mul_ldr(self, M, F):
        L, d, R = (_f64(x) for x in F); M = _f64(M); n = M.shape[0]
        Lo, do, Ro = np.empty((n, n), order="F"), np.empty(n), np.empty((n, n), order="F")
        self.check(self._sym("mat_mul_ldr")(n, _p(M), _p(L), _p(d), _p(R), _p(Lo), _p(do), _p(Ro)))
        return Lo, do, Ro

    def ldr_mul_ldr(self, F1, F2):
        L1, d1, R1 = (_f64(x) for x in F1); L2, d2, R2 = (_f64(x) for x in F2); n = L1.shape[0]
        Lo, do, Ro = np.empty((n, n), order="F"), np.empty(n), np.empty((n, n), order="F")
        self.check(self._sym("ldr_mul_ldr")(n, _p(L1), _p(d1), _p(R1), _p(L2), _p(d2), _p(R2), _p(Lo), _p(do), _p(Ro)))
        return Lo, do, Ro

    def inv_I_plus_ldr(self, F) -> Tuple[np.ndarray, float]:
        L, d, R = (_f64(x) for x in F); n = L.shape[0]
        G = np.empty((n, n), order="F"); ld = C.c_double(0.0)
        self.check(self._sym("inv_I_plus_ldr")(n, _p(L), _p(d), _p(R), _p(G), C.byref(ld)))
        return G, ld.value

    def inv_I_plus_ldr_mul_ldr(self, F1, F2) -> np.ndarray:
        L1, d1, R1 = (_f64(x) for x in F1); L2, d2, R2 = (_f64(x) for x in F2); n = L1.shape[0]
        G = np.empty((n, n), order="F")
        self.check(self._sym("inv_I_plus_ldr_mul_ldr")(n, _p(L1), _p(d1), _p(R1), _p(L2), _p(d2), _p(R2), _p(G)))
        return G

    def gemm(self, A, B, transA: bool = False, transB: bool = False) -> np.ndarray:
        A = _f64(A); B = _f64(B); n = A.shape[0]
        Cm = np.empty((n, n), order="F")
        self.check(self._sym("gemm")(n, _p(A), int(transA), _p(B), int(transB), _p(Cm)))
        return Cm

    def rank1_update(self, G, i: int, delta: float) -> np.ndarray:
        G = np.array(G, dtype=np.float64, order="F", copy=True); n = G.shape[0]
        self.check(self._sym("rank1_update")(n, _p(G), int(i), float(delta)))
        return G

    def engine(self, *args, **kw) -> "Engine":
        return Engine(self, *args, **kw)

    # -- replica exchange: a dqmc_comm plays MPI_COMM_WORLD (source/update.cpp:47-117) --
    def partner_rank(self, rank: int, world: int, attempt: int) -> int:
        return int(self._sym("partner_rank")(rank, world, attempt))

    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(UNIQUE_ID_BYTES)
        self.check(self._sym("comm_unique_id")(buf))
        return buf.raw

    def comm_rccl(self, uid: bytes, world: int, rank: int, device: int) -> "Comm":
        h = C.c_void_p()
        self.check(self._sym("comm_create_rccl")(C.byref(h), C.c_char_p(uid), world, rank, device))
        return Comm(self, h)

    def comm_callbacks(self, world: int, rank: int, fn) -> "Comm":
        """fn(send: bytes, partner: int, tag: int) -> bytes: a blocking pairwise exchange (MPI_Sendrecv shape)."""
        def tramp(_user, send, recv, nbytes, partner, tag):
            try:
                got = fn(C.string_at(send, nbytes), int(partner), int(tag))
                if len(got) != nbytes:
                    return -2
                C.memmove(recv, got, nbytes)
                return 0
            except Exception:
                return -1
        cb = SENDRECV_FN(tramp)
        h = C.c_void_p()
        self.check(self._sym("comm_create_callbacks")(C.byref(h), world, rank, cb, None))
        return Comm(self, h, keep=cb)


class Comm:
    """Handle onto a dqmc_comm."""

    def __init__(self, lib: "DqmcLib", h, keep=None):
        self.lib, self._h, self._keep = lib, h, keep

    @property
    def rank(self) -> int:
        return int(self.lib._sym("comm_rank")(self._h))

    @property
    def world(self) -> int:
        return int(self.lib._sym("comm_world_size")(self._h))

    @property
    def transport(self) -> str:
        return self.lib._sym("comm_transport")(self._h).decode()

    def selftest(self):
        """dqmc_comm_selftest: loop-back of the RCCL transport (every rank calls it)."""
        self.lib.check(self.lib._sym("comm_selftest")(self._h))

    def barrier(self):
        self.lib.check(self.lib._sym("comm_barrier")(self._h))

    def allreduce_sum(self, x) -> np.ndarray:
        v = np.ascontiguousarray(x, dtype=np.float64).copy()
        self.lib.check(self.lib._sym("comm_allreduce_sum")(self._h, _p(v), int(v.size)))
        return v

    def exchange_round(self, engine: "Engine", attempt: int, u: float) -> ExchangeResult:
        """dqmc_replica_exchange_round: `attempt` is the counter after the reference's exchange_attempt++."""
        res = ExchangeResult()
        self.lib.check(self.lib._sym("replica_exchange_round")(engine._h, self._h, int(attempt), float(u), C.byref(res)))
        return res

    def close(self):
        if self._h is not None:
            self.lib._sym("comm_destroy")(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """Handle onto one dqmc_engine: class DQMC of the reference
    (include/dqmc.h:21-93) with the model/field state it reads.

    n_chains = None -> the single-chain ABI (dqmc_create); arrays have the
    reference's shapes.  n_chains = C -> dqmc_create_batch; every per-chain
    array gains a leading C dimension (g / expK / invexpK may be given once and
    are then replicated)."""

    def __init__(self, lib: DqmcLib, n_sites: int, nt: int, n_stab: int, g, gamma, eta,
                 expK, invexpK, device: int = 0, n_chains: Optional[int] = None):
        self.lib, self.n, self.nt, self.n_stab = lib, int(n_sites), int(nt), int(n_stab)
        self.batched = n_chains is not None
        self.C = int(n_chains) if self.batched else 1
        gamma = np.ascontiguousarray(gamma, dtype=np.float64); eta = np.ascontiguousarray(eta, dtype=np.float64)
        h = C.c_void_p()
        if not self.batched:
            eK, iK = _f64(expK), _f64(invexpK)
            assert eK.shape == (self.n, self.n) and iK.shape == (self.n, self.n)
            lib.check(lib._sym("create")(C.byref(h), device, self.n, self.nt, self.n_stab, float(g),
                                         _p(gamma), _p(eta), _p(eK), _p(iK)))
        else:
            gv = np.ascontiguousarray(np.broadcast_to(np.asarray(g, dtype=np.float64), (self.C,)))
            eK = self._mats_in(np.broadcast_to(np.asarray(expK), (self.C, self.n, self.n)) if np.ndim(expK) == 2 else expK)
            iK = self._mats_in(np.broadcast_to(np.asarray(invexpK), (self.C, self.n, self.n)) if np.ndim(invexpK) == 2 else invexpK)
            lib.check(lib._sym("create_batch")(C.byref(h), device, self.C, self.n, self.nt, self.n_stab, _p(gv),
                                               _p(gamma), _p(eta), _p(eK), _p(iK)))
        self._h: Optional[C.c_void_p] = h

    # (C, n, n) python-indexed [c, i, j]  <->  C consecutive column-major matrices
    def _mats_in(self, a) -> np.ndarray:
        a = np.asarray(a, dtype=np.float64).reshape(self.C, self.n, self.n)
        return np.ascontiguousarray(a.transpose(0, 2, 1))

    def _mats_out(self, buf: np.ndarray) -> np.ndarray:
        m = buf.reshape(self.C, self.n, self.n).transpose(0, 2, 1)
        return m if self.batched else np.asfortranarray(m[0])

    def _vec_out(self, buf: np.ndarray) -> np.ndarray:
        return buf.reshape(self.C, -1) if self.batched else buf.reshape(-1)

    def close(self):
        if self._h is not None:
            self.lib._sym("destroy")(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _c(self, name, *args):
        self.lib.check(self.lib._sym(name)(self._h, *args))

    # fields cross the ABI as arma::imat memory: nt x nv column-major int64
    def set_fields(self, fields):
        """fields: integers in {0,1,2,3}, shape (nt, n_sites) [(C, nt, n_sites) when batched]."""
        f = np.asarray(fields).reshape(self.C, self.nt, self.n)
        f = np.ascontiguousarray(f.transpose(0, 2, 1), dtype=np.int64)     # [c][i][l] == column-major nt x nv
        self._c("set_fields", f.ctypes.data_as(c_int64_p))

    def get_fields(self) -> np.ndarray:
        f = np.empty((self.C, self.n, self.nt), dtype=np.int64)
        self._c("get_fields", f.ctypes.data_as(c_int64_p))
        f = f.transpose(0, 2, 1)
        return f if self.batched else f[0]

    def init(self):
        self._c("init")

    def get_G(self) -> np.ndarray:
        G = np.empty(self.C * self.n * self.n); self._c("get_G", _p(G)); return self._mats_out(G)

    def set_G(self, G):
        G = self._mats_in(G); self._c("set_G", _p(G))

    def get_logdet(self):
        v = np.zeros(self.C); self._c("get_logdet", _p(v)); return v if self.batched else float(v[0])

    def n_stack(self) -> int:
        return int(self.lib._sym("n_stack")(self._h))

    def get_stack(self, i: int):
        L, d, R = np.empty(self.C * self.n * self.n), np.empty(self.C * self.n), np.empty(self.C * self.n * self.n)
        self._c("get_stack", int(i), _p(L), _p(d), _p(R))
        return self._mats_out(L), self._vec_out(d), self._mats_out(R)

    def _stream(self, perm, kprop, u, rows):
        perm = np.ascontiguousarray(np.asarray(perm).reshape(self.C, rows, self.n), dtype=np.int32)
        kprop = np.ascontiguousarray(np.asarray(kprop).reshape(self.C, rows, self.n), dtype=np.uint8)
        u = np.ascontiguousarray(np.asarray(u).reshape(self.C, rows, self.n), dtype=np.float64)
        return perm, kprop, u

    def sweep_0_to_beta(self, perm, kprop, u):
        perm, kprop, u = self._stream(perm, kprop, u, self.nt)
        self._c("sweep_0_to_beta", perm.ctypes.data_as(c_int32_p), kprop.ctypes.data_as(c_uint8_p), _p(u))

    def sweep_beta_to_0(self, perm, kprop, u):
        perm, kprop, u = self._stream(perm, kprop, u, self.nt)
        self._c("sweep_beta_to_0", perm.ctypes.data_as(c_int32_p), kprop.ctypes.data_as(c_uint8_p), _p(u))

    def sync(self):
        self._c("sync")

    def stats(self):
        arr = (Stats * self.C)(); self._c("get_stats", arr)
        return list(arr) if self.batched else arr[0]

    def wrap_forward(self, l: int):
        self._c("wrap_forward", int(l))

    def wrap_backward(self, l: int):
        self._c("wrap_backward", int(l))

    def local_update_slice(self, l: int, perm, kprop, u):
        perm, kprop, u = self._stream(perm, kprop, u, 1)
        acc = (C.c_int * self.C)()
        self._c("local_update_slice", int(l), perm.ctypes.data_as(c_int32_p), kprop.ctypes.data_as(c_uint8_p), _p(u), acc)
        return list(acc) if self.batched else acc[0]

    def calculate_Bbar(self, i_stack: int) -> np.ndarray:
        B = np.empty(self.C * self.n * self.n); self._c("calculate_Bbar", int(i_stack), _p(B)); return self._mats_out(B)

    def set_checkerboard(self, groups, cosh_t, sinh_t, diag_factor):
        """dqmc_set_checkerboard: `groups` = list of bond groups, each a sequence of (i, j) site pairs; the three
        parameters are scalars or one value per chain.  Call before init()."""
        sizes = np.ascontiguousarray([len(g) for g in groups], np.int32)
        flat = [p for g in groups for p in g]
        bonds = np.ascontiguousarray(np.asarray(flat, np.int32).reshape(-1, 2) if flat else np.zeros((1, 2), np.int32))
        par = [np.ascontiguousarray(np.broadcast_to(np.asarray(x, np.float64), (self.C,))) for x in (cosh_t, sinh_t, diag_factor)]
        self._c("set_checkerboard", len(groups), bonds.ctypes.data_as(c_int32_p), sizes.ctypes.data_as(c_int32_p), _p(par[0]), _p(par[1]), _p(par[2]))

    def global_action(self):
        v = np.zeros(self.C); self._c("global_action", _p(v)); return v if self.batched else float(v[0])

    # ---- equal-time observables ----
    def measure_equal_time(self, L1: int, L2: int):
        """(scalars[3] = density, doubleOcc, swave; chi_r[L1, L2]) of the current G; leading chain axis when batched."""
        sc = np.empty((self.C, 3)); chi = np.empty((self.C, L1 * L2))
        self._c("measure_equal_time", int(L1), int(L2), _p(sc), _p(chi))
        chi = chi.reshape(self.C, L2, L1).transpose(0, 2, 1)            # element (dx_idx, dy_idx) at dx_idx + L1*dy_idx
        return (sc, chi) if self.batched else (sc[0], chi[0])

    def measure_accumulate(self, L1: int, L2: int):
        self._c("measure_accumulate", int(L1), int(L2))

    def measure_fetch(self, L1: int, L2: int, reset: bool = True):
        sc = np.empty((self.C, 3)); chi = np.empty((self.C, L1 * L2)); cnt = np.zeros(1, dtype=np.int64)
        self._c("measure_fetch", _p(sc), _p(chi), cnt.ctypes.data_as(c_int64_p), 1 if reset else 0)
        chi = chi.reshape(self.C, L2, L1).transpose(0, 2, 1)
        return ((sc, chi) if self.batched else (sc[0], chi[0])) + (int(cnt[0]),)

    # ---- unequal-time path ----
    def sweep_unequal_time(self):
        self._c("sweep_unequal_time")

    def get_G_tau(self, which: str, l: int) -> np.ndarray:
        """which in {"tt", "t0", "0t"}: Gtt[l], Gt0[l] = G(tau_l, 0), G0t[l] = G(0, tau_l) of the last sweep_unequal_time."""
        B = np.empty(self.C * self.n * self.n); self._c("get_G_tau", {"tt": 0, "t0": 1, "0t": 2}[which], int(l), _p(B)); return self._mats_out(B)

    def half_warp(self, expK_half=None, invexpK_half=None, which: str = "G", l: int = 0) -> np.ndarray:
        """DQMC::half_warp (source/dqmc.cpp:288-315): invexpK_half @ M @ expK_half, M = the current G ("G") or Gtt / Gt0 / G0t[l]
        ("tt" / "t0" / "0t"); the matrices may be omitted after the first call."""
        a = _f64(expK_half) if expK_half is not None else None; b = _f64(invexpK_half) if invexpK_half is not None else None
        B = np.empty(self.C * self.n * self.n)
        self._c("half_warp", _p(a) if a is not None else None, _p(b) if b is not None else None, {"G": -1, "tt": 0, "t0": 1, "0t": 2}[which], int(l), _p(B))
        return self._mats_out(B)

    def _ut_cube(self, flat, L1, L2):
        a = flat.reshape(self.C, 3, self.nt + 1, L2, L1).transpose(0, 1, 2, 4, 3)      # [chain][obs][tau][dx_idx][dy_idx]
        return a if self.batched else a[0]

    def measure_unequal_time(self, L1: int, L2: int, accumulate: bool = False):
        """greenTau, doublonTau, currxxTau [3][nt + 1][L1][L2] of the last sweep_unequal_time (or, accumulate=True, add them to the bin)."""
        if accumulate:
            self._c("measure_unequal_time", int(L1), int(L2), 1, None); return None
        out = np.empty(self.C * 3 * (self.nt + 1) * L1 * L2)
        self._c("measure_unequal_time", int(L1), int(L2), 0, _p(out)); return self._ut_cube(out, L1, L2)

    def measure_unequal_fetch(self, L1: int, L2: int, reset: bool = True):
        out = np.empty(self.C * 3 * (self.nt + 1) * L1 * L2); cnt = np.zeros(1, dtype=np.int64)
        self._c("measure_unequal_fetch", _p(out), cnt.ctypes.data_as(c_int64_p), 1 if reset else 0)
        return self._ut_cube(out, L1, L2), int(cnt[0])

    def debug_snapshot(self):
        """dqmc_debug_snapshot of chain 0: dict(wrap_err[n_stack], accepted[nt] of the last half sweep, sync_words[80], slice_epoch)."""
        ns = self.n_stack()
        we = np.zeros(ns); acc = np.zeros(self.nt, np.int32); sw = np.zeros(80, np.uint32); ep = C.c_uint(0)
        self._c("debug_snapshot", _p(we), acc.ctypes.data, sw.ctypes.data, C.byref(ep))
        return dict(wrap_err=we, accepted=acc, sync_words=sw, slice_epoch=int(ep.value))

    def slice_path(self) -> int:
        """dqmc_slice_path: 0 = scan / flush kernel pairs, 1 = persistent single-launch slice kernel, 2 = persistent, and at least one
        launch fell back (solo walk / untouched slice) because a flush workgroup was not resident in time."""
        return int(self.lib._sym("slice_path")(self._h))

    def set_profiling(self, on: bool):
        self._c("set_profiling", int(bool(on)))

    def update_kernel_time(self):
        ms = C.c_double(0.0); nl = C.c_int64(0); na = C.c_int64(0)
        self._c("update_kernel_time", C.cast(C.byref(ms), c_double_p), C.cast(C.byref(nl), c_int64_p), C.cast(C.byref(na), c_int64_p))
        return ms.value, nl.value, na.value
