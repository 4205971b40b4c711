#!/usr/bin/env python3
"""Bin statistics of a run's results/data_<rank>.h5 files (the layout of the reference's MeasurementManager::saveToHDF5,
include/measurementh5.h:277-362, as written by dqmc_amd/host/results_h5.hpp).

What the reference's scripts/analysis.py does for the scalar observables -- concatenate the bins of all ranks, mean and standard
error -- without h5py, which this image lacks: datasets are read through the host library's libhdf5 binding.

    python scripts/results_summary.py [results_dir]
"""
import ctypes as C
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _lib():
    import dqmc_amd
    h = C.CDLL(dqmc_amd.HOST_LIB_PATH)
    h.dqmc_host_results_read.restype = C.c_longlong
    h.dqmc_host_results_read.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_char_p, C.c_int]
    return h


def read_dataset(h, path, dataset):
    """numpy array of a fp64 dataset, or None when it does not exist."""
    nd = C.c_int(0); dims = (C.c_ulonglong * 8)(); err = C.create_string_buffer(256)
    cnt = h.dqmc_host_results_read(path.encode(), dataset.encode(), C.byref(nd), dims, None, 0, err, 256)
    if cnt < 0:
        return None
    data = np.empty(cnt)
    h.dqmc_host_results_read(path.encode(), dataset.encode(), None, None, data.ctypes.data, cnt, err, 256)
    return data.reshape([dims[k] for k in range(nd.value)])


def load_bins(results_dir="results"):
    """{observable: array over all bins of all ranks}; scalars as numbers, equal-time / unequal-time data as arrays."""
    h = _lib()
    out = {}
    for path in sorted(glob.glob(os.path.join(results_dir, "data_*.h5"))):
        k = 0
        while True:
            first = read_dataset(h, path, f"/bin_{k}/scalar/density")
            if first is None:
                break
            for name in ("density", "doubleOcc", "swave"):
                out.setdefault("scalar/" + name, []).append(float(read_dataset(h, path, f"/bin_{k}/scalar/{name}")[0]))
            out.setdefault("equaltime/densityCorr", []).append(read_dataset(h, path, f"/bin_{k}/equaltime/densityCorr"))
            for name in ("greenTau", "doublonTau", "currxxTau"):
                d = read_dataset(h, path, f"/bin_{k}/unequaltime/{name}")
                if d is not None:
                    out.setdefault("unequaltime/" + name, []).append(d)
            k += 1
    return {k: np.asarray(v) for k, v in out.items()}


def mean_and_stderr(x):
    x = np.asarray(x, dtype=float)
    n = x.shape[0]
    return x.mean(axis=0), (x.std(axis=0, ddof=1) / np.sqrt(n) if n > 1 else np.zeros_like(x.mean(axis=0)))


def main():
    d = sys.argv[1] if len(sys.argv) > 1 else "results"
    bins = load_bins(d)
    if not bins:
        raise SystemExit(f"no data_*.h5 with bins under {d}")
    n = len(bins["scalar/density"])
    print(f"{n} bins from {len(glob.glob(os.path.join(d, 'data_*.h5')))} file(s)")
    for name in ("density", "doubleOcc", "swave"):
        m, e = mean_and_stderr(bins["scalar/" + name])
        print(f"  {name:10s} = {m:.8f} +- {e:.2e}")
    m, e = mean_and_stderr(bins["equaltime/densityCorr"])
    L1, L2 = m.shape[0], m.shape[1]
    print(f"  densityCorr(r = 0) = {m[L1 // 2 - 1, L2 // 2 - 1, 0]:.8f} +- {e[L1 // 2 - 1, L2 // 2 - 1, 0]:.2e}")
    if "unequaltime/greenTau" in bins:
        m, e = mean_and_stderr(bins["unequaltime/greenTau"])
        nt = m.shape[2] - 1
        for t in (0, nt // 2, nt):
            print(f"  greenTau(r = 0, tau index {t}) = {m[L1 // 2 - 1, L2 // 2 - 1, t]:.8f} +- {e[L1 // 2 - 1, L2 // 2 - 1, t]:.2e}")


if __name__ == "__main__":
    main()
