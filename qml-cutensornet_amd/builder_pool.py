"""Build many MPS on the host cores in parallel (input production, outside the hot path).

Stands in for the per-rank ``simulate`` loop of the reference
(/root/reference/gpu_backend/kernel_state_ansatz.py:213-231): every data point
is independent, so the points are dealt to a process pool with BLAS pinned to one
thread per worker.
"""
from __future__ import annotations

import multiprocessing as mp
import os
import time

import numpy as np

from .mps import MPS, simulate

_G = {}


def _init(ansatz, fidelity, max_bond=None):
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[v] = "1"
    try:
        from threadpoolctl import threadpool_limits

        _G["tp"] = threadpool_limits(1)
    except Exception:  # pragma: no cover - threadpoolctl is optional
        pass
    _G["ansatz"], _G["fid"], _G["chi"] = ansatz, fidelity, max_bond


def _one(x):
    t0 = time.perf_counter()
    m = simulate(_G["ansatz"].circuit_for_data(x), _G["fid"], max_bond=_G.get("chi"))
    return m.tensors, m.fidelity, time.perf_counter() - t0


def default_workers() -> int:
    """Host cores this process may really use: the affinity mask, clipped by the cgroup CPU
    quota and by QK_BUILD_WORKERS (default cap 16: a GPU box hands each GPU a 16-core share
    even though the affinity mask shows every core of the node)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    cap = int(os.environ.get("QK_BUILD_WORKERS", "16"))
    return max(1, min(n, cap))


def build_states(ansatz, X, truncation_fidelity, workers=None, max_bond=None):
    """Return (list[MPS], per-state build seconds) for the rows of ``X``."""
    X = np.asarray(X, dtype=np.float64)
    workers = default_workers() if workers is None else int(workers)
    workers = min(workers, len(X)) or 1
    if workers <= 1:
        _init(ansatz, truncation_fidelity, max_bond)
        res = [_one(x) for x in X]
    else:
        ctx = mp.get_context("fork")
        with ctx.Pool(workers, initializer=_init, initargs=(ansatz, truncation_fidelity, max_bond)) as pool:
            res = pool.map(_one, list(X), chunksize=1)
    return [MPS(t, f) for t, f, _ in res], [dt for _, _, dt in res]
