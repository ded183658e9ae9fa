"""Matrix-product states: the container that crosses the hot-path boundary, and the
host-side builder that produces the path's inputs.

``MPS`` mirrors what the reference uses of pytket-cutensornet's MPS object
(/root/reference/gpu_backend/kernel_state_ansatz.py:223,290,295-296,370,374,380):
``tensors``, ``get_virtual_dimensions``, ``fidelity``, ``copy``,
``update_libhandle``, ``len`` and ``vdot``.  Site tensors are complex128 numpy
arrays indexed ``[left bond, physical, right bond]``.

``simulate`` is the *input producer* (SURVEY.md section 8, row A8 / N1), not the hot
path: it applies the bound gate program to |0...0> gate by gate with one SVD per
two-qubit gate, truncating at ``truncation_fidelity`` exactly where the reference
does (ref :141-144, :221; criterion as ITensors ``cutoff``,
/root/reference/KernelPkg/src/KernelPkg.jl:68).  It runs on the host (LAPACK); a
device builder is the next row of the scope table.
"""
from __future__ import annotations

import os

import numpy as np
from scipy.linalg import qr as _qr
from scipy.linalg import svd as _svd

from .ansatz import OP_H, OP_RZ, OP_SWAP, OP_XX, BoundCircuit

_SQRT_HALF = 0.7071067811865476


class MPS:
    """n-site matrix-product state, open boundaries, physical dimension 2."""

    __slots__ = ("tensors", "fidelity", "_handle")

    def __init__(self, tensors, fidelity: float = 1.0):
        self.tensors = list(tensors)
        self.fidelity = float(fidelity)
        self._handle = None
        left = 1
        for k, t in enumerate(self.tensors):
            if t.ndim != 3 or t.shape[1] != 2 or t.shape[0] != left:
                raise RuntimeError(f"site {k}: tensor shape {t.shape} does not chain (left bond {left}, physical 2)")
            left = t.shape[2]
        if left != 1:
            raise RuntimeError("the last right bond must have dimension 1")

    def __len__(self) -> int:
        return len(self.tensors)

    def get_virtual_dimensions(self, position: int) -> tuple[int, int]:
        t = self.tensors[position]
        return (t.shape[0], t.shape[2])

    def bond_dims(self) -> np.ndarray:
        """chi[0..n]: chi[0] = chi[n] = 1."""
        return np.asarray([1] + [t.shape[2] for t in self.tensors], dtype=np.int32)

    def max_bond(self) -> int:
        return int(self.bond_dims().max())

    def nbytes(self) -> int:
        return int(sum(t.nbytes for t in self.tensors))

    def copy(self) -> "MPS":
        return MPS([t.copy() for t in self.tensors], self.fidelity)

    def update_libhandle(self, handle) -> None:
        """Accepted for interface parity (ref :370,:374); states are immutable host
        arrays here and device residency is owned by the engine's MPS sets."""
        self._handle = handle

    def vdot(self, other: "MPS") -> complex:
        """<self|other> through the HIP engine (single pair; the Gram path is batch-first)."""
        from .engine import default_context

        if len(other) != len(self):
            raise RuntimeError("the two MPS must have the same number of sites")
        ctx = default_context()
        with ctx.upload([self]) as xs, ctx.upload([other]) as ys:
            return complex(ctx.overlaps(xs, ys)[0, 0])


def random_mps(n_sites: int, bond_dims, rng) -> MPS:
    """Random normalised MPS with a prescribed bond profile ``bond_dims[0..n]`` (pure-kernel
    benchmark inputs): complex Gaussian tensors, right-orthonormalised by QR."""
    chi = [int(c) for c in bond_dims]
    assert len(chi) == n_sites + 1 and chi[0] == 1 and chi[-1] == 1
    ts = [
        (rng.standard_normal((chi[k], 2, chi[k + 1])) + 1j * rng.standard_normal((chi[k], 2, chi[k + 1])))
        for k in range(n_sites)
    ]
    for k in range(n_sites - 1, 0, -1):
        l, _, r = ts[k].shape
        q, rr = np.linalg.qr(ts[k].reshape(l, 2 * r).T)  # (2r, l) = q (2r, m) rr (m, l)
        m = q.shape[1]
        if m != l:
            raise ValueError(f"bond {k}: dimension {l} exceeds what its neighbours allow ({m})")
        ts[k] = q.T.reshape(l, 2, r)
        ts[k - 1] = np.tensordot(ts[k - 1], rr.T, axes=(2, 0))
    ts[0] = ts[0] / np.linalg.norm(ts[0])
    return MPS(ts)


def _kept(s: np.ndarray, discard_budget: float, zero: float) -> tuple[int, float]:
    """How many leading singular values survive, and the kept fraction of the weight.

    Values <= ``zero`` (absolute; the state is normalised) are dropped first, then the
    smallest values are dropped while their summed weight stays within
    ``discard_budget`` of the total -- both sums accumulated from the small end.
    """
    w = s * s
    total = float(w.sum())
    keep = int(np.count_nonzero(s > zero))
    keep = max(keep, 1)
    tail = np.cumsum(w[:keep][::-1])  # tail[m-1] = weight of the m smallest candidates
    drop = int(np.searchsorted(tail, discard_budget * total, side="right"))
    keep = max(keep - drop, 1)
    return keep, float(w[:keep].sum()) / total


_NATIVE = None  # ctypes handle of libqkbuilder.so, False once loading has failed


def _native_builder():
    """The native host builder (csrc/qk_builder.cpp), or None: it needs libqkbuilder.so (built by
    ``__graft_entry__.build()``) and the OpenBLAS that scipy ships (for zgesdd / zgeqrf / zungqr / zgemm)."""
    global _NATIVE
    if _NATIVE is None:
        import ctypes as C
        import glob
        import os

        _NATIVE = False
        try:
            import scipy

            here = os.path.dirname(os.path.abspath(__file__))
            blas = sorted(glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas*.so")))
            lib = C.CDLL(os.path.join(here, "libqkbuilder.so"))
            lib.qkb_last_error.restype = C.c_char_p
            lib.qkb_init.argtypes = [C.c_char_p]
            lib.qkb_simulate_chi.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int32,
                                             C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_double)]
            lib.qkb_free.argtypes = [C.c_void_p]
            if blas and lib.qkb_init(os.path.realpath(blas[0]).encode()) == 0:
                _NATIVE = lib
        except OSError:
            _NATIVE = False
    return _NATIVE or None


def simulate_native(circuit: BoundCircuit, truncation_fidelity: float = 1.0 - 1e-16, value_of_zero: float = 1e-16, max_bond: int | None = None) -> MPS:
    """``simulate`` through the native builder: same algorithm and LAPACK routines, no interpreter in the gate loop."""
    import ctypes as C

    lib = _native_builder()
    if lib is None:
        raise RuntimeError("native MPS builder unavailable (libqkbuilder.so or scipy's OpenBLAS not found)")
    n = circuit.n_qubits
    op = np.ascontiguousarray(circuit.op, dtype=np.int8)
    q0 = np.ascontiguousarray(circuit.q0, dtype=np.int32)
    alpha = np.ascontiguousarray(circuit.alpha, dtype=np.float64)
    dims = np.zeros(n + 1, dtype=np.int32)
    block, count, fid = C.c_void_p(), C.c_int64(), C.c_double()
    rc = lib.qkb_simulate_chi(n, int(op.shape[0]), op.ctypes.data, q0.ctypes.data, alpha.ctypes.data, max(0.0, 1.0 - float(truncation_fidelity)),
                              float(value_of_zero), int(max_bond or 0), dims.ctypes.data, C.byref(block), C.byref(count), C.byref(fid))
    if rc != 0:
        raise RuntimeError(f"native MPS builder failed: {lib.qkb_last_error().decode()}")
    try:
        flat = np.ctypeslib.as_array(C.cast(block, C.POINTER(C.c_double)), shape=(2 * count.value,)).view(np.complex128).copy()
    finally:
        lib.qkb_free(block)
    tensors, pos = [], 0
    for k in range(n):
        sz = int(dims[k]) * 2 * int(dims[k + 1])
        tensors.append(flat[pos : pos + sz].reshape(int(dims[k]), 2, int(dims[k + 1])))
        pos += sz
    return MPS(tensors, fid.value)


def simulate(circuit: BoundCircuit, truncation_fidelity: float = 1.0 - 1e-16, value_of_zero: float = 1e-16, max_bond: int | None = None) -> MPS:
    """MPS of circuit|0...0>, "MPSxGate" style: one SVD per two-qubit gate (ref :221).  ``max_bond``: at most that many singular
    values survive a gate -- the ``chi`` of pytket-cutensornet's ``Config`` (ref :141-144 is where it would go); the weight it
    costs goes into ``fidelity`` like any other truncation.

    The matrices are small (tens to a few hundred rows): a multi-threaded BLAS spends its time waking
    threads (measured 24 s instead of 0.9 s per 60-qubit state on 8 cores), so the LAPACK calls run
    single-threaded here; parallelism is across states (``builder_pool.build_states``)."""
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:  # pragma: no cover - optional dependency
        return _simulate(circuit, truncation_fidelity, value_of_zero, max_bond)
    with threadpool_limits(limits=1):
        if _use_native():
            return simulate_native(circuit, truncation_fidelity, value_of_zero, max_bond)
        return _simulate(circuit, truncation_fidelity, value_of_zero, max_bond)


def simulate_many(circuits, truncation_fidelity: float = 1.0 - 1e-16, value_of_zero: float = 1e-16, workers: int | None = None, progress=None, max_bond: int | None = None):
    """``simulate`` for a list of circuits on several host cores **without forking** (safe once the GPU is initialised).
    The first circuit is built in this process and sizes the job: long jobs go to one worker *process* per core
    (``builder_worker.py`` over pipes: 10.8x on 16 cores for cfg4-shaped circuits, where threads in one process reach 2x
    because concurrent LAPACK calls contend inside OpenBLAS), medium ones to a thread pool (the native builder is a ctypes
    call that releases the GIL), short ones stay serial.  QK_BUILDER_POOL=procs|threads|serial overrides the choice.
    Returns (list[MPS], seconds per circuit)."""
    import time

    circuits = list(circuits)
    if not circuits:
        return [], []
    if workers is None:
        from .builder_pool import default_workers

        workers = default_workers()
    workers = max(1, min(int(workers), len(circuits) - 1))

    def one(c):
        t0 = time.perf_counter()
        m = simulate(c, truncation_fidelity, value_of_zero, max_bond)
        if progress is not None:
            progress()
        return m, time.perf_counter() - t0

    first, t_first = one(circuits[0])
    rest = circuits[1:]
    estimate = t_first * len(rest)  # serial seconds still to do
    mode = os.environ.get("QK_BUILDER_POOL", "auto")
    if mode == "auto":
        mode = "procs" if estimate / max(1, workers) > 2.0 else ("threads" if estimate > 0.5 else "serial")
    if not rest or workers <= 1:
        mode = "serial"
    if mode == "threads" and not _use_native():
        mode = "serial"
    if mode == "procs":
        out, secs = _simulate_many_procs(rest, truncation_fidelity, value_of_zero, workers, progress, max_bond)
    elif mode == "threads":
        from concurrent.futures import ThreadPoolExecutor

        def one_native(c):  # simulate() would enter / leave the BLAS thread limit per call, racing across threads
            t0 = time.perf_counter()
            m = simulate_native(c, truncation_fidelity, value_of_zero, max_bond)
            if progress is not None:
                progress()
            return m, time.perf_counter() - t0

        try:
            from threadpoolctl import threadpool_limits
        except ImportError:  # pragma: no cover - optional dependency
            threadpool_limits = None
        limit = threadpool_limits(limits=1) if threadpool_limits else None
        try:
            with ThreadPoolExecutor(max_workers=workers) as pool:
                res = list(pool.map(one_native, rest))
        finally:
            if limit is not None:
                limit.restore_original_limits()
        out, secs = [m for m, _ in res], [dt for _, dt in res]
    else:
        res = [one(c) for c in rest]
        out, secs = [m for m, _ in res], [dt for _, dt in res]
    return [first] + out, [t_first] + secs


def _simulate_many_procs(circuits, truncation_fidelity, value_of_zero, workers, progress, max_bond=None):
    """One interpreter per core (builder_worker.py) fed over pipes by one thread each; tasks are handed out one at a time,
    so uneven circuits balance themselves."""
    import pickle
    import subprocess
    import sys
    import threading

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"import sys; sys.path.insert(0, {root!r}); import qml_cutensornet_amd.builder_worker as w; w.main()"
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, "-c", code], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env) for _ in range(workers)]
    results, errors = [None] * len(circuits), []
    lock, nxt = threading.Lock(), [0]

    def feed(pr):
        try:
            while True:
                with lock:
                    i = nxt[0]
                    nxt[0] += 1
                if i >= len(circuits) or errors:
                    break
                pickle.dump((i, circuits[i], truncation_fidelity, value_of_zero, max_bond), pr.stdin, protocol=4)
                pr.stdin.flush()
                idx, tensors, fid, dt, err = pickle.load(pr.stdout)
                if err is not None:
                    raise RuntimeError(f"builder worker failed on circuit {idx}: {err}")
                results[idx] = (MPS(tensors, fid), dt)
                if progress is not None:
                    progress()
        except Exception as exc:  # noqa: BLE001 - collected and re-raised in the caller's thread
            errors.append(exc)
        finally:
            try:
                pr.stdin.close()
            except OSError:
                pass

    threads = [threading.Thread(target=feed, args=(pr,), daemon=True) for pr in procs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for pr in procs:
        try:
            pr.wait(timeout=30)
        except subprocess.TimeoutExpired:
            pr.kill()
    if errors:
        raise RuntimeError(f"host builder pool: {errors[0]}")
    return [r[0] for r in results], [r[1] for r in results]


def _use_native() -> bool:
    """QK_NATIVE_BUILDER=0 forces the numpy/scipy loop; otherwise the native builder is used when it loads."""
    import os

    return os.environ.get("QK_NATIVE_BUILDER", "1") != "0" and _native_builder() is not None


def _simulate(circuit: BoundCircuit, truncation_fidelity: float, value_of_zero: float, max_bond: int | None = None) -> MPS:
    n = circuit.n_qubits
    budget = max(0.0, 1.0 - float(truncation_fidelity))
    A = []
    for _ in range(n):
        t = np.zeros((1, 2, 1), dtype=np.complex128)
        t[0, 0, 0] = 1.0
        A.append(t)
    ops = circuit.op.tolist()
    qs = circuit.q0.tolist()
    alphas = circuit.alpha.tolist()
    two_q_pos = [q for o, q in zip(ops, qs) if o in (OP_XX, OP_SWAP)]
    fidelity = 1.0
    centre = 0  # sites < centre are left-orthonormal, sites > centre right-orthonormal
    g2 = 0  # running index into two_q_pos

    for o, q, a in zip(ops, qs, alphas):
        if o == OP_H:
            t = A[q]
            A[q] = np.stack((t[:, 0] + t[:, 1], t[:, 0] - t[:, 1]), axis=1) * _SQRT_HALF
            continue
        if o == OP_RZ:
            th = 0.5 * np.pi * a
            ph = complex(np.cos(th), np.sin(th))
            t = A[q].copy()
            t[:, 0] *= ph.conjugate()
            t[:, 1] *= ph
            A[q] = t
            continue

        # ---- two-qubit gate on (q, q+1): bring the orthogonality centre onto the pair
        while centre < q:
            l, _, r = A[centre].shape
            qq, rr = _qr(A[centre].reshape(l * 2, r), mode="economic", check_finite=False)
            A[centre] = qq.reshape(l, 2, -1)
            A[centre + 1] = np.tensordot(rr, A[centre + 1], axes=(1, 0))
            centre += 1
        while centre > q + 1:
            l, _, r = A[centre].shape
            qq, rr = _qr(A[centre].reshape(l, 2 * r).T, mode="economic", check_finite=False)
            A[centre] = qq.T.reshape(-1, 2, r)
            A[centre - 1] = np.tensordot(A[centre - 1], rr.T, axes=(2, 0))
            centre -= 1

        l = A[q].shape[0]
        r = A[q + 1].shape[2]
        theta = np.tensordot(A[q], A[q + 1], axes=(2, 0))  # [l, p, p', r]
        if o == OP_SWAP:
            theta = theta.transpose(0, 2, 1, 3)
        else:  # XXPhase: cos(th) 1 - i sin(th) X(x)X
            th = 0.5 * np.pi * a
            theta = np.cos(th) * theta - 1j * np.sin(th) * theta[:, ::-1, ::-1, :]
        u, s, vh = _svd(
            np.ascontiguousarray(theta).reshape(l * 2, 2 * r),
            full_matrices=False,
            lapack_driver="gesdd",
            check_finite=False,
            overwrite_a=True,
        )
        keep, frac = _kept(s, budget, value_of_zero)
        if max_bond and keep > max_bond:  # the chi cap
            keep = int(max_bond)
            frac = float((s[:keep] ** 2).sum() / (s ** 2).sum())
        fidelity *= frac
        s = s[:keep]
        s = s / np.sqrt(float((s * s).sum()))
        g2 += 1
        nxt = two_q_pos[g2] if g2 < len(two_q_pos) else q
        if nxt >= q + 1 or (nxt == q and True):
            # leave the centre on q+1 (also right for a repeat on the same pair)
            A[q] = u[:, :keep].reshape(l, 2, keep)
            A[q + 1] = (s[:, None] * vh[:keep]).reshape(keep, 2, r)
            centre = q + 1
        else:
            A[q] = (u[:, :keep] * s[None, :]).reshape(l, 2, keep)
            A[q + 1] = vh[:keep].reshape(keep, 2, r)
            centre = q
    return MPS(A, fidelity)
