"""ctypes binding of the C ABI in ``include/qkgram.h`` (library: ``libqkgram.so``, built in-tree
by ``__graft_entry__.build()``).

This is the only door to the hot path.  There is no CPU fallback: if the library
is missing, or no gfx950 device is usable, every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from contextlib import contextmanager

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libqkgram.so")

QK_LAYOUT_LPR, QK_LAYOUT_LRP = 0, 1
QK_PLAN_SYMMETRIC = 1
QK_PLAN_QUADS = 2  # 2x2 blocks of pairs per workgroup (include/qkgram.h)
QK_PLAN_ORIENT = 4  # symmetric plans: list each pair in the cheaper order of contraction


class QkError(RuntimeError):
    pass


class QkStats(C.Structure):
    _fields_ = [
        ("pairs", C.c_int64),
        ("flops", C.c_double),
        ("padded_flops", C.c_double),
        ("bytes", C.c_double),
        ("kernel_ms", C.c_double),
        ("grid", C.c_int32),
        ("max_bond", C.c_int32),
        ("kernel", C.c_int32),
        ("precision", C.c_int32),
        ("second_pairs", C.c_int64),
        ("second_flops", C.c_double),
        ("second_padded_flops", C.c_double),
        ("second_bytes", C.c_double),
        ("second_ms", C.c_double),
        ("second_kernel", C.c_int32),
        ("queues", C.c_int32),
        ("tail_frac", C.c_double),
        ("second_tail_frac", C.c_double),
        ("derive_ms", C.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None

# every symbol include/qkgram.h declares: (name, restype, argtypes)
_P = C.c_void_p
_SIGNATURES = [
    ("qk_last_error", C.c_char_p, []),
    ("qk_device_count", C.c_int, []),
    ("qk_ctx_create", C.c_int, [C.c_int, C.POINTER(_P)]),
    ("qk_ctx_destroy", C.c_int, [_P]),
    ("qk_ctx_set_stream", C.c_int, [_P, _P]),
    ("qk_ctx_use_own_stream", C.c_int, [_P]),
    ("qk_ctx_synchronize", C.c_int, [_P]),
    ("qk_ctx_trim", C.c_int, [_P]),
    ("qk_mps_set_create", C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, C.c_int32, C.POINTER(_P)]),
    ("qk_mps_set_destroy", C.c_int, [_P]),
    ("qk_mps_set_info", C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    ("qk_mps_set_image", C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(_P), _P, _P]),
    ("qk_mps_set_copy_image", C.c_int, [_P, _P, C.c_int64]),
    ("qk_mps_set_from_packed", C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P, C.c_int64, C.POINTER(_P)]),
    ("qk_mps_set_precision", C.c_int, [_P]),
    ("qk_mps_set_to_f32", C.c_int, [_P, _P, C.POINTER(_P)]),
    ("qk_pack_state_size", C.c_int64, [C.c_int32, _P]),
    ("qk_pack_state", C.c_int, [C.c_int32, _P, _P, C.c_int32, _P, _P]),
    ("qk_plan_create", C.c_int, [C.c_int32, C.c_int32, _P, C.c_int32, _P, C.c_uint32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P)]),
    ("qk_plan_destroy", C.c_int, [_P]),
    ("qk_plan_num_pairs", C.c_int64, [_P]),
    ("qk_plan_total_pairs", C.c_int64, [_P]),
    ("qk_plan_max_pairs_per_rank", C.c_int64, [_P]),
    ("qk_plan_pairs", _P, [_P]),
    ("qk_plan_stats", C.c_int, [_P, C.POINTER(QkStats)]),
    ("qk_plan_first_run", C.c_int64, [_P]),
    ("qk_plan_queues", C.c_int, [_P, _P]),
    ("qk_plan_edge_sites", C.c_int32, [_P]),
    ("qk_plan_create_all", C.c_int, [C.c_int32, C.c_int32, _P, C.c_int32, _P, C.c_uint32, C.c_int32, _P]),
    ("qk_plan_cost", C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    ("qk_gram_values", C.c_int, [_P, _P, _P, _P, _P, _P]),
    ("qk_gram_values_host", C.c_int, [_P, _P, _P, _P, _P, _P]),
    ("qk_scatter", C.c_int, [_P, _P, _P, C.c_int64, _P, C.c_int64, C.c_int32]),
    ("qk_gram_host", C.c_int, [_P, _P, _P, _P, C.c_int64]),
    ("qk_overlaps_host", C.c_int, [_P, _P, _P, _P]),
    ("qk_get_stats", C.c_int, [_P, C.POINTER(QkStats)]),
    ("qk_kernel_name", C.c_char_p, [C.c_int32, C.c_int32]),
    ("qk_selftest_mfma", C.c_int, [_P]),
    ("qk_build_mps", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_double, C.c_double, C.c_int32, C.c_uint32, C.POINTER(_P)]),
    ("qk_built_info", C.c_int, [_P, _P, _P, _P, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    ("qk_built_download", C.c_int, [_P, _P]),
    ("qk_built_destroy", C.c_int, [_P]),
    ("qk_mps_set_from_built", C.c_int, [_P, _P, C.POINTER(_P)]),
    ("qk_debug_jacobi", C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    ("qk_debug_jacobi_precond", C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    ("qk_range_push", C.c_int, [C.c_char_p]),
    ("qk_range_pop", C.c_int, []),
    ("qk_comm_init_all", C.c_int, [C.c_int32, _P, C.POINTER(_P)]),
    ("qk_comm_destroy", C.c_int, [_P]),
    ("qk_comm_size", C.c_int32, [_P]),
    ("qk_comm_ctx", _P, [_P, C.c_int32]),
    ("qk_mps_set_allgather", C.c_int, [_P, _P, _P, C.c_int32, _P]),
    ("qk_gram_sharded", C.c_int, [_P, _P, _P, _P, C.c_int64]),
    ("qk_comm_device_gram", C.c_int, [_P, C.c_int32, C.POINTER(_P)]),
    ("qk_comm_stats", C.c_int, [_P, C.c_int32, C.POINTER(QkStats), C.POINTER(C.c_double)]),
]
EXPORTED_SYMBOLS = [s[0] for s in _SIGNATURES]
# entry points of lab/qk_lab.h: only the lab library (lab/libqklab.so, loaded by lab/tools via use_lab_library()) has them
_LAB_SIGNATURES = [
    ("qk_debug_profile", C.c_int, [_P, _P]),
    ("qk_debug_mma_bench", C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
]
LAB_LIB_PATH = os.path.join(os.path.dirname(_HERE), "lab", "libqklab.so")


def use_lab_library(path=None):
    """lab/tools only: load the lab library (experimental kernels, QK_VARIANT, instrumented builds) instead of the
    shipped one.  Must be called before the first use of the engine."""
    global LIB_PATH
    if _lib is not None:
        raise QkError("use_lab_library() must be called before the library is loaded")
    LIB_PATH = path or os.environ.get("QK_LIB") or LAB_LIB_PATH


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own ``libamdhip64.so``
    (SONAME libamdhip64.so.7, looked up by file name from libtorch_hip.so); libqkgram.so needs
    ``libamdhip64.so.7``.  Loaded in the wrong order the process ends up with two runtimes and
    the second one sees no device.  Loading torch's copy first (if torch is installed) makes
    both resolve to the same object; without torch the system runtime is used."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    """Load ``libqkgram.so`` (once).  Raises ``QkError`` if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise QkError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the Gram path."
            )
        _preload_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, res, args in _SIGNATURES:
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        for name, res, args in _LAB_SIGNATURES:  # present in the lab library only
            if hasattr(L, name):
                f = getattr(L, name)
                f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def _check(rc: int, what: str):
    if rc != 0:
        msg = lib().qk_last_error()
        raise QkError(f"{what}: error {rc}: {msg.decode() if msg else '?'}")


def device_count() -> int:
    return int(lib().qk_device_count())


def range_push(name: str) -> None:
    """Open a roctx range (visible to ``rocprofv3 --marker-trace``; a no-op unless a profiler is attached or QK_ROCTX=1)."""
    lib().qk_range_push(name.encode())


def range_pop() -> None:
    lib().qk_range_pop()


def _dims_table(states) -> np.ndarray:
    return np.ascontiguousarray(np.stack([np.asarray(m.bond_dims(), dtype=np.int32) for m in states]))


def pack_state(mps, layout=QK_LAYOUT_LPR):
    """Host-only: the padded split-plane image of one MPS and its per-site offsets."""
    L = lib()
    dims = np.ascontiguousarray(mps.bond_dims(), dtype=np.int32)
    n = len(mps)
    tens = [np.ascontiguousarray(t, dtype=np.complex128) for t in mps.tensors]
    ptrs = (C.c_void_p * n)(*[t.ctypes.data for t in tens])
    size = L.qk_pack_state_size(n, dims.ctypes.data)
    out = np.empty(size, dtype=np.float64)
    offs = np.empty(n, dtype=np.int64)
    _check(L.qk_pack_state(n, dims.ctypes.data, ptrs, layout, out.ctypes.data, offs.ctypes.data), "qk_pack_state")
    return out, offs


class Plan:
    """Ordered share of the Gram's (x, y) pairs for one rank (host object)."""

    def __init__(self, x_dims, y_dims=None, world_size=1, rank=0, block=0, quads=False, orient=None):
        L = lib()
        xd = np.ascontiguousarray(x_dims, dtype=np.int32)
        self.symmetric = y_dims is None
        yd = None if self.symmetric else np.ascontiguousarray(y_dims, dtype=np.int32)
        self.nx = xd.shape[0]
        self.ny = self.nx if self.symmetric else yd.shape[0]
        n_sites = xd.shape[1] - 1
        if orient is None:  # default: on (QK_PLAN_ORIENT=0 lists every pair of a symmetric plan as i <= j)
            orient = os.environ.get("QK_PLAN_ORIENT", "1") != "0"
        self.orient = bool(orient) and self.symmetric and not quads
        h = _P()
        _check(
            L.qk_plan_create(
                n_sites, self.nx, xd.ctypes.data, self.ny, None if yd is None else yd.ctypes.data,
                (QK_PLAN_SYMMETRIC if self.symmetric else 0) | (QK_PLAN_QUADS if quads else 0) | (QK_PLAN_ORIENT if self.orient else 0), world_size, rank, block, C.byref(h),
            ),
            "qk_plan_create",
        )
        self._h = h
        self.world_size, self.rank, self.quads = world_size, rank, bool(quads)

    @classmethod
    def create_all(cls, x_dims, y_dims=None, world_size=1):
        """The plans of all ``world_size`` ranks from ONE cost pass (``qk_plan_create_all``): equal to ``[Plan(x_dims, y_dims,
        world_size, r) for r in range(world_size)]`` at a fraction of the host time."""
        xd = np.ascontiguousarray(x_dims, dtype=np.int32)
        sym = y_dims is None
        yd = None if sym else np.ascontiguousarray(y_dims, dtype=np.int32)
        orient = sym and os.environ.get("QK_PLAN_ORIENT", "1") != "0"
        out = (_P * int(world_size))()
        _check(lib().qk_plan_create_all(xd.shape[1] - 1, xd.shape[0], xd.ctypes.data, xd.shape[0] if sym else yd.shape[0], None if sym else yd.ctypes.data,
                                        (QK_PLAN_SYMMETRIC if sym else 0) | (QK_PLAN_ORIENT if orient else 0), int(world_size), out), "qk_plan_create_all")
        plans = []
        for r in range(int(world_size)):
            p = cls.__new__(cls)
            p.symmetric, p.nx, p.ny = sym, xd.shape[0], xd.shape[0] if sym else yd.shape[0]
            p.orient, p._h, p.world_size, p.rank, p.quads = orient, _P(out[r]), int(world_size), r, False
            plans.append(p)
        return plans

    @property
    def handle(self):
        return self._h

    @property
    def num_pairs(self) -> int:
        return int(lib().qk_plan_num_pairs(self._h))

    @property
    def total_pairs(self) -> int:
        return int(lib().qk_plan_total_pairs(self._h))

    @property
    def max_pairs_per_rank(self) -> int:
        return int(lib().qk_plan_max_pairs_per_rank(self._h))

    def pairs(self) -> np.ndarray:
        n = self.num_pairs
        if n == 0:
            return np.zeros((0, 2), dtype=np.int32)
        ptr = lib().qk_plan_pairs(self._h)
        buf = (C.c_int32 * (2 * n)).from_address(ptr)
        return np.frombuffer(buf, dtype=np.int32).reshape(n, 2).copy()

    @property
    def first_run(self) -> int:
        """Pairs [first_run, num_pairs) are the run the site-fused sweep takes with its two-workgroups-per-CU shape."""
        return int(lib().qk_plan_first_run(self._h))

    def cost(self) -> dict:
        """Host cost of making this plan and the tile-reuse lower bound on its bytes (qk_plan_cost)."""
        ms, th, tr = C.c_double(0), C.c_int32(0), C.c_double(0)
        _check(lib().qk_plan_cost(self._h, C.byref(ms), C.byref(th), C.byref(tr)), "qk_plan_cost")
        return {"plan_ms": ms.value, "threads": int(th.value), "tile_reuse_bytes": tr.value}

    @property
    def edge_sites(self) -> int:
        """Sites at either end of the chain that the site-fused sweep takes from the sets' edge blocks (0: none)."""
        return int(lib().qk_plan_edge_sites(self._h))

    def queues(self):
        """(number of device work queues, qstart[17]): queue s = pairs [qstart[s], qstart[s+1]) -- 8 per run of the list, one
        per XCD (include/qkgram.h: qk_plan_queues); 1 queue = the flat cost-ordered list."""
        qs = np.zeros(17, dtype=np.int64)
        return int(lib().qk_plan_queues(self._h, qs.ctypes.data)), qs

    def stats(self) -> dict:
        st = QkStats()
        _check(lib().qk_plan_stats(self._h, C.byref(st)), "qk_plan_stats")
        return st.as_dict()

    def close(self):
        if self._h:
            lib().qk_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MpsSet:
    """Device-resident list of MPS (opaque handle)."""

    def __init__(self, ctx, handle, dims):
        self.ctx, self._h, self.dims = ctx, handle, dims
        ctx._adopt(self)  # the context closes the sets that are still alive before it goes (their handles point into it)

    @property
    def handle(self):
        return self._h

    def __len__(self):
        return self.dims.shape[0]

    def info(self) -> dict:
        a, b, c, d = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
        _check(lib().qk_mps_set_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)), "qk_mps_set_info")
        return {"n_states": a.value, "n_sites": b.value, "max_padded_bond": c.value, "device_bytes": d.value}

    def image(self):
        """The fp64 device image of the set: (number of doubles, device address of the planes, true bonds
        [n_states, n_sites + 1], re-plane offsets in doubles [n_states, n_sites])."""
        n, ptr = C.c_int64(), _P()
        ns, nsites = self.dims.shape[0], self.dims.shape[1] - 1
        dims = np.zeros((ns, nsites + 1), dtype=np.int32)
        offs = np.zeros((ns, nsites), dtype=np.int64)
        _check(lib().qk_mps_set_image(self._h, C.byref(n), C.byref(ptr), dims.ctypes.data, offs.ctypes.data), "qk_mps_set_image")
        return int(n.value), int(ptr.value or 0), dims, offs

    def copy_image(self, dst_ptr: int, n_doubles: int):
        """Copy the planes into a device or host buffer of ``n_doubles`` doubles (the send buffer of the all-gather)."""
        _check(lib().qk_mps_set_copy_image(self._h, _P(dst_ptr), int(n_doubles)), "qk_mps_set_copy_image")

    @property
    def precision(self) -> int:
        """Bits of a real of the device image: 64 (complex128) or 32 (complex64)."""
        return int(lib().qk_mps_set_precision(self._h))

    def to_f32(self) -> "MpsSet":
        """A complex64 copy of this set on the same device; sweeps over fp32 sets run the
        fp32-MFMA kernel.  Both sets of a Gram must have the same precision."""
        h = _P()
        _check(lib().qk_mps_set_to_f32(self.ctx.handle, self._h, C.byref(h)), "qk_mps_set_to_f32")
        return MpsSet(self.ctx, h, self.dims)

    def close(self):
        if self._h:
            lib().qk_mps_set_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One engine context per device (what ``CuTensorNetHandle(device_id)`` is to the reference,
    /root/reference/gpu_backend/kernel_state_ansatz.py:213)."""

    def __init__(self, device_id: int = 0):
        h = _P()
        _check(lib().qk_ctx_create(int(device_id), C.byref(h)), "qk_ctx_create")
        self._h, self.device_id = h, int(device_id)
        self._sets = weakref.WeakSet()

    def _adopt(self, mps_set):
        if not hasattr(self, "_sets"):
            self._sets = weakref.WeakSet()
        self._sets.add(mps_set)

    def _close_sets(self):
        """Destroy every MpsSet made on this context that is still alive: ``qk_mps_set_destroy`` dereferences the set's context,
        so a set must never outlive it (a later ``close()`` / ``__del__`` of such a set is then a no-op)."""
        for s in list(getattr(self, "_sets", ())):
            s.close()

    @property
    def handle(self):
        return self._h

    def set_stream(self, hip_stream: int | None):
        """Use exactly this hipStream_t (0 = HIP's null stream = torch's default stream); ``None``
        goes back to the context's private stream."""
        if hip_stream is None:
            _check(lib().qk_ctx_use_own_stream(self._h), "qk_ctx_use_own_stream")
        else:
            _check(lib().qk_ctx_set_stream(self._h, _P(int(hip_stream))), "qk_ctx_set_stream")

    def synchronize(self):
        _check(lib().qk_ctx_synchronize(self._h), "qk_ctx_synchronize")

    def trim(self):
        """Release the device memory the context keeps between calls (sweep scratch, the device builder's arena and workspace)."""
        _check(lib().qk_ctx_trim(self._h), "qk_ctx_trim")

    def debug_mma_bench(self, which, wgs_per_cu, reps=2000):
        out = C.c_double()
        _check(lib().qk_debug_mma_bench(self._h, which, wgs_per_cu, reps, C.byref(out)), "qk_debug_mma_bench")
        return out.value

    def debug_profile(self):
        out = (C.c_uint64 * 8)()
        _check(lib().qk_debug_profile(self._h, out), "qk_debug_profile")
        return list(out)

    def selftest(self):
        _check(lib().qk_selftest_mfma(self._h), "qk_selftest_mfma")

    def debug_jacobi(self, a):
        """The device builder's one-sided Jacobi on one matrix: returns (A V, V, column norms, order by decreasing norm)."""
        a = np.array(a, dtype=np.complex128, order="C")
        p, q = a.shape
        v = np.zeros((q, q), dtype=np.complex128)
        sig = np.zeros(q, dtype=np.float64)
        order = np.zeros(q, dtype=np.int32)
        _check(lib().qk_debug_jacobi(self._h, p, q, a.ctypes.data, v.ctypes.data, sig.ctypes.data, order.ctypes.data), "qk_debug_jacobi")
        return a, v, sig, order

    def debug_jacobi_precond(self, a):
        """The builder's preconditioned block factorisation on one matrix (p x q, 16 <= q): (W = A V, V, sig, ord, sweeps, ms)."""
        a = np.ascontiguousarray(a, dtype=np.complex128).copy()
        p, q = a.shape
        v = np.zeros((q, q), dtype=np.complex128)
        sig = np.zeros(q, dtype=np.float64)
        ord_ = np.zeros(q, dtype=np.int32)
        st = np.zeros(6, dtype=np.int32)
        _check(lib().qk_debug_jacobi_precond(self._h, p, q, a.ctypes.data, v.ctypes.data, sig.ctypes.data, ord_.ctypes.data, st.ctypes.data), "qk_debug_jacobi_precond")
        self.last_precond_ms = [float(np.uint32(t)) / 1e5 for t in st[1:]]  # all, sort + copy, Gram-Schmidt, sweeps, V and W = A V
        return a, v, sig, ord_, int(st[0]), self.last_precond_ms[0]

    def build_mps(self, circuits, truncation_fidelity: float = 1.0 - 1e-16, value_of_zero: float = 1e-16, max_bond: int = 256, partial: bool = False, truncate: bool = False):
        """Device MPS builder (/root/reference/gpu_backend/kernel_state_ansatz.py:221, 263): the MPS of every
        bound circuit of the list (``ansatz.BoundCircuit``; they must share one gate structure, as the data points of one
        ansatz do) in ONE launch.  Returns (list[MPS], info) with info = {"kernel_ms", "total_complex", "dropped"}.  With
        ``partial`` a state that outgrows ``max_bond`` does not fail the call: its entry in the list is ``None`` and its index is
        in info["dropped"] (build it with the host builder)."""
        from .mps import MPS

        circuits = list(circuits)
        if not circuits:
            raise QkError("build_mps needs at least one circuit")
        c0 = circuits[0]
        op = np.ascontiguousarray(c0.op, dtype=np.int8)
        q0 = np.ascontiguousarray(c0.q0, dtype=np.int32)
        for c in circuits[1:]:
            if c.n_qubits != c0.n_qubits or not np.array_equal(c.op, c0.op) or not np.array_equal(c.q0, c0.q0):
                raise QkError("build_mps: the circuits of one call must share their gate structure")
        alpha = np.ascontiguousarray(np.stack([np.asarray(c.alpha, dtype=np.float64) for c in circuits]))
        n, ns = int(c0.n_qubits), len(circuits)
        h = _P()
        _check(lib().qk_build_mps(self._h, ns, n, int(op.shape[0]), op.ctypes.data, q0.ctypes.data, alpha.ctypes.data,
                                  max(0.0, 1.0 - float(truncation_fidelity)), float(value_of_zero), int(max_bond), (1 if partial else 0) | (2 if truncate else 0), C.byref(h)), "qk_build_mps")
        try:
            dims = np.zeros((ns, n + 1), dtype=np.int32)
            fid = np.zeros(ns, dtype=np.float64)
            offs = np.zeros(ns, dtype=np.int64)
            total, ms = C.c_int64(), C.c_double()
            _check(lib().qk_built_info(h, dims.ctypes.data, fid.ctypes.data, offs.ctypes.data, C.byref(total), C.byref(ms)), "qk_built_info")
            flat = np.empty(total.value, dtype=np.complex128)
            _check(lib().qk_built_download(h, flat.ctypes.data), "qk_built_download")
        finally:
            lib().qk_built_destroy(h)
        states, dropped = [], []
        for s_ in range(ns):
            if fid[s_] < 0:
                states.append(None)
                dropped.append(s_)
                continue
            pos, tensors = int(offs[s_]), []
            for k in range(n):
                sz = int(dims[s_, k]) * 2 * int(dims[s_, k + 1])
                tensors.append(flat[pos : pos + sz].reshape(int(dims[s_, k]), 2, int(dims[s_, k + 1])))
                pos += sz
            states.append(MPS(tensors, float(fid[s_])))
        return states, {"kernel_ms": ms.value, "total_complex": int(total.value), "dropped": dropped}

    def build_mps_set(self, circuits, truncation_fidelity: float = 1.0 - 1e-16, value_of_zero: float = 1e-16, max_bond: int = 256, truncate: bool = False):
        """Like ``build_mps`` but the states never leave the device: returns (MpsSet, info) with info = {"kernel_ms", "dims",
        "fidelity"}; the set feeds ``gram`` / ``gram_values`` directly."""
        circuits = list(circuits)
        if not circuits:
            raise QkError("build_mps_set needs at least one circuit")
        c0 = circuits[0]
        op = np.ascontiguousarray(c0.op, dtype=np.int8)
        q0 = np.ascontiguousarray(c0.q0, dtype=np.int32)
        for c in circuits[1:]:
            if c.n_qubits != c0.n_qubits or not np.array_equal(c.op, c0.op) or not np.array_equal(c.q0, c0.q0):
                raise QkError("build_mps_set: the circuits of one call must share their gate structure")
        alpha = np.ascontiguousarray(np.stack([np.asarray(c.alpha, dtype=np.float64) for c in circuits]))
        n, ns = int(c0.n_qubits), len(circuits)
        h, hs = _P(), _P()
        _check(lib().qk_build_mps(self._h, ns, n, int(op.shape[0]), op.ctypes.data, q0.ctypes.data, alpha.ctypes.data,
                                  max(0.0, 1.0 - float(truncation_fidelity)), float(value_of_zero), int(max_bond), 2 if truncate else 0, C.byref(h)), "qk_build_mps")
        try:
            dims = np.zeros((ns, n + 1), dtype=np.int32)
            fid = np.zeros(ns, dtype=np.float64)
            ms = C.c_double()
            _check(lib().qk_built_info(h, dims.ctypes.data, fid.ctypes.data, None, None, C.byref(ms)), "qk_built_info")
            _check(lib().qk_mps_set_from_built(self._h, h, C.byref(hs)), "qk_mps_set_from_built")
        finally:
            lib().qk_built_destroy(h)
        return MpsSet(self, hs, dims), {"kernel_ms": ms.value, "dims": dims, "fidelity": fid}

    def set_from_packed(self, dims_true, offsets, planes_ptr: int, n_doubles: int) -> MpsSet:
        """A set assembled from packed images (``MpsSet.image`` of several ranks, gathered into one buffer on the device or
        on the host): see ``qk_mps_set_from_packed``."""
        dims = np.ascontiguousarray(dims_true, dtype=np.int32)
        offs = np.ascontiguousarray(offsets, dtype=np.int64)
        h = _P()
        _check(lib().qk_mps_set_from_packed(self._h, dims.shape[0], dims.shape[1] - 1, dims.ctypes.data, offs.ctypes.data, _P(planes_ptr), int(n_doubles), C.byref(h)),
               "qk_mps_set_from_packed")
        return MpsSet(self, h, dims)

    def build_share(self, circuits, truncation_fidelity: float = 1.0 - 1e-16, value_of_zero: float = 1e-16, max_bond: int = 256, partial: bool = False, truncate: bool = False):
        """Device builder for one rank's share of a data set, the states staying on the device whenever possible: returns
        (MpsSet, None, info) when every state fitted ``max_bond`` (packed on the device by ``qk_mps_set_from_built``: nothing
        is downloaded), else (None, list[MPS | None], info) with the dropped states ``None`` (``partial`` only), to be
        completed by the host builder.  info = {"kernel_ms", "dims", "fidelity", "dropped"}."""
        from .mps import MPS

        circuits = list(circuits)
        if not circuits:
            raise QkError("build_share needs at least one circuit")
        c0 = circuits[0]
        op = np.ascontiguousarray(c0.op, dtype=np.int8)
        q0 = np.ascontiguousarray(c0.q0, dtype=np.int32)
        for c in circuits[1:]:
            if c.n_qubits != c0.n_qubits or not np.array_equal(c.op, c0.op) or not np.array_equal(c.q0, c0.q0):
                raise QkError("build_share: the circuits of one call must share their gate structure")
        alpha = np.ascontiguousarray(np.stack([np.asarray(c.alpha, dtype=np.float64) for c in circuits]))
        n, ns = int(c0.n_qubits), len(circuits)
        h = _P()
        _check(lib().qk_build_mps(self._h, ns, n, int(op.shape[0]), op.ctypes.data, q0.ctypes.data, alpha.ctypes.data,
                                  max(0.0, 1.0 - float(truncation_fidelity)), float(value_of_zero), int(max_bond), (1 if partial else 0) | (2 if truncate else 0), C.byref(h)), "qk_build_mps")
        try:
            dims = np.zeros((ns, n + 1), dtype=np.int32)
            fid = np.zeros(ns, dtype=np.float64)
            offs = np.zeros(ns, dtype=np.int64)
            total, ms = C.c_int64(), C.c_double()
            _check(lib().qk_built_info(h, dims.ctypes.data, fid.ctypes.data, offs.ctypes.data, C.byref(total), C.byref(ms)), "qk_built_info")
            dropped = [int(k) for k in np.nonzero(fid < 0)[0]]
            info = {"kernel_ms": ms.value, "dims": dims, "fidelity": fid, "dropped": dropped}
            if not dropped:
                hs = _P()
                _check(lib().qk_mps_set_from_built(self._h, h, C.byref(hs)), "qk_mps_set_from_built")
                return MpsSet(self, hs, dims), None, info
            flat = np.empty(total.value, dtype=np.complex128)
            _check(lib().qk_built_download(h, flat.ctypes.data), "qk_built_download")
        finally:
            lib().qk_built_destroy(h)
        states = []
        for s_ in range(ns):
            if fid[s_] < 0:
                states.append(None)
                continue
            pos, tensors = int(offs[s_]), []
            for k in range(n):
                sz = int(dims[s_, k]) * 2 * int(dims[s_, k + 1])
                tensors.append(flat[pos : pos + sz].reshape(int(dims[s_, k]), 2, int(dims[s_, k + 1])))
                pos += sz
            states.append(MPS(tensors, float(fid[s_])))
        return None, states, info

    def upload(self, states, layout=QK_LAYOUT_LPR) -> MpsSet:
        states = list(states)
        if not states:
            raise QkError("cannot upload an empty list of MPS")
        n_sites = len(states[0])
        if any(len(m) != n_sites for m in states):
            raise QkError("all MPS of a set must have the same number of sites")
        dims = _dims_table(states)
        keep = [[np.ascontiguousarray(t, dtype=np.complex128) for t in m.tensors] for m in states]
        flat = [t.ctypes.data for ts in keep for t in ts]
        ptrs = (C.c_void_p * len(flat))(*flat)
        h = _P()
        _check(
            lib().qk_mps_set_create(self._h, len(states), n_sites, dims.ctypes.data, ptrs, layout, C.byref(h)),
            "qk_mps_set_create",
        )
        return MpsSet(self, h, dims)

    def gram_values(self, xset: MpsSet, yset: MpsSet | None, plan: Plan, values_ptr: int, z_ptr: int | None = None):
        """Asynchronous: enqueue the sweep of ``plan``'s pairs; device pointers are plain addresses."""
        _check(
            lib().qk_gram_values(self._h, xset.handle, None if yset is None else yset.handle, plan.handle, _P(values_ptr), _P(z_ptr) if z_ptr else None),
            "qk_gram_values",
        )

    def gram_values_host(self, xset: MpsSet, yset: MpsSet | None, plan: Plan, want_z: bool = False):
        """Synchronous sweep of ``plan``'s pairs; returns |z|^2 (and z if asked) as host arrays."""
        n = plan.num_pairs
        vals = np.zeros(n, dtype=np.float64)
        z = np.zeros((n, 2), dtype=np.float64) if want_z else None
        _check(
            lib().qk_gram_values_host(self._h, xset.handle, None if yset is None else yset.handle, plan.handle,
                                      vals.ctypes.data, None if z is None else z.ctypes.data),
            "qk_gram_values_host",
        )
        return (vals, z[:, 0] + 1j * z[:, 1]) if want_z else vals

    def scatter(self, pairs_ptr: int, values_ptr: int, n: int, k_ptr: int, ld: int, mirror: bool):
        _check(lib().qk_scatter(self._h, _P(pairs_ptr), _P(values_ptr), int(n), _P(k_ptr), int(ld), 1 if mirror else 0), "qk_scatter")

    def gram(self, xset: MpsSet, yset: MpsSet | None = None) -> np.ndarray:
        """Synchronous whole Gram: rows = Y (or X), cols = X."""
        nx = len(xset)
        ny = nx if yset is None else len(yset)
        out = np.zeros((ny, nx), dtype=np.float64)
        _check(lib().qk_gram_host(self._h, xset.handle, None if yset is None else yset.handle, out.ctypes.data, nx), "qk_gram_host")
        return out

    def overlaps(self, xset: MpsSet, yset: MpsSet | None = None) -> np.ndarray:
        """Synchronous complex overlaps z[j, i] = <x_i|y_j>."""
        nx = len(xset)
        ny = nx if yset is None else len(yset)
        out = np.zeros((ny, nx, 2), dtype=np.float64)
        _check(lib().qk_overlaps_host(self._h, xset.handle, None if yset is None else yset.handle, out.ctypes.data), "qk_overlaps_host")
        return out[..., 0] + 1j * out[..., 1]

    def stats(self) -> dict:
        st = QkStats()
        _check(lib().qk_get_stats(self._h, C.byref(st)), "qk_get_stats")
        d = st.as_dict()
        d["kernel_name"] = lib().qk_kernel_name(st.kernel, st.precision).decode()
        d["second_kernel_name"] = lib().qk_kernel_name(st.second_kernel, st.precision).decode() if st.second_kernel else ""
        return d

    def close(self):
        if self._h:
            self._close_sets()
            if not getattr(self, "_borrowed", False):  # a communicator's contexts die with the communicator
                lib().qk_ctx_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class Comm:
    """The multi-GPU part of the C ABI (``qk_comm_*``): ONE process drives k MI355X of a node; RCCL over xGMI is reached
    inside the library (``ncclCommInitAll`` / ``ncclAllGather``), not through torch.  What the reference does with an
    mpi4py communicator and one process per GPU (/root/reference/gpu_backend/kernel_state_ansatz.py:149-199, 415-428)."""

    def __init__(self, n_devices: int | None = None, device_ids=None):
        if device_ids is not None:
            ids = np.ascontiguousarray(device_ids, dtype=np.int32)
            n_devices = int(ids.shape[0])
        else:
            ids = None
            n_devices = int(n_devices or device_count())
        h = _P()
        _check(lib().qk_comm_init_all(n_devices, None if ids is None else ids.ctypes.data, C.byref(h)), "qk_comm_init_all")
        self._h, self.size = h, int(lib().qk_comm_size(h))
        self._ctx = []
        for r in range(self.size):
            c = Context.__new__(Context)
            c._h, c.device_id, c._borrowed = _P(lib().qk_comm_ctx(h, r)), int(ids[r]) if ids is not None else r, True
            self._ctx.append(c)

    def ctx(self, rank: int) -> "Context":
        return self._ctx[rank]

    def allgather_sets(self, local, lo, total: int):
        """``local[r]``: the MpsSet of the states ``[lo[r], lo[r] + len(local[r]))`` on rank r's context (``None`` = empty
        share).  ONE all-gather of the packed images; returns the whole set on every device."""
        hs = (_P * self.size)(*[(m.handle if m is not None else None) for m in local])
        lo_a = np.ascontiguousarray(lo, dtype=np.int32)
        out = (_P * self.size)()
        _check(lib().qk_mps_set_allgather(self._h, hs, lo_a.ctypes.data, int(total), out), "qk_mps_set_allgather")
        n_sites = next(m for m in local if m is not None).dims.shape[1] - 1
        dims = np.zeros((total, n_sites + 1), dtype=np.int32)
        for m, l0 in zip(local, lo):
            if m is not None:
                dims[l0 : l0 + len(m)] = m.dims
        return [MpsSet(self._ctx[r], _P(out[r]), dims.copy()) for r in range(self.size)]

    def gram(self, xsets, ysets=None) -> np.ndarray:
        """The sharded Gram: one sweep launch per device, ONE all-gather of the packed values, a scatter per device;
        returns rank 0's dense matrix (rows = Y, cols = X)."""
        nx = len(xsets[0])
        ny = nx if ysets is None else len(ysets[0])
        xs = (_P * self.size)(*[m.handle for m in xsets])
        ys = None if ysets is None else (_P * self.size)(*[m.handle for m in ysets])
        out = np.zeros((ny, nx), dtype=np.float64)
        _check(lib().qk_gram_sharded(self._h, xs, ys, out.ctypes.data, nx), "qk_gram_sharded")
        return out

    def device_gram_ptr(self, rank: int) -> int:
        p = _P()
        _check(lib().qk_comm_device_gram(self._h, rank, C.byref(p)), "qk_comm_device_gram")
        return int(p.value or 0)

    def stats(self, rank: int = 0):
        st, ms = QkStats(), C.c_double()
        _check(lib().qk_comm_stats(self._h, rank, C.byref(st), C.byref(ms)), "qk_comm_stats")
        d = st.as_dict()
        d["kernel_name"] = lib().qk_kernel_name(st.kernel, st.precision).decode()
        d["allgather_ms"] = ms.value
        return d

    def close(self):
        if self._h:
            for c in self._ctx:  # sets made on the communicator's contexts (uploads, all-gathered sets) go first: their handles point
                c._close_sets()  # into contexts that qk_comm_destroy deletes
                c._h = None
            lib().qk_comm_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


_default_ctx: dict[int, Context] = {}


def default_context(device_id: int = 0) -> Context:
    if device_id not in _default_ctx:
        _default_ctx[device_id] = Context(device_id)
    return _default_ctx[device_id]


@contextmanager
def context(device_id: int = 0):
    ctx = Context(device_id)
    try:
        yield ctx
    finally:
        ctx.close()
