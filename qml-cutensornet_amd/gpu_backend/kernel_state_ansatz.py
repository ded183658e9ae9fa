"""MI355X drop-in for the reference module of the same name.

Surface kept (reference: /root/reference/gpu_backend/kernel_state_ansatz.py):
  * ``KernelStateAnsatz(num_qubits, reps, gamma, entanglement_map, hadamard_init=True)``  (ref :16-103)
  * ``build_kernel_matrix(mpi_comm, ansatz, X, Y=None, info_file=None, truncation_error=None,
    loglevel=30) -> np.ndarray``                                                          (ref :106-452)
    same argument meaning, same exceptions (ref :136-139), same orientation
    ``K[len(Y) or len(X), len(X)]`` with rows = Y (ref :325-326, :387), valid on rank 0,
    same profiling-JSON keys (ref :160-162, 205, 238-244, 301-320, 434-444).

What differs is how the work is done: the reference loops over pairs in Python and calls
cuTensorNet once per entry, rotating pickled MPS between ranks; here a rank builds its share of
the states, keeps it as ONE packed device image, the images are exchanged as flat buffers (one
RCCL all-gather, ``dist.exchange_sets``) so that every rank holds all MPS on its MI355X, each
rank sweeps its share of the pairs in one persistent HIP kernel launch and the shares meet in a
single all-gather (RCCL through torch.distributed when it is initialised with the nccl backend,
the communicator's own ``allgather`` otherwise).
"""
from __future__ import annotations

import json
import sys
import time
from statistics import mean, median

import numpy as np

try:  # normal case: imported as qml_cutensornet_amd.gpu_backend.kernel_state_ansatz
    from ..ansatz import KernelStateAnsatz  # noqa: F401
    from .. import engine as _engine
    from ..dist import assemble_gram, comm_allgather, exchange_sets
    from ..mps import MPS, simulate, simulate_many  # noqa: F401
except ImportError:  # imported top-level as gpu_backend.kernel_state_ansatz
    import qml_cutensornet_amd as _pkg  # noqa: F401
    from qml_cutensornet_amd.ansatz import KernelStateAnsatz  # noqa: F401
    from qml_cutensornet_amd import engine as _engine
    from qml_cutensornet_amd.dist import assemble_gram, comm_allgather, exchange_sets
    from qml_cutensornet_amd.mps import MPS, simulate, simulate_many  # noqa: F401

ROOT_RANK = 0


def _say(is_root, text):
    if is_root:
        print(text)
        sys.stdout.flush()


_PILOT_MIN_STATES = 24  # below this a share goes to the device builder as a whole (states that outgrow the cap: host)


def _entangling_weight(circuit):
    """Cost proxy of a circuit: sum of sin^2(pi alpha) over its XXPhase gates (alpha in half-turns: 0 and 1 do not
    entangle).  The device builder orders its queue by the same quantity; it tracks the bonds a state will reach."""
    xx = np.asarray(circuit.op) == 2
    return float((np.sin(np.pi * np.asarray(circuit.alpha)[xx]) ** 2).sum())


def _hybrid_build(ctx, circuits, fidelity, cap, host_workers, is_root, label):
    """QK_BUILDER=hybrid for a large share: the device builder (bonds capped at ``cap``) and the host pool work AT THE SAME
    TIME, and a state predicted to outgrow the cap never visits the device.
      1. the heaviest quarter (by the cost proxy) starts on the host pool at once -- whatever the prediction will say,
         these are the states the host is the better tool for;
      2. meanwhile a pilot of 8 states spread over the rest runs on the device (partial): the lightest state it drops
         sets the threshold of the prediction;
      3. predicted-to-fit states go to the device in one launch while the host pool takes the others;
      4. what the device still drops (misprediction) is built on the host at the end.
    Returns (list[MPS], seconds per state), or None when the device builder fails (the caller falls back)."""
    import threading

    m = len(circuits)
    w = np.array([_entangling_weight(c) for c in circuits])
    order = np.argsort(w)  # lightest first
    states, secs = [None] * m, [0.0] * m

    def host(idx, box):
        t0 = time.perf_counter()
        built, bsecs = simulate_many([circuits[k] for k in idx], fidelity, workers=host_workers)
        for k, mps, dt in zip(idx, built, bsecs):
            states[k], secs[k] = mps, dt
        box.append(time.perf_counter() - t0)

    def device(idx):
        t0 = time.perf_counter()
        built, info = ctx.build_mps([circuits[k] for k in idx], fidelity, max_bond=cap, partial=True)
        dt = (time.perf_counter() - t0) / max(1, len(idx))
        dropped = []
        for pos, k in enumerate(idx):
            if built[pos] is None:
                dropped.append(k)
            else:
                states[k], secs[k] = built[pos], dt
        return dropped

    heavy = [int(k) for k in order[m - m // 4 :]]
    rest = [int(k) for k in order[: m - m // 4]]
    pilot = sorted({rest[int(round(f * (len(rest) - 1)))] for f in np.linspace(0.0, 1.0, 8)})
    box_a = []
    th = threading.Thread(target=host, args=(heavy, box_a))
    th.start()
    try:
        pilot_dropped = device(pilot)
    except _engine.QkError as exc:
        th.join()
        _say(is_root, f"{label}: device builder gave up on the pilot ({exc}); building on the host")
        return None
    thr = min((w[k] for k in pilot_dropped), default=np.inf)  # lightest state the device could not hold
    others = [k for k in rest if k not in pilot]
    dev_idx = [k for k in others if w[k] < thr]
    host_idx = [k for k in others if w[k] >= thr] + pilot_dropped
    _say(is_root, f"{label}: pilot of {len(pilot)}: {len(pilot_dropped)} outgrew bond {cap}; device builder takes {len(dev_idx)} states, host pool {len(heavy) + len(host_idx)}")
    th.join()
    box_b = []
    th = threading.Thread(target=host, args=(host_idx, box_b)) if host_idx else None
    if th:
        th.start()
    late = []
    try:
        if dev_idx:
            late = device(dev_idx)
    except _engine.QkError as exc:
        _say(is_root, f"{label}: device builder gave up ({exc}); its states go to the host")
        late = dev_idx
    if th:
        th.join()
    if late:
        host(late, [])
    return states, secs


def _auto_builder(circuits, host_workers):
    """QK_BUILDER=auto: device or host for this share, from its size, the host cores at hand and the spread of the cost proxy.
    Calibration (one MI355X box with a 16-core share, profiles/r03/builder_policy.txt): the device builder runs a state per
    workgroup -- 256 to 1024 in flight -- and a launch ends with its heaviest state, which takes about 2.5 x what one host
    core needs for it (60 qubits x 6 layers, gamma = 1: 4.3 s for the longest of 500 states, whose sum is 230-430 cpu-s).
    The host pool needs (states x mean cost) / workers.  With the heaviest state at (w_max / w_mean)^2 times the mean cost the
    device wins from about 2.5 x workers x (w_max / w_mean)^2 states on -- at 16 workers: 137 states of that config (ratio 1.85),
    250 of the 100-qubit gamma = 0.1 one (2.5), 180 of the 40-qubit x 4-layer one (2.1): the full data sets go to the device, an
    eighth of the 60-qubit one (63 states) stays on the host cores."""
    m = len(circuits)
    if m == 0:
        return "host"
    w = np.array([_entangling_weight(c) for c in circuits])
    ratio = float(w.max() / w.mean()) if w.mean() > 0 else 1.0
    need = 2.5 * max(1, host_workers) * max(1.0, ratio) ** 2
    return "device" if m >= need else "host"


SMALL_BOND_CAP = 64  # the device builder's cap for shares whose bonds are expected to stay small (_expect_small_bonds)


def _expect_small_bonds(ansatz, circuits):
    """Will the bonds of this share stay small (a few tens)?  Then the device builder's 256-thread shape with two workgroups per CU
    (512 states in flight, most factorisations in LDS; bond cap 64) is the right one -- 100 qubits x 10 layers at gamma = 0.1
    (final bonds <= 32, a few more in mid-circuit): 1000 states -- against one 512-thread workgroup per CU, the shape for bonds in
    the hundreds (17.9 s for the same 1000 states).  Yes when the analytic bound 2^(distance x layers) says so, or when the
    heaviest circuit's entangling weight is small (calibration: 40 qubits x 4 layers at gamma = 0.5, bonds 20-26, has w_max = 8.4;
    gamma = 1 configurations start at w = 11 and reach bonds 60-250).  A wrong yes costs one short launch: a state that outgrows
    the small cap is dropped at the gate where it does and the share is rebuilt with the large one."""
    try:
        dist_max = max((abs(int(b_) - int(a_)) for a_, b_ in ansatz.entanglement_map), default=0)
        if 2 ** min(dist_max * int(ansatz.reps), int(ansatz.num_qubits) // 2) <= 64:
            return True
    except (AttributeError, TypeError, ValueError):
        pass
    return max(_entangling_weight(c) for c in circuits) <= 10.0


def _simulate_share(ansatz, points, rank, n_procs, fidelity, is_root, label, device_id=0, host_workers=1, want_set=True):
    """This rank's slice of the data set (contiguous chunks of ceil(N/P), as ref :154,:171-174) -> (first index, the
    states as ONE packed device set -- ``None`` for an empty share --, seconds per state, fidelities).  ``want_set=False``
    (host-only callers: the CPU tests) keeps the host builder's list of MPS instead of uploading it.
    QK_MAX_BOND (environment): a bond cap for either builder -- the ``chi`` pytket-cutensornet's ``Config`` would take at
    ref :141-144 (the reference leaves it unset: the default is no cap)."""
    import os

    per_rank = -(-len(points) // n_procs)
    lo = min(len(points), rank * per_rank)
    hi = min(len(points), lo + per_rank)
    chi = int(os.environ.get("QK_MAX_BOND", "0")) or None
    which = os.environ.get("QK_BUILDER", "auto") if want_set else "host"  # auto | device | hybrid | host
    forced = which  # what the caller asked for: only a FORCED device build may fail the call
    circuits = [ansatz.circuit_for_data(points[k, :]) for k in range(lo, hi)] if hi > lo else []
    if which == "auto":
        try:
            which = _auto_builder(circuits, host_workers)
        except (AttributeError, TypeError, ValueError):  # an ansatz without a compiled gate program: the host loop handles it
            which = "host"
    if which in ("device", "hybrid") and hi > lo:
        # the rank's whole share in ONE launch of the device builder (csrc/qk_build.hip): what the reference does with
        # simulate(libhandle, ...) on the rank's GPU (ref :221,:263).  "device" / "auto": bonds up to QK_BUILDER_MAX_BOND (320:
        # bonds in mid-circuit exceed the final ones), a state that outgrows it is built on the host afterwards (a forced
        # "device" fails instead); "hybrid" caps at 64 and builds what outgrows the cap on the host pool -- concurrently, behind a
        # pilot, for shares of >= 24 states (_hybrid_build); "host" skips it.
        t0 = time.perf_counter()
        cap = chi or int(os.environ.get("QK_BUILDER_MAX_BOND", "64" if which == "hybrid" else "320"))
        partial = which == "hybrid" or (forced == "auto" and chi is None)
        ctx = _engine.default_context(device_id)
        if which == "hybrid" and hi - lo >= _PILOT_MIN_STATES and chi is None:
            out = _hybrid_build(ctx, circuits, fidelity, cap, host_workers, is_root, label)
            if out is not None:
                states, secs = out
                _say(is_root, f"{label}: 100%")
                return lo, ctx.upload(states), secs, [m.fidelity for m in states]
        try:
            dset = binfo = states = None
            if chi is None and cap > SMALL_BOND_CAP and which != "hybrid" and _expect_small_bonds(ansatz, circuits):
                dset, states, binfo = ctx.build_share(circuits, fidelity, max_bond=SMALL_BOND_CAP, partial=True)
                if dset is None:  # some state outgrew the small cap after all: the whole share again, with the large one
                    _say(is_root, f"{label}: {len(binfo['dropped'])} of {hi - lo} states outgrew bond {SMALL_BOND_CAP}; rebuilding with bonds up to {cap}")
                    dset = binfo = states = None
            if binfo is None:
                dset, states, binfo = ctx.build_share(circuits, fidelity, max_bond=cap, partial=partial, truncate=chi is not None)
        except _engine.QkError as exc:
            if forced == "device":
                raise
            _say(is_root, f"{label}: device builder gave up ({exc}); building on the host")
            dset, states, binfo = None, None, None
        if binfo is not None:
            dt = (time.perf_counter() - t0) / (hi - lo)
            secs = [dt] * (hi - lo)
            if dset is not None:  # every state fitted: the share is already a packed device set, nothing was downloaded
                _say(is_root, f"{label}: 100%")
                return lo, dset, secs, [float(f) for f in binfo["fidelity"]]
            if binfo["dropped"]:  # states whose bonds outgrew the cap: the host builder is the better tool for those
                _say(is_root, f"{label}: {len(binfo['dropped'])} of {hi - lo} states outgrew bond {cap}; building them on the host")
                built, bsecs = simulate_many([circuits[k] for k in binfo["dropped"]], fidelity, workers=host_workers)
                for k, m, dt_k in zip(binfo["dropped"], built, bsecs):
                    states[k], secs[k] = m, dt_k
            _say(is_root, f"{label}: 100%")
            return lo, ctx.upload(states), secs, [m.fidelity for m in states]
    # host builder: one circuit per core on a thread pool (no fork: the GPU may already be initialised; the native builder
    # releases the GIL) -- the reference's loop is serial because its simulate() runs on the GPU (ref :213-231)
    tick, done = max(1, per_rank // 10), [0]

    def progress():
        done[0] += 1
        if (done[0] - 1) % tick == 0:
            _say(is_root, f"{label}: {10 * ((done[0] - 1) // tick)}%")

    states, secs = simulate_many(circuits, fidelity, workers=host_workers, progress=progress, max_bond=chi)
    if not want_set:
        return lo, states, secs, [m.fidelity for m in states]
    if not states:
        return lo, None, secs, []
    return lo, _engine.default_context(device_id).upload(states), secs, [m.fidelity for m in states]


def _gram_on_device(comm, rank, n_procs, ctx, xset, yset):
    """The hot path.  Returns (K on the host or None, seconds in the final exchange)."""
    use_torch = False
    if n_procs > 1:
        try:
            import torch.distributed as dist

            use_torch = dist.is_initialized() and dist.get_world_size() == n_procs and dist.get_backend() == "nccl"
        except ImportError:
            use_torch = False
    if n_procs == 1:
        return ctx.gram(xset, yset), 0.0
    if use_torch:
        import importlib

        GramJob = importlib.import_module("qml_cutensornet_amd.gram").GramJob

        job = GramJob(ctx, xset, yset, n_procs, rank)
        t0 = time.perf_counter()
        K = job.run()
        job.close()
        return K, time.perf_counter() - t0
    # host communicator (mpi4py or gloo): sweep on the GPU, all-gather the packed values on the host
    plan = _engine.Plan(xset.dims, None if yset is None else yset.dims, n_procs, rank)
    vals = ctx.gram_values_host(xset, yset, plan)
    t0 = time.perf_counter()
    shares = comm_allgather(comm, (plan.pairs(), vals))
    exchange = time.perf_counter() - t0
    ny = len(xset) if yset is None else len(yset)
    K = assemble_gram(ny, len(xset), [s[0] for s in shares], [s[1] for s in shares], yset is None)
    plan.close()
    return K, exchange


def _set_mib(dims):
    """MiB of the complex128 tensors of the states with bond table ``dims`` (what the reference sums from .nbytes, ref :295)."""
    d = np.asarray(dims, dtype=np.float64)
    return float((32.0 * d[:, :-1] * d[:, 1:]).sum() / 2**20)


def build_kernel_matrix(mpi_comm, ansatz, X, Y=None, info_file=None, truncation_error=None, loglevel=30):
    """Fill the kernel (Gram) matrix ``K[j, i] = |<psi(X_i)|psi(Y_j)>|^2``; ``Y=None`` means ``Y = X``.

    Returns the ``len(Y) x len(X)`` float64 matrix on rank 0 and ``None`` elsewhere (the reference
    returns the result of ``reduce(..., root=0)``, ref :428,:452).
    """
    if Y is not None and len(X) < len(Y):
        raise ValueError("X must not be smaller than Y. Swap input order and transpose output.")
    if truncation_error is None:
        raise ValueError("You must specify a truncation error.")
    X = np.asarray(X, dtype=np.float64)
    Y = None if Y is None else np.asarray(Y, dtype=np.float64)
    fidelity = 1.0 - float(truncation_error)

    rank, n_procs = mpi_comm.Get_rank(), mpi_comm.Get_size()
    is_root = rank == ROOT_RANK
    n_dev = _engine.device_count()
    if n_dev <= 0:
        raise _engine.QkError("no gfx950 device visible: the Gram path has no CPU fallback")
    device_id = rank % n_dev
    from qml_cutensornet_amd.builder_pool import default_workers

    host_workers = max(1, default_workers() // max(1, min(n_procs, n_dev)))  # host cores of this rank's share of the node
    prof = {}
    t_start = time.perf_counter()
    if is_root:
        prof["n_procs"] = [n_procs, "gpus"]
        prof["lenX"] = [len(X), "entries"]
        prof["lenY"] = [None if Y is None else len(Y), "entries"]

    # circuits are bound lazily inside the simulation loop; the reference times their generation apart
    prof["r0_circ_gen"] = [0.0, "seconds"]
    _say(is_root, "\nContracting the MPS of the circuits from the X dataset...")
    ctx = _engine.default_context(device_id)
    x_lo, x_local, x_secs, x_fid = _simulate_share(ansatz, X, rank, n_procs, fidelity, is_root, "X", device_id, host_workers)
    y_lo, y_local, y_secs, y_fid = (0, None, [], [])
    if Y is not None:
        _say(is_root, "\nContracting the MPS of the circuits from the Y dataset...")
        y_lo, y_local, y_secs, y_fid = _simulate_share(ansatz, Y, rank, n_procs, fidelity, is_root, "Y", device_id, host_workers)
    sim_secs = x_secs + y_secs
    # the device builder keeps its per-workgroup arena and workspace on the context (tens of GB at large bond caps): they go back before
    # the exchange and the sweep need the memory (several ranks may share one GPU: device = rank % n_devices, ref :152)
    ctx.trim()

    # every rank gets the whole set: the packed device images of the shares, one all-gather (ref :341-352, 415-419)
    xset, gather_secs = exchange_sets(mpi_comm, ctx, x_local, x_lo, len(X))
    yset = None
    if Y is not None:
        yset, dt = exchange_sets(mpi_comm, ctx, y_local, y_lo, len(Y))
        gather_secs += dt
    for loc, full in ((x_local, xset), (y_local, yset)):
        if loc is not None and loc is not full:
            loc.close()

    try:
        if is_root:
            mine_fid = x_fid + y_fid
            prof["r0_circ_sim"] = [sum(sim_secs), "seconds"]
            if sim_secs:
                prof["avg_circ_sim"] = [mean(sim_secs), "seconds"]
                prof["median_circ_sim"] = [median(sim_secs), "seconds"]
                prof["q1_circ_sim"] = [float(np.percentile(sim_secs, 25)), "seconds"]
                prof["q3_circ_sim"] = [float(np.percentile(sim_secs, 75)), "seconds"]
            total_mib = _set_mib(xset.dims) + (0.0 if yset is None else _set_mib(yset.dims))
            n_all = len(xset) + (0 if yset is None else len(yset))
            prof["gpu_mps_mem"] = [total_mib, "MiB"]  # every GPU holds the whole set here
            prof["avg_mps_mem"] = [total_mib / n_all, "MiB"]
            prof["avg_fidelity"] = [sum(mine_fid) / max(1, len(mine_fid)), ""]
            chi_x = xset.dims.max(axis=1)
            prof["ave max chi x"] = (float(chi_x.mean()), "chi x")
            prof["ave max chi y"] = (float((chi_x if yset is None else yset.dims.max(axis=1)).mean()), "chi y")
            prof["r_nonRR_recv"] = [0, "seconds"]  # no ranks outside a ring: there is no ring
            prof["r0_RR_recv"] = [gather_secs, "seconds"]  # exchange of the packed sets; the Gram all-gather is added below
            _say(True, "\nFinished contracting all MPS.\n\nCalculating kernel matrix...")

        t_tiles = time.perf_counter()
        kernel_mat, exchange = _gram_on_device(mpi_comm, rank, n_procs, ctx, xset, yset)
        tiles = time.perf_counter() - t_tiles
    finally:
        xset.close()
        if yset is not None:
            yset.close()

    if not is_root:
        return None
    n_entries = kernel_mat.size if Y is not None else len(X) * (len(X) + 1) // 2
    per_entry = tiles / max(1, n_entries)
    prof["r0_RR_recv"][0] += exchange
    prof["kernel_mat_time"] = [tiles, "seconds"]
    prof["total_time"] = [time.perf_counter() - t_start, "seconds"]
    # one launch computes every overlap: per-product statistics collapse to the mean
    prof["r0_product"] = [tiles - exchange, "seconds"]
    for key in ("avg_product", "median_product", "q1_product", "q3_product"):
        prof[key] = [per_entry, "seconds"]
    _say(True, f"\nFinished calculating all inner products.\n\tAverage time per inner product: {per_entry:.3e} seconds.\n")
    if info_file is not None:
        with open(info_file + ".json", "w") as fp:
            json.dump(prof, fp, indent=4)
    return kernel_mat
