#!/usr/bin/env python3
"""Headline benchmark: Gram kernel entries/sec at 60 qubits x 6 layers (BASELINE.json cfg4).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full fill of the symmetric 500x500 training Gram
K[j,i] = |<psi(x_i)|psi(x_j)>|^2 from MPS that are already resident in HBM: one persistent
sweep launch per rank over its share of the 125 250 unique pairs, one RCCL all-gather of the
packed values, a scatter (with mirroring) into the dense matrix and its copy to the host.
MPS construction (the reference's `simulate` loop, the step before the path) happens before
the timed region and is reported separately, like the reference's `r0_circ_sim`.

Rank 0 prints ONE JSON line (contract in the task statement), with `roofline` (dominant
kernel = the sweep kernel the engine selected, named in the line; fp64 MFMA bound) and, at N=1,
`cpu_baseline` (the oracle's C restatement of the ITensors `inner` loop on BLAS zgemm, timed on this
host's cores over a bounded sample; the hand-written loop is timed beside it).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import pickle
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (qubits, layers, distance, points)   -- BASELINE.json "configs"
    "cfg2": (20, 2, 1, 100),
    "cfg3": (40, 4, 2, 200),
    "cfg4": (60, 6, 2, 500),
    "cfg5": (100, 10, 4, 1000),  # only tractable at small gamma (SURVEY.md addendum): pass --gamma 0.1
}
# fp64 matrix peak of MI355X: 256 CU x 4 SIMD x 32 flop/clk (v_mfma_f64_16x16x4: 2048 flop / 64 clk) x 2.4 GHz
PEAK_F64_MFMA_TFLOPS = 256 * 4 * 32 * 2.4e9 / 1e12
PEAK_F32_MFMA_TFLOPS = 256 * 4 * 64 * 2.4e9 / 1e12  # v_mfma_f32_16x16x4_f32: 64 flop/clk/SIMD (MI355X_MICROARCH.md)
PEAK_HBM_BYTES = 8.0e12  # HBM3E, bytes/s (MI355X_MICROARCH.md)


def _host_cpu():
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
    except OSError:
        model = "unknown"
    return f"{model}; {len(os.sched_getaffinity(0))} cores available to this process"


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def build_or_load_states(name, n, reps, d, gamma, npts, seed, rank, world, workers):
    """All `npts` MPS of the config; chunks are built by rank (chunk % world) and cached on disk."""
    import qml_cutensornet_amd as Q
    from qml_cutensornet_amd.builder_pool import build_states
    from qml_cutensornet_amd.data import synthetic_features

    X = synthetic_features(npts, n, seed)
    ansatz = Q.KernelStateAnsatz(n, reps, gamma, Q.entanglement_graph(n, d))
    key = hashlib.sha1(f"{name}|{n}|{reps}|{d}|{gamma}|{npts}|{seed}|v1".encode()).hexdigest()[:16]
    cdir = os.path.join(os.environ.get("QK_CACHE_DIR", "/tmp/qkgram_cache"), key)
    os.makedirs(cdir, exist_ok=True)
    chunk = 25
    nchunks = (npts + chunk - 1) // chunk
    t0 = time.perf_counter()
    todo = [c for c in range(rank, nchunks, world) if not os.path.exists(os.path.join(cdir, f"chunk_{c:04d}.pkl"))]
    built = 0
    if todo:
        idx = np.concatenate([np.arange(c * chunk, min(npts, (c + 1) * chunk)) for c in todo])
        states, secs = build_states(ansatz, X[idx], 1.0 - 1e-16, workers)  # one pool over all missing points
        pos = 0
        for c in todo:
            m = min(npts, (c + 1) * chunk) - c * chunk
            part, psecs = states[pos : pos + m], secs[pos : pos + m]
            pos += m
            path = os.path.join(cdir, f"chunk_{c:04d}.pkl")
            tmp = path + f".tmp{os.getpid()}"
            with open(tmp, "wb") as f:
                pickle.dump(([s_.tensors for s_ in part], [s_.fidelity for s_ in part], psecs), f, protocol=4)
            os.replace(tmp, path)
        built = len(states)
    build_wall = time.perf_counter() - t0
    # file barrier (no torch.distributed yet: worker processes are forked BEFORE the GPU is touched)
    deadline = time.time() + 3600
    while not all(os.path.exists(os.path.join(cdir, f"chunk_{c:04d}.pkl")) for c in range(nchunks)):
        if time.time() > deadline:
            raise SystemExit("timed out waiting for the other ranks' MPS chunks")
        time.sleep(0.2)
    tensors, fids, secs = [], [], []
    for c in range(nchunks):
        with open(os.path.join(cdir, f"chunk_{c:04d}.pkl"), "rb") as f:
            t, fd, s = pickle.load(f)
        tensors += t
        fids += fd
        secs += s
    states = [Q.MPS(t, fd) for t, fd in zip(tensors, fids)]
    return states, {"built_here": built, "build_wall_s": build_wall, "cpu_s_per_state": float(np.mean(secs)), "ansatz": ansatz, "X": X}


SWEEP_SOURCES = ("qk_fused.h", "qk_ring.h", "qk_device.h", "qkgram.hip")  # what a sweep kernel is compiled from
PROFILE_ROUNDS = ("r04", "r03")


def sweep_source_sha():
    """Digest of the sweep kernels' sources: a committed PMC summary is quoted only while it describes THIS code."""
    h = hashlib.sha256()
    for f in SWEEP_SOURCES:
        with open(os.path.join(ROOT, "qml-cutensornet_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def quoted_traffic(cfg_name, kernels):
    """HBM / fabric bytes per launch from the committed rocprofv3 --pmc summary of the same workload (PMC counters cannot be
    read from inside this process): {kernel name: (bytes, executed matrix flops or None)}, the file, or (None, reason) when the summary
    is missing or was taken on other sources."""
    pmc, why = None, "no committed summary for this workload"
    for rnd in PROFILE_ROUNDS:  # newest first: the first summary taken on THIS build's sweep sources is the one quoted
        pmc_file = os.path.join(ROOT, "profiles", rnd, cfg_name, "pmc_summary.json")
        if not os.path.exists(pmc_file):
            continue
        try:
            cand = json.load(open(pmc_file))
        except ValueError:
            why = f"unreadable summary {os.path.relpath(pmc_file, ROOT)}"
            continue
        if cand.get("source_sha") != sweep_source_sha():
            why = f"stale: {os.path.relpath(pmc_file, ROOT)} was taken on sources {cand.get('source_sha')}, this run is {sweep_source_sha()}"
            continue
        pmc = cand
        break
    if pmc is None:
        return None, why
    out = {}
    for k in kernels:
        ent = next((v for name, v in pmc.get("kernels", {}).items() if k and k in name), None)
        if ent is None or "traffic_bytes_per_launch" not in ent:
            return None, f"the summary has no entry for {k}"
        out[k] = (ent["traffic_bytes_per_launch"], ent.get("mfma_flops_executed"),  # (bytes, flops the matrix cores executed: SQ_INSTS_VALU_MFMA_MOPS_F64,
                  {q_: ent[q_] for q_ in ("l2_hit_rate", "lds_active_share_of_busy", "lds_bank_conflict_share_of_lds_active", "mfma_util") if ent.get(q_) is not None})  # other quoted counters)
    return out, os.path.relpath(pmc_file, ROOT)


def device_build_leg(ctx, ansatz, X, states, log_):
    """The input producer on the device: every circuit of the data set in ONE launch of the device builder, timed;
    the Gram of the device-built set is returned so that the caller can compare it with the Gram of the host-built states
    (the ones the timed steps run on).  Untimed region."""
    from qml_cutensornet_amd.gpu_backend.kernel_state_ansatz import SMALL_BOND_CAP, _expect_small_bonds

    circs = [ansatz.circuit_for_data(x) for x in X]
    big = int(os.environ.get("QK_BUILDER_MAX_BOND", "320"))
    caps = [SMALL_BOND_CAP, big] if _expect_small_bonds(ansatz, circs) else [big]  # build_kernel_matrix's escalation: the small-bond shape first where it is expected to do
    t0 = time.perf_counter()
    dset = info = None
    for cap in caps:
        try:
            dset, info = ctx.build_mps_set(circs, max_bond=cap)
            break
        except Exception as exc:  # noqa: BLE001 - reported; the timed Gram does not depend on it
            log_(f"device builder at bonds <= {cap}: {exc}")
            err = str(exc)
    if dset is None:
        return {"error": err}, None
    wall = time.perf_counter() - t0
    host_dims = np.array([m.bond_dims() for m in states])
    differ = int((info["dims"] != host_dims).any(axis=1).sum())
    K = ctx.gram(dset)
    dset.close()
    ctx.trim()  # the builder's arena and workspace (tens of GB) go back before the timed Gram
    return {"device_kernel_s": info["kernel_ms"] / 1e3, "device_wall_s": wall, "max_bond_cap": cap, "states_whose_bonds_differ_from_host": differ,
            "largest_bond_difference": int(np.abs(info["dims"] - host_dims).max())}, K


def cpu_baseline(states, pairs, total_unique, npts, seconds, gpu_vals, threads):
    """Time the oracle's C restatement on a bounded random sample of this Gram's pairs: the zgemm-based leg
    (oracle/overlap_blas.c on scipy's OpenBLAS -- what ITensors' `inner`, KernelPkg.jl:106, runs on) is the
    reported baseline; the hand-written loop (oracle/overlap_ref.c) is timed on the same pairs beside it."""
    from oracle import c_oracle

    rng = np.random.default_rng(0)
    order = rng.permutation(pairs.shape[0])
    dims = np.stack([m.bond_dims() for m in states])
    # choose the sample size from the algorithmic flops, assuming ~40 GFlop/s per core for the two legs together
    est_rate = 40e9 * threads
    chosen, acc = [], 0.0
    for t in order:
        i, j = pairs[t]
        a, b = dims[i].astype(float), dims[j].astype(float)
        f = 8 * np.minimum(a[:-1] * b[:-1] * 2 * b[1:] + 2 * a[:-1] * a[1:] * b[1:], a[:-1] * b[:-1] * 2 * a[1:] + 2 * b[:-1] * a[1:] * b[1:]).sum()
        chosen.append(t)
        acc += f
        if acc / est_rate > seconds or len(chosen) >= 60000:
            break
    chosen = np.asarray(chosen)
    ts = [m.tensors for m in states]
    legs = {}
    for name, fn in (("blas", c_oracle.gram_pairs_blas), ("hand_loop", c_oracle.gram_pairs)):
        t0 = time.perf_counter()
        vals, _, used = fn(ts, None, pairs[chosen], threads=threads)
        dt = time.perf_counter() - t0
        legs[name] = {
            "value": npts * npts / (dt / len(chosen) * total_unique),
            "seconds": dt,
            "gflops": acc / dt / 1e9,
            "cores": int(used),
            "parity_max_abs_err_vs_gpu": float(np.abs(vals - gpu_vals[chosen]).max()),
        }
    b = legs["blas"]
    return {
        "value": b["value"],
        "unit": "entries/s",
        "cores": b["cores"],
        "kind": "port",
        "sample": f"{len(chosen)} random pairs of the same Gram ({b['seconds']:.1f} s on {b['cores']} threads, {b['gflops']:.1f} GFlop/s); "
        f"rate extrapolated to all {total_unique} unique pairs; oracle/overlap_blas.c: C restatement of KernelPkg.jl:101-109 with every "
        "contraction on zgemm (scipy's OpenBLAS, one pair per thread)",
        "gflops": b["gflops"],
        "parity_max_abs_err_vs_gpu": b["parity_max_abs_err_vs_gpu"],
        "hand_loop": {k: legs["hand_loop"][k] for k in ("value", "gflops", "cores", "parity_max_abs_err_vs_gpu")},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg4", choices=sorted(CONFIGS))
    ap.add_argument("--gamma", type=float, default=None, help="default 1.0 (0.1 for cfg5, whose bonds explode at larger gamma)")
    ap.add_argument("--points", type=int, default=0, help="override the number of data points")
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline budget (0 = skip)")
    ap.add_argument("--workers", type=int, default=0, help="host processes for MPS building (0 = all cores / ranks)")
    ap.add_argument("--max-bond", type=int, default=0,
                    help="bond cap chi (the `chi` of pytket-cutensornet's Config, ref gpu_backend/kernel_state_ansatz.py:141-144; 0 = none, like the reference): "
                    "the states are then built by the DEVICE builder, truncating at chi -- what makes cfg5 runnable beyond gamma = 0.1")
    ap.add_argument("--precision", default="f64", choices=["f64", "f32"],
                    help="f64 = the reference's precision and the headline metric; f32 = the complex64 sweep (SURVEY 8f N4), a supplementary line")
    args = ap.parse_args()
    if args.gamma is None:
        args.gamma = 0.1 if args.config == "cfg5" else 1.0

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    # ---- phase 0 (no GPU touched yet): native build + MPS construction on the host cores -------------
    import __graft_entry__ as graft

    if rank == 0:
        graft.build()
    from qml_cutensornet_amd.builder_pool import default_workers

    n, reps, d, npts = CONFIGS[args.config]
    if args.points:
        npts = args.points
    workers = args.workers or max(1, default_workers() // world)
    log(rank, f"{args.config}: {n} qubits x {reps} layers, d={d}, gamma={args.gamma}, {npts} points; {world} GPU(s); {workers} builder procs/rank")
    if args.max_bond:  # capped bonds: the device builder makes the states once the GPU is up (every rank its own copy)
        import qml_cutensornet_amd as Q
        from qml_cutensornet_amd.data import synthetic_features

        states = None
        binfo = {"built_here": 0, "build_wall_s": 0.0, "cpu_s_per_state": float("nan"), "X": synthetic_features(npts, n, args.seed),
                 "ansatz": Q.KernelStateAnsatz(n, reps, args.gamma, Q.entanglement_graph(n, d))}
    else:
        states, binfo = build_or_load_states(args.config, n, reps, d, args.gamma, npts, args.seed, rank, world, workers)
        chi_max = np.array([m.max_bond() for m in states])
        log(rank, f"states ready: built {binfo['built_here']} here in {binfo['build_wall_s']:.1f}s ({binfo['cpu_s_per_state']:.2f} cpu-s/state); max bond mean {chi_max.mean():.1f} max {chi_max.max()}")

    # ---- phase 1: GPU --------------------------------------------------------------------------------
    import torch
    import torch.distributed as dist

    # rehearsal hooks (one-GPU box): QK_FORCE_DEVICE puts every rank on that GPU, QK_DIST_BACKEND=gloo
    # replaces RCCL, which needs one GPU per rank.  The driver's multi-GPU runs use neither.
    if "QK_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["QK_FORCE_DEVICE"])
    backend = os.environ.get("QK_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def barrier():
        if world > 1:
            dist.barrier()

    barrier()
    from qml_cutensornet_amd import engine
    from qml_cutensornet_amd.gram import GramJob

    ctx = engine.Context(local_rank)
    ctx.selftest()
    capped_build = None
    if args.max_bond:
        circs = [binfo["ansatz"].circuit_for_data(x) for x in binfo["X"]]
        t0 = time.perf_counter()
        states, dinfo = ctx.build_mps(circs, max_bond=args.max_bond, truncate=True)
        ctx.trim()
        fids = np.array([m.fidelity for m in states])
        chi_max = np.array([m.max_bond() for m in states])
        capped_build = {"device_kernel_s": dinfo["kernel_ms"] / 1e3, "device_wall_s_with_download": time.perf_counter() - t0, "max_bond_cap": args.max_bond, "truncated_at_cap": True,
                        "fidelity_median": float(np.median(fids)), "fidelity_min": float(fids.min())}
        log(rank, f"device MPS builder, bonds cut at {args.max_bond}: {npts} states in {capped_build['device_kernel_s']:.2f} s; max bond mean {chi_max.mean():.1f} max {chi_max.max()}; "
                  f"truncation fidelity median {capped_build['fidelity_median']:.3g}")
    t0 = time.perf_counter()
    xset = ctx.upload(states)
    upload_s = time.perf_counter() - t0
    info = xset.info()
    k64_ref = None
    if args.precision == "f32":  # supplementary line: same workload on fp32 planes, checked against one fp64 Gram (untimed)
        ref_job = GramJob(ctx, xset, None, world, rank)
        k64_ref = ref_job.run()
        ref_job.close()
        x64 = xset
        xset = x64.to_f32()
        x64.close()
        info = xset.info()
    # the device MPS builder on the same circuits (rank 0, one GPU; QK_BENCH_DEVICE_BUILD=0 skips it): reported beside the host pool
    dev_build, K_dev = None, None
    if world == 1 and args.precision == "f64" and not args.max_bond and os.environ.get("QK_BENCH_DEVICE_BUILD", "1") != "0":
        dev_build, K_dev = device_build_leg(ctx, binfo["ansatz"], binfo["X"], states, lambda t: log(rank, t))
        if "error" not in dev_build:
            log(rank, f"device MPS builder: {npts} states in {dev_build['device_kernel_s']:.2f} s (wall {dev_build['device_wall_s']:.2f} s); host pool: {binfo['cpu_s_per_state'] * npts / workers:.1f} s on {workers} workers")
    host_K = torch.empty((npts, npts), dtype=torch.float64, pin_memory=True)
    # ---- the COLD Gram: what the product pays, since it computes every Gram once (the reference's kernel_mat_time, G:322, 432-434,
    # brackets the set-up of the tiling phase and the tiles alike).  A fresh set (its derived images -- interleaved, edge blocks, merged
    # steps -- do not exist yet) and a fresh plan -> K on the host; max over ranks.  The steady-state steps below reuse this job.
    ctx.trim()  # (scratch of earlier legs: the cold step allocates its own, as a first call does)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    engine.range_push("bench:cold_step")
    job = GramJob(ctx, xset, None, world, rank)
    t_planned = time.perf_counter()
    host_K.copy_(job.enqueue(), non_blocking=True)
    torch.cuda.current_stream().synchronize()
    engine.range_pop()
    t_done = time.perf_counter()
    barrier()
    cold_all = time.perf_counter() - t0
    cold_st = ctx.stats()
    pcost = job.plan.cost()
    cold = {"cold_step_ms": 1e3 * cold_all, "rank_cold_ms": 1e3 * (t_done - t0), "job_setup_ms": 1e3 * (t_planned - t0), "plan_ms": pcost["plan_ms"], "plan_threads": pcost["threads"],
            "derive_ms": cold_st["derive_ms"], "sweep_ms": cold_st["kernel_ms"], "allgather_ms": job.allgather_ms()}
    my = job.work[rank]
    log(rank, f"uploaded {info['device_bytes'] / 2**30:.2f} GiB in {upload_s:.1f}s; rank 0 share: {my['pairs']} pairs, {my['flops'] / 1e12:.2f} TFlop algorithmic ({my['padded_flops'] / 1e12:.2f} padded)")
    log(rank, f"cold Gram {cold['cold_step_ms']:.1f} ms: plan {cold['plan_ms']:.1f} ms on {cold['plan_threads']} threads (job set-up {cold['job_setup_ms']:.1f}), derived images {cold['derive_ms']:.1f} ms, sweep {cold['sweep_ms']:.1f} ms")

    drain = []
    tails, second_ms = [], []  # per step: (tail share of the first launch, of the second); device time of the second launch of a split sweep

    def step():
        engine.range_push("bench:step")
        K = job.enqueue()
        host_K.copy_(K, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        engine.range_pop()
        st_ = ctx.stats()
        second_ms.append(st_["second_ms"])
        tails.append((st_["tail_frac"], st_["second_tail_frac"]))
        # share of the whole sweep during which the chip was draining: both launches' tails in ms over the sweep's time
        drain.append((st_["tail_frac"] * (st_["kernel_ms"] - st_["second_ms"]) + st_["second_tail_frac"] * st_["second_ms"]) / st_["kernel_ms"] if st_["kernel_ms"] > 0 else 0.0)
        return st_["kernel_ms"]

    kernel_name = None

    for _ in range(args.warmup):
        step()
    kernel_name = ctx.stats()["kernel_name"] if args.warmup else None
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ms = []
    second_ms.clear()
    tails.clear()
    drain.clear()
    for _ in range(args.steps):
        kernel_ms.append(step())
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    ms_per_step = 1e3 * elapsed / max(1, args.steps)
    kms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
    kernel_name = kernel_name or ctx.stats()["kernel_name"]

    # the cold Gram through the C ABI's one-call route (qk_gram_host: what build_kernel_matrix runs on one rank, gpu_backend/kernel_state_ansatz.py):
    # a fresh set again -- fresh plan, fresh derived images, its own buffers -> K in host memory; rank 0, after the timed region
    cold_abi = None
    if world == 1 and states is not None:
        x2 = ctx.upload(states) if args.precision == "f64" else None
        if x2 is not None:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            K2 = ctx.gram(x2)
            cold_abi = {"ms": 1e3 * (time.perf_counter() - t0), "max_abs_vs_steady": float(np.abs(K2 - host_K.numpy()).max())}
            x2.close()
    # sanity on the result itself (invariants of a Gram of normalised states)
    Kh = host_K.numpy()
    diag_err = float(np.abs(np.diag(Kh) - 1).max())
    sym_err = float(np.abs(Kh - Kh.T).max())
    # after the timed region: every rank's kernel time and a digest of its copy of K (all ranks hold the full matrix)
    tail_all = float(np.mean(drain)) if drain else 0.0
    per_rank = [(kms, hashlib.sha1(np.ascontiguousarray(Kh).tobytes()).hexdigest(), my["padded_flops"] / 1e12, tail_all)]
    cold_rank = [cold]
    steady_gather_ms = job.allgather_ms()
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, (per_rank[0], cold, steady_gather_ms))
        per_rank = [g_[0] for g_ in gathered]
        cold_rank = [g_[1] for g_ in gathered]
        steady_gather_ms = [g_[2] for g_ in gathered]

    out = None
    if rank == 0:
        peak = PEAK_F64_MFMA_TFLOPS if args.precision == "f64" else PEAK_F32_MFMA_TFLOPS
        # A split sweep is two launches (qk_stats.second_*).  The roofline object prices the WHOLE sweep -- all pairs' algorithmic
        # flops over the device time of both launches -- and lists each launch beside it.
        last = ctx.stats()
        s_ms = float(np.mean(second_ms)) if second_ms else 0.0
        split = last["second_kernel"] != 0 and s_ms > 0
        first_ms = kms - s_ms if split else kms
        first_flops = my["flops"] - (last["second_flops"] if split else 0.0)
        bytes_scale = 1.0 if args.precision == "f64" else 0.5
        launches = [{"kernel": kernel_name, "kernel_ms": first_ms, "pairs": int(my["pairs"] - (last["second_pairs"] if split else 0)), "algorithmic_tflop": first_flops / 1e12,
                     "padded_4m_tflop": (my["padded_flops"] - (last["second_padded_flops"] if split else 0.0)) / 1e12,
                     "algorithmic_gbytes": (my["bytes"] - (last["second_bytes"] if split else 0.0)) / 1e9 * bytes_scale,
                     "achieved_tflops": first_flops / (first_ms * 1e-3) / 1e12 if first_ms > 0 else 0.0, "tail_frac": float(np.mean([t[0] for t in tails])) if tails else None}]
        if split:
            launches.append({"kernel": last["second_kernel_name"], "kernel_ms": s_ms, "pairs": int(last["second_pairs"]), "algorithmic_tflop": last["second_flops"] / 1e12,
                             "padded_4m_tflop": last["second_padded_flops"] / 1e12, "algorithmic_gbytes": last["second_bytes"] / 1e9 * bytes_scale,
                             "achieved_tflops": last["second_flops"] / (s_ms * 1e-3) / 1e12, "tail_frac": float(np.mean([t[1] for t in tails])) if tails else None})
        for ln in launches:
            ln["frac_of_mfma_peak"] = ln["achieved_tflops"] / peak
            ln["hbm_algorithmic_tb_per_s"] = ln["algorithmic_gbytes"] / ln["kernel_ms"] if ln["kernel_ms"] > 0 else 0.0  # GB / ms = TB / s
        whole_tflops = my["flops"] / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
        alg_bytes = my["bytes"] * bytes_scale
        intensity = my["flops"] / alg_bytes if alg_bytes > 0 else 0.0  # algorithmic flop per byte
        ridge = peak * 1e12 / PEAK_HBM_BYTES  # ~9.8 flop/B for fp64
        bound = "mfma" if intensity >= ridge else "hbm"
        # HBM / fabric bytes per sweep: quoted from the committed rocprofv3 --pmc summary of the SAME workload -- only while that summary
        # was taken on the sources this run was built from (profiles/run_rocprof.sh records their digest)
        executed = None
        traffic, traffic_src = None, "none (a reduced or non-default workload: no committed PMC summary describes it)"
        cfg_dir = args.config if (world == 1 and args.precision == "f64" and not args.max_bond and not args.points and args.seed == 5 and args.gamma == (0.1 if args.config == "cfg5" else 1.0)) else None
        if cfg_dir:
            per_kernel, traffic_src = quoted_traffic(cfg_dir, [ln["kernel"] for ln in launches])
            if per_kernel:
                for ln in launches:
                    ln["traffic_bytes"], ex, more = per_kernel[ln["kernel"]]
                    ln.update({f"pmc_{k_}": v_ for k_, v_ in more.items()})  # L2 hit rate, LDS shares, matrix-pipe busy under the profiler
                    if ex:  # what the matrix cores executed (3M complex product, k-steps trimmed to the true bonds, edge blocks, merged steps)
                        ln["executed_mfma_tflop"] = ex / 1e12
                        ln["executed_over_algorithmic_4m"] = ex * 4.0 / 3.0 / (ln["algorithmic_tflop"] * 1e12) if ln["algorithmic_tflop"] > 0 else None
                traffic = float(sum(v_[0] for v_ in per_kernel.values()))
                executed = sum(v_[1] for v_ in per_kernel.values() if v_[1]) if all(v_[1] for v_ in per_kernel.values()) else None
                for ln in launches:  # what binds THIS launch, from the measured resources
                    ln["traffic_tb_per_s"] = ln["traffic_bytes"] / (ln["kernel_ms"] * 1e-3) / 1e12 if ln["kernel_ms"] > 0 else None
                    if ln.get("executed_mfma_tflop"):  # share of the launch during which the matrix pipes would be busy at peak clock: executed flops / time / peak
                        ln["matrix_pipe_frac"] = ln["executed_mfma_tflop"] / (ln["kernel_ms"] * 1e-3) / peak
                traffic_src = f"quoted: {traffic_src} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of these kernels on this workload, gfx950 FETCH_SIZE correction as calibrated in profiles/r03/fetch_calibration.txt; source digest {sweep_source_sha()} matches)"
            else:
                traffic_src = f"none ({traffic_src})"
        # per launch: `fabric` when the measured L2 <-> fabric traffic runs at >= 6 TB/s (the guide rates the fabric at ~6.3 TB/s achievable),
        # else the roof the algorithmic intensity points at
        for ln in launches:
            li = (ln["algorithmic_tflop"] * 1e12) / (ln["algorithmic_gbytes"] * 1e9) if ln["algorithmic_gbytes"] > 0 else 0.0
            ln["bound"] = "fabric" if (ln.get("traffic_tb_per_s") or 0.0) >= 6.0 else ("mfma" if li >= ridge else "hbm")
        plan_cost = job.plan.cost()
        out = {
            "metric": "Gram kernel entries/sec @ 60 qubits x 6 layers" if args.config == "cfg4" else f"Gram kernel entries/sec @ {n} qubits x {reps} layers",
            "value": npts * npts / (ms_per_step * 1e-3),
            "unit": "entries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            # the COLD Gram (fresh set, fresh plan -> K on the host; max over ranks): what the product pays, it computes every Gram once
            "cold_step_ms": cold["cold_step_ms"],
            "cold_over_steady": cold["cold_step_ms"] / ms_per_step if ms_per_step > 0 else None,
            "plan_ms": cold["plan_ms"],
            "derive_ms": cold["derive_ms"],
            "cold_c_abi_ms": cold_abi["ms"] if cold_abi else None,  # the same through qk_gram_host alone (no torch): the product's one-rank route
            "dtype": "f64 (complex128)" if args.precision == "f64" else "f32 (complex64)",
            "data": "synthetic features (normal -> standardise -> MinMax[0,2], seed %d); real ansatz MPS built %s" % (args.seed, f"on the device, bonds cut at {args.max_bond}" if args.max_bond else "on the host"),
            "config": {
                "workload": f"{args.config}: {n} qubits x {reps} layers, d={d}, gamma={args.gamma}, truncation 1e-16{f', bond cap {args.max_bond}' if args.max_bond else ''}, {npts}x{npts} symmetric training Gram",
                "unique_pairs": int(job.plan.total_pairs),
                "overlaps_per_s": job.plan.total_pairs / (ms_per_step * 1e-3),
                "parallelism": f"states sorted by weight, Gram cut into 8 x 8 tiles of pairs, tiles dealt by cost to {world} rank(s) and, per rank, to 8 per-XCD work queues; one {'RCCL' if backend == 'nccl' else backend} all-gather of packed values",
                "max_bond_mean": float(chi_max.mean()),
                "max_bond_max": int(chi_max.max()),
                "mps_gib": info["device_bytes"] / 2**30,
                # the input producer (not part of the timed step): host pool (cpu-s per state, measured when built here) and the device builder
                "mps_build": capped_build if capped_build else {"host_cpu_s_per_state": binfo["cpu_s_per_state"], "host_workers": workers, "host_pool_s": binfo["cpu_s_per_state"] * npts / workers,
                              "host_pool_wall_s_this_run": binfo["build_wall_s"] if binfo["built_here"] == npts else None, **(dev_build or {})},
                "diag_err": diag_err,
                "sym_err": sym_err,
                "rank_kernel_ms": [round(float(t), 3) for t, _, _, _ in per_rank],
                "rank_padded_tflop": [round(float(f), 5) for _, _, f, _ in per_rank],
                "rank_tail_frac": [round(float(t), 5) for _, _, _, t in per_rank],
                "k_identical_on_all_ranks": len({h for _, h, _, _ in per_rank}) == 1,
                # the cold Gram by rank: host planning (threads stated), the job's set-up around it (buffers, the all-gather of the pair tables),
                # the kernels that make the set's derived images, the first sweep, the all-gather of the values (it waits for the slowest rank)
                "rank_cold_ms": [round(c_["rank_cold_ms"], 2) for c_ in cold_rank],
                "rank_plan_ms": [round(c_["plan_ms"], 2) for c_ in cold_rank],
                "rank_job_setup_ms": [round(c_["job_setup_ms"], 2) for c_ in cold_rank],
                "rank_derive_ms": [round(c_["derive_ms"], 2) for c_ in cold_rank],
                "rank_cold_sweep_ms": [round(c_["sweep_ms"], 2) for c_ in cold_rank],
                "rank_allgather_ms": [round(float(v), 3) for v in (steady_gather_ms if isinstance(steady_gather_ms, list) else [steady_gather_ms])],
                "plan_threads": int(cold["plan_threads"]),
                "host_cpu": _host_cpu(),
                **({"cold_c_abi_vs_steady_max_abs": cold_abi["max_abs_vs_steady"]} if cold_abi else {}),
                **({"f32_vs_f64_max_abs": float(np.abs(Kh - k64_ref).max()), "f32_vs_f64_median_abs": float(np.median(np.abs(Kh - k64_ref)))} if k64_ref is not None else {}),
                **({"device_built_vs_host_built_gram_max_abs": float(np.abs(K_dev - Kh).max())} if K_dev is not None else {}),
            },
            "roofline": {
                # which roof binds THIS workload: algorithmic flop per algorithmic byte against the ridge of the dtype's matrix peak over HBM
                "bound": bound,
                "kernel": kernel_name if not split else f"{kernel_name} + {last['second_kernel_name']} (one sweep, two launches)",
                "achieved": whole_tflops if bound == "mfma" else alg_bytes / (kms * 1e-3) / 1e9,
                "peak": peak if bound == "mfma" else PEAK_HBM_BYTES / 1e9,
                "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                # (kept at or below 1: on full 16-wide tiles the algorithmic count -- 4 real products per complex product -- can pass the peak of
                #  kernels that issue 3; the unclamped figure is frac_of_mfma_peak, the pipe's real load matrix_pipe_frac)
                "frac": min(1.0, (whole_tflops / peak) if bound == "mfma" else (alg_bytes / (kms * 1e-3) / PEAK_HBM_BYTES)),
                # ALGORITHMIC flops count a complex product as 4 real ones (8 flop per complex multiply-add, true bonds, the cheaper
                # association per site); the sweep kernels issue 3 (3M form), so on full 16-wide tiles `frac` can pass 1 -- its ceiling is 4/3
                # (a set cut at bond 64 reaches 1.03 = 77 % of the matrix pipe); padding to 16 pulls the other way (executed_over_algorithmic_4m)
                "flop_convention": "algorithmic, 4 real products per complex product; the kernels issue 3 (3M): ceiling of frac = 4/3",
                "traffic": traffic,
                "traffic_source": traffic_src,
                "traffic_tb_per_s": (traffic / (kms * 1e-3) / 1e12) if (traffic and kms > 0) else None,
                "kernel_ms": kms,
                "algorithmic_tflop_per_sweep": my["flops"] / 1e12,
                "algorithmic_gbytes_per_sweep": alg_bytes / 1e9,
                # the tile-reuse lower bound on the bytes: every state read once per plan tile (8 x 8 pairs) it takes part in, instead of
                # once per pair -- what a sweep that shared operands perfectly inside a tile would move
                "tile_reuse_gbytes": plan_cost["tile_reuse_bytes"] * bytes_scale / 1e9,
                "traffic_over_tile_reuse": (traffic / (plan_cost["tile_reuse_bytes"] * bytes_scale)) if (traffic and plan_cost["tile_reuse_bytes"] > 0) else None,
                # share of the sweep during which the matrix pipes would be busy at peak clock = executed MFMA flops / time / peak (quoted with the traffic)
                "matrix_pipe_frac": (executed / (kms * 1e-3) / 1e12 / peak) if (executed and kms > 0) else None,
                "algorithmic_flop_per_byte": intensity,
                "ridge_flop_per_byte": ridge,
                "frac_of_mfma_peak": whole_tflops / peak,
                "frac_of_hbm_peak_algorithmic": alg_bytes / (kms * 1e-3) / PEAK_HBM_BYTES if kms > 0 else 0.0,
                # the same flop count with every bond rounded up to the 16-wide MFMA tile (four-product form); the sweep kernels issue
                # 3/4 of the K-trimmed part of it (3M complex product): see profiles/r03/<config>/pmc_summary.json
                "padded_4m_tflop_per_sweep": my["padded_flops"] / 1e12,
                # what the matrix cores executed per sweep (quoted with the traffic, same PMC summary): in four-product equivalents (x 4/3)
                # over the algorithmic count = the padding the kernels really pay (the planner's figure above pads K as well, which they do not)
                "executed_mfma_tflop_per_sweep": executed / 1e12 if executed else None,
                "executed_over_algorithmic_4m": (executed * 4.0 / 3.0 / my["flops"]) if (executed and my["flops"] > 0) else None,
                "work_queues": int(last["queues"]),
                "edge_sites": int(job.plan.edge_sites),  # sites at either end of the chain taken from per-state edge blocks (contraction order chosen on the host)
                "launches": launches,
            },
        }
        if world == 1 and args.cpu_seconds > 0:
            pairs = job.plan.pairs()
            gpu_vals = job.my_vals.cpu().numpy()[: pairs.shape[0]]
            threads = default_workers()
            log(rank, f"CPU baseline: oracle C restatement (zgemm leg + hand loop) on {threads} threads, ~{args.cpu_seconds:.0f}s sample each ...")
            out["cpu_baseline"] = cpu_baseline(states, pairs, job.plan.total_pairs, npts, args.cpu_seconds, gpu_vals, threads)
        print(json.dumps(out), flush=True)
    barrier()
    job.close()
    xset.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
