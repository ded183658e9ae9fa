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
    return states, {"built_here": built, "build_wall_s": build_wall, "cpu_s_per_state": float(np.mean(secs))}


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
    job = GramJob(ctx, xset, None, world, rank)
    my = job.work[rank]
    log(rank, f"uploaded {info['device_bytes'] / 2**30:.2f} GiB in {upload_s:.1f}s; rank 0 share: {my['pairs']} pairs, {my['flops'] / 1e12:.2f} TFlop algorithmic ({my['padded_flops'] / 1e12:.2f} padded)")

    host_K = torch.empty((npts, npts), dtype=torch.float64, pin_memory=True)

    def step():
        K = job.enqueue()
        host_K.copy_(K, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        st_ = ctx.stats()
        second_ms.append(st_["second_ms"])
        return st_["kernel_ms"]

    kernel_name = None
    second_ms = []  # a split sweep (two launches, two shapes of the site-fused kernel): device time of the second launch

    for _ in range(args.warmup):
        step()
    kernel_name = ctx.stats()["kernel_name"] if args.warmup else None
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ms = []
    second_ms.clear()
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

    # sanity on the result itself (invariants of a Gram of normalised states)
    Kh = host_K.numpy()
    diag_err = float(np.abs(np.diag(Kh) - 1).max())
    sym_err = float(np.abs(Kh - Kh.T).max())
    # after the timed region: every rank's kernel time and a digest of its copy of K (all ranks hold the full matrix)
    per_rank = [(kms, hashlib.sha1(np.ascontiguousarray(Kh).tobytes()).hexdigest(), my["padded_flops"] / 1e12)]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered

    out = None
    if rank == 0:
        peak = PEAK_F64_MFMA_TFLOPS if args.precision == "f64" else PEAK_F32_MFMA_TFLOPS
        # A split sweep is two launches (qk_stats.second_*): the roofline object is that of the DOMINANT one -- the first,
        # `kernel_name`, with its own pairs' flops and its own duration -- and the second is listed beside it.
        last = ctx.stats()
        s_ms = float(np.mean(second_ms)) if second_ms else 0.0
        split = last["second_kernel"] != 0 and s_ms > 0
        first_ms = kms - s_ms if split else kms
        first_flops = my["flops"] - (last["second_flops"] if split else 0.0)
        achieved = first_flops / (first_ms * 1e-3) / 1e12 if first_ms > 0 else 0.0
        second = None
        if split:
            a2 = last["second_flops"] / (s_ms * 1e-3) / 1e12
            second = {"kernel": last["second_kernel_name"], "kernel_ms": s_ms, "pairs": int(last["second_pairs"]), "algorithmic_tflop_per_launch": last["second_flops"] / 1e12,
                      "padded_4m_tflop_per_launch": last["second_padded_flops"] / 1e12, "achieved": a2, "frac": a2 / peak}
        # HBM/fabric bytes per launch of the dominant kernel: PMC numbers cannot be collected from inside this
        # process, so the committed rocprofv3 --pmc summary of the SAME workload (profiles/run_rocprof.sh) is quoted.
        traffic = None
        pmc_file = os.path.join(ROOT, "profiles", "r02", "pmc_summary.json")
        if world == 1 and args.precision == "f64" and args.config == "cfg4" and args.gamma == 1.0 and not args.points and args.seed == 5 and os.path.exists(pmc_file):
            try:
                pmc = json.load(open(pmc_file))
                traffic = pmc["derived"]["traffic_bytes_per_launch"] if kernel_name in pmc["kernel"] else None
            except (KeyError, ValueError):
                traffic = None
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
            "dtype": "f64 (complex128)" if args.precision == "f64" else "f32 (complex64)",
            "data": "synthetic features (normal -> standardise -> MinMax[0,2], seed %d); real ansatz MPS built on the host" % args.seed,
            "config": {
                "workload": f"{args.config}: {n} qubits x {reps} layers, d={d}, gamma={args.gamma}, truncation 1e-16, {npts}x{npts} symmetric training Gram",
                "unique_pairs": int(job.plan.total_pairs),
                "overlaps_per_s": job.plan.total_pairs / (ms_per_step * 1e-3),
                "parallelism": f"pairs in cost order dealt in serpentine order to {world} rank(s); one {'RCCL' if backend == 'nccl' else backend} all-gather of packed values",
                "max_bond_mean": float(chi_max.mean()),
                "max_bond_max": int(chi_max.max()),
                "mps_gib": info["device_bytes"] / 2**30,
                "mps_build_cpu_s_per_state": binfo["cpu_s_per_state"],
                "diag_err": diag_err,
                "sym_err": sym_err,
                "rank_kernel_ms": [round(float(t), 3) for t, _, _ in per_rank],
                "rank_padded_tflop": [round(float(f), 5) for _, _, f in per_rank],
                "k_identical_on_all_ranks": len({h for _, h, _ in per_rank}) == 1,
                **({"f32_vs_f64_max_abs": float(np.abs(Kh - k64_ref).max()), "f32_vs_f64_median_abs": float(np.median(np.abs(Kh - k64_ref)))} if k64_ref is not None else {}),
            },
            "roofline": {
                "bound": "mfma",  # fp64 matrix cores for f64, fp32 matrix cores for f32
                "kernel": kernel_name,
                "achieved": achieved,
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": achieved / peak,
                "traffic": traffic,
                "traffic_source": "profiles/r02/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel on this workload, FETCH_SIZE x2 gfx950 correction), bytes per launch" if traffic else None,
                # L2<->fabric rate those bytes imply at this run's kernel time (Infinity-Cache hits included; HBM peak ~8 TB/s)
                "traffic_tb_per_s": (traffic / (first_ms * 1e-3) / 1e12) if (traffic and first_ms > 0) else None,
                "kernel_ms": first_ms,
                "algorithmic_tflop_per_launch": first_flops / 1e12,
                # the same count with every bond rounded up to the 16-wide MFMA tile (four-product form); the sweep kernels
                # issues 3/4 of the K-trimmed part of it (3M complex product): see profiles/r02/pmc_summary.json
                "padded_4m_tflop_per_launch": (my["padded_flops"] - (last["second_padded_flops"] if split else 0.0)) / 1e12,
                "algorithmic_gbytes_per_launch": (my["bytes"] - (last["second_bytes"] if split else 0.0)) / 1e9 * (1.0 if args.precision == "f64" else 0.5),
                # both launches of a split sweep together: all pairs' flops over the whole device time
                "whole_sweep": {"kernel_ms": kms, "algorithmic_tflop": my["flops"] / 1e12, "achieved": (my["flops"] / (kms * 1e-3) / 1e12) if kms > 0 else 0.0,
                                "frac": (my["flops"] / (kms * 1e-3) / 1e12 / peak) if kms > 0 else 0.0},
                "second_launch": second,
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
