#!/usr/bin/env python3
"""Time dist.exchange_sets (packed set images) on the GPU box: (a) 2 ranks over gloo sharing GPU 0 -- the host route:
D2H of the own image, allgather, one upload --, (b) a 1-rank nccl group forced through all_gather_into_tensor -- the device
route of a real multi-GPU node (RCCL, nothing crosses PCIe).  Workload: cfg3-sized set (200 states, 40 sites, bonds <= 100).
usage: python tools/exchange_bench.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def states_for(lo, hi):
    sys.path.insert(0, ROOT)
    import qml_cutensornet_amd as Q

    out = []
    for k in range(lo, hi):
        rng = np.random.default_rng(1000 + k)
        cap = int(rng.integers(40, 101))
        out.append(Q.random_mps(40, [min(2 ** min(s, 40 - s), cap) for s in range(41)], rng))
    return out


def worker(rank, world, backend, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from qml_cutensornet_amd import engine
    from qml_cutensornet_amd.dist import TorchComm, exchange_sets

    total = 200
    per = -(-total // world)
    lo, hi = rank * per, min(total, (rank + 1) * per)
    ctx = engine.Context(0)
    local = ctx.upload(states_for(lo, hi))
    comm = TorchComm()
    times = []
    for _ in range(3):
        dist.barrier()
        t0 = time.perf_counter()
        full, secs = exchange_sets(comm, ctx, local, lo, total, force_collective=(world == 1))
        times.append(time.perf_counter() - t0)
        mib = full.info()["device_bytes"] / 2**20
        if full is not local:
            full.close()
    q.put((rank, backend, world, min(times), mib))
    local.close(), ctx.close()
    dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp

    mpc = mp.get_context("spawn")
    for backend, world in (("gloo", 2), ("nccl", 1)):
        q = mpc.Queue()
        port = 29700 + (os.getpid() % 200) + (7 if backend == "nccl" else 0)
        procs = [mpc.Process(target=worker, args=(r, world, backend, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=600) for _ in procs)
        for p in procs:
            p.join(60)
        for rank, be, w, t, mib in res:
            print(f"{be} world {w} rank {rank}: exchange_sets {1e3 * t:.1f} ms for a {mib:.0f} MiB set (best of 3)", flush=True)


if __name__ == "__main__":
    main()
