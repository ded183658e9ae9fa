"""Two ranks sharing GPU 0 over gloo: the N>1 code paths with the real HIP kernels.
(NCCL/RCCL itself needs one GPU per rank, which only the driver's multi-GPU node has; everything
around the collective -- plans per rank, padded pair table, scatter, root-only result -- runs here.)"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import qml_cutensornet_amd as Q
        from helpers import golden
        from qml_cutensornet_amd import engine
        from qml_cutensornet_amd.dist import TorchComm
        from qml_cutensornet_amd.gpu_backend.kernel_state_ansatz import KernelStateAnsatz, build_kernel_matrix
        from qml_cutensornet_amd.gram import GramJob

        out = {}
        # (1) the reference's module surface with a 2-rank communicator (host all-gather route)
        g = golden("deep_10q_r3_d3.npz")
        n, reps, gamma, d = int(g["n"]), int(g["reps"]), float(g["gamma"]), int(g["d"])
        ans = KernelStateAnsatz(n, reps, gamma, Q.entanglement_graph(n, d))
        comm = TorchComm()
        out["train"] = build_kernel_matrix(comm, ans, X=g["X_train"], truncation_error=1e-16)
        out["test"] = build_kernel_matrix(comm, ans, X=g["X_train"], Y=g["X_test"], truncation_error=1e-16)
        # (2) the bench/driver device route: GramJob with world = 2 (gloo gather of device buffers)
        rng = np.random.default_rng(3)
        prof = [1, 2, 4, 8, 16, 24, 20, 12, 8, 4, 2, 1]
        states = [Q.random_mps(11, prof, rng) for _ in range(13)]
        torch.cuda.set_device(0)
        ctx = engine.Context(0)
        xs = ctx.upload(states)
        job = GramJob(ctx, xs, None, world, rank)
        out["job"] = job.run()
        out["job_pairs"] = job.plan.num_pairs
        ys = ctx.upload(states[:5])
        job2 = GramJob(ctx, xs, ys, world, rank)
        out["job_rect"] = job2.run()
        out["states"] = [m.tensors for m in states]
        job.close(), job2.close(), xs.close(), ys.close(), ctx.close()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_one_gpu(built):
    import torch.multiprocessing as mp

    from helpers import golden
    from oracle import restatement as R

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g = golden("deep_10q_r3_d3.npz")
    assert res[1]["train"] is None and res[1]["test"] is None  # result lives on rank 0, like reduce(root=0)
    assert np.abs(res[0]["train"] - g["K_train"]).max() < 1e-8
    assert res[0]["test"].shape == g["K_test"].shape and np.abs(res[0]["test"] - g["K_test"]).max() < 1e-8
    ref = R.gram_from_mps(res[0]["states"])
    for r in (0, 1):
        assert np.abs(res[r]["job"] - ref).max() < 1e-11  # every rank holds the full matrix after the gather
        assert np.abs(res[r]["job_rect"] - ref[:5, :]).max() < 1e-11
    assert res[0]["job_pairs"] + res[1]["job_pairs"] == 13 * 14 // 2
    assert abs(res[0]["job_pairs"] - res[1]["job_pairs"]) <= 1
