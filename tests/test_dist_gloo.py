"""The N>1 path on CPU: two gloo ranks shard the pairs with the product's planner, exchange the
packed values through the product's communicator adapter and rebuild the dense Gram.  The
values themselves come from the oracle here (there is no GPU in this tier), so what is under
test is exactly the multi-rank logic: partition, all-gather, scatter/mirror, root-only result."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import qml_cutensornet_amd as Q
        from oracle import restatement as R
        from qml_cutensornet_amd import engine
        from qml_cutensornet_amd.dist import TorchComm, assemble_gram, comm_allgather

        comm = TorchComm()
        assert comm.Get_rank() == rank and comm.Get_size() == world
        n = 8
        X = R.synthetic_features(9, n, 2)
        ans = Q.KernelStateAnsatz(n, 2, 1.0, Q.entanglement_graph(n, 2))
        # each rank builds its contiguous share, then everybody gets everything (as build_kernel_matrix does)
        from qml_cutensornet_amd.gpu_backend.kernel_state_ansatz import _simulate_share
        from qml_cutensornet_amd.mps import MPS

        lo, mine, _, _ = _simulate_share(ans, X, rank, world, 1 - 1e-16, False, "X", want_set=False)
        states = [None] * len(X)  # (on a GPU the shares travel as packed device images: dist.exchange_sets, tests/test_gpu_nccl.py)
        for start, items in comm_allgather(comm, (lo, [m.tensors for m in mine])):
            for off, tensors in enumerate(items):
                states[start + off] = MPS(tensors)
        dims = np.stack([m.bond_dims() for m in states])
        out = {}
        for sym in (True, False):
            ys = None if sym else states[:4]
            plan = engine.Plan(dims, None if sym else dims[:4], world, rank)
            pairs = plan.pairs()
            vals = np.array([abs(R.mps_inner(states[i].tensors, (states if sym else ys)[j].tensors)) ** 2 for i, j in pairs])
            shares = comm_allgather(comm, (pairs, vals))
            K = assemble_gram(len(states) if sym else 4, len(states), [s[0] for s in shares], [s[1] for s in shares], sym)
            out["sym" if sym else "rect"] = K
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_gram(built):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    import qml_cutensornet_amd as Q
    from oracle import restatement as R

    n = 8
    X = R.synthetic_features(9, n, 2)
    K_sv = R.gram_statevector(X, None, 2, 1.0, R.entanglement_graph(n, 2))
    for r in (0, 1):
        assert np.abs(results[r]["sym"] - K_sv).max() < 1e-9
        assert np.abs(results[r]["rect"] - K_sv[:4, :]).max() < 1e-9
    assert np.array_equal(results[0]["sym"], results[1]["sym"])
