"""RCCL on the one GPU of the test box, and a 4-rank rehearsal of the multi-GPU bench.

The driver's 8-GPU node is the only place where the nccl branches run with more than one rank.  What can be run here:
(1) a ONE-rank nccl process group that pushes GramJob and the packed-set exchange through ``all_gather_into_tensor``
    (RCCL is loaded, the communicator is created, the collective calls execute on the GPU);
(2) the bench itself with 4 ranks over gloo on GPU 0 (the GPU box admits 6 processes on its card: the test runner, the
    launcher and four ranks): identical K on every rank, balanced shares.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _nccl_worker(port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        import qml_cutensornet_amd as Q
        from qml_cutensornet_amd import engine
        from qml_cutensornet_amd.dist import TorchComm, exchange_sets
        from qml_cutensornet_amd.gram import GramJob

        rng = np.random.default_rng(8)
        prof = [1, 2, 4, 8, 16, 32, 40, 32, 16, 8, 4, 2, 1]
        states = [Q.random_mps(12, prof, rng) for _ in range(9)]
        ctx = engine.Context(0)
        local = ctx.upload(states)
        # the packed-set exchange through RCCL all_gather_into_tensor (one rank: the gathered image is the local one)
        full, secs = exchange_sets(TorchComm(), ctx, local, 0, len(states), force_collective=True)
        job = GramJob(ctx, full, None, 1, 0, force_collective=True)  # all_gather_into_tensor of pairs and of values
        K = job.run()
        K_plain = ctx.gram(local)
        # a LARGE share (hundreds of MB: a fill of the send buffer on another stream would still be running when the image copy
        # lands -- the race the advisor found): the gathered image must be the local one, byte for byte
        big_prof = [min(2 ** min(k, 24 - k), 160) for k in range(25)]
        big = [Q.random_mps(24, big_prof, rng) for _ in range(40)]
        big_local = ctx.upload(big)
        big_full, _ = exchange_sets(TorchComm(), ctx, big_local, 0, len(big), force_collective=True)
        n_a, ptr_a, dims_a, offs_a = big_local.image()
        n_b, ptr_b, dims_b, offs_b = big_full.image()
        img_a = torch.empty(n_a, dtype=torch.float64, device="cuda:0")
        img_b = torch.empty(n_b, dtype=torch.float64, device="cuda:0")
        big_local.copy_image(img_a.data_ptr(), n_a), big_full.copy_image(img_b.data_ptr(), n_b)
        same_image = bool(torch.equal(img_a, img_b[:n_a])) and np.array_equal(dims_a, dims_b) and np.array_equal(offs_a, offs_b)
        q.put({"K": K, "K_plain": K_plain, "backend": dist.get_backend(), "states": [m.tensors for m in states], "exchange_s": secs,
               "big_image_mib": n_a * 8 / 2**20, "big_image_identical": same_image})
        job.close(), full.close(), local.close(), big_full.close(), big_local.close(), ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_one_rank_nccl_group_runs_the_collectives(built):
    import torch.multiprocessing as mp

    from oracle import restatement as R

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(29500 + (os.getpid() % 400), q))
    p.start()
    res = q.get(timeout=240)
    p.join(60)
    assert p.exitcode == 0
    assert res["backend"] == "nccl"
    ref = R.gram_from_mps(res["states"])
    assert np.abs(res["K"] - ref).max() < 1e-11
    # the gathered set and the gathered values change nothing (two launches of the fused sweep may differ in the last bit:
    # the sum over the tiles of a column is taken by LDS atomics, in arrival order)
    assert np.abs(res["K"] - res["K_plain"]).max() < 1e-14
    assert res["big_image_mib"] > 200 and res["big_image_identical"]


@pytest.mark.timeout(900)
def test_four_rank_bench_rehearsal(built, tmp_path):
    """bench.py --gpus 4 on GPU 0 over gloo (QK_FORCE_DEVICE / QK_DIST_BACKEND are the bench's rehearsal hooks): 120 points of
    cfg4.  Every rank must end with the same K and the ranks' shares of the padded work must be equal."""
    env = dict(os.environ, QK_FORCE_DEVICE="0", QK_DIST_BACKEND="gloo", QK_CACHE_DIR=str(tmp_path / "cache"), MASTER_ADDR="127.0.0.1")
    port = 29900 + (os.getpid() % 90)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "4", "--points", "120", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 4 and d["config"]["unique_pairs"] == 120 * 121 // 2
    assert d["config"]["k_identical_on_all_ranks"] is True
    assert d["config"]["diag_err"] < 1e-11 and d["config"]["sym_err"] == 0.0
    ms, work = np.array(d["config"]["rank_kernel_ms"]), np.array(d["config"]["rank_padded_tflop"])
    assert len(ms) == 4 and ms.min() > 0  # (four processes time-share ONE GPU here: their kernel times say nothing about balance)
    assert work.max() / work.min() < 1.02, work  # tiles dealt by cost, the lightest pair by pair: level shares of the padded work
    assert len(d["config"]["rank_tail_frac"]) == 4
    # the cold Gram (fresh set, fresh plan -> K on the host) with its parts by rank: planning, the set's derived images, the first sweep, the all-gather
    assert d["cold_step_ms"] > 0 and d["plan_ms"] > 0 and d["derive_ms"] > 0
    for key in ("rank_cold_ms", "rank_plan_ms", "rank_job_setup_ms", "rank_derive_ms", "rank_cold_sweep_ms", "rank_allgather_ms"):
        assert len(d["config"][key]) == 4 and min(d["config"][key]) >= 0, key
    assert max(d["config"]["rank_cold_ms"]) <= d["cold_step_ms"] + 1e-6 and min(d["config"]["rank_allgather_ms"]) > 0


@pytest.mark.timeout(600)
def test_bench_line_contract(built, tmp_path):
    """`python bench.py` on one GPU (a reduced workload: 60 points of cfg4, whose states differ enough for the split sweep):
    ONE JSON line with the driver's keys, the roofline object of the whole sweep (its launches listed beside it),
    the CPU baseline timed on a bounded sample and checked against the GPU values."""
    env = dict(os.environ, QK_CACHE_DIR=str(tmp_path / "cache"))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--points", "60", "--steps", "2", "--warmup", "1", "--cpu-seconds", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=550, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # one line on stdout, everything else on stderr
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "entries/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 60 * 60 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel", "kernel_ms", "launches", "work_queues"):
        assert key in rf, key
    # the roofline object prices the WHOLE sweep (both launches of a split one); the roof is chosen from the algorithmic intensity
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0 < rf["frac"] < 1
    assert rf["algorithmic_flop_per_byte"] > rf["ridge_flop_per_byte"]
    assert rf["kernel"].startswith("qk_sweep_") and rf["work_queues"] == 8
    assert rf["traffic"] is None and rf["traffic_source"].startswith("none")  # a reduced workload: no committed PMC summary to quote
    ls = rf["launches"]
    assert 1 <= len(ls) <= 2
    # what the steady-state number hides and what binds: the cold Gram, the tile-reuse lower bound on the bytes, a roof per launch
    assert d["cold_step_ms"] >= d["plan_ms"] > 0 and d["derive_ms"] > 0 and d["cold_over_steady"] > 0.9
    assert 0 < rf["tile_reuse_gbytes"] < rf["algorithmic_gbytes_per_sweep"] and rf["matrix_pipe_frac"] is None
    assert all(x["bound"] in ("mfma", "hbm", "fabric") for x in ls)
    assert abs(sum(x["kernel_ms"] for x in ls) - rf["kernel_ms"]) < 1e-6 * rf["kernel_ms"]
    assert abs(sum(x["algorithmic_tflop"] for x in ls) - rf["algorithmic_tflop_per_sweep"]) < 1e-9 * rf["algorithmic_tflop_per_sweep"]
    assert abs(rf["achieved"] - rf["algorithmic_tflop_per_sweep"] / (rf["kernel_ms"] * 1e-3)) < 1e-9 * rf["achieved"]
    for x in ls:
        assert x["kernel"].startswith("qk_sweep_fused") and 0 <= x["tail_frac"] < 0.5
    if len(ls) == 2:
        assert ls[1]["kernel"].startswith("qk_sweep_fused_kernel<8")
    mb = d["config"]["mps_build"]  # the input producer, host pool and device builder side by side (untimed)
    assert mb["host_pool_s"] > 0 and mb["device_kernel_s"] > 0 and mb["largest_bond_difference"] <= 1
    assert d["config"]["device_built_vs_host_built_gram_max_abs"] < 1e-9
    cb = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["parity_max_abs_err_vs_gpu"] < 1e-10
    assert d["config"]["diag_err"] < 1e-11 and d["config"]["sym_err"] == 0.0
