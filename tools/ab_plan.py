#!/usr/bin/env python3
"""A/B of planner settings on one config: flat cost-ordered list against XCD-aware tiled queues (QK_PLAN_XCD, QK_PLAN_TILE).
    python tools/ab_plan.py [cfg4] [steps]     -> one line per setting: kernel ms (first / second launch), tail fractions"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    settings = [s.split(",") for s in (sys.argv[3:] or ["QK_PLAN_XCD=0", "QK_PLAN_TILE=8", "QK_PLAN_TILE=4", "QK_PLAN_TILE=16", "QK_PLAN_XCD=0"])]
    import __graft_entry__ as graft
    from qml_cutensornet_amd import engine

    if os.environ.get("QK_AB_LIB"):  # another build of the library (e.g. the previous commit's) on the same box: BEFORE build() loads the shipped one
        engine.LIB_PATH = os.path.abspath(os.environ["QK_AB_LIB"])
    graft.build()
    print(f"library: {engine.LIB_PATH}", flush=True)
    from qml_cutensornet_amd.builder_pool import default_workers

    n, reps, d, npts = bench.CONFIGS[cfg]
    gamma = 0.1 if cfg == "cfg5" else 1.0
    states, binfo = bench.build_or_load_states(cfg, n, reps, d, gamma, npts, 5, 0, 1, default_workers())
    print(f"states: {binfo}", flush=True)
    import torch

    from qml_cutensornet_amd.gram import GramJob

    ref = None
    for st in settings:
        for k in ("QK_PLAN_XCD", "QK_PLAN_TILE", "QK_PLAN_TILE_X", "QK_PLAN_TILE_Y", "QK_PLAN_ORIENT_TILE", "QK_DETERMINISTIC", "QK_FUSED_SPLIT", "QK_PLAN_SPLIT", "QK_EDGE", "QK_MERGE", "QK_DEBUG_ALIAS", "QK_PLAN_FIT", "QK_GANG"):
            os.environ.pop(k, None)
        for kv in st:
            k, v = kv.split("=")
            os.environ[k] = v
        ctx = engine.Context(0)  # (a context per setting: some switches are read when it is made)
        xset = ctx.upload(states)
        world = int(os.environ.get("QK_AB_WORLD", "1"))  # QK_AB_WORLD=8: rank 0's share of an 8-rank job (values only, no exchange)
        if world > 1:
            plan = engine.Plan(xset.dims, None, world, 0, 0, False)
            vals = torch.zeros(max(1, plan.max_pairs_per_rank), dtype=torch.float64, device="cuda")
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            run = lambda: ctx.gram_values(xset, None, plan, vals.data_ptr())  # noqa: E731
            run()
            torch.cuda.synchronize()
            K = vals.cpu().numpy()
        else:
            job = GramJob(ctx, xset, None, 1, 0)
            plan, run = job.plan, job.enqueue
            K = job.run()
        ms, ms2, tf, tf2 = [], [], [], []
        for _ in range(steps):
            run()
            torch.cuda.synchronize()
            s_ = ctx.stats()
            ms.append(s_["kernel_ms"]), ms2.append(s_["second_ms"]), tf.append(s_["tail_frac"]), tf2.append(s_["second_tail_frac"])
        if ref is None:
            ref = K
        err = float(np.abs(K - ref).max()) if K.shape == ref.shape else float("nan")  # (shares of a multi-rank plan differ with the plan)
        print(f"{' '.join(st):32s} kernel {np.mean(ms):8.2f} ms (second launch {np.mean(ms2):7.2f}), tail {np.mean(tf):.4f} / {np.mean(tf2):.4f}, queues {s_['queues']}, edge sites {plan.edge_sites}, {s_['kernel_name']}; max |K - K_first| {err:.2e}", flush=True)
        plan.close()
        xset.close()
        ctx.close()


if __name__ == "__main__":
    main()
