#!/usr/bin/env python3
"""Every rank's share of a multi-rank Gram, swept ONE AFTER THE OTHER on the one GPU of the test box: what the strong-scaling curve
of `bench.py --gpus N` is made of before any exchange (the all-gather of the packed values moves 8 bytes per pair).
    python tools/share_times.py [cfg4] [steps] [worlds, e.g. 1,2,4,8] [KEY=V,KEY=V environment of the engine]
Per world size: each rank's pairs, padded work, kernel ms (mean of `steps` sweeps) and tail fraction; the slowest rank against the
one-GPU time = the kernel-time efficiency a node of identical GPUs would reach."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    worlds = [int(w) for w in (sys.argv[3] if len(sys.argv) > 3 else "1,2,4,8").split(",")]
    for kv in (sys.argv[4].split(",") if len(sys.argv) > 4 else []):
        k, v = kv.split("=")
        os.environ[k] = v
    import __graft_entry__ as graft

    graft.build()
    from qml_cutensornet_amd import engine
    from qml_cutensornet_amd.builder_pool import default_workers

    n, reps, d, npts = bench.CONFIGS[cfg]
    gamma = 0.1 if cfg == "cfg5" else 1.0
    states, _ = bench.build_or_load_states(cfg, n, reps, d, gamma, npts, 5, 0, 1, default_workers())
    import torch

    ctx = engine.Context(0)
    xset = ctx.upload(states)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    base = None
    for world in worlds:
        rows = []
        for rank in range(world):
            plan = engine.Plan(xset.dims, None, world, rank, 0, False)
            vals = torch.zeros(max(1, plan.max_pairs_per_rank), dtype=torch.float64, device="cuda")
            ctx.gram_values(xset, None, plan, vals.data_ptr())  # warm-up (the set's edge blocks / merged image for this plan's k)
            torch.cuda.synchronize()
            ms, tails, ms2, t1s, t2s = [], [], [], [], []
            for _ in range(steps):
                ctx.gram_values(xset, None, plan, vals.data_ptr())
                torch.cuda.synchronize()
                s_ = ctx.stats()
                ms.append(s_["kernel_ms"]), ms2.append(s_["second_ms"]), t1s.append(s_["tail_frac"]), t2s.append(s_["second_tail_frac"])
                # share of the WHOLE sweep during which the chip was draining: both launches' tails in ms over the sweep's time
                tails.append((s_["tail_frac"] * (s_["kernel_ms"] - s_["second_ms"]) + s_["second_tail_frac"] * s_["second_ms"]) / s_["kernel_ms"])
            st = plan.stats()
            rows.append((rank, st["pairs"], st["padded_flops"] / 1e12, float(np.mean(ms)), float(np.mean(tails)), plan.edge_sites, float(np.mean(ms2)), float(np.mean(t1s)), float(np.mean(t2s))))
            plan.close()
        slow = max(r[3] for r in rows)
        if base is None:
            base = slow
        print(f"{cfg} world {world}: slowest rank {slow:8.2f} ms, mean {np.mean([r[3] for r in rows]):8.2f} ms; against {base:.2f} ms on one GPU: "
              f"kernel-time efficiency {base / (world * slow):.3f}", flush=True)
        for r in rows:
            print(f"    rank {r[0]}: {r[1]:7d} pairs, {r[2]:7.3f} padded TFlop, {r[3]:8.2f} ms (second launch {r[6]:6.2f}), draining {r[4]:.4f} of the sweep (launch tails {r[7]:.4f} / {r[8]:.4f}), edge sites {r[5]}", flush=True)
    xset.close()
    ctx.close()


if __name__ == "__main__":
    main()
