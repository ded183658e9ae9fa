#!/usr/bin/env python3
"""Kernel time of ONE rank's share of the cfg4 Gram for a given world size and planner block (1-GPU box):
how the persistent launch behaves when the share is 1/8 of the pairs (tail effects, ordering).
usage: python lab/tools/rank_share_bench.py [world] [blocks...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from qml_cutensornet_amd import engine


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    blocks = [int(b) for b in sys.argv[2:]] or [16, 512]
    n, reps, d, npts = bench.CONFIGS["cfg4"]
    states, _ = bench.build_or_load_states("cfg4", n, reps, d, 1.0, npts, 5, 0, 1, 16)
    ctx = engine.Context(0)
    xs = ctx.upload(states)
    for block in blocks:
        for rank in (0, world - 1):
            plan = engine.Plan(xs.dims, None, world, rank, block)
            best = 1e9
            for _ in range(3):
                ctx.gram_values_host(xs, None, plan)
                best = min(best, ctx.stats()["kernel_ms"])
            st = plan.stats()
            print(f"world {world} rank {rank} block {block}: {plan.num_pairs} pairs, {st['flops'] / 1e12:.3f} TFlop, kernel {best:.2f} ms, {st['flops'] / best / 1e9:.2f} TFLOP/s algorithmic")
            plan.close()
    xs.close()
    ctx.close()


if __name__ == "__main__":
    main()
