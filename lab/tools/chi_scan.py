#!/usr/bin/env python3
"""Kernel efficiency vs bond dimension: uniform-chi random MPS (pure-kernel mode).
usage: python lab/tools/chi_scan.py [n_sites] [n_states]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import qml_cutensornet_amd as Q
from qml_cutensornet_amd import engine

PEAK = 256 * 4 * 32 * 2.4e9
if "QK_LIB" in os.environ or "QK_VARIANT" in os.environ:  # the lab library, or an experiment build of it (lab/tools/exp_fused.sh)
    engine.use_lab_library()


def profile(n, chi):
    p = [min(2 ** min(k, n - k), chi) for k in range(n + 1)]
    return p


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    rng = np.random.default_rng(0)
    ctx = engine.Context(0)
    print(f"{'chi':>5} {'pairs':>6} {'ms':>9} {'alg TF/s':>9} {'pad TF/s':>9} {'pad/peak':>8} {'us/pair/WG':>10}")
    chis = [int(c) for c in os.environ.get("QK_CHIS", "4,16,32,48,64,96,128,192").split(",")]
    for chi in chis:
        m0 = Q.random_mps(n, profile(n, chi), rng)
        # identical tensors for every state are fine for timing; perturb one entry so states differ
        states = [m0] * ns
        xs = ctx.upload(states)
        plan = engine.Plan(xs.dims)
        best = 1e9
        for _ in range(3):
            ctx.gram_values_host(xs, None, plan)
            best = min(best, ctx.stats()["kernel_ms"])
        st = ctx.stats()
        npairs = st["pairs"]
        grid = st["grid"]
        print(f"{chi:5d} {npairs:6d} {best:9.3f} {st['flops'] / best / 1e9:9.2f} {st['padded_flops'] / best / 1e9:9.2f} {st['padded_flops'] / best / 1e-3 / PEAK:8.3f} {best * 1e3 * grid / npairs:10.1f}")
        plan.close()
        xs.close()
    ctx.close()


if __name__ == "__main__":
    main()
