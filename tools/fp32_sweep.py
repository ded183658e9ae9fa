#!/usr/bin/env python3
"""fp32-vs-fp64 tolerance of the Gram sweep: max / median |K32 - K64| and
|z32 - z64| against the number of sites and the bond dimension, on random MPS and on ansatz states.
usage: python tools/fp32_sweep.py   (GPU box)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qml_cutensornet_amd as Q
from qml_cutensornet_amd import engine


def profile(n, chi):
    return [min(2 ** min(k, n - k, 20), chi) for k in range(n + 1)]


def main():
    rng = np.random.default_rng(0)
    ctx = engine.Context(0)
    print("random MPS (uniform bond cap chi), 12 states, all 144 ordered pairs")
    print(f"{'sites':>6} {'chi':>5} {'max|dz|':>10} {'med|dz|':>10} {'max|dK|':>10} {'med|dK|':>10} {'med|z|':>10}")
    for n in (20, 60, 100):
        for chi in (8, 32, 64, 128):
            states = [Q.random_mps(n, profile(n, chi), rng) for _ in range(12)]
            with ctx.upload(states) as d64, d64.to_f32() as d32:
                z64, z32 = ctx.overlaps(d64), ctx.overlaps(d32)
                k64, k32 = ctx.gram(d64), ctx.gram(d32)
            dz, dk = np.abs(z32 - z64), np.abs(k32 - k64)
            print(f"{n:6d} {chi:5d} {dz.max():10.2e} {np.median(dz):10.2e} {dk.max():10.2e} {np.median(dk):10.2e} {np.median(np.abs(z64)):10.2e}")
    print("ansatz states (d=2, truncation 1e-16), 16 points")
    print(f"{'qubits':>6} {'reps':>5} {'gamma':>6} {'chi max':>8} {'max|dK|':>10} {'med|dK|':>10} {'max|diag-1|':>12}")
    from qml_cutensornet_amd.data import synthetic_features
    for n, reps, gamma in ((20, 2, 1.0), (40, 4, 0.5), (40, 4, 1.0), (60, 6, 0.5)):
        X = synthetic_features(16, n, 5)
        ans = Q.KernelStateAnsatz(n, reps, gamma, Q.entanglement_graph(n, 2))
        states = [Q.simulate(ans.circuit_for_data(x), 1 - 1e-16) for x in X]
        with ctx.upload(states) as d64, d64.to_f32() as d32:
            k64, k32 = ctx.gram(d64), ctx.gram(d32)
        dk = np.abs(k32 - k64)
        print(f"{n:6d} {reps:5d} {gamma:6.1f} {max(m.max_bond() for m in states):8d} {dk.max():10.2e} {np.median(dk):10.2e} {np.abs(np.diag(k32) - 1).max():12.2e}")
    # the real states of the configs (first points of each config's own data set, host builder): what cfg5's "fp32 vs fp64 tolerance sweep" asks for
    print("real states of the configs (seed 5, truncation 1e-16): complex64 sets against complex128")
    print(f"{'config':>22} {'points':>6} {'chi max':>8} {'kernel (complex64)':>34} {'max|dK|':>10} {'med|dK|':>10} {'max|diag-1|':>12} {'min K64':>10}")
    from qml_cutensornet_amd.builder_pool import build_states

    for name, n, reps, d, gamma, full, pts in (("cfg3 40q x 4, d=2, g=1", 40, 4, 2, 1.0, 200, 24), ("cfg4 60q x 6, d=2, g=1", 60, 6, 2, 1.0, 500, 24), ("cfg5 100q x 10, d=4, g=.1", 100, 10, 4, 0.1, 1000, 32)):
        X = synthetic_features(full, n, 5)[:pts]
        ans = Q.KernelStateAnsatz(n, reps, gamma, Q.entanglement_graph(n, d))
        states, _ = build_states(ans, X, 1 - 1e-16, min(pts, os.cpu_count() or 1))
        with ctx.upload(states) as d64, d64.to_f32() as d32:
            k64, k32 = ctx.gram(d64), ctx.gram(d32)
            kern = ctx.stats()["kernel_name"]
        dk = np.abs(k32 - k64)
        print(f"{name:>22} {pts:6d} {max(m.max_bond() for m in states):8d} {kern:>34} {dk.max():10.2e} {np.median(dk):10.2e} {np.abs(np.diag(k32) - 1).max():12.2e} {k64.min():10.2e}")
    ctx.close()


if __name__ == "__main__":
    main()
