#!/usr/bin/env python3
"""Randomised comparison of the device MPS builder with the host builder (run on the GPU box).
usage: python lab/tools/dev_builder_fuzz.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import qml_cutensornet_amd as Q
from qml_cutensornet_amd import engine
from qml_cutensornet_amd.mps import simulate
from oracle import restatement as R


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    ctx = engine.Context(0)
    worst = 0.0
    for c in range(cases):
        n = int(rng.integers(1, 25))
        reps = int(rng.integers(0, 5))
        d = int(rng.integers(0, min(4, max(1, n - 1)) + 1)) if n > 1 else 0
        gamma = float(rng.choice([0.1, 0.5, 1.0, 2.0]))
        npts = int(rng.integers(2, 7))
        fid = float(rng.choice([1 - 1e-16, 1 - 1e-16, 1 - 1e-8, 1 - 1e-3]))
        had = bool(rng.integers(0, 2))
        X = rng.uniform(0, 2, size=(npts, n))
        an = Q.KernelStateAnsatz(n, reps, gamma, Q.entanglement_graph(n, d) if d > 0 else [], hadamard_init=had)
        circs = [an.circuit_for_data(x) for x in X]
        dev, _ = ctx.build_mps(circs, fid)
        host = [simulate(ci, fid) for ci in circs]
        z = np.array([abs(R.mps_inner(a.tensors, b.tensors)) ** 2 for a, b in zip(dev, host)])
        nd = np.array([abs(R.mps_inner(a.tensors, a.tensors)) for a in dev])
        fe = max(abs(a.fidelity - b.fidelity) for a, b in zip(dev, host))
        tol = 1e-10 if fid > 1 - 1e-12 else 50 * (1 - fid)
        err = max(abs(z - 1).max(), abs(nd - 1).max())
        worst = max(worst, err if fid > 1 - 1e-12 else 0.0)
        status = "ok" if err < tol and fe < max(1e-10, 50 * (1 - fid)) else "FAIL"
        print(f"{c:3d} n={n:2d} reps={reps} d={d} gamma={gamma} pts={npts} fid=1-{1 - fid:.0e} H={int(had)}: |<d|h>|^2-1 {abs(z - 1).max():.1e} norm {abs(nd - 1).max():.1e} "
              f"fidelity diff {fe:.1e} bonds dev {max(m.max_bond() for m in dev)} host {max(m.max_bond() for m in host)} {status}", flush=True)
        if status != "ok":
            raise SystemExit(1)
    print(f"all {cases} cases agree; worst deviation at full fidelity {worst:.1e}")


if __name__ == "__main__":
    main()
