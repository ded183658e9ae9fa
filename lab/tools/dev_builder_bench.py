#!/usr/bin/env python3
"""Device MPS builder against the host builder, throughput on one MI355X vs one host core (run on the GPU box).
usage: python lab/tools/dev_builder_bench.py "n,reps,d,gamma,npts[,max_bond]" ..."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import qml_cutensornet_amd as Q
from qml_cutensornet_amd import engine
from qml_cutensornet_amd.data import synthetic_features
from qml_cutensornet_amd.mps import simulate


def main():
    ctx = engine.Context(0)
    for spec in sys.argv[1:]:
        f = spec.split(",")
        n, reps, d, gamma, npts = int(f[0]), int(f[1]), int(f[2]), float(f[3]), int(f[4])
        cap = int(f[5]) if len(f) > 5 else 256
        X = synthetic_features(npts, n, 5)
        an = Q.KernelStateAnsatz(n, reps, gamma, Q.entanglement_graph(n, d))
        circs = [an.circuit_for_data(x) for x in X]
        t0 = time.perf_counter()
        dev, info = ctx.build_mps(circs, max_bond=cap)
        wall = time.perf_counter() - t0
        k = min(npts, 6)
        t0 = time.perf_counter()
        host = [simulate(c) for c in circs[:k]]
        th = (time.perf_counter() - t0) / k
        with ctx.upload(dev[:k]) as xs, ctx.upload(host) as ys:
            z = np.abs(np.diag(ctx.overlaps(xs, ys))) ** 2
        print(f"{n}q x {reps} layers d={d} gamma={gamma}: {npts} states, device kernel {info['kernel_ms'] / 1e3:.2f} s (wall {wall:.2f} s) = "
              f"{info['kernel_ms'] / npts:.1f} ms/state; host builder {th * 1e3:.1f} ms/state/core = {th * npts / 16:.2f} s on 16 cores; "
              f"max bond {max(m.max_bond() for m in dev)}; |<dev|host>|^2 - 1 = {np.abs(z - 1).max():.1e}", flush=True)


if __name__ == "__main__":
    main()
