#!/usr/bin/env python3
"""QK_BUILDER=auto in numbers: device builder with capped bonds + threaded host builder for the states that outgrow the cap.
usage: python lab/tools/dev_builder_hybrid.py "n,reps,d,gamma,npts,cap" ..."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import qml_cutensornet_amd as Q
from qml_cutensornet_amd import engine
from qml_cutensornet_amd.data import synthetic_features
from qml_cutensornet_amd.mps import simulate, simulate_many


def main():
    ctx = engine.Context(0)
    for spec in sys.argv[1:]:
        f = spec.split(",")
        n, reps, d, gamma, npts, cap = int(f[0]), int(f[1]), int(f[2]), float(f[3]), int(f[4]), int(f[5])
        X = synthetic_features(npts, n, 5)
        an = Q.KernelStateAnsatz(n, reps, gamma, Q.entanglement_graph(n, d))
        circs = [an.circuit_for_data(x) for x in X]
        t0 = time.perf_counter()
        dev, info = ctx.build_mps(circs, max_bond=cap, partial=True)
        t_dev = time.perf_counter() - t0
        t0 = time.perf_counter()
        built, _ = simulate_many([circs[k] for k in info["dropped"]]) if info["dropped"] else ([], [])
        for k, m in zip(info["dropped"], built):
            dev[k] = m
        t_host = time.perf_counter() - t0
        k = 6
        t0 = time.perf_counter()
        ref = [simulate(c) for c in circs[:k]]
        t_ref = (time.perf_counter() - t0) / k
        with ctx.upload(dev[:k]) as xs, ctx.upload(ref) as ys:
            z = np.abs(np.diag(ctx.overlaps(xs, ys))) ** 2
        print(f"{n}q x {reps} layers d={d} gamma={gamma}, {npts} states, cap {cap}: device {t_dev:.2f} s (kernel {info['kernel_ms'] / 1e3:.2f} s), "
              f"{len(info['dropped'])} dropped states on the host pool {t_host:.2f} s; all on the host: {t_ref * npts:.1f} s on one core = {t_ref * npts / 16:.2f} s on 16; "
              f"|<dev|host>|^2 - 1 = {np.abs(z - 1).max():.1e}", flush=True)


if __name__ == "__main__":
    main()
