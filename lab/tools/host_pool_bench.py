#!/usr/bin/env python3
"""Host builder pools on this machine: serial vs threads vs worker processes (mps.simulate_many).
usage: python lab/tools/host_pool_bench.py n reps d gamma npts"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import qml_cutensornet_amd as Q
from qml_cutensornet_amd.builder_pool import default_workers
from qml_cutensornet_amd.data import synthetic_features
from qml_cutensornet_amd.mps import simulate, simulate_many

n, reps, d, gamma, npts = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])
X = synthetic_features(npts, n, 5)
an = Q.KernelStateAnsatz(n, reps, gamma, Q.entanglement_graph(n, d))
cs = [an.circuit_for_data(x) for x in X]
w = default_workers()
t0 = time.perf_counter()
ref = [simulate(c) for c in cs[:8]]
t_serial = (time.perf_counter() - t0) / 8 * npts
for mode in ("threads", "procs"):
    os.environ["QK_BUILDER_POOL"] = mode
    t0 = time.perf_counter()
    out, secs = simulate_many(cs, workers=w)
    print(f"{n}q x {reps} d={d} gamma={gamma}, {npts} circuits, {w} workers, {mode}: {time.perf_counter() - t0:.2f} s wall "
          f"(sum of per-circuit times {sum(secs):.1f} s; serial estimate {t_serial:.1f} s)", flush=True)
