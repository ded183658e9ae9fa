#!/usr/bin/env python3
"""QK_BUILDER=auto (pilot + concurrent device / host builders) against the host pool alone, on the GPU box.
usage: python lab/tools/builder_policy_bench.py "n,reps,d,gamma,npts" ..."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import qml_cutensornet_amd as Q
from qml_cutensornet_amd import engine
from qml_cutensornet_amd.builder_pool import default_workers
from qml_cutensornet_amd.data import synthetic_features
from qml_cutensornet_amd.gpu_backend import kernel_state_ansatz as M
from qml_cutensornet_amd.mps import simulate_many


def main():
    ctx = engine.default_context(0)
    workers = default_workers()
    for spec in sys.argv[1:]:
        f = spec.split(",")
        n, reps, d, gamma, npts = int(f[0]), int(f[1]), int(f[2]), float(f[3]), int(f[4])
        X = synthetic_features(npts, n, 5)
        an = Q.KernelStateAnsatz(n, reps, gamma, Q.entanglement_graph(n, d))
        circs = [an.circuit_for_data(x) for x in X]
        t0 = time.perf_counter()
        states, _ = M._hybrid_build(ctx, circs, 1 - 1e-16, 64, workers, True, "X")
        t_auto = time.perf_counter() - t0
        t0 = time.perf_counter()
        ref, _ = simulate_many(circs, 1 - 1e-16, workers=workers)
        t_host = time.perf_counter() - t0
        chi = max(m.max_bond() for m in states)
        print(f"{n}q x {reps} layers d={d} gamma={gamma}, {npts} states (max bond {chi}): auto {t_auto:.2f} s, host pool alone ({workers} workers) {t_host:.2f} s", flush=True)


if __name__ == "__main__":
    main()
