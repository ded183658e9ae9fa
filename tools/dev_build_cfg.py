#!/usr/bin/env python3
"""The device MPS builder on a whole BASELINE config (run on the GPU box): every state of the data set in one launch, against the
host pool on the same box; states compared by overlap and bonds.
usage: python tools/dev_build_cfg.py [cfg4] [max_bond] [host_states]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import qml_cutensornet_amd as Q  # noqa: E402
from qml_cutensornet_amd import engine  # noqa: E402
from qml_cutensornet_amd.builder_pool import build_states, default_workers  # noqa: E402
from qml_cutensornet_amd.data import synthetic_features  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    cap = int(sys.argv[2]) if len(sys.argv) > 2 else 320
    nhost = int(sys.argv[3]) if len(sys.argv) > 3 else 48
    n, reps, d, npts = bench.CONFIGS[cfg]
    gamma = 0.1 if cfg == "cfg5" else 1.0
    X = synthetic_features(npts, n, 5)
    an = Q.KernelStateAnsatz(n, reps, gamma, Q.entanglement_graph(n, d))
    circs = [an.circuit_for_data(x) for x in X]
    # host pool first (worker processes are forked before the GPU is touched): a sample, extrapolated
    idx = np.linspace(0, npts - 1, nhost).astype(int)
    t0 = time.perf_counter()
    host, secs = build_states(an, X[idx], 1.0 - 1e-16, default_workers())
    t_host = time.perf_counter() - t0
    print(f"host pool: {nhost} states in {t_host:.2f} s on {default_workers()} workers ({np.mean(secs):.2f} cpu-s/state, max {np.max(secs):.2f}) => {np.sum(secs) / nhost * npts / default_workers():.1f} s for {npts}", flush=True)
    ctx = engine.Context(0)
    os.environ["QK_BUILD_DEBUG"] = "1"
    for rep in range(2):
        t0 = time.perf_counter()
        dset, info = ctx.build_mps_set(circs, max_bond=cap)
        wall = time.perf_counter() - t0
        print(f"device builder: {npts} states, kernel {info['kernel_ms'] / 1e3:.2f} s, wall {wall:.2f} s (incl. packing the set on the device); max bond {int(info['dims'].max())}", flush=True)
        if rep == 0:
            with ctx.upload(host) as hs:
                K = ctx.gram(dset)
                Kh = ctx.gram(hs)
            sub = K[np.ix_(idx, idx)]
            same = np.array_equal(info["dims"][idx], np.array([m.bond_dims() for m in host]))
            print(f"Gram of the device-built sample against the host-built one: max |dK| = {np.abs(sub - Kh).max():.2e}; same bonds: {same}", flush=True)
        dset.close()


if __name__ == "__main__":
    main()
