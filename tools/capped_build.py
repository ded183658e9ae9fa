#!/usr/bin/env python3
"""The device MPS builder with a bond cap (chi) on a BASELINE config, one launch per workgroup shape:
    python tools/capped_build.py [cfg5] [gamma] [chi] [states] [QK_BUILD_WGS values, e.g. 2 1]
prints kernel seconds, the builder's own statistics (QK_BUILD_DEBUG) and the truncation fidelities."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import qml_cutensornet_amd as Q  # noqa: E402
from qml_cutensornet_amd import engine  # noqa: E402
from qml_cutensornet_amd.data import synthetic_features  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
    gamma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    chi = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    ns = int(sys.argv[4]) if len(sys.argv) > 4 else 128
    shapes = sys.argv[5:] or ["2", "1"]
    n, reps, d, npts = bench.CONFIGS[cfg]
    X = synthetic_features(npts, n, 5)[:ns]
    an = Q.KernelStateAnsatz(n, reps, gamma, Q.entanglement_graph(n, d))
    circs = [an.circuit_for_data(x) for x in X]
    os.environ["QK_BUILD_DEBUG"] = "1"
    ref = None
    for wgs in shapes:
        os.environ["QK_BUILD_WGS"] = wgs
        ctx = engine.Context(0)
        t0 = time.perf_counter()
        dset, info = ctx.build_mps_set(circs, max_bond=chi, truncate=True)
        wall = time.perf_counter() - t0
        K = ctx.gram(dset)
        ref = K if ref is None else ref
        print(f"QK_BUILD_WGS={wgs}: {ns} states of {cfg} at gamma {gamma}, bonds cut at {chi}: kernel {info['kernel_ms'] / 1e3:.2f} s (wall {wall:.2f} s); largest bond {int(info['dims'].max())}, "
              f"fidelity median {np.median(info['fidelity']):.6f} min {info['fidelity'].min():.6f}; max |K - K_first| {np.abs(K - ref).max():.2e}", flush=True)
        dset.close()
        ctx.close()


if __name__ == "__main__":
    main()
