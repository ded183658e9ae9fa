#!/usr/bin/env python3
"""The device builder on the HEAVIEST states of cfg4 alone (indices by final bond weight), with its statistics: where does the time of a
heavy state go (block factorisations: sort / Gram-Schmidt / sweeps / W = A V; scalar sweeps; the rest)?
usage: python lab/tools/heavy_states.py 8"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import qml_cutensornet_amd as Q  # noqa: E402
from qml_cutensornet_amd import engine  # noqa: E402
from qml_cutensornet_amd.data import synthetic_features  # noqa: E402

if os.environ.get("QK_AB_LIB"):  # another build of the library on the same box
    engine.LIB_PATH = os.path.abspath(os.environ["QK_AB_LIB"])

HEAVY = [212, 37, 121, 14, 127, 42, 123, 77]  # cfg4, seed 5: the states with the largest sum of chi^3 (host-built bonds)


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n, reps, d, npts = 60, 6, 2, 500
    X = synthetic_features(npts, n, 5)
    an = Q.KernelStateAnsatz(n, reps, 1.0, Q.entanglement_graph(n, d))
    circs = [an.circuit_for_data(X[i]) for i in HEAVY[:k]]
    ctx = engine.Context(0)
    os.environ["QK_BUILD_DEBUG"] = "1"
    ref = None
    for rep in range(2):
        dset, info = ctx.build_mps_set(circs, max_bond=320)
        K = ctx.gram(dset)
        if ref is None:
            ref = (info["dims"].copy(), K.copy())
        print(f"{k} heaviest states: kernel {info['kernel_ms'] / 1e3:.3f} s; max bonds {info['dims'].max(axis=1)}; same bonds as the first build: "
              f"{np.array_equal(info['dims'], ref[0])}; max |K - K_first| = {np.abs(K - ref[1]).max():.2e}", flush=True)
        dset.close()


if __name__ == "__main__":
    main()
