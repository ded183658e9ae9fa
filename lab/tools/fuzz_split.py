#!/usr/bin/env python3
"""GPU box: random heterogeneous sets (small and large states side by side) through the default Gram path -- split sweep,
dual form, wave sweeps, whatever the planner picks -- against the oracle's C restatement.
usage: python lab/tools/fuzz_split.py [cases]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as graft
import qml_cutensornet_amd as Q
from qml_cutensornet_amd import engine

if os.environ.get("QK_AB_LIB"):  # another build of the library (lab/libqkgram_<variant>.so): before build() loads the shipped one
    engine.LIB_PATH = os.path.abspath(os.environ["QK_AB_LIB"])
graft.build()
from oracle import c_oracle


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(12345)
    ctx = engine.Context(0)
    worst, kernels = 0.0, {}
    for case in range(cases):
        n = int(rng.integers(6, 26))
        ns = int(rng.integers(3, 12))
        caps = rng.choice([8, 20, 30, 40, 56, 64, 80, 100, 130, 180], size=ns)
        states = []
        for c in caps:
            prof = [1]
            for k in range(1, n):
                cap = min(2 ** min(k, n - k, 20), int(c), 2 * prof[-1])
                prof.append(int(rng.integers(max(1, cap // 2), cap + 1)))
            prof.append(1)
            for k in range(n - 1, 0, -1):
                prof[k] = min(prof[k], 2 * prof[k + 1])
            states.append(Q.random_mps(n, prof, rng))
        with ctx.upload(states) as d:
            K = ctx.gram(d)
            st = ctx.stats()
        pairs = np.array([(i, j) for j in range(ns) for i in range(j + 1)], dtype=np.int32)
        v_ref, _, _ = c_oracle.gram_pairs([m.tensors for m in states], [m.tensors for m in states], pairs)
        K_ref = np.zeros((ns, ns))
        K_ref[pairs[:, 1], pairs[:, 0]] = v_ref
        K_ref[pairs[:, 0], pairs[:, 1]] = v_ref
        err = float(np.abs(K - K_ref).max())
        worst = max(worst, err)
        name = st["kernel_name"] + (" + " + st["second_kernel_name"] if st["second_kernel"] else "")
        kernels[name] = kernels.get(name, 0) + 1
        print(f"case {case}: n={n} states={ns} caps={sorted(caps.tolist())} {name}: max |K - K_ref| = {err:.2e}", flush=True)
    print("kernels:", kernels)
    print("worst", worst)
    if not worst < 1e-11:
        raise SystemExit("FUZZ FAILED")
    ctx.close()


if __name__ == "__main__":
    main()
