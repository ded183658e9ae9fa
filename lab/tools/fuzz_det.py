#!/usr/bin/env python3
"""GPU box: random heterogeneous sets -- symmetric and rectangular Grams, forced and automatic splits, both forms of the 12-wave shape --
through the DET forms of the site-fused kernels (QK_DETERMINISTIC=1): against the oracle's C restatement, and twice for bit identity.
usage: python lab/tools/fuzz_det.py [cases]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as graft

graft.build()
import qml_cutensornet_amd as Q
from oracle import c_oracle
from qml_cutensornet_amd import engine


def states_of(rng, n, caps):
    out = []
    for c in caps:
        prof = [1]
        for k in range(1, n):
            cap = min(2 ** min(k, n - k, 20), int(c), 2 * prof[-1])
            prof.append(int(rng.integers(max(1, cap // 2), cap + 1)))
        prof.append(1)
        for k in range(n - 1, 0, -1):
            prof[k] = min(prof[k], 2 * prof[k + 1])
        out.append(Q.random_mps(n, prof, rng))
    return out


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rng = np.random.default_rng(2024)
    os.environ["QK_DETERMINISTIC"] = "1"
    worst, kernels, bad = 0.0, {}, 0
    for case in range(cases):
        os.environ["QK_FUSED_SPLIT"] = str(int(rng.integers(1, 3)))
        os.environ["QK_FUSED_DUAL"] = str(int(rng.integers(0, 2)))
        n = int(rng.integers(8, 28))
        nx, ny = int(rng.integers(3, 10)), int(rng.integers(2, 6))
        pool = [40, 56, 64, 80, 100, 130, 180, 220, 300]
        xs = states_of(rng, n, rng.choice(pool, size=nx))
        rect = bool(rng.integers(0, 2)) and ny <= nx
        ys = states_of(rng, n, rng.choice(pool, size=ny)) if rect else None
        runs = []
        for _ in range(2):
            ctx = engine.Context(0)
            dx = ctx.upload(xs)
            dy = ctx.upload(ys) if rect else None
            runs.append(ctx.gram(dx, dy))
            st = ctx.stats()
            ctx.close()
        tx, ty = [m.tensors for m in xs], [m.tensors for m in (ys if rect else xs)]
        pairs = np.array([(i, j) for j in range(len(ty)) for i in range(len(tx))], dtype=np.int32)
        v_ref, _, _ = c_oracle.gram_pairs(tx, ty, pairs)
        K_ref = v_ref.reshape(len(ty), len(tx))
        err = float(np.abs(runs[0] - K_ref).max())
        same = np.array_equal(runs[0], runs[1])
        worst = max(worst, err)
        bad += (not same) or err > 1e-11
        name = st["kernel_name"] + (" + " + st["second_kernel_name"] if st["second_kernel"] else "")
        kernels[name] = kernels.get(name, 0) + 1
        print(f"case {case}: n={n} {'rect' if rect else 'sym '} {nx}x{len(ty)} split={os.environ['QK_FUSED_SPLIT']} dual={os.environ['QK_FUSED_DUAL']} {name}: max |K - K_ref| = {err:.2e}, bit-identical twice: {same}", flush=True)
    print(f"worst {worst:.2e}; failures {bad}; kernels {kernels}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
