#!/usr/bin/env python3
"""GPU box: the one-tile wave sweep (bonds <= 16) against the 2 x 2-tile wave sweep on the same small-bond sets.
usage: python lab/tools/wave_compare.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as graft

graft.build()
import qml_cutensornet_amd as Q
from qml_cutensornet_amd import engine

rng = np.random.default_rng(1)
for n, chi, ns in ((60, 2, 400), (60, 4, 400), (60, 8, 400), (165, 4, 300), (60, 16, 181)):
    prof = [min(chi, 2 ** min(k, n - k)) for k in range(n + 1)]
    base = [Q.random_mps(n, prof, rng) for _ in range(8)]
    st = (base * (ns // 8 + 1))[:ns]
    for mode in ("1", "0"):
        os.environ["QK_WAVE"] = mode
        c = engine.Context(0)
        with c.upload(st) as d:
            c.gram(d)
            K = c.gram(d)
            s = c.stats()
        print(f"n={n} chi={chi} states={ns} QK_WAVE={mode}: {s['kernel_name']} {s['kernel_ms']:.3f} ms, {s['pairs'] / s['kernel_ms'] / 1e3:.2f} M overlaps/s, diag err {np.abs(np.diag(K) - 1).max():.1e}", flush=True)
        c.close()
