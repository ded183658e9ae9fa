#!/usr/bin/env python3
"""GPU box: single-tile against dual-tile form of the site-fused sweep on uniform-bond sets (60 sites, 181 states).
usage: python lab/tools/dual_compare.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["QK_FUSED"] = "2"
os.environ["QK_FUSED_WGS"] = "1"
import __graft_entry__ as graft

graft.build()
import qml_cutensornet_amd as Q
from qml_cutensornet_amd import engine

rng = np.random.default_rng(7)
for chi in (48, 64, 80, 96, 128, 160, 192, 256):
    prof = [min(chi, 2 ** min(k, 60 - k)) for k in range(61)]
    states = [Q.random_mps(60, prof, rng) for _ in range(8)]
    states = (states * 23)[:181]
    for mode in ("0", "1"):
        os.environ["QK_FUSED_DUAL"] = mode
        c2 = engine.Context(0)
        with c2.upload(states) as d2:
            c2.gram(d2)
            K = c2.gram(d2)
            st = c2.stats()
        print(f"chi={chi} dual={mode}: kernel {st['kernel_ms']:.2f} ms, {st['padded_flops'] / st['kernel_ms'] / 1e9:.1f} TFLOP/s padded-4M; diag err {np.abs(np.diag(K) - 1).max():.1e}", flush=True)
        c2.close()
