#!/usr/bin/env python3
"""What does a SMALL site cost in the site-fused sweep?  Uniform chains of 60 sites whose bonds are capped at 16 / 32 / 48 (so every
site is 1 x 1, 2 x 2 or 3 x 3 tiles), forced onto the fused kernels (QK_FUSED=2, one-wave and small-bond sweeps off): kernel time
per pair and site = the per-site overhead plus a few matrix instructions.  usage: python tools/site_overhead.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(QK_FUSED="2", QK_WAVE="0", QK_WAVE2="0", QK_SMALL="0")
import qml_cutensornet_amd as Q  # noqa: E402
from qml_cutensornet_amd import engine  # noqa: E402


def main():
    rng = np.random.default_rng(1)
    n, ns = 60, 181
    ctx = engine.Context(0)
    for cap in (17, 32, 48, 64):
        prof = [min(2 ** min(k, n - k), cap) for k in range(n + 1)]
        states = [Q.random_mps(n, prof, rng) for _ in range(ns)]
        for wgs in ("1", "2"):
            os.environ["QK_FUSED_WGS"] = wgs
            c2 = engine.Context(0)
            with c2.upload(states) as xs:
                c2.gram(xs)
                ms = []
                for _ in range(3):
                    c2.gram(xs)
                    ms.append(c2.stats()["kernel_ms"])
                st = c2.stats()
            pairs = ns * (ns + 1) // 2
            wg = st["grid"]
            per_site_us = np.mean(ms) * 1e3 * wg / pairs / n
            print(f"cap {cap:3d} (padded {(-(-cap // 16)) * 16}), {st['kernel_name']}, grid {wg}: {np.mean(ms):7.3f} ms per Gram = {per_site_us:6.2f} us per pair and site per workgroup", flush=True)
            c2.close()


if __name__ == "__main__":
    main()
