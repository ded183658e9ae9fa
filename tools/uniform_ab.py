#!/usr/bin/env python3
"""A/B of an environment switch on UNIFORM chains (every state the same bond cap): python tools/uniform_ab.py CAP[,CAP..] KEY=V [KEY=V ..]
Forces the fused kernels (QK_FUSED=2, one-wave sweeps off); prints kernel ms per Gram for each setting."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(QK_FUSED="2", QK_WAVE="0", QK_WAVE2="0", QK_SMALL="0")
import qml_cutensornet_amd as Q  # noqa: E402
from qml_cutensornet_amd import engine  # noqa: E402

if os.environ.get("QK_AB_LIB"):  # another build of the library (build_variant) on the same box
    engine.LIB_PATH = os.path.abspath(os.environ["QK_AB_LIB"])


def main():
    caps = [int(c) for c in sys.argv[1].split(",")]
    settings = [s.split(",") for s in sys.argv[2:]]
    rng = np.random.default_rng(1)
    n = 60
    for cap in caps:
        prof = [min(2 ** min(k, n - k), cap) for k in range(n + 1)]
        ns = 181 if cap <= 96 else 61  # (fewer states at large bonds: the host makes them)
        print(f"cap {cap}: making {ns} states", flush=True)
        states = [Q.random_mps(n, prof, rng) for _ in range(ns)]
        ref = None
        for st in settings:
            keys = [kv.split("=")[0] for kv in st]
            for kv in st:
                k, v = kv.split("=")
                os.environ[k] = v
            ctx = engine.Context(0)
            with ctx.upload(states) as xs:
                K = ctx.gram(xs)
                ms = []
                for _ in range(3):
                    ctx.gram(xs)
                    ms.append(ctx.stats()["kernel_ms"])
                name = ctx.stats()["kernel_name"]
            ctx.close()
            for k in keys:
                os.environ.pop(k, None)
            ref = K if ref is None else ref
            print(f"cap {cap:3d} {' '.join(st):40s} {np.mean(ms):8.3f} ms  {name}  max |K - K_first| {np.abs(K - ref).max():.1e}", flush=True)


if __name__ == "__main__":
    main()
