#!/usr/bin/env python3
"""cfg5's states whose bonds are all <= 16: their Gram on the one-tile wave kernel (the default for such a set) against the 2 x 2-tile
wave2 kernel (QK_WAVE=0) -- would a separate run for the pairs of two such states pay inside a bonds-<=-32 set?
    python tools/small_pairs_ab.py [steps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    import __graft_entry__ as graft

    graft.build()
    from qml_cutensornet_amd import engine
    from qml_cutensornet_amd.builder_pool import default_workers

    n, reps, d, npts = bench.CONFIGS["cfg5"]
    states, _ = bench.build_or_load_states("cfg5", n, reps, d, 0.1, npts, 5, 0, 1, default_workers())
    mb = np.array([m.max_bond() for m in states])
    small = [m for m, b in zip(states, mb) if b <= 16]
    print(f"cfg5: {len(states)} states, largest bond: mean {mb.mean():.1f}, <= 16 for {len(small)} states ({(len(small) * (len(small) + 1) // 2) / (npts * (npts + 1) // 2):.2f} of the pairs)", flush=True)
    ref = None
    for label, env in (("all states, default", {}), ("bonds <= 16 only, default (wave kernel)", {}), ("bonds <= 16 only, QK_WAVE=0 (wave2 kernel)", {"QK_WAVE": "0"})):
        for k in ("QK_WAVE",):
            os.environ.pop(k, None)
        os.environ.update(env)
        sel = states if label.startswith("all") else small
        ctx = engine.Context(0)
        with ctx.upload(sel) as xs:
            K = ctx.gram(xs)
            ms = []
            for _ in range(steps):
                ctx.gram(xs)
                ms.append(ctx.stats()["kernel_ms"])
            name = ctx.stats()["kernel_name"]
        ctx.close()
        if not label.startswith("all"):
            ref = K if ref is None else ref
            err = float(np.abs(K - ref).max())
        else:
            err = 0.0
        print(f"{label:48s} {np.mean(ms):8.3f} ms  {name}  max |K - K_ref| {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
