#!/usr/bin/env python3
"""Three-term cost model of the ring sweep (CPU only): time = cg * GEMM calls + c0 * steps + c1 * tile-steps per workgroup.

The constants are fitted on the uniform-bond scan of the shipped kernel (profiles/r01/chi_scan_shipped_kernel.txt:
chi = 48, 64, 128; chi = 96 is then reproduced to 0.6 %): cg = 2.79 us per GEMM call (first K-tile latency, store drain,
end barrier), c0 = 0.249 us per K-tile step (DMA issue, wait, barrier), c1 = 0.0868 us per 16x16 tile and K-tile of 8
(six MFMAs: 92 % of the pipe rate with two workgroups per CU).  Applied to the bonds of a config it splits the launch
time into the three terms and evaluates alternative pass shapes (how many steps a GEMM needs when a pass may be
pm x pn tiles with pm + pn <= cols and pm * pn <= tiles instead of the fixed 4 x 4).
usage: python lab/tools/cost_model.py [config] [sampled pairs]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench

CG, C0, C1 = 2.79, 0.249, 0.0868  # microseconds
WGS = 512


def steps_for(mt, nt, nk, shapes):
    return min((-(-mt // pm)) * (-(-nt // pn)) * nk for pm, pn in shapes)


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    nsample = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    n, reps, d, npts = bench.CONFIGS[cfg]
    gamma = 0.1 if cfg == "cfg5" else 1.0
    states, _ = bench.build_or_load_states(cfg, n, reps, d, gamma, npts, 5, 0, 1, os.cpu_count() or 1)
    dims = np.array([[1] + [t.shape[2] for t in s.tensors] for s in states])
    pad = lambda x: (x + 15) // 16 * 16
    I, J = np.triu_indices(len(dims))
    rng = np.random.default_rng(0)
    sel = rng.choice(len(I), min(nsample, len(I)), replace=False)
    variants = {
        "shipped 4x4 passes": [(4, 4)],
        "adaptive, 8 tile columns / 16 tiles": [(m, k) for m in range(1, 8) for k in range(1, 8) if m + k <= 8 and m * k <= 16],
        "adaptive, 10 tile columns / 24 tiles": [(m, k) for m in range(1, 9) for k in range(1, 9) if m + k <= 10 and m * k <= 24],
    }
    tot = {k: np.zeros(3) for k in variants}
    cache = {}
    for i, j in zip(I[sel], J[sel]):
        xa, ya = dims[i], dims[j]
        for k in range(n):
            a, a2, b2 = pad(xa[k]), pad(xa[k + 1]), pad(ya[k + 1])
            for g in ((a // 16, 2 * b2 // 16, -(-ya[k] // 8)), (b2 // 16, a2 // 16, -(-(2 * xa[k]) // 8))):
                if g not in cache:
                    cache[g] = {name: (1, steps_for(*g, sh), g[0] * g[1] * g[2]) for name, sh in variants.items()}
                for name in variants:
                    tot[name] += cache[g][name]
    scale = len(I) / len(sel) / WGS / 1e3  # -> ms per workgroup
    print(f"{cfg}: {len(I)} pairs, {len(sel)} sampled; constants cg {CG} us/call, c0 {C0} us/step, c1 {C1} us/tile-step")
    for name, v in tot.items():
        call, step, mfma = CG * v[0] * scale, C0 * v[1] * scale, C1 * v[2] * scale
        print(f"  {name:40s} calls {call:6.1f} ms + steps {step:6.1f} ms + MFMA {mfma:6.1f} ms = {call + step + mfma:6.1f} ms   (fill {v[2] / v[1] / 16:.2f})")


if __name__ == "__main__":
    main()
