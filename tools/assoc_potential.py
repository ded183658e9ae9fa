#!/usr/bin/env python3
"""What a per-SITE choice of the contraction association could still buy (X1, north star: "contraction order chosen on the host").

The site-fused sweep contracts a site as  T = X^T B_k,  X' = T^T conj(A_k)  ("B first"); the other association is
U = X conj(A_k),  X' = B_k^T U  ("A first" = the same kernel code with the roles of the two states exchanged and X transposed).
The planner already lists each pair of a symmetric Gram in the cheaper of the two orders for the WHOLE chain (QK_PLAN_ORIENT).
This script prices, in matrix instructions as the planner's `fused_cost` does (tiles of 16, K trimmed to the true bond in steps of
4, 3M product), what choosing per site would add on top of that, over a random sample of the pairs of a bench config's states:

  * with free switches (a lower bound on the instructions, not reachable: X has to be transposed at every switch);
  * with a switch priced at SW instructions (two barriers and a transpose of X through L2: about the fixed cost of a site,
    375 instructions = 2.5 us of a 12-wave workgroup, tools/site_overhead.py), by dynamic programming along the chain.

    python tools/assoc_potential.py [cfg4] [pairs] [edge_k]     (CPU only; builds or loads the config's states like bench.py)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    ek = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    n, reps, d, npts = bench.CONFIGS[cfg]
    gamma = 0.1 if cfg == "cfg5" else 1.0
    states, _ = bench.build_or_load_states(cfg, n, reps, d, gamma, npts, 5, 0, 1, os.cpu_count() or 1)
    dims = np.array([m.bond_dims() for m in states], dtype=np.int64)
    ns = dims.shape[1] - 1
    t16 = lambda v: (v + 15) // 16  # noqa: E731
    k4 = lambda v: (v + 3) // 4  # noqa: E731

    def site_costs(a, b):
        s = slice(ek, ns - ek)
        a0, a1, b0, b1 = a[:-1][s], a[1:][s], b[:-1][s], b[1:][s]
        b_first = 6 * t16(a0) * t16(b1) * k4(b0) + 6 * t16(b1) * t16(a1) * k4(a0)
        a_first = 6 * t16(b0) * t16(a1) * k4(a0) + 6 * t16(a1) * t16(b1) * k4(b0)
        return b_first, a_first

    rng = np.random.default_rng(0)
    iu = np.triu_indices(npts, 1)
    sel = rng.choice(iu[0].shape[0], min(n_pairs, iu[0].shape[0]), replace=False)
    cur = free = 0.0
    dp = {100.0: 0.0, 200.0: 0.0, 400.0: 0.0}
    switches = 0
    for t in sel:
        i, j = iu[0][t], iu[1][t]
        bf, af = site_costs(dims[i], dims[j])
        if af.sum() < bf.sum():  # QK_PLAN_ORIENT: the pair is listed the other way round
            bf, af = af, bf
        cur += bf.sum()
        free += np.minimum(bf, af).sum()
        switches += int(np.count_nonzero(np.diff((af < bf).astype(np.int8))))
        for sw in dp:
            c0 = c1 = 0.0
            for s in range(bf.shape[0]):
                c0, c1 = min(c0, c1 + sw) + bf[s], min(c1, c0 + sw) + af[s]
            dp[sw] += min(c0, c1)
    print(f"{cfg}: {len(sel)} pairs, sites {ek}..{ns - ek - 1} (edge blocks outside), matrix instructions per pair (mean) {cur / len(sel):.0f}")
    print(f"  per-site minimum, switches free: {free / cur:.4f} of the per-pair choice ({switches / len(sel):.1f} switches per chain)")
    for sw, v in dp.items():
        print(f"  per-site choice, {sw:.0f} instructions per switch: {v / cur:.4f}")


if __name__ == "__main__":
    main()
