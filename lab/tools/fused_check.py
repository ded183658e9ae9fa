#!/usr/bin/env python3
"""GPU box: the site-fused sweep against the oracle's C restatement on random ragged MPS of several bond regimes
(LDS-resident sites, strips, several strips, mode changes inside one chain) and a timing of cfg4-like uniform sets.
usage: python lab/tools/fused_check.py [quick]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("QK_FUSED", "2")
import __graft_entry__ as graft

graft.build()
import qml_cutensornet_amd as Q
from oracle import c_oracle
from qml_cutensornet_amd import engine


def ragged(rng, n, chi_max, lo=1):
    chi = [1]
    for k in range(1, n):
        cap = min(2 ** min(k, n - k, 20), chi_max, 2 * chi[-1])
        chi.append(int(rng.integers(min(lo, cap), cap + 1)))
    chi.append(1)
    for k in range(n - 1, 0, -1):
        chi[k] = min(chi[k], 2 * chi[k + 1])
    return chi


def main():
    rng = np.random.default_rng(7)
    ctx = engine.Context(0)
    ctx.selftest()
    worst = 0.0
    cases = [(10, 20, 4, 3), (14, 32, 5, 4), (16, 48, 4, 4), (18, 64, 4, 3), (20, 90, 3, 4), (20, 128, 3, 3), (22, 200, 2, 3), (24, 300, 2, 2), (30, 100, 6, 5)]
    for n, chi_max, nx, ny in cases:
        xs = [Q.random_mps(n, ragged(rng, n, chi_max, lo=max(1, chi_max // 3)), rng) for _ in range(nx)]
        ys = [Q.random_mps(n, ragged(rng, n, chi_max, lo=max(1, chi_max // 3)), rng) for _ in range(ny)]
        pairs = np.array([(i, j) for j in range(ny) for i in range(nx)], dtype=np.int32)
        _, z_ref, _ = c_oracle.gram_pairs([m.tensors for m in xs], [m.tensors for m in ys], pairs)
        with ctx.upload(xs) as dx, ctx.upload(ys) as dy:
            z = ctx.overlaps(dx, dy)
        err = float(np.abs(z.reshape(-1) - z_ref).max())
        worst = max(worst, err)
        print(f"n={n} chi<={chi_max} {nx}x{ny}: max |z - z_ref| = {err:.2e}  (max bonds x {max(m.max_bond() for m in xs)}, y {max(m.max_bond() for m in ys)})", flush=True)
    print("worst", worst)
    if not worst < 1e-11:
        raise SystemExit("FUSED PARITY FAILED")
    if len(sys.argv) > 1 and sys.argv[1] == "quick":
        return
    # timing: uniform-bond sets (60 sites, 181 identical-profile states -> 16471 pairs), fused vs ring
    for chi in (32, 48, 64, 96, 128):
        prof = [min(chi, 2 ** min(k, 60 - k)) for k in range(61)]
        states = [Q.random_mps(60, prof, rng) for _ in range(8)]
        states = (states * 23)[:181]
        for mode in ("2", "0"):
            os.environ["QK_FUSED"] = mode
            c2 = engine.Context(0)
            with c2.upload(states) as d2:
                c2.gram(d2)
                K = c2.gram(d2)
                st = c2.stats()
            print(f"chi={chi} QK_FUSED={mode}: kernel {st['kernel_ms']:.2f} ms, {st['padded_flops'] / st['kernel_ms'] / 1e9:.1f} TFLOP/s padded-4M, {st['flops'] / st['kernel_ms'] / 1e9:.1f} algorithmic; diag err {np.abs(np.diag(K) - 1).max():.1e}", flush=True)
            c2.close()
    ctx.close()


if __name__ == "__main__":
    main()
