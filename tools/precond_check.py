#!/usr/bin/env python3
"""First light of the builder's preconditioned block factorisation (csrc/qk_build.hip: jacobi_precond) on the GPU box:
the primitive against LAPACK on random, rank-deficient and GRADED matrices (the shape of a gate's theta), with timings;
then cfg4-shaped circuits through the builder against the host builder.
usage: python tools/precond_check.py [n_states]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qml_cutensornet_amd as Q
from qml_cutensornet_amd import engine
from qml_cutensornet_amd.data import synthetic_features
from qml_cutensornet_amd.mps import simulate


def graded(rng, p, q, decades):
    u, _ = np.linalg.qr(rng.standard_normal((p, q)) + 1j * rng.standard_normal((p, q)))
    v, _ = np.linalg.qr(rng.standard_normal((q, q)) + 1j * rng.standard_normal((q, q)))
    s = 10.0 ** (-decades * np.arange(q) / max(1, q - 1))
    return (u * s) @ v.conj().T


def check_primitive(ctx, rng):
    cases = [("random", 64, 48, None), ("random", 130, 100, None), ("rank", 96, 64, 20), ("graded22", 78, 66, 22), ("graded22", 160, 128, 22), ("graded22", 256, 256, 22),
             ("graded22", 384, 320, 22), ("graded22", 512, 512, 22), ("graded12", 512, 256, 12), ("random", 512, 512, None)]
    for kind, p, q, par in cases:
        if kind == "random":
            a = rng.standard_normal((p, q)) + 1j * rng.standard_normal((p, q))
        elif kind == "rank":
            a = (rng.standard_normal((p, par)) + 1j * rng.standard_normal((p, par))) @ (rng.standard_normal((par, q)) + 1j * rng.standard_normal((par, q)))
        else:
            a = graded(rng, p, q, par)
        w, v, sig, order, sweeps, ms = ctx.debug_jacobi_precond(a)
        u_ref, s_ref, vh_ref = np.linalg.svd(a, full_matrices=False)
        tot = (s_ref ** 2).sum()
        keep = int((np.cumsum((s_ref ** 2)[::-1])[::-1] > 1e-16 * tot).sum())
        sg = sig[order]
        e_s = (np.abs(sg[:keep] - s_ref[:keep]) / s_ref[:keep]).max()
        wk, vk = w[:, order[:keep]], v[:, order[:keep]]
        e_av = np.abs(a @ vk - wk).max() / s_ref[0]
        e_rec = np.abs(wk @ vk.conj().T - (u_ref[:, :keep] * s_ref[:keep]) @ vh_ref[:keep]).max() / s_ref[0]
        e_v = np.abs(vk.conj().T @ vk - np.eye(keep)).max()
        g = wk.conj().T @ wk
        g = g / np.sqrt(np.outer(np.diag(g).real, np.diag(g).real))
        e_w = np.abs(g - np.eye(keep)).max()
        rank = int((sig > 0).sum())
        print(f"precond {kind:9s} {p:4d}x{q:4d}: rank {rank:4d} keep {keep:4d} sweeps {sweeps:2d} {ms:8.2f} ms (sort {ctx.last_precond_ms[1]:.2f} mgs {ctx.last_precond_ms[2]:.2f} sweeps {ctx.last_precond_ms[3]:.2f} VW {ctx.last_precond_ms[4]:.2f}) | sigma rel {e_s:.1e} AV-W {e_av:.1e} recon(kept) {e_rec:.1e} V orth {e_v:.1e} W orth {e_w:.1e}", flush=True)
        assert e_av < 1e-13 and e_rec < 1e-12 and e_v < 1e-11, "primitive out of tolerance"  # (e_s: relative error of the smallest kept values, 1e-9 at most: LAPACK itself is good to eps x the largest)


def compare(ctx, n, reps, d, gamma, npts, pick, label, cap=256):
    X = synthetic_features(500 if n == 60 else npts, n, 5)
    an = Q.KernelStateAnsatz(n, reps, gamma, Q.entanglement_graph(n, d))
    circs = [an.circuit_for_data(x) for x in X]
    if pick == "heavy":
        w = np.array([float((np.sin(np.pi * np.asarray(c.alpha)[np.asarray(c.op) == 2]) ** 2).sum()) for c in circs])
        circs = [circs[i] for i in np.argsort(-w)[:npts]]
    else:
        circs = circs[:npts]
    for blk in ("1", "0") if (npts <= 16 and n < 60) else ("1",):
        os.environ["QK_BUILD_BLOCK"] = blk
        os.environ["QK_BUILD_DEBUG"] = "1"
        t0 = time.perf_counter()
        dev, info = ctx.build_mps(circs, max_bond=cap)
        t_dev = time.perf_counter() - t0
        if blk == "1":
            t0 = time.perf_counter()
            host = [simulate(c) for c in circs[: min(len(circs), 6)]]
            t_host = (time.perf_counter() - t0) / len(host)
        with ctx.upload(dev[: len(host)]) as xs, ctx.upload(host) as ys:
            z = np.abs(np.diag(ctx.overlaps(xs, ys))) ** 2
        same = all(np.array_equal(a.bond_dims(), b.bond_dims()) for a, b in zip(dev, host))
        print(f"{label} block={blk}: {len(circs)} states, device {info['kernel_ms'] / 1e3:.2f} s (wall {t_dev:.2f}); host {t_host:.2f} s/state/core; max bond {max(m.max_bond() for m in dev)}; "
              f"|<dev|host>|^2 - 1 = {np.abs(z - 1).max():.1e}; same bonds: {same}; fidelity diff {max(abs(a.fidelity - b.fidelity) for a, b in zip(dev, host)):.1e}", flush=True)


def main():
    ns = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    ctx = engine.Context(0)
    rng = np.random.default_rng(7)
    check_primitive(ctx, rng)
    compare(ctx, 24, 4, 2, 1.0, 8, "first", "24q x 4 layers")
    compare(ctx, 40, 4, 2, 1.0, 8, "first", "40q x 4 layers (cfg3)")
    compare(ctx, 60, 6, 2, 1.0, ns, "heavy", f"60q x 6 layers (cfg4), the {ns} heaviest by proxy", cap=320)


if __name__ == "__main__":
    main()
