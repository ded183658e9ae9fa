#!/usr/bin/env python3
"""First-light checks of the device MPS builder (run on the GPU box): the Jacobi primitive against numpy's SVD, small
circuits against the host builder, and a timing against the host builder on cfg4-shaped circuits.
usage: python lab/tools/dev_builder_check.py [n_states_timing]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import qml_cutensornet_amd as Q
from qml_cutensornet_amd import engine
from qml_cutensornet_amd.mps import simulate
from qml_cutensornet_amd.data import synthetic_features


def check_jacobi(ctx, rng):
    for p, q, rank in [(4, 2, None), (20, 12, None), (47, 43, None), (64, 64, None), (130, 100, None), (60, 40, 17), (33, 33, 1)]:
        a = rng.standard_normal((p, q)) + 1j * rng.standard_normal((p, q))
        if rank:
            a = (rng.standard_normal((p, rank)) + 1j * rng.standard_normal((p, rank))) @ (rng.standard_normal((rank, q)) + 1j * rng.standard_normal((rank, q)))
        t0 = time.perf_counter()
        w, v, sig, order = ctx.debug_jacobi(a)
        dt = time.perf_counter() - t0
        s_ref = np.linalg.svd(a, compute_uv=False)
        e_s = np.abs(sig[order] - s_ref).max() / s_ref[0]
        e_rec = np.abs(w @ v.conj().T - a).max() / np.abs(a).max()
        e_uni = np.abs(v.conj().T @ v - np.eye(q)).max()
        g = w.conj().T @ w
        nz = sig > 1e-12 * sig.max()
        e_orth = np.abs((g - np.diag(np.diag(g)))[np.ix_(nz, nz)] / np.sqrt(np.outer(np.diag(g)[nz], np.diag(g)[nz]).real)).max() if nz.sum() > 1 else 0.0
        print(f"jacobi {p}x{q} rank {rank}: sigma err {e_s:.1e} recon {e_rec:.1e} unitarity {e_uni:.1e} orth {e_orth:.1e}  ({dt*1e3:.1f} ms incl. copies)", flush=True)
        assert e_s < 1e-12 and e_rec < 1e-12 and e_uni < 1e-12 and e_orth < 1e-12


def compare(ctx, n, reps, d, gamma, npts, seed, label, max_bond=256):
    X = synthetic_features(npts, n, seed)
    an = Q.KernelStateAnsatz(n, reps, gamma, Q.entanglement_graph(n, d))
    circs = [an.circuit_for_data(x) for x in X]
    t0 = time.perf_counter()
    dev, info = ctx.build_mps(circs, max_bond=max_bond)
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    host = [simulate(c) for c in circs[: min(npts, 8)]]
    t_host = (time.perf_counter() - t0) / len(host)
    with ctx.upload(dev[: len(host)]) as xs, ctx.upload(host) as ys:
        z = ctx.overlaps(xs, ys)
    diag = np.abs(np.diag(z)) ** 2
    with ctx.upload(dev[: len(host)]) as xs:
        zz = ctx.overlaps(xs, xs)
    nrm = np.abs(np.diag(zz))
    print(f"{label}: device {info['kernel_ms']:.1f} ms kernel ({t_dev:.2f} s wall) for {npts} states; host {t_host:.3f} s/state; "
          f"|<dev|host>|^2 min {diag.min():.12f}  |<dev|dev>| in [{nrm.min():.12f}, {nrm.max():.12f}]; "
          f"max bond dev {max(m.max_bond() for m in dev)} host {max(m.max_bond() for m in host)}; fidelity dev {min(m.fidelity for m in dev):.3e}", flush=True)
    assert abs(diag - 1).max() < 1e-9 and abs(nrm - 1).max() < 1e-9
    return info["kernel_ms"], t_host


def main():
    rng = np.random.default_rng(1)
    ctx = engine.Context(0)
    check_jacobi(ctx, rng)
    compare(ctx, 8, 2, 1, 1.0, 6, 3, "8q r2 d1")
    compare(ctx, 12, 3, 2, 1.0, 6, 4, "12q r3 d2")
    compare(ctx, 20, 4, 2, 1.0, 8, 5, "20q r4 d2")
    if len(sys.argv) > 1:
        ns = int(sys.argv[1])
        ms, th = compare(ctx, 60, 6, 2, 1.0, ns, 5, f"cfg4-shaped, {ns} states", max_bond=int(sys.argv[2]) if len(sys.argv) > 2 else 256)
        print(f"cfg4-shaped: device {ms/1e3:.2f} s for {ns} states vs host {th*ns/16:.2f} s on 16 cores (extrapolated)")
    ctx.close() if hasattr(ctx, "close") else None


if __name__ == "__main__":
    main()
