#!/usr/bin/env python3
"""Regenerates the golden fixtures in this directory:  python tests/golden/make_golden.py

The reference ships no golden vectors for the Gram path, and it cannot
be imported here (pytket / cuquantum / Julia are absent), so the fixtures come from this
repo's own INDEPENDENT oracles, none of which shares code with the MPS sweep being tested:

  * exact dense state vectors of the ansatz circuit          (oracle.restatement.statevector)
  * the closed form for an empty entanglement map            (gram_product_closed_form)
  * MPS -> dense vector by brute-force contraction + np.vdot  (dense_from_mps below)

A fixture is data only: inputs (features / MPS tensors, circuit parameters) and expected outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import restatement as R  # noqa: E402


def dense_from_mps(tensors):
    """Full 2^n vector of an MPS given as [chi_l, 2, chi_r] tensors (no sweep code involved)."""
    v = np.ones((1, 1), dtype=complex)
    for t in tensors:
        v = np.tensordot(v, t, axes=(1, 0)).reshape(-1, t.shape[2])
    return v[:, 0]


def save(name, **kw):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **kw)
    print(f"{name}: {os.path.getsize(path)} bytes")


def ansatz_case(name, n, reps, gamma, d, n_train, n_test, seed):
    X = R.synthetic_features(n_train + n_test, n, seed)
    Xtr, Xte = X[:n_train], X[n_train:]
    edges = R.entanglement_graph(n, d)
    K_train = R.gram_statevector(Xtr, None, reps, gamma, edges)
    K_test = R.gram_statevector(Xtr, Xte, reps, gamma, edges) if n_test else np.zeros((0, n_train))
    save(name, X_train=Xtr, X_test=Xte, n=n, reps=reps, gamma=gamma, d=d, K_train=K_train, K_test=K_test)


def main():
    # cfg1 of BASELINE.json: 8 qubits, 1 layer, d=1, 10+10 points -> 16 train + 4 test (80/20 split, main.py:62)
    ansatz_case("cfg1_8q_r1_d1.npz", 8, 1, 1.0, 1, 16, 4, 5)
    # deeper / longer-range circuit where truncation at 1e-16 actually bites
    ansatz_case("deep_10q_r3_d3.npz", 10, 3, 0.8, 3, 6, 2, 8)
    # a slice of cfg2 (20 qubits, 2 layers, d=1): 16 MiB per dense state, 6 + 2 points
    ansatz_case("cfg2_20q_r2_d1_subset.npz", 20, 2, 1.0, 1, 6, 2, 5)
    # closed form, no entanglement
    X = R.synthetic_features(5, 12, 20)
    save("d0_closed_12q_r3.npz", X_train=X, n=12, reps=3, gamma=0.7, K_train=R.gram_product_closed_form(X, None, 3, 0.7))
    # raw MPS pairs: inputs of the hot path itself
    rng = np.random.default_rng(77)
    n = 9
    sets = {}
    for tag, prof in (("x0", [1, 2, 3, 5, 7, 6, 4, 3, 2, 1]), ("x1", [1, 1, 2, 4, 8, 16, 8, 4, 2, 1]), ("y0", [1, 2, 4, 6, 5, 9, 5, 3, 1, 1]), ("y1", [1, 2, 2, 2, 3, 3, 2, 2, 2, 1])):
        ts = [rng.standard_normal((prof[k], 2, prof[k + 1])) + 1j * rng.standard_normal((prof[k], 2, prof[k + 1])) for k in range(n)]
        nrm = np.linalg.norm(dense_from_mps(ts))
        ts[0] = ts[0] / nrm
        sets[tag] = ts
    dense = {k: dense_from_mps(v) for k, v in sets.items()}
    z = np.array([[np.vdot(dense[x], dense[y]) for x in ("x0", "x1")] for y in ("y0", "y1")])  # z[j, i] = <x_i|y_j>
    flat = {f"{tag}_{k}": t for tag, ts in sets.items() for k, t in enumerate(ts)}
    save("mps_pairs_9q.npz", n=n, z=z, **flat)


if __name__ == "__main__":
    main()
