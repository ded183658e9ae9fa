"""The oracle itself, pinned against the golden fixtures (independent dense computations)."""
import numpy as np
import pytest

from helpers import golden, golden_mps_sets
from oracle import restatement as R


def test_entanglement_graph_worked_examples():
    # SURVEY.md addendum (restated from main.py:21-45)
    assert R.entanglement_graph(8, 1) == [(0, 1), (2, 3), (4, 5), (6, 7), (1, 2), (3, 4), (5, 6)]
    assert R.entanglement_graph(8, 2) == [(0, 1), (2, 3), (4, 5), (6, 7), (1, 2), (3, 4), (5, 6), (0, 2), (1, 3), (4, 6), (5, 7), (2, 4), (3, 5)]


@pytest.mark.parametrize("n,r,d,gates,edges", [(8, 1, 1, 23, 7), (20, 2, 1, 98, 19), (40, 4, 2, 812, 77), (60, 6, 2, 1818, 117), (100, 10, 4, 16600, 390)])
def test_gate_counts_of_the_configs(n, r, d, gates, edges):
    e = R.entanglement_graph(n, d)
    assert len(e) == edges
    assert len(R.ansatz_gates(np.ones(n), r, 1.0, e)) == gates


def test_mps_inner_matches_dense_vdot_golden():
    xs, ys, z = golden_mps_sets()
    got = np.array([[R.mps_inner(x, y) for x in xs] for y in ys])
    assert np.abs(got - z).max() < 1e-13


@pytest.mark.parametrize("name,tol", [("cfg1_8q_r1_d1.npz", 1e-12), ("deep_10q_r3_d3.npz", 1e-8), ("cfg2_20q_r2_d1_subset.npz", 1e-11)])
def test_mps_path_reproduces_statevector_golden(name, tol):
    g = golden(name)
    n, reps, gamma, d = int(g["n"]), int(g["reps"]), float(g["gamma"]), int(g["d"])
    e = R.entanglement_graph(n, d)
    xs = [R.mps_simulate(n, R.ansatz_gates(x, reps, gamma, e)) for x in g["X_train"]]
    ys = [R.mps_simulate(n, R.ansatz_gates(x, reps, gamma, e)) for x in g["X_test"]]
    assert np.abs(R.gram_from_mps(xs) - g["K_train"]).max() < tol
    assert np.abs(R.gram_from_mps(xs, ys) - g["K_test"]).max() < tol


def test_untruncated_mps_is_exact():
    g = golden("deep_10q_r3_d3.npz")
    n, reps, gamma, d = int(g["n"]), int(g["reps"]), float(g["gamma"]), int(g["d"])
    e = R.entanglement_graph(n, d)
    xs = [R.mps_simulate(n, R.ansatz_gates(x, reps, gamma, e), cutoff=0.0) for x in g["X_train"]]
    assert np.abs(R.gram_from_mps(xs) - g["K_train"]).max() < 1e-13


def test_closed_form_without_entanglement():
    g = golden("d0_closed_12q_r3.npz")
    n, reps, gamma = int(g["n"]), int(g["reps"]), float(g["gamma"])
    xs = [R.mps_simulate(n, R.ansatz_gates(x, reps, gamma, [])) for x in g["X_train"]]
    assert max(t.shape[2] for m in xs for t in m) == 1
    assert np.abs(R.gram_from_mps(xs) - g["K_train"]).max() < 1e-13


def test_gram_invariants():
    g = golden("cfg1_8q_r1_d1.npz")
    K = g["K_train"]
    assert np.abs(np.diag(K) - 1).max() < 1e-13
    assert np.abs(K - K.T).max() < 1e-14
    assert K.min() >= 0 and K.max() <= 1 + 1e-13
    assert np.linalg.eigvalsh(K).min() > -1e-12  # Schur product of a Gram matrix with its conjugate
    X = g["X_train"][:3]
    assert np.abs(R.gram_statevector(X, None, 1, 0.0, R.entanglement_graph(8, 1)) - 1).max() < 1e-13  # gamma = 0
    assert np.abs(R.gram_statevector(X, None, 0, 1.0, R.entanglement_graph(8, 1)) - 1).max() < 1e-13  # no layers


def test_c_restatement_agrees_with_numpy(built):
    from oracle import c_oracle

    xs, ys, z = golden_mps_sets()
    pairs = [(i, j) for j in range(2) for i in range(2)]
    vals, zc, used = c_oracle.gram_pairs(xs, ys, pairs, threads=2)
    ref = np.array([z[j, i] for i, j in pairs])
    assert np.abs(zc - ref).max() < 1e-13
    assert np.abs(vals - np.abs(ref) ** 2).max() < 1e-13
    # symmetric form
    v2, z2, _ = c_oracle.gram_pairs(xs, None, [(0, 0), (0, 1), (1, 1)], threads=1)
    assert abs(v2[0] - abs(R.mps_inner(xs[0], xs[0])) ** 2) < 1e-13
    assert abs(z2[1] - R.mps_inner(xs[0], xs[1])) < 1e-13


def test_blas_leg_matches_hand_loop():
    """oracle/overlap_blas.c (every contraction on zgemm: the CPU baseline of bench.py) against oracle/overlap_ref.c and
    the numpy restatement on ragged MPS."""
    import qml_cutensornet_amd as Q
    from oracle import c_oracle

    rng = np.random.default_rng(12)
    n = 14
    xs = [Q.random_mps(n, [min(2 ** min(k, n - k), c) for k in range(n + 1)], rng) for c in (3, 17, 40)]
    ys = [Q.random_mps(n, [min(2 ** min(k, n - k), c) for k in range(n + 1)], rng) for c in (8, 33)]
    pairs = np.array([(i, j) for j in range(2) for i in range(3)], dtype=np.int32)
    v_ref, z_ref, _ = c_oracle.gram_pairs([m.tensors for m in xs], [m.tensors for m in ys], pairs)
    v_blas, z_blas, _ = c_oracle.gram_pairs_blas([m.tensors for m in xs], [m.tensors for m in ys], pairs, threads=2)
    z_np = np.array([R.mps_inner(xs[i].tensors, ys[j].tensors) for i, j in pairs])
    assert np.abs(z_blas - z_ref).max() < 1e-13 and np.abs(z_blas - z_np).max() < 1e-13
    assert np.abs(v_blas - np.abs(z_np) ** 2).max() < 1e-13
