"""CPU tier: the numpy model of the device builder's factorisation (oracle/jacobi_model.py) against LAPACK -- the algorithm
of csrc/qk_build.hip checked without a GPU; the kernel itself is checked against the same quantities in test_gpu_builder.py."""
import numpy as np
import pytest

from oracle import jacobi_model as J


@pytest.mark.parametrize("p,q,rank", [(1, 1, None), (4, 2, None), (12, 7, None), (20, 20, None), (30, 12, 5), (9, 9, 1)])
def test_model_matches_lapack_svd(p, q, rank):
    rng = np.random.default_rng(10 * p + q)
    a = rng.standard_normal((p, q)) + 1j * rng.standard_normal((p, q))
    if rank:
        a = (rng.standard_normal((p, rank)) + 1j * rng.standard_normal((p, rank))) @ (rng.standard_normal((rank, q)) + 1j * rng.standard_normal((rank, q)))
    w, v, sig, sweeps = J.jacobi(a)
    s_ref = np.linalg.svd(a, compute_uv=False)
    assert sweeps <= 12
    assert np.abs(np.sort(sig)[::-1] - s_ref).max() < 1e-13 * s_ref[0]
    assert np.abs(a @ v - w).max() < 1e-13 * np.abs(a).max()
    assert np.abs(v.conj().T @ v - np.eye(q)).max() < 1e-13
    u, s, vh = J.svd_model(a)
    assert np.abs((u * s) @ vh - a).max() < 1e-13 * np.abs(a).max()
    u, s, vh = J.svd_model(a.T.copy())  # the wide case goes through the transpose
    assert np.abs((u * s) @ vh - a.T).max() < 1e-13 * np.abs(a).max()


def test_model_settles_on_a_matrix_with_noise_columns():
    """A rank-1 matrix whose second column is a denormal multiple of the first (what a product state's theta looks like):
    a purely relative rotation test never settles on it; the floor in the test does."""
    col = np.array([-0.64297 - 0.29426j, -0.64297 - 0.29426j])
    a = np.stack([col, col * 2e-160], axis=1)
    w, v, sig, sweeps = J.jacobi(a)
    assert sweeps <= 2 and abs(sig.max() - np.linalg.norm(col)) < 1e-15


@pytest.mark.parametrize("n,reps,d,gamma", [(6, 2, 1, 1.0), (8, 2, 2, 1.0), (9, 1, 3, 0.5)])
def test_builder_with_the_model_reproduces_the_lapack_builder(n, reps, d, gamma, monkeypatch):
    """mps._simulate with the model in place of gesdd / QR builds the same states (the device builder's algorithm end to end)."""
    import qml_cutensornet_amd as Q
    from oracle import restatement as R
    from qml_cutensornet_amd import mps as M

    X = R.synthetic_features(3, n, 5)
    ans = Q.KernelStateAnsatz(n, reps, gamma, Q.entanglement_graph(n, d))
    for x in X:
        c = ans.circuit_for_data(x)
        ref = M._simulate(c, 1 - 1e-16, 1e-16)
        with monkeypatch.context() as mp:
            mp.setattr(M, "_svd", J.svd_model)
            mp.setattr(M, "_qr", J.qr_model)
            mod = M._simulate(c, 1 - 1e-16, 1e-16)
        assert abs(abs(R.mps_inner(mod.tensors, ref.tensors)) ** 2 - 1) < 1e-10
        assert max(mod.bond_dims()) <= max(ref.bond_dims())
        assert abs(mod.fidelity - ref.fidelity) < 1e-10


@pytest.mark.parametrize("p,q,decades", [(40, 32, 20), (78, 66, 22), (96, 96, 12)])
def test_preconditioned_block_model_on_graded_matrices(p, q, decades):
    """The device builder's factorisation for large matrices, restated (jacobi_precond: sorted columns, Gram-Schmidt R, block Jacobi
    of R^H without accumulating rotations, W = A V), on GRADED matrices -- what a gate's theta looks like: about half the sweeps of
    the plain method, the kept part of the decomposition as LAPACK gives it, V orthonormal."""
    rng = np.random.default_rng(p * q)
    u, _ = np.linalg.qr(rng.standard_normal((p, q)) + 1j * rng.standard_normal((p, q)))
    v0, _ = np.linalg.qr(rng.standard_normal((q, q)) + 1j * rng.standard_normal((q, q)))
    a = (u * 10.0 ** (-decades * np.arange(q) / (q - 1))) @ v0.conj().T
    w, v, sig, sweeps = J.jacobi_precond(a)
    _, _, _, plain = J.jacobi(a)
    assert sweeps <= 12 and sweeps <= plain
    u_ref, s_ref, vh_ref = np.linalg.svd(a, full_matrices=False)
    keep = int((np.cumsum((s_ref ** 2)[::-1])[::-1] > 1e-16 * (s_ref ** 2).sum()).sum())
    o = np.argsort(-sig, kind="stable")
    assert np.abs(sig[o][:keep] - s_ref[:keep]).max() < 1e-14 * s_ref[0]  # (LAPACK's own values are good to eps x the largest one)
    wk, vk = w[:, o[:keep]], v[:, o[:keep]]
    assert np.abs(wk @ vk.conj().T - (u_ref[:, :keep] * s_ref[:keep]) @ vh_ref[:keep]).max() < 1e-12
    assert np.abs(vk.conj().T @ vk - np.eye(keep)).max() < 1e-11


def test_block_model_matches_the_scalar_model():
    """Block Jacobi (8-column blocks, all pairs of a panel in a sweep's first round, cross pairs afterwards) is a cyclic ordering of
    the scalar method: same singular values, A V = W, V unitary."""
    rng = np.random.default_rng(3)
    a = rng.standard_normal((50, 37)) + 1j * rng.standard_normal((50, 37))
    w, v, sig, sweeps = J.jacobi_block(a)
    s_ref = np.linalg.svd(a, compute_uv=False)
    assert sweeps <= 12
    assert np.abs(np.sort(sig)[::-1] - s_ref).max() < 1e-13 * s_ref[0]
    assert np.abs(a @ v - w).max() < 1e-13 * np.abs(a).max()
    assert np.abs(v.conj().T @ v - np.eye(37)).max() < 1e-13


def test_builder_with_the_preconditioned_model_reproduces_the_lapack_builder(monkeypatch):
    import qml_cutensornet_amd as Q
    from oracle import restatement as R
    from qml_cutensornet_amd import mps as M

    n, reps, d = 10, 3, 3
    X = R.synthetic_features(2, n, 5)
    ans = Q.KernelStateAnsatz(n, reps, 1.0, Q.entanglement_graph(n, d))
    for x in X:
        c = ans.circuit_for_data(x)
        ref = M._simulate(c, 1 - 1e-16, 1e-16)
        with monkeypatch.context() as mp:
            mp.setattr(M, "_svd", lambda a, **k: J.svd_precond_model(a) if min(a.shape) >= 16 else J.svd_model(a))
            mp.setattr(M, "_qr", lambda a, **k: J.qr_precond_model(a) if min(a.shape) >= 16 else J.qr_model(a))
            mod = M._simulate(c, 1 - 1e-16, 1e-16)
        assert abs(abs(R.mps_inner(mod.tensors, ref.tensors)) ** 2 - 1) < 1e-10
        assert np.array_equal(mod.bond_dims(), ref.bond_dims())
