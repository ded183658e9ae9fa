"""GPU parity: the HIP path (through the C ABI) against the oracle on identical MPS tensors.

Tolerance: the north star asks overlaps to match the CPU backend to 1e-10 (fp64); both sides
see the same tensors here, so the bound asserted is 1e-11 on |z| <= 1 quantities.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-11


def _ragged_profile(rng, n, chi_max):
    """Random admissible bond profile (bond k bounded by 2^k, 2^(n-k) and its neighbours * 2)."""
    chi = [1]
    for k in range(1, n):
        cap = min(2 ** min(k, n - k, 20), chi_max, 2 * chi[-1])
        chi.append(int(rng.integers(1, cap + 1)))
    chi.append(1)
    for k in range(n - 1, 0, -1):  # right-to-left admissibility
        chi[k] = min(chi[k], 2 * chi[k + 1])
    return chi


def test_mfma_fragment_maps(gpu_ctx):
    gpu_ctx.selftest()


@pytest.mark.parametrize("n,chi_max,nx,ny,seed", [(6, 4, 3, 2, 1), (12, 16, 5, 4, 2), (14, 40, 4, 5, 3), (16, 100, 3, 3, 4), (18, 150, 2, 3, 5)])
def test_overlaps_random_ragged(gpu_ctx, n, chi_max, nx, ny, seed):
    import qml_cutensornet_amd as Q
    from oracle import restatement as R

    rng = np.random.default_rng(seed)
    xs = [Q.random_mps(n, _ragged_profile(rng, n, chi_max), rng) for _ in range(nx)]
    ys = [Q.random_mps(n, _ragged_profile(rng, n, chi_max), rng) for _ in range(ny)]
    with gpu_ctx.upload(xs) as dx, gpu_ctx.upload(ys) as dy:
        z = gpu_ctx.overlaps(dx, dy)
        K = gpu_ctx.gram(dx, dy)
    z_ref = np.array([[R.mps_inner(x.tensors, y.tensors) for x in xs] for y in ys])
    assert np.abs(z - z_ref).max() < TOL
    assert np.abs(K - np.abs(z_ref) ** 2).max() < TOL


def test_self_overlap_is_one(gpu_ctx):
    import qml_cutensornet_amd as Q

    rng = np.random.default_rng(11)
    xs = [Q.random_mps(20, _ragged_profile(rng, 20, 70), rng) for _ in range(4)]
    with gpu_ctx.upload(xs) as dx:
        K = gpu_ctx.gram(dx)
    assert np.abs(np.diag(K) - 1).max() < 1e-12
    assert np.abs(K - K.T).max() == 0.0
    assert K.min() >= 0 and K.max() <= 1 + 1e-12


@pytest.mark.parametrize("n,reps,gamma,d,npts", [(8, 1, 1.0, 1, 16), (10, 3, 0.8, 3, 6), (20, 2, 1.0, 1, 12)])
def test_gram_of_ansatz_states(gpu_ctx, n, reps, gamma, d, npts):
    import qml_cutensornet_amd as Q
    from oracle import restatement as R

    X = R.synthetic_features(npts, n, 5)
    ans = Q.KernelStateAnsatz(n, reps, gamma, Q.entanglement_graph(n, d))
    states = [Q.simulate(ans.circuit_for_data(x), 1 - 1e-16) for x in X]
    with gpu_ctx.upload(states) as dx:
        K = gpu_ctx.gram(dx)
    K_ref = R.gram_from_mps([m.tensors for m in states])
    assert np.abs(K - K_ref).max() < TOL
    if n <= 10:  # end to end against the exact state vector (truncation 1e-16 => ~1e-9)
        K_sv = R.gram_statevector(X, None, reps, gamma, Q.entanglement_graph(n, d))
        assert np.abs(K - K_sv).max() < 1e-8
