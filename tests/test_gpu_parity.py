"""GPU parity: the HIP path (through the C ABI) against the oracle on identical MPS tensors.

Tolerance: the north star asks overlaps to match the CPU backend to 1e-10 (fp64); both sides
see the same tensors here, so the bound asserted is 1e-11 on |z| <= 1 quantities.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-11
# Two launches on the same inputs may differ in the last bits: the site-fused sweep sums the tiles of a column of X' with
# LDS atomics (ds_add_f64), i.e. in arrival order.  Entries are <= 1, so this is an absolute bound of a few ulp.
LAUNCH_EPS = 1e-14


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


# ------------------------------------------------------------------ golden fixtures, drop-in API
def test_golden_mps_pairs(gpu_ctx):
    import qml_cutensornet_amd as Q
    from helpers import golden_mps_sets

    xs, ys, z = golden_mps_sets()
    with gpu_ctx.upload([Q.MPS(t) for t in xs]) as dx, gpu_ctx.upload([Q.MPS(t) for t in ys]) as dy:
        got = gpu_ctx.overlaps(dx, dy)
    assert np.abs(got - z).max() < 1e-12


@pytest.mark.parametrize("name,tol", [("cfg1_8q_r1_d1.npz", 1e-10), ("deep_10q_r3_d3.npz", 1e-8), ("cfg2_20q_r2_d1_subset.npz", 1e-10), ("d0_closed_12q_r3.npz", 1e-10)])
def test_build_kernel_matrix_against_golden(built, name, tol, tmp_path):
    """The reference's module surface end to end (train Gram, test Gram, JSON keys)."""
    import json

    import qml_cutensornet_amd as Q
    from helpers import golden
    from qml_cutensornet_amd.dist import SingleComm
    from qml_cutensornet_amd.gpu_backend.kernel_state_ansatz import KernelStateAnsatz, build_kernel_matrix

    g = golden(name)
    n, reps, gamma = int(g["n"]), int(g["reps"]), float(g["gamma"])
    emap = Q.entanglement_graph(n, int(g["d"])) if "d" in g else []
    ans = KernelStateAnsatz(num_qubits=n, reps=reps, gamma=gamma, entanglement_map=emap, hadamard_init=True)
    info = str(tmp_path / "train_info")
    K = build_kernel_matrix(SingleComm(), ans, X=g["X_train"], info_file=info, truncation_error=1e-16)
    assert K.shape == g["K_train"].shape and K.dtype == np.float64
    assert np.abs(K - g["K_train"]).max() < tol
    prof = json.load(open(info + ".json"))
    for key in ["n_procs", "lenX", "lenY", "r0_circ_gen", "r0_circ_sim", "avg_circ_sim", "median_circ_sim", "q1_circ_sim", "q3_circ_sim",
                "gpu_mps_mem", "avg_mps_mem", "avg_fidelity", "ave max chi x", "ave max chi y", "r_nonRR_recv", "r0_RR_recv",
                "kernel_mat_time", "total_time", "r0_product", "avg_product", "median_product", "q1_product", "q3_product"]:
        assert key in prof, key
    assert prof["lenX"][0] == len(g["X_train"]) and prof["lenY"][0] is None
    if "X_test" in g and len(g["X_test"]):
        Kt = build_kernel_matrix(SingleComm(), ans, X=g["X_train"], Y=g["X_test"], truncation_error=1e-16)
        assert Kt.shape == (len(g["X_test"]), len(g["X_train"]))  # rows = Y, cols = X
        assert np.abs(Kt - g["K_test"]).max() < tol


def test_mps_vdot_single_pair(built):
    import qml_cutensornet_amd as Q
    from helpers import golden_mps_sets

    xs, ys, z = golden_mps_sets()
    assert abs(Q.MPS(xs[1]).vdot(Q.MPS(ys[0])) - z[0, 1]) < 1e-12  # conjugates self, like the reference's x_mps.vdot(y_mps)


# ------------------------------------------------------------------ sharding on one GPU
@pytest.mark.parametrize("world", [2, 3])
def test_rank_shares_reassemble(gpu_ctx, world):
    """Every rank's share computed by the sweep kernel, joined the way the all-gather joins them."""
    import qml_cutensornet_amd as Q
    from oracle import restatement as R
    from qml_cutensornet_amd import engine
    from qml_cutensornet_amd.dist import assemble_gram

    rng = np.random.default_rng(9)
    xs = [Q.random_mps(10, _ragged_profile(rng, 10, 24), rng) for _ in range(11)]
    ys = [Q.random_mps(10, _ragged_profile(rng, 10, 24), rng) for _ in range(5)]
    with gpu_ctx.upload(xs) as dx, gpu_ctx.upload(ys) as dy:
        for sym in (True, False):
            pairs, vals = [], []
            for r in range(world):
                plan = engine.Plan(dx.dims, None if sym else dy.dims, world, r)
                pairs.append(plan.pairs())
                vals.append(gpu_ctx.gram_values_host(dx, None if sym else dy, plan))
                plan.close()
            K = assemble_gram(len(xs) if sym else len(ys), len(xs), pairs, vals, sym)
            ref = R.gram_from_mps([m.tensors for m in xs], None if sym else [m.tensors for m in ys])
            assert np.abs(K - ref).max() < TOL
            assert np.abs(K - gpu_ctx.gram(dx, None if sym else dy)).max() < 1e-14  # the 1-rank path (see LAUNCH_EPS below)


def test_gram_job_device_path(gpu_ctx):
    """The bench/driver path: torch buffers, scatter kernel, mirrored fill."""
    import qml_cutensornet_amd as Q
    from oracle import restatement as R
    from qml_cutensornet_amd.gram import GramJob

    rng = np.random.default_rng(21)
    xs = [Q.random_mps(12, _ragged_profile(rng, 12, 40), rng) for _ in range(9)]
    try:
        with gpu_ctx.upload(xs) as dx:
            job = GramJob(gpu_ctx, dx)
            K = job.run()  # very first launch: plan upload + scratch allocation happen here
            K2 = job.run()  # re-enqueue on the same buffers
            job.close()
    finally:
        gpu_ctx.set_stream(None)  # back to the context's private stream for the other tests
    assert np.abs(K - K2).max() < LAUNCH_EPS
    assert np.abs(K - R.gram_from_mps([m.tensors for m in xs])).max() < TOL


# ------------------------------------------------------------------ size-independent properties at benchmark scale
def test_cfg4_scale_properties(gpu_ctx):
    """60-site states with bonds up to ~130 (cfg4 regime): unit diagonal, exact symmetry of the mirrored fill,
    0 <= K <= 1, positive semidefinite, and <x|y> = conj(<y|x>) between the two sweep orders."""
    import qml_cutensornet_amd as Q

    rng = np.random.default_rng(4)
    xs = [Q.random_mps(60, _ragged_profile(rng, 60, 130), rng) for _ in range(24)]
    # make overlaps non-trivial: y_j is x_j with every site tensor slightly perturbed (same bond profile)
    with gpu_ctx.upload(xs) as dx:
        K = gpu_ctx.gram(dx)
        z = gpu_ctx.overlaps(dx)
        st = gpu_ctx.stats()
    assert np.abs(np.diag(K) - 1).max() < 1e-11
    assert np.array_equal(K, K.T)
    assert K.min() >= 0 and K.max() <= 1 + 1e-11
    assert np.linalg.eigvalsh(K).min() > -1e-10
    assert np.abs(z - z.conj().T).max() < 1e-12  # independent sweeps of (i,j) and (j,i)
    assert np.abs(np.abs(z) ** 2 - K).max() < 1e-12
    assert st["pairs"] == 24 * 24 and st["padded_flops"] >= st["flops"] > 0 and st["kernel_ms"] > 0


def test_ansatz_states_cfg3_slice(gpu_ctx):
    """Real circuits of cfg3 (40 qubits, 4 layers, d=2, gamma=1): HIP vs the C and numpy oracles on the same tensors."""
    import qml_cutensornet_amd as Q
    from oracle import c_oracle, restatement as R
    from qml_cutensornet_amd.data import synthetic_features

    n = 40
    X = synthetic_features(10, n, 5)
    ans = Q.KernelStateAnsatz(n, 4, 1.0, Q.entanglement_graph(n, 2))
    states = [Q.simulate(ans.circuit_for_data(x), 1 - 1e-16) for x in X]
    with gpu_ctx.upload(states) as dx, gpu_ctx.upload(states[:4]) as dy:
        K = gpu_ctx.gram(dx)
        Kt = gpu_ctx.gram(dx, dy)
    pairs = [(i, j) for j in range(10) for i in range(10)]
    vals, _, _ = c_oracle.gram_pairs([m.tensors for m in states], None, pairs, threads=4)
    assert np.abs(K - vals.reshape(10, 10)).max() < TOL
    assert np.abs(Kt - K[:4, :]).max() < TOL
    assert np.abs(K[:3, :3] - R.gram_from_mps([m.tensors for m in states[:3]])).max() < TOL


def test_driver_end_to_end(built, tmp_path, monkeypatch):
    """N3: the main_no_test-shaped driver writes kernels/<stem>.npy and <stem>.json with the reference's names/keys."""
    import json

    from helpers import golden
    from qml_cutensornet_amd import driver
    from qml_cutensornet_amd.data import synthetic_features

    monkeypatch.chdir(tmp_path)
    K = driver.main(["GPU", "8", "1", "1.0", "1", "10", "10", "5", "nodata.csv"])
    stem = "train_Nf8_r1_g1.0_p0.0_nn1_mslinear_Ntr10_s5_nodata"
    saved = np.load(tmp_path / "kernels" / f"{stem}.npy")
    prof = json.load(open(tmp_path / f"{stem}.json"))
    assert saved.shape == (16, 16) and np.array_equal(saved, K)
    assert prof["lenX"][0] == 16 and prof["n_procs"][0] == 1 and "kernel_mat_time" in prof
    # same circuits as cfg1's golden fixture generator (8 q, 1 layer, d=1, gamma=1): compare on identical features
    g = golden("cfg1_8q_r1_d1.npz")
    from oracle import restatement as R

    X = synthetic_features(16, 8, 5)
    assert np.abs(K - R.gram_statevector(X, None, 1, 1.0, R.entanglement_graph(8, 1))).max() < 1e-10


# ------------------------------------------------------------------ edge cases
def test_edge_shapes(gpu_ctx):
    """One-site chains, a 1x1 Gram, bonds straddling the 16-wide MFMA tile (15, 16, 17, 33), product states."""
    import qml_cutensornet_amd as Q
    from oracle import restatement as R

    rng = np.random.default_rng(42)
    # one site: the overlap is a plain 2-vector dot product
    one = [Q.random_mps(1, [1, 1], rng) for _ in range(3)]
    with gpu_ctx.upload(one) as d:
        z = gpu_ctx.overlaps(d)
    ref = np.array([[np.vdot(x.tensors[0].ravel(), y.tensors[0].ravel()) for x in one] for y in one])
    assert np.abs(z - ref).max() < 1e-14
    # single state
    m = Q.random_mps(9, [1, 2, 4, 8, 15, 8, 4, 2, 1, 1], rng)
    with gpu_ctx.upload([m]) as d:
        K = gpu_ctx.gram(d)
    assert K.shape == (1, 1) and abs(K[0, 0] - 1) < 1e-13
    # tile-boundary bonds on both operands
    profs = [[1, 2, 4, 8, 16, 17, 33, 17, 9, 5, 3, 2, 1], [1, 2, 4, 8, 15, 16, 32, 16, 8, 4, 2, 1, 1], [1, 2, 3, 5, 9, 17, 31, 16, 15, 8, 4, 2, 1], [1] * 13]
    xs = [Q.random_mps(12, p, rng) for p in profs]
    with gpu_ctx.upload(xs) as d:
        z = gpu_ctx.overlaps(d)
    ref = np.array([[R.mps_inner(x.tensors, y.tensors) for x in xs] for y in xs])
    assert np.abs(z - ref).max() < TOL


def test_host_layout_lrp(gpu_ctx):
    """Site tensors handed over as [left][right][physical] (the order pytket-cutensornet is recalled to use)."""
    import qml_cutensornet_amd as Q
    from qml_cutensornet_amd import engine

    rng = np.random.default_rng(8)
    xs = [Q.random_mps(8, [1, 2, 4, 7, 9, 6, 3, 2, 1], rng) for _ in range(3)]

    class Swapped:
        def __init__(self, m):
            self._m = m
            self.tensors = [np.ascontiguousarray(t.transpose(0, 2, 1)) for t in m.tensors]

        def bond_dims(self):
            return self._m.bond_dims()

        def __len__(self):
            return len(self._m)

    with gpu_ctx.upload(xs) as a, gpu_ctx.upload([Swapped(m) for m in xs], layout=engine.QK_LAYOUT_LRP) as b:
        assert np.abs(gpu_ctx.gram(a) - gpu_ctx.gram(b)).max() < LAUNCH_EPS


def test_bad_inputs_fail_loudly(gpu_ctx):
    import qml_cutensornet_amd as Q
    from qml_cutensornet_amd import engine

    rng = np.random.default_rng(1)
    a = Q.random_mps(4, [1, 2, 2, 2, 1], rng)
    b = Q.random_mps(5, [1, 2, 2, 2, 2, 1], rng)
    with pytest.raises(engine.QkError):
        gpu_ctx.upload([])
    with pytest.raises(engine.QkError):
        gpu_ctx.upload([a, b])  # different site counts in one set
    with gpu_ctx.upload([a]) as da, gpu_ctx.upload([b]) as db:
        with pytest.raises(engine.QkError, match="site counts"):
            gpu_ctx.gram(da, db)
    with pytest.raises(RuntimeError):
        a.vdot(b)


# ------------------------------------------------------------------ N4: complex64 sweep (fp32 MFMA)
F32_TOL = 2e-5  # |z32 - z64| on |z| <= 1 quantities; measured 1e-7 ... 3e-6 (profiles/r01/fp32_tolerance.txt)


@pytest.mark.parametrize("n,chi_max,nx,ny,seed", [(6, 4, 3, 2, 1), (14, 40, 4, 5, 3), (18, 150, 2, 3, 5), (40, 70, 3, 3, 7)])
def test_f32_sweep_close_to_f64(gpu_ctx, n, chi_max, nx, ny, seed):
    import qml_cutensornet_amd as Q
    from oracle import restatement as R

    rng = np.random.default_rng(seed)
    xs = [Q.random_mps(n, _ragged_profile(rng, n, chi_max), rng) for _ in range(nx)]
    ys = [Q.random_mps(n, _ragged_profile(rng, n, chi_max), rng) for _ in range(ny)]
    z_ref = np.array([[R.mps_inner(x.tensors, y.tensors) for x in xs] for y in ys])
    with gpu_ctx.upload(xs) as dx, gpu_ctx.upload(ys) as dy, dx.to_f32() as fx, dy.to_f32() as fy:
        assert dx.precision == 64 and fx.precision == 32
        assert fx.info()["device_bytes"] * 2 == dx.info()["device_bytes"]
        z32 = gpu_ctx.overlaps(fx, fy)
        K32 = gpu_ctx.gram(fx, fy)
        with pytest.raises(Exception):  # mixed precision is refused, loudly
            gpu_ctx.overlaps(fx, dy)
    assert np.abs(z32 - z_ref).max() < F32_TOL
    assert np.abs(K32 - np.abs(z_ref) ** 2).max() < F32_TOL
    assert np.abs(z32 - z_ref).max() > 0  # it really is a different arithmetic


def test_f32_gram_of_ansatz_states(gpu_ctx):
    import qml_cutensornet_amd as Q
    from oracle import restatement as R

    X = R.synthetic_features(10, 16, 5)
    ans = Q.KernelStateAnsatz(16, 3, 1.0, Q.entanglement_graph(16, 2))
    states = [Q.simulate(ans.circuit_for_data(x), 1 - 1e-16) for x in X]
    with gpu_ctx.upload(states) as dx, dx.to_f32() as fx:
        K64 = gpu_ctx.gram(dx)
        K32 = gpu_ctx.gram(fx)
    assert np.abs(K32 - K64).max() < F32_TOL
    assert np.abs(np.diag(K32) - 1).max() < F32_TOL
    assert np.abs(K32 - K32.T).max() == 0.0


# ------------------------------------------------------------------ 2x2 pair blocks (QK_PLAN_QUADS): lab library only
def test_quad_plans_are_refused_by_the_product_library(gpu_ctx):
    """QK_PLAN_QUADS plans are swept by an experimental kernel that only libqklab.so (tools/) contains: the shipped
    library must say so instead of running anything."""
    import qml_cutensornet_amd as Q
    from qml_cutensornet_amd import engine

    rng = np.random.default_rng(3)
    xs = [Q.random_mps(10, _ragged_profile(rng, 10, 40), rng) for _ in range(4)]
    with gpu_ctx.upload(xs) as dx:
        plan = engine.Plan(dx.dims, None, quads=True)
        with pytest.raises(engine.QkError, match="libqklab"):
            gpu_ctx.gram_values_host(dx, None, plan)
        plan.close()


# ------------------------------------------------------------------ small-bond sweep (all bonds <= 32: X, T resident in LDS)
@pytest.mark.parametrize("n,chi_max,nx,ny,seed", [(5, 2, 3, 3, 1), (24, 16, 4, 3, 2), (60, 9, 6, 5, 6), (40, 32, 5, 4, 3), (100, 27, 3, 3, 4)])
def test_small_bond_kernel(gpu_ctx, n, chi_max, nx, ny, seed, monkeypatch):
    import qml_cutensornet_amd as Q
    from qml_cutensornet_amd import engine
    from oracle import restatement as R

    rng = np.random.default_rng(seed)
    xs = [Q.random_mps(n, _ragged_profile(rng, n, chi_max), rng) for _ in range(nx)]
    ys = [Q.random_mps(n, _ragged_profile(rng, n, chi_max), rng) for _ in range(ny)]
    z_ref = np.array([[R.mps_inner(x.tensors, y.tensors) for x in xs] for y in ys])
    with gpu_ctx.upload(xs) as dx, gpu_ctx.upload(ys) as dy:
        assert max(dx.info()["max_padded_bond"], dy.info()["max_padded_bond"]) <= 32  # takes a wave sweep (fp64) / the small-bond sweep (complex64)
        z = gpu_ctx.overlaps(dx, dy)
        K = gpu_ctx.gram(dx)
        with dx.to_f32() as fx, dy.to_f32() as fy:
            z32 = gpu_ctx.overlaps(fx, fy)
    assert np.abs(z - z_ref).max() < TOL
    assert np.abs(np.diag(K) - 1).max() < 1e-12 and np.abs(K - K.T).max() == 0.0
    assert np.abs(z32 - z_ref).max() < F32_TOL
    # every other kernel that can take these sets agrees to rounding on the same inputs: the 2 x 2-tile wave sweep (where the
    # 16-bond wave sweep was the one selected above), the LDS-resident small-bond sweep, the site-fused sweep, the ring sweep
    seen = {gpu_ctx.stats()["kernel_name"]}
    for env in ({"QK_WAVE": "0"}, {"QK_WAVE": "0", "QK_WAVE2": "0"}, {"QK_WAVE": "0", "QK_WAVE2": "0", "QK_SMALL": "0", "QK_FUSED": "2"},
                {"QK_WAVE": "0", "QK_WAVE2": "0", "QK_SMALL": "0", "QK_FUSED": "0"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        with engine.context(0) as ctx1, ctx1.upload(xs) as dx1, ctx1.upload(ys) as dy1:
            z_other = ctx1.overlaps(dx1, dy1)
            seen.add(ctx1.stats()["kernel_name"])
        assert np.abs(z - z_other).max() < 1e-13, env
    assert {"qk_sweep_wave2_kernel<3, double>", "qk_sweep_small_kernel<double>", "qk_sweep_ring_kernel<double>"} <= seen
    assert any("fused" in name_ for name_ in seen) or max(dx.dims.max(), dy.dims.max()) <= 16


# ------------------------------------------------------------------ randomised sweep over shapes (wave, small-bond and site-fused kernels)
def test_randomised_shapes_against_oracle(gpu_ctx):
    """40 seeded random (sites, bond cap, set sizes): bond caps 2...70 exercise the register (<= 16), LDS-resident (<= 32)
    and site-fused kernels, ragged profiles exercise the K-trim, ragged tile counts and the ping-pong / in-place X buffers."""
    import qml_cutensornet_amd as Q
    from oracle import restatement as R

    rng = np.random.default_rng(2024)
    worst = 0.0
    for case in range(40):
        n = int(rng.integers(2, 26))
        chi_max = int(rng.choice([2, 3, 5, 8, 13, 16, 17, 24, 31, 32, 33, 40, 48, 64, 70]))
        nx, ny = int(rng.integers(1, 5)), int(rng.integers(1, 5))
        xs = [Q.random_mps(n, _ragged_profile(rng, n, chi_max), rng) for _ in range(nx)]
        ys = [Q.random_mps(n, _ragged_profile(rng, n, chi_max), rng) for _ in range(ny)]
        z_ref = np.array([[R.mps_inner(x.tensors, y.tensors) for x in xs] for y in ys])
        with gpu_ctx.upload(xs) as dx, gpu_ctx.upload(ys) as dy:
            z = gpu_ctx.overlaps(dx, dy)
        err = float(np.abs(z - z_ref).max())
        assert err < TOL, (case, n, chi_max, nx, ny, err)
        worst = max(worst, err)
    assert worst < TOL


def test_c_abi_example_runs(gpu_ctx, tmp_path):
    """The plain-C program of examples/ (no Python between it and the library) reproduces a closed-form Gram."""
    import os
    import shutil
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "qk_example")
    libdir = os.path.join(root, "qml-cutensornet_amd")
    subprocess.run([shutil.which("gcc") or "gcc", "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "c_abi_example.c"),
                    "-o", exe, "-L", libdir, "-lqkgram", f"-Wl,-rpath,{libdir}", "-lm"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "max |K - closed form|" in r.stdout
    assert "sharded over" in r.stdout  # the multi-GPU entry points (qk_comm_*), here a communicator of one rank


def test_c_abi_communicator_sharded_gram(gpu_ctx):
    """The multi-GPU part of the C ABI on a one-device communicator: shares -> qk_mps_set_allgather (ncclAllGather of the
    packed images) -> qk_gram_sharded (sweep, ncclAllGather of the values, scatter) against the single-context Gram and
    the oracle; rectangular Gram; plans are reused when the same sets come back; bad input fails loudly."""
    import qml_cutensornet_amd as Q
    from oracle import restatement as R
    from qml_cutensornet_amd import engine

    rng = np.random.default_rng(23)
    n = 18
    xs = [Q.random_mps(n, _ragged_profile(rng, n, c), rng) for c in (70, 33, 9, 120, 48, 16, 80)]
    ys = [Q.random_mps(n, _ragged_profile(rng, n, c), rng) for c in (40, 90, 12)]
    K_ref = R.gram_from_mps([m.tensors for m in xs])
    Kxy_ref = R.gram_from_mps([m.tensors for m in xs], [m.tensors for m in ys])
    with engine.Comm(1) as comm:
        assert comm.size == 1
        c0 = comm.ctx(0)
        share = c0.upload(xs)
        full = comm.allgather_sets([share], [0], len(xs))
        assert np.array_equal(full[0].dims, share.dims)
        n_img, _, _, offs = full[0].image()
        assert n_img >= share.image()[0] and np.array_equal(offs, share.image()[3])
        K = comm.gram(full)
        st = comm.stats(0)
        assert st["pairs"] == len(xs) * (len(xs) + 1) // 2 and st["kernel_ms"] > 0 and st["allgather_ms"] >= 0
        assert np.abs(K - K_ref).max() < TOL and np.array_equal(K, K.T)
        assert np.abs(K - gpu_ctx.gram(gpu_ctx.upload(xs))).max() < 1e-13
        K2 = comm.gram(full)  # same sets: the cached plans and buffers
        assert np.abs(K2 - K).max() < 1e-14
        ysets = [c0.upload(ys)]
        Kxy = comm.gram(full, ysets)
        assert Kxy.shape == (len(ys), len(xs)) and np.abs(Kxy - Kxy_ref).max() < TOL
        with pytest.raises(engine.QkError, match="belongs to no share"):
            comm.allgather_sets([share], [0], len(xs) + 1)
        with pytest.raises(engine.QkError):
            comm.allgather_sets([gpu_ctx.upload(xs)], [0], len(xs))  # a set of another context
        # a destroyed set whose address may come back with other bonds must not meet the cached plan of the old one (the job is keyed
        # on the sets' uids as well as on their addresses): same communicator, new sets with different bonds, right answers
        ysets[0].close()
        zs = [Q.random_mps(n, _ragged_profile(rng, n, c), rng) for c in (100, 20, 61)]
        zsets = [c0.upload(zs)]
        Kxz = comm.gram(full, zsets)
        assert np.abs(Kxz - R.gram_from_mps([m.tensors for m in xs], [m.tensors for m in zs])).max() < TOL
        with pytest.raises(engine.QkError, match="NULL"):  # a ysets array holding a NULL is rejected, not dereferenced
            L = engine.lib()
            xs_h = (C.c_void_p * 1)(full[0].handle)
            ys_h = (C.c_void_p * 1)(None)
            out = np.zeros((3, len(xs)))
            engine._check(L.qk_gram_sharded(comm._h, xs_h, ys_h, out.ctypes.data, len(xs)), "qk_gram_sharded")
        leaked = (share, full[0], zsets[0])  # NOT closed by hand: the communicator closes what lives on its contexts before it goes
    for m in leaked:
        assert m.handle is None
        m.close()  # a no-op now (no use-after-free of the destroyed contexts)
    with pytest.raises(engine.QkError):
        engine.Comm(device_ids=[0, 0])  # one rank per GPU


def test_large_bonds(gpu_ctx):
    """Bonds up to 300 (padded 304: five 64-wide passes per dimension, K tails): three states against the oracle."""
    import qml_cutensornet_amd as Q
    from oracle import restatement as R

    rng = np.random.default_rng(17)
    n = 22
    caps = (300, 257, 129)
    xs = [Q.random_mps(n, [min(2 ** min(k, n - k), c) for k in range(n + 1)], rng) for c in caps]
    z_ref = np.array([[R.mps_inner(x.tensors, y.tensors) for x in xs] for y in xs])
    with gpu_ctx.upload(xs) as dx:
        assert dx.info()["max_padded_bond"] == 304
        z = gpu_ctx.overlaps(dx)
        with dx.to_f32() as fx:
            z32 = gpu_ctx.overlaps(fx)
    assert np.abs(z - z_ref).max() < TOL
    assert np.abs(np.diag(z) - 1).max() < 1e-12
    assert np.abs(z32 - z_ref).max() < F32_TOL


# ------------------------------------------------------------------ the real workloads of cfg4 and cfg5 (BASELINE.json configs[3], [4])
def _real_states(n, reps, d, gamma, points, seed=5):
    """The first `points` states of the config's own data set (synthetic features of the FULL set, seed 5), host builder."""
    import os

    import qml_cutensornet_amd as Q
    from qml_cutensornet_amd.builder_pool import build_states
    from qml_cutensornet_amd.data import synthetic_features

    full = {60: 500, 100: 1000}[n]
    X = synthetic_features(full, n, seed)[:points]
    ans = Q.KernelStateAnsatz(n, reps, gamma, Q.entanglement_graph(n, d))
    states, _ = build_states(ans, X, 1 - 1e-16, min(points, os.cpu_count() or 1))
    return states


def _check_real_workload(ctx, states, ny):
    """Full symmetric Gram + a rectangular slice (the first ny states as Y) against oracle/overlap_ref.c on the same
    tensors: absolute 1e-11 on z and K, and RELATIVE 1e-9 on every overlap (the off-diagonal overlaps of these sets
    are tiny; the 3M complex product is only normwise stable in the imaginary part)."""
    from oracle import c_oracle

    ts = [m.tensors for m in states]
    n = len(states)
    pairs = np.array([(i, j) for j in range(n) for i in range(n)], dtype=np.int32)
    _, z_ref, _ = c_oracle.gram_pairs(ts, None, pairs)
    z_ref = z_ref.reshape(n, n)
    with ctx.upload(states) as dx, ctx.upload(states[:ny]) as dy:
        K = ctx.gram(dx)
        z = ctx.overlaps(dx)
        K_rect = ctx.gram(dx, dy)
    assert np.abs(z - z_ref).max() < TOL
    assert np.abs(K - np.abs(z_ref) ** 2).max() < TOL
    assert K_rect.shape == (ny, n) and np.abs(K_rect - np.abs(z_ref[:ny]) ** 2).max() < TOL
    rel = np.abs(z - z_ref) / np.abs(z_ref)
    assert rel.max() < 1e-9, (rel.max(), np.abs(z_ref).min())
    assert np.abs(np.diag(K) - 1).max() < 1e-11 and np.array_equal(K, K.T)
    return float(np.abs(z_ref).min())


def test_cfg4_real_states(gpu_ctx, monkeypatch):
    """cfg4 (60 qubits x 6 layers, d=2, gamma=1, seed 5): 12 of its 500 states through the shipped path (site-fused
    sweep) and, on a second context, through the ring sweep."""
    from qml_cutensornet_amd import engine

    states = _real_states(60, 6, 2, 1.0, 12)
    assert max(m.max_bond() for m in states) > 64  # matrix-core regime
    smallest = _check_real_workload(gpu_ctx, states, 5)
    assert smallest < 1e-3  # the relative bound above was exercised on small overlaps
    monkeypatch.setenv("QK_FUSED", "0")
    with engine.context(0) as ctx_ring:
        _check_real_workload(ctx_ring, states, 5)


def test_cfg5_real_states(gpu_ctx, monkeypatch):
    """cfg5's real circuit (100 qubits x 10 layers, d=4, gamma=0.1, seed 5): 8 of its 1000 states through every kernel that
    takes bonds <= 32: the 2 x 2-tile wave sweep (the default), the LDS-resident small-bond sweep (QK_WAVE2=0), the site-fused
    sweep (QK_FUSED=2) and the ring sweep (everything else off)."""
    from qml_cutensornet_amd import engine

    states = _real_states(100, 10, 4, 0.1, 8)
    assert 16 < max(m.max_bond() for m in states) <= 32
    _check_real_workload(gpu_ctx, states, 3)
    assert gpu_ctx.stats()["kernel_name"] == "qk_sweep_wave2_kernel<3, double>"
    for env, name in (({"QK_WAVE2": "0"}, "small"), ({"QK_FUSED": "2"}, "fused"), ({"QK_FUSED": "0", "QK_SMALL": "0"}, "ring")):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        with engine.context(0) as ctx_other:
            _check_real_workload(ctx_other, states, 3)
            assert name in ctx_other.stats()["kernel_name"]


def test_complex64_on_the_real_states_of_cfg5_and_cfg4(gpu_ctx):
    """The fp32 leg of cfg5's "fp32 vs fp64 tolerance sweep" (BASELINE.json configs[4]; the reference never sets float_precision, ref
    gpu_backend/kernel_state_ansatz.py:141-144) on the config's REAL states, and on cfg4's.  cfg5 (bonds <= 32): complex64 STORAGE with
    fp64 arithmetic (qk_sweep_wave2_kernel<3, float>) is the fp64 sweep of the rounded tensors -- checked against the oracle ON the
    rounded tensors to 1e-11 -- and differs from the exact Gram by the input rounding only.  cfg4 (bonds to 248): complex64 ARITHMETIC on
    the fp32 matrix cores (ring sweep), within the complex64 tolerance of the fp64 Gram.  The table of both over n and chi:
    tools/fp32_sweep.py -> profiles/r04/fp32_tolerance.txt."""
    from oracle import c_oracle

    for (n, reps, d, gamma, pts), want in (((100, 10, 4, 0.1, 8), "qk_sweep_wave2_kernel<3, float>"), ((60, 6, 2, 1.0, 6), "qk_sweep_ring_kernel<float>")):
        states = _real_states(n, reps, d, gamma, pts)
        rounded = [[t.astype(np.complex64).astype(np.complex128) for t in m.tensors] for m in states]
        pairs = np.array([(i, j) for j in range(pts) for i in range(pts)], dtype=np.int32)
        _, z_round, _ = c_oracle.gram_pairs(rounded, None, pairs)
        _, z_exact, _ = c_oracle.gram_pairs([m.tensors for m in states], None, pairs)
        z_round, z_exact = z_round.reshape(pts, pts), z_exact.reshape(pts, pts)
        with gpu_ctx.upload(states) as d64, d64.to_f32() as d32:
            z32 = gpu_ctx.overlaps(d32)
            assert gpu_ctx.stats()["kernel_name"] == want
            K32, K64 = gpu_ctx.gram(d32), gpu_ctx.gram(d64)
        if "wave2" in want:
            assert np.abs(z32 - z_round).max() < TOL  # fp64 arithmetic on rounded inputs: exact to fp64 accuracy
        assert np.abs(z32 - z_exact).max() < F32_TOL and np.abs(K32 - K64).max() < F32_TOL
        assert np.abs(np.diag(K32) - 1).max() < F32_TOL and np.array_equal(K32, K32.T)


def test_both_contraction_orders_agree(gpu_ctx):
    """X1: <x_i|x_j> contracted with x_j as the Y state (the order QK_PLAN_ORIENT may pick) and with x_i as the Y state give
    conjugate overlaps to 1e-13 -- on ragged states where the two orders do different amounts of padded work."""
    import qml_cutensornet_amd as Q
    from qml_cutensornet_amd import engine

    rng = np.random.default_rng(31)
    n = 18

    def skewed(cap, late):  # bonds that peak late (slow rise, fast fall) or early: the two orders then differ a lot in padded work
        up = [min(cap, int(1.45**k) + (k > 0)) for k in range(n + 1)]
        dn = [min(cap, 2 ** (n - k)) for k in range(n + 1)]
        prof = [min(u, d, 2 ** min(k, n - k)) for k, (u, d) in enumerate(zip(up, dn))]
        return prof if late else prof[::-1]

    xs = [Q.random_mps(n, skewed(cap, late), rng) for cap, late in ((90, True), (120, False), (60, True), (130, True), (100, False), (70, False))]
    with gpu_ctx.upload(xs) as dx:
        plain, orient = engine.Plan(dx.dims, orient=False), engine.Plan(dx.dims, orient=True)
        v0, z0 = gpu_ctx.gram_values_host(dx, None, plain, want_z=True)
        v1, z1 = gpu_ctx.gram_values_host(dx, None, orient, want_z=True)
        zmap = {(i, j): z for (i, j), z in zip(plain.pairs().tolist(), z0)}
        turned = 0
        for (i, j), v, z in zip(orient.pairs().tolist(), v1, z1):
            ref = zmap[(i, j)] if (i, j) in zmap else np.conj(zmap[(j, i)])
            turned += (i, j) not in zmap
            assert abs(z - ref) < 1e-13 and abs(v - abs(ref) ** 2) < 1e-13
        assert turned > 0
        K = gpu_ctx.gram(dx)  # (plans of the convenience calls are oriented too)
        assert np.array_equal(K, K.T) and np.abs(np.diag(K) - 1).max() < 1e-12
        plain.close(), orient.close()


def test_bonds_beyond_the_fused_sweep_take_the_ring_sweep(gpu_ctx):
    """Bonds > 512 (one 16-row strip of X' no longer fits the fused sweep's LDS buffer): the engine falls back to the ring
    sweep on its own, and says so in the stats."""
    import qml_cutensornet_amd as Q
    from oracle import restatement as R

    rng = np.random.default_rng(23)
    n = 22
    xs = [Q.random_mps(n, [min(2 ** min(k, n - k), c) for k in range(n + 1)], rng) for c in (530, 90)]
    z_ref = np.array([[R.mps_inner(x.tensors, y.tensors) for x in xs] for y in xs])
    with gpu_ctx.upload(xs) as dx:
        assert dx.info()["max_padded_bond"] == 544
        z = gpu_ctx.overlaps(dx)
        assert "ring" in gpu_ctx.stats()["kernel_name"]
    assert np.abs(z - z_ref).max() < TOL
    with gpu_ctx.upload(xs[1:]) as dy:  # the same state alone: the fused sweep
        gpu_ctx.overlaps(dy)
        assert "fused" in gpu_ctx.stats()["kernel_name"]


@pytest.mark.parametrize("n,chi_max,nx,ny,seed", [(14, 40, 3, 3, 1), (18, 100, 3, 2, 2), (20, 150, 2, 3, 3), (22, 260, 2, 2, 4)])
def test_dual_form_of_the_fused_sweep(gpu_ctx, n, chi_max, nx, ny, seed, monkeypatch):
    """qk_sweep_fused_dual_kernel (pairs of tiles per wave, the default form of the 12-wave shape; QK_FUSED_DUAL=0: single tiles): against the
    oracle on ragged sets of every site class -- LDS-resident in place and ping-pong, single and several strips, odd strip
    widths and odd tile counts (single tiles at the end) -- and against the single-tile form on the same inputs."""
    import qml_cutensornet_amd as Q
    from qml_cutensornet_amd import engine
    from oracle import restatement as R

    rng = np.random.default_rng(seed)

    def profile():  # ragged, but every inner bond at least a third of the cap
        chi = [1]
        for k in range(1, n):
            cap = min(2 ** min(k, n - k, 20), chi_max, 2 * chi[-1])
            chi.append(int(rng.integers(min(max(1, chi_max // 3), cap), cap + 1)))
        chi.append(1)
        for k in range(n - 1, 0, -1):
            chi[k] = min(chi[k], 2 * chi[k + 1])
        return chi

    xs = [Q.random_mps(n, profile(), rng) for _ in range(nx)]
    ys = [Q.random_mps(n, profile(), rng) for _ in range(ny)]
    z_ref = np.array([[R.mps_inner(x.tensors, y.tensors) for x in xs] for y in ys])
    out = {}
    for dual in ("1", "0"):
        monkeypatch.setenv("QK_FUSED_DUAL", dual)
        monkeypatch.setenv("QK_FUSED_WGS", "1")
        monkeypatch.setenv("QK_FUSED", "2")  # (the fused sweep also for sets whose bonds happen to stay <= 32)
        with engine.context(0) as ctx1, ctx1.upload(xs) as dx, ctx1.upload(ys) as dy:
            out[dual] = ctx1.overlaps(dx, dy)
            assert ("dual" in ctx1.stats()["kernel_name"]) == (dual == "1")
    assert np.abs(out["1"] - z_ref).max() < TOL
    assert np.abs(out["1"] - out["0"]).max() < 1e-13
    # left to itself the 12-wave shape runs in the dual form
    monkeypatch.delenv("QK_FUSED_DUAL")
    monkeypatch.delenv("QK_FUSED_WGS")
    monkeypatch.delenv("QK_FUSED")
    if seed == 1:
        big = [Q.random_mps(16, [min(2 ** min(k, 16 - k), 112) for k in range(17)], rng) for _ in range(3)]
        with engine.context(0) as ctx1, ctx1.upload(big) as db:
            z = ctx1.overlaps(db)
            assert "dual" in ctx1.stats()["kernel_name"]
        z_big = np.array([[R.mps_inner(x.tensors, y.tensors) for x in big] for y in big])
        assert np.abs(z - z_big).max() < TOL


def test_split_sweep_two_shapes_one_gram(gpu_ctx, monkeypatch):
    """A set of small and large states: the plan's second run (pairs whose sites fit the smaller LDS buffer) is swept by the
    two-workgroups-per-CU shape right after the first; same values as the one-shape sweep and as the oracle."""
    import qml_cutensornet_amd as Q
    from qml_cutensornet_amd import engine
    from oracle import restatement as R

    rng = np.random.default_rng(31)
    n = 16
    caps = [40, 48, 56, 60, 150, 120, 100, 64]
    xs = [Q.random_mps(n, [min(2 ** min(k, n - k), c) for k in range(n + 1)], rng) for c in caps]
    K_ref = np.array([[abs(R.mps_inner(x.tensors, y.tensors)) ** 2 for x in xs] for y in xs])
    monkeypatch.setenv("QK_FUSED_SPLIT", "2")  # (a share of fewer than 100 pairs per CU is ONE launch by default: its second launch would be mostly tail)
    with engine.context(0) as ctx2, ctx2.upload(xs) as dx:
        K = ctx2.gram(dx)
        st = ctx2.stats()
        assert st["kernel_name"].startswith("qk_sweep_fused_dual_kernel<12") and st["second_kernel_name"].startswith("qk_sweep_fused_kernel<8")
        assert 0 < st["second_pairs"] < st["pairs"] and 0 < st["second_ms"] < st["kernel_ms"] and 0 < st["second_flops"] < st["flops"]
    monkeypatch.setenv("QK_FUSED_SPLIT", "0")
    with engine.context(0) as ctx1, ctx1.upload(xs) as dx1:
        K1 = ctx1.gram(dx1)
        assert ctx1.stats()["second_kernel"] == 0 and ctx1.stats()["second_ms"] == 0
    assert np.abs(K - K_ref).max() < TOL and np.abs(K - K1).max() < 1e-13 and np.array_equal(K, K.T)


def test_gang_start_changes_no_value(gpu_ctx, monkeypatch):
    """QK_GANG=1 (the workgroups of an XCD begin their pairs together: qk_device.h, qk_gang_sync) is a matter of timing only: same
    Gram as the free-running launch (up to the arrival order of the LDS adds) and as the oracle; 120 pairs = 15 workgroups per XCD."""
    import qml_cutensornet_amd as Q
    from qml_cutensornet_amd import engine
    from oracle import restatement as R

    rng = np.random.default_rng(77)
    n = 14
    caps = [40, 48, 56, 60, 100, 90, 80, 64, 70, 44, 36, 52, 96, 72, 66]
    xs = [Q.random_mps(n, [min(2 ** min(k, n - k), c) for k in range(n + 1)], rng) for c in caps]
    K_ref = np.array([[abs(R.mps_inner(x.tensors, y.tensors)) ** 2 for x in xs] for y in xs])
    monkeypatch.setenv("QK_GANG", "1")
    with engine.context(0) as ctx1, ctx1.upload(xs) as dx:
        K_gang = ctx1.gram(dx)
        assert "fused" in ctx1.stats()["kernel_name"]
    monkeypatch.delenv("QK_GANG")
    with engine.context(0) as ctx2, ctx2.upload(xs) as dx:
        K = ctx2.gram(dx)
    assert np.abs(K_gang - K_ref).max() < TOL and np.abs(K_gang - K).max() < 1e-13 and np.array_equal(K_gang, K_gang.T)


def test_one_class_sets_pick_their_shape_by_site_size(gpu_ctx):
    """A set with ONE class of pairs whose sites all fit the smaller LDS buffer: bonds capped at 48 (3 x 3 tiles per site) stay on
    the two-workgroup shape, bonds capped at 64 (4 x 4 tiles: what QK_MAX_BOND=64 produces) take the 12-wave dual shape -- the
    planner's fit_narrow share decides (qk_gram_values); both against the oracle."""
    import qml_cutensornet_amd as Q
    from oracle import restatement as R

    rng = np.random.default_rng(47)
    n = 14
    for cap, want in ((48, "qk_sweep_fused_kernel<8, 1, 4608, 4, false>"), (64, "qk_sweep_fused_dual_kernel<12, 8192, 3, false>")):
        xs = [Q.random_mps(n, [min(2 ** min(k, n - k), cap) for k in range(n + 1)], rng) for _ in range(6)]
        K_ref = np.array([[abs(R.mps_inner(x.tensors, y.tensors)) ** 2 for x in xs] for y in xs])
        with gpu_ctx.upload(xs) as dx:
            K = gpu_ctx.gram(dx)
            st = gpu_ctx.stats()
        assert st["kernel_name"] == want and st["second_kernel"] == 0, (cap, st["kernel_name"])
        assert np.abs(K - K_ref).max() < TOL and np.array_equal(K, K.T)


def test_complex64_storage_wave_sweep(gpu_ctx, monkeypatch):
    """Complex64 sets with bonds <= 32 take qk_sweep_wave2_kernel<3, float>: single-precision STORAGE, fp64 arithmetic.  Its
    result is the fp64 sweep of the rounded tensors -- checked against the oracle ON the rounded tensors to fp64 accuracy --
    and differs from the exact overlap by the input rounding only (well inside the complex64 tolerance)."""
    import qml_cutensornet_amd as Q
    from qml_cutensornet_amd import engine
    from oracle import restatement as R

    rng = np.random.default_rng(77)
    for n, chi_max, nx, ny in ((30, 32, 4, 3), (12, 9, 3, 3), (64, 20, 3, 4)):
        xs = [Q.random_mps(n, _ragged_profile(rng, n, chi_max), rng) for _ in range(nx)]
        ys = [Q.random_mps(n, _ragged_profile(rng, n, chi_max), rng) for _ in range(ny)]
        rounded = lambda m: [t.astype(np.complex64).astype(np.complex128) for t in m.tensors]
        z_exact = np.array([[R.mps_inner(x.tensors, y.tensors) for x in xs] for y in ys])
        z_round = np.array([[R.mps_inner(rounded(x), rounded(y)) for x in xs] for y in ys])
        with gpu_ctx.upload(xs) as dx, gpu_ctx.upload(ys) as dy, dx.to_f32() as fx, dy.to_f32() as fy:
            z32 = gpu_ctx.overlaps(fx, fy)
            assert gpu_ctx.stats()["kernel_name"] == "qk_sweep_wave2_kernel<3, float>"
            K32 = gpu_ctx.gram(fx)
        assert np.abs(z32 - z_round).max() < TOL
        assert np.abs(z32 - z_exact).max() < F32_TOL
        assert np.array_equal(K32, K32.T)
    # QK_WAVE2=0: the complex64 arithmetic path (LDS-resident small-bond sweep) on the same sets
    monkeypatch.setenv("QK_WAVE2", "0")
    with engine.context(0) as ctx1, ctx1.upload(xs) as dx, ctx1.upload(ys) as dy, dx.to_f32() as fx, dy.to_f32() as fy:
        z_small = ctx1.overlaps(fx, fy)
        assert "small" in ctx1.stats()["kernel_name"]
    assert np.abs(z_small - z32).max() < F32_TOL


def test_deterministic_mode_is_bit_reproducible(built, monkeypatch):
    """QK_DETERMINISTIC=1 switches the site-fused sweep to its DET forms: the contributions to a block of X' are added in a fixed
    order (turn counters in LDS) instead of in arrival order.  Grams of the same ragged set are then bit-identical, launch after
    launch and context after context, in every shape of the kernel -- the split sweep (12-wave dual shape + two workgroups per CU),
    the single-tile 12-wave form, sites beyond the LDS buffer (strips), one class of small sites -- and agree with the oracle and
    with the default (arrival-order) path to rounding."""
    import qml_cutensornet_amd as Q
    from oracle import restatement as R
    from qml_cutensornet_amd import engine

    rng = np.random.default_rng(41)
    n = 24
    cases = {
        "split": [Q.random_mps(n, [min(2 ** min(k, n - k), c) for k in range(n + 1)], rng) for c in (90, 140, 40, 70, 33, 120, 64, 18)],
        "small": [Q.random_mps(14, [min(2 ** min(k, 14 - k), 48) for k in range(15)], rng) for _ in range(6)],
        "strips": [Q.random_mps(18, [min(2 ** min(k, 18 - k), c) for k in range(19)], rng) for c in (200, 150, 33, 97)],
    }
    for name, xs in cases.items():
        K_ref = R.gram_from_mps([m.tensors for m in xs])
        monkeypatch.delenv("QK_DETERMINISTIC", raising=False)
        monkeypatch.delenv("QK_FUSED_DUAL", raising=False)
        with engine.context(0) as ctx, ctx.upload(xs) as dx:
            K_default = ctx.gram(dx)
            assert "fused" in ctx.stats()["kernel_name"] and "true" not in ctx.stats()["kernel_name"]
        monkeypatch.setenv("QK_DETERMINISTIC", "1")
        monkeypatch.setenv("QK_FUSED_SPLIT", "2" if name == "split" else "1")  # (short shares are one launch by default)
        for dual in (("1", "0") if name != "small" else ("1",)):
            monkeypatch.setenv("QK_FUSED_DUAL", dual)
            runs = []
            for _ in range(2):
                with engine.context(0) as ctx, ctx.upload(xs) as dx:
                    runs.append(ctx.gram(dx))
                    runs.append(ctx.gram(dx))
                    st = ctx.stats()
                    assert "fused" in st["kernel_name"] and st["kernel_name"].endswith("true>"), st["kernel_name"]
                    if name == "split":
                        assert st["second_kernel_name"] == "qk_sweep_fused_kernel<8, 1, 4608, 4, true>"
            for K in runs[1:]:
                assert np.array_equal(K, runs[0]), (name, dual)
            assert np.abs(runs[0] - K_ref).max() < TOL and np.abs(runs[0] - K_default).max() < 1e-13, (name, dual)


def test_mixed_set_keeps_small_pairs_on_the_one_wave_sweep(gpu_ctx, monkeypatch):
    """A set of small states (every bond <= 32) with a few large ones: the plan lists the small-small pairs as its second run and the
    engine sweeps them with the one-wave kernel right behind the fused launch -- one large state no longer drags every pair of the
    set onto the multi-wave kernels.  Same Gram as the oracle and as the unmixed plan."""
    import qml_cutensornet_amd as Q
    from oracle import restatement as R
    from qml_cutensornet_amd import engine

    rng = np.random.default_rng(77)
    n = 20
    caps = [30, 24, 17, 32, 9, 28, 31, 12, 20, 26, 16, 29, 90, 140, 60]
    xs = [Q.random_mps(n, [min(2 ** min(k, n - k), c) for k in range(n + 1)], rng) for c in caps]
    K_ref = R.gram_from_mps([m.tensors for m in xs])
    plan = engine.Plan(np.array([m.bond_dims() for m in xs]))
    small = {i for i, c in enumerate(caps) if c <= 32}
    pr, first = plan.pairs(), plan.first_run
    assert len(pr) - first == len(small) * (len(small) + 1) // 2
    assert all((i in small and j in small) == (t >= first) for t, (i, j) in enumerate(pr.tolist()))
    plan.close()
    with gpu_ctx.upload(xs) as dx:
        K = gpu_ctx.gram(dx)
        st = gpu_ctx.stats()
        assert "fused" in st["kernel_name"] and "wave2" in st["second_kernel_name"] and st["second_pairs"] == len(pr) - first and st["second_ms"] > 0
        monkeypatch.setenv("QK_PLAN_NO_MIXED", "1")
        K_plain = gpu_ctx.gram(dx)
        assert "wave2" not in gpu_ctx.stats()["second_kernel_name"]
    assert np.abs(K - K_ref).max() < TOL and np.array_equal(K, K.T) and np.abs(K - K_plain).max() < 1e-13


@pytest.mark.parametrize("edge", ["4", "6", "8"])
def test_edge_blocks_agree_with_the_plain_chain(built, monkeypatch, edge):
    """The ends of the chain from the sets' edge blocks (QK_EDGE: the first and last k sites of every state contracted across their
    physical legs, X = Ly^T conj(Lx) at the left, sum X . (Ry^T conj(Rx)) at the right) against the plain sweep and the oracle, on
    ragged states whose environment behind the edge is anything from one tile to larger than the LDS buffer; symmetric and
    rectangular Grams, complex overlaps included."""
    import qml_cutensornet_amd as Q
    from oracle import restatement as R
    from qml_cutensornet_amd import engine

    rng = np.random.default_rng(int(edge) + 5)
    n = 26
    caps = (300, 150, 96, 60, 33, 17, 200)
    xs = [Q.random_mps(n, [min(2 ** min(k, n - k), c) for k in range(n + 1)], rng) for c in caps]
    ys = [Q.random_mps(n, _ragged_profile(rng, n, c), rng) for c in (120, 40, 70)]
    z_ref = np.array([[R.mps_inner(x.tensors, y.tensors) for x in xs] for y in ys])
    K_ref = R.gram_from_mps([m.tensors for m in xs])
    out = {}
    monkeypatch.setenv("QK_MERGE", "0")  # (merged steps have their own test, with and without edge blocks; here the byte count below is the edge blocks' alone)
    for mode in ("0", edge):
        monkeypatch.setenv("QK_EDGE", mode)
        with engine.context(0) as ctx, ctx.upload(xs) as dx, ctx.upload(ys) as dy:
            out[mode] = (ctx.gram(dx), ctx.overlaps(dx, dy), dx.info()["device_bytes"])
            assert "fused" in ctx.stats()["kernel_name"]
    for mode in out:
        assert np.abs(out[mode][0] - K_ref).max() < TOL and np.abs(out[mode][1] - z_ref).max() < TOL
        assert np.array_equal(out[mode][0], out[mode][0].T)
    assert np.abs(out[edge][0] - out["0"][0]).max() < 1e-13 and np.abs(out[edge][1] - out["0"][1]).max() < 1e-13
    assert out[edge][2] > out["0"][2]  # the edge blocks are counted in the set's device bytes


@pytest.mark.parametrize("n,edge", [(26, "0"), (26, "4"), (27, "4"), (27, "0"), (30, "6"), (12, "4")])
def test_merged_sites_agree_with_the_plain_chain(built, monkeypatch, n, edge):
    """The chain walked in merged steps (QK_MERGE: two neighbouring sites contracted over their bond into one tensor of physical
    dimension 4, a last single site when the chain is odd) against the plain walk and the oracle: with and without edge blocks,
    chains of even and odd length, ragged states from one tile to environments larger than the LDS buffer, complex overlaps."""
    import qml_cutensornet_amd as Q
    from oracle import restatement as R
    from qml_cutensornet_amd import engine

    rng = np.random.default_rng(n + int(edge))
    caps = (300, 150, 96, 60, 33, 17, 200)
    xs = [Q.random_mps(n, [min(2 ** min(k, n - k), c) for k in range(n + 1)], rng) for c in caps]
    ys = [Q.random_mps(n, _ragged_profile(rng, n, c), rng) for c in (120, 40, 70)]
    z_ref = np.array([[R.mps_inner(x.tensors, y.tensors) for x in xs] for y in ys])
    K_ref = R.gram_from_mps([m.tensors for m in xs])
    monkeypatch.setenv("QK_EDGE", edge)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("QK_MERGE", mode)
        with engine.context(0) as ctx, ctx.upload(xs) as dx, ctx.upload(ys) as dy:
            out[mode] = (ctx.gram(dx), ctx.overlaps(dx, dy), dx.info()["device_bytes"])
            assert "fused" in ctx.stats()["kernel_name"]
    for mode in out:
        assert np.abs(out[mode][0] - K_ref).max() < TOL and np.abs(out[mode][1] - z_ref).max() < TOL
        assert np.array_equal(out[mode][0], out[mode][0].T)
    assert np.abs(out["1"][0] - out["0"][0]).max() < 1e-13 and np.abs(out["1"][1] - out["0"][1]).max() < 1e-13
    assert out["1"][2] > out["0"][2]  # the merged image is counted in the set's device bytes
