"""GPU tests of the device MPS builder: its Jacobi primitive against numpy's SVD, built states against the
host builder (same algorithm on LAPACK) and against exact state vectors (the golden fixtures and the oracle).

Tolerances: the two builders truncate along different numerical paths, so states agree to the truncation error
(1e-16 of weight per gate): |<dev|host>|^2 = 1 and Gram entries within 1e-9; against the exact state vector 1e-8, the
same bound the host builder is held to (tests/test_gpu_parity.py)."""
import numpy as np
import pytest

from helpers import golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("p,q,rank", [(1, 1, None), (2, 2, None), (4, 2, None), (20, 12, None), (47, 43, None), (64, 64, None), (130, 100, None),
                                      (60, 40, 17), (33, 33, 1), (16, 8, 4)])
def test_jacobi_primitive_against_numpy_svd(gpu_ctx, p, q, rank):
    rng = np.random.default_rng(p * 1000 + q)
    a = rng.standard_normal((p, q)) + 1j * rng.standard_normal((p, q))
    if rank:
        a = (rng.standard_normal((p, rank)) + 1j * rng.standard_normal((p, rank))) @ (rng.standard_normal((rank, q)) + 1j * rng.standard_normal((rank, q)))
    w, v, sig, order = gpu_ctx.debug_jacobi(a)
    s_ref = np.linalg.svd(a, compute_uv=False)
    assert sorted(order.tolist()) == list(range(q))
    assert np.all(np.diff(sig[order]) <= 0)
    assert np.abs(sig[order] - s_ref).max() < 1e-13 * s_ref[0]
    assert np.abs(w @ v.conj().T - a).max() < 1e-13 * np.abs(a).max()  # A V = W with V unitary
    assert np.abs(v.conj().T @ v - np.eye(q)).max() < 1e-13
    g = w.conj().T @ w  # columns of W mutually orthogonal: absolute accuracy eps |A|^2
    assert np.abs(g - np.diag(np.diag(g))).max() < 1e-13 * s_ref[0] ** 2


@pytest.mark.parametrize("n,reps,gamma,d,npts", [(8, 2, 1.0, 1, 6), (12, 3, 1.0, 2, 6), (20, 4, 1.0, 2, 5), (9, 2, 0.3, 4, 4)])
def test_device_builder_matches_host_builder(gpu_ctx, n, reps, gamma, d, npts):
    import qml_cutensornet_amd as Q
    from oracle import restatement as R

    X = R.synthetic_features(npts, n, 7)
    ans = Q.KernelStateAnsatz(n, reps, gamma, Q.entanglement_graph(n, d))
    circuits = [ans.circuit_for_data(x) for x in X]
    dev, info = gpu_ctx.build_mps(circuits)
    host = [Q.simulate(c, 1 - 1e-16) for c in circuits]
    assert info["kernel_ms"] > 0 and len(dev) == npts
    for md, mh in zip(dev, host):
        assert md.bond_dims()[0] == 1 and md.bond_dims()[-1] == 1
        assert md.max_bond() <= max(2 * mh.max_bond(), 2)
        assert abs(md.fidelity - 1.0) < 1e-12
    z = np.array([R.mps_inner(md.tensors, mh.tensors) for md, mh in zip(dev, host)])
    assert np.abs(np.abs(z) ** 2 - 1).max() < 1e-10
    with gpu_ctx.upload(dev) as dx, gpu_ctx.upload(host) as hx:
        assert np.abs(gpu_ctx.gram(dx) - gpu_ctx.gram(hx)).max() < 1e-9


@pytest.mark.parametrize("name", ["cfg1_8q_r1_d1.npz", "deep_10q_r3_d3.npz"])
def test_device_built_gram_against_exact_statevectors(gpu_ctx, name):
    import qml_cutensornet_amd as Q

    g = golden(name)
    n, reps, gamma, d = int(g["n"]), int(g["reps"]), float(g["gamma"]), int(g["d"])
    ans = Q.KernelStateAnsatz(n, reps, gamma, Q.entanglement_graph(n, d))
    tr, _ = gpu_ctx.build_mps([ans.circuit_for_data(x) for x in g["X_train"]])
    te, _ = gpu_ctx.build_mps([ans.circuit_for_data(x) for x in g["X_test"]])
    with gpu_ctx.upload(tr) as dx, gpu_ctx.upload(te) as dy:
        assert np.abs(gpu_ctx.gram(dx) - g["K_train"]).max() < 1e-8
        assert np.abs(gpu_ctx.gram(dx, dy) - g["K_test"]).max() < 1e-8


def test_device_builder_truncates_like_the_host_builder(gpu_ctx):
    """A loose fidelity really truncates: same kept bonds and fidelity product as the host builder, state by state."""
    import qml_cutensornet_amd as Q
    from oracle import restatement as R

    n, reps, d = 14, 3, 2
    X = R.synthetic_features(4, n, 11)
    ans = Q.KernelStateAnsatz(n, reps, 1.0, Q.entanglement_graph(n, d))
    circuits = [ans.circuit_for_data(x) for x in X]
    dev, _ = gpu_ctx.build_mps(circuits, truncation_fidelity=1 - 1e-4)
    host = [Q.simulate(c, 1 - 1e-4) for c in circuits]
    for md, mh in zip(dev, host):
        assert md.max_bond() <= mh.max_bond() + 1
        assert abs(md.fidelity - mh.fidelity) < 1e-6
        assert abs(abs(R.mps_inner(md.tensors, mh.tensors)) ** 2 - 1) < 1e-3


def test_device_builder_rejects_bad_input(gpu_ctx):
    import qml_cutensornet_amd as Q
    from qml_cutensornet_amd.engine import QkError
    from oracle import restatement as R

    ans = Q.KernelStateAnsatz(10, 3, 1.0, Q.entanglement_graph(10, 3))
    c = ans.circuit_for_data(R.synthetic_features(3, 10, 3)[0])
    with pytest.raises(QkError, match="max_bond"):
        gpu_ctx.build_mps([c], max_bond=2)  # the state needs more than 2
    other = Q.KernelStateAnsatz(10, 2, 1.0, Q.entanglement_graph(10, 3)).circuit_for_data(R.synthetic_features(3, 10, 3)[0])
    with pytest.raises(QkError, match="gate structure"):
        gpu_ctx.build_mps([c, other])
    with pytest.raises(QkError):
        gpu_ctx.build_mps([])


@pytest.mark.parametrize("name,tol", [("cfg1_8q_r1_d1.npz", 1e-9), ("cfg2_20q_r2_d1_subset.npz", 1e-9), ("deep_10q_r3_d3.npz", 1e-8)])
def test_build_kernel_matrix_with_device_builder(built, name, tol, tmp_path, monkeypatch):
    """The reference's module surface with QK_BUILDER=device: states built on the GPU, train and test Gram against the
    exact-statevector fixtures, profiling JSON still complete."""
    import json

    import qml_cutensornet_amd as Q
    from qml_cutensornet_amd.dist import SingleComm
    from qml_cutensornet_amd.gpu_backend.kernel_state_ansatz import KernelStateAnsatz, build_kernel_matrix

    monkeypatch.setenv("QK_BUILDER", "device")
    g = golden(name)
    n, reps, gamma = int(g["n"]), int(g["reps"]), float(g["gamma"])
    ans = KernelStateAnsatz(num_qubits=n, reps=reps, gamma=gamma, entanglement_map=Q.entanglement_graph(n, int(g["d"])), hadamard_init=True)
    info = str(tmp_path / "train_info")
    K = build_kernel_matrix(SingleComm(), ans, X=g["X_train"], info_file=info, truncation_error=1e-16)
    assert np.abs(K - g["K_train"]).max() < tol
    prof = json.load(open(info + ".json"))
    assert "avg_circ_sim" in prof and "avg_fidelity" in prof and abs(prof["avg_fidelity"][0] - 1) < 1e-9
    Kt = build_kernel_matrix(SingleComm(), ans, X=g["X_train"], Y=g["X_test"], truncation_error=1e-16)
    assert np.abs(Kt - g["K_test"]).max() < tol


def test_built_states_stay_on_the_device(gpu_ctx):
    """build_mps_set packs the built states into a Gram-engine set on the device: same Gram as download + upload."""
    import qml_cutensornet_amd as Q
    from oracle import restatement as R

    n, reps, d = 16, 3, 2
    X = R.synthetic_features(7, n, 21)
    ans = Q.KernelStateAnsatz(n, reps, 0.7, Q.entanglement_graph(n, d))
    circuits = [ans.circuit_for_data(x) for x in X]
    states, _ = gpu_ctx.build_mps(circuits)
    dset, info = gpu_ctx.build_mps_set(circuits)
    assert np.array_equal(info["dims"], np.array([m.bond_dims() for m in states]))
    with gpu_ctx.upload(states) as up:
        assert np.abs(gpu_ctx.gram(dset) - gpu_ctx.gram(up)).max() < 1e-12
        assert dset.info()["device_bytes"] == up.info()["device_bytes"]
    K_sv = R.gram_statevector(X, None, reps, 0.7, Q.entanglement_graph(n, d))
    assert np.abs(gpu_ctx.gram(dset) - K_sv).max() < 1e-8
    dset.close()


def test_hybrid_builder_falls_back_to_the_host(built, monkeypatch, capsys):
    """QK_BUILDER=hybrid: the device builder with bonds capped (here at 4, so that it must give up), then the host builder."""
    import qml_cutensornet_amd as Q
    from qml_cutensornet_amd.dist import SingleComm
    from qml_cutensornet_amd.gpu_backend.kernel_state_ansatz import KernelStateAnsatz, build_kernel_matrix

    g = golden("deep_10q_r3_d3.npz")
    n, reps, gamma = int(g["n"]), int(g["reps"]), float(g["gamma"])
    ans = KernelStateAnsatz(num_qubits=n, reps=reps, gamma=gamma, entanglement_map=Q.entanglement_graph(n, int(g["d"])), hadamard_init=True)
    monkeypatch.setenv("QK_BUILDER", "hybrid")
    monkeypatch.setenv("QK_BUILDER_MAX_BOND", "4")
    K = build_kernel_matrix(SingleComm(), ans, X=g["X_train"], truncation_error=1e-16)
    assert "on the host" in capsys.readouterr().out
    assert np.abs(K - g["K_train"]).max() < 1e-8
    monkeypatch.setenv("QK_BUILDER_MAX_BOND", "64")
    K2 = build_kernel_matrix(SingleComm(), ans, X=g["X_train"], truncation_error=1e-16)
    assert "on the host" not in capsys.readouterr().out
    assert np.abs(K2 - g["K_train"]).max() < 1e-8


def test_partial_build_drops_outgrown_states(gpu_ctx):
    """partial=True: a state that outgrows max_bond is dropped (None, listed in info["dropped"]); the others are complete."""
    import qml_cutensornet_amd as Q
    from oracle import restatement as R

    n, reps, d = 12, 3, 2
    X = R.synthetic_features(8, n, 4)
    X[0] = X[1] = 2.0  # features at 2 give exponents 0: product states, bonds of 1
    ans = Q.KernelStateAnsatz(n, reps, 1.0, Q.entanglement_graph(n, d))
    circuits = [ans.circuit_for_data(x) for x in X]
    full, _ = gpu_ctx.build_mps(circuits)
    cap = 8
    part, info = gpu_ctx.build_mps(circuits, max_bond=cap, partial=True)
    big = [i for i, m in enumerate(full) if m.max_bond() > cap]
    assert big and len(big) < len(circuits)
    assert set(big) <= set(info["dropped"])  # (a transient bond may exceed the cap even if the final ones do not)
    for i, m in enumerate(part):
        if i in info["dropped"]:
            assert m is None
        else:
            assert m.max_bond() <= cap and abs(abs(R.mps_inner(m.tensors, full[i].tensors)) ** 2 - 1) < 1e-10


def test_build_kernel_matrix_with_host_builder_pool(built, monkeypatch):
    """QK_BUILDER=host: the module surface builds its states with the threaded host builder."""
    import qml_cutensornet_amd as Q
    from qml_cutensornet_amd.dist import SingleComm
    from qml_cutensornet_amd.gpu_backend.kernel_state_ansatz import KernelStateAnsatz, build_kernel_matrix

    monkeypatch.setenv("QK_BUILDER", "host")
    g = golden("cfg2_20q_r2_d1_subset.npz")
    n, reps, gamma = int(g["n"]), int(g["reps"]), float(g["gamma"])
    ans = KernelStateAnsatz(num_qubits=n, reps=reps, gamma=gamma, entanglement_map=Q.entanglement_graph(n, int(g["d"])), hadamard_init=True)
    K = build_kernel_matrix(SingleComm(), ans, X=g["X_train"], truncation_error=1e-16)
    assert np.abs(K - g["K_train"]).max() < 1e-10


def test_hybrid_builder_policy(gpu_ctx, monkeypatch, capsys):
    """QK_BUILDER=hybrid on a share large enough for the pilot (>= 24 states): the heaviest quarter starts on the host pool, a
    pilot of 8 runs on the device at the same time and sets the prediction threshold, predicted-to-fit states are built
    on the device while the host pool takes the rest.  With a cap of 8 on a 14-qubit, 3-layer, d=2 ansatz some states fit
    and some do not.  Every state must equal the host builder's (|<a|b>|^2 = 1), whoever built it."""
    import qml_cutensornet_amd as Q
    from oracle import restatement as R
    from qml_cutensornet_amd.gpu_backend import kernel_state_ansatz as M

    n, reps, d = 14, 3, 2
    X = R.synthetic_features(32, n, 9)
    X[:6] = 1.0 + 0.02 * (X[:6] - 1.0)  # six nearly product states (tiny XXPhase angles): light ones that fit any cap
    ans = Q.KernelStateAnsatz(n, reps, 1.0, Q.entanglement_graph(n, d))
    circuits = [ans.circuit_for_data(x) for x in X]
    weights = [M._entangling_weight(c) for c in circuits]
    assert max(weights[:6]) < min(weights[6:])
    states, secs = M._hybrid_build(gpu_ctx, circuits, 1 - 1e-16, 8, 4, True, "X")
    out = capsys.readouterr().out
    assert "pilot of" in out and "device builder takes" in out
    assert len(states) == 32 and all(m is not None for m in states) and all(t >= 0 for t in secs)
    ref = [Q.simulate(c, 1 - 1e-16) for c in circuits]
    for a, b in zip(states, ref):
        assert abs(abs(R.mps_inner(a.tensors, b.tensors)) ** 2 - 1) < 1e-9
    assert max(m.max_bond() for m in states) > 8  # (so the cap did bite: those states came from the host pool)


def test_auto_builder_policy_and_fallback(built, monkeypatch, capsys):
    """QK_BUILDER=auto (the default): the device builder takes a share when it holds at least 2.5 x host workers x (w_max / w_mean)^2
    states (w = the entangling weight of a circuit: the launch ends with its heaviest state), the host pool otherwise -- same Gram
    either way; a device failure in auto mode falls back to the host, a FORCED device build raises; QK_MAX_BOND caps the bonds of
    either builder the same way."""
    import qml_cutensornet_amd as Q
    from oracle import restatement as R
    from qml_cutensornet_amd import engine
    from qml_cutensornet_amd.dist import SingleComm
    from qml_cutensornet_amd.gpu_backend import kernel_state_ansatz as M

    n = 16
    X = R.synthetic_features(40, n, 3)
    ans = Q.KernelStateAnsatz(n, 3, 1.0, Q.entanglement_graph(n, 2))
    circuits = [ans.circuit_for_data(x) for x in X]
    w = np.array([M._entangling_weight(c) for c in circuits])
    ratio2 = (w.max() / w.mean()) ** 2
    assert M._auto_builder(circuits, 1) == "device" and 40 >= 2.5 * ratio2  # one host core: the device takes 40 states
    assert M._auto_builder(circuits, 64) == "host"  # 64 cores would need >= 160 states
    assert M._auto_builder(circuits[:3], 2) == "host"
    calls = []
    real = M._engine.Context.build_share
    monkeypatch.setattr(M._engine.Context, "build_share", lambda self, *a, **k: (calls.append(len(a[0])), real(self, *a, **k))[1])
    K_ref = R.gram_from_mps([Q.simulate(c, 1 - 1e-16).tensors for c in circuits[:6]])
    for workers, expect_device in (("1", True), ("64", False)):
        calls.clear()
        monkeypatch.setenv("QK_BUILDER", "auto")
        monkeypatch.setattr(M, "default_workers", lambda: int(workers), raising=False)
        monkeypatch.setenv("QK_BUILD_WORKERS", workers)
        K = M.build_kernel_matrix(SingleComm(), ans, X=X, truncation_error=1e-16)
        assert np.abs(K[:6, :6] - K_ref).max() < 1e-9
        if expect_device:
            assert calls == [40]
    # auto falls back to the host when the device builder gives up; a forced device build does not
    def boom(self, *a, **k):
        raise engine.QkError("qk_build_mps: injected failure")

    monkeypatch.setattr(M._engine.Context, "build_share", boom)
    monkeypatch.setenv("QK_BUILD_WORKERS", "1")
    monkeypatch.setenv("QK_BUILDER", "auto")
    K = M.build_kernel_matrix(SingleComm(), ans, X=X[:12], truncation_error=1e-16)
    assert np.abs(K[:6, :6] - K_ref).max() < 1e-9
    monkeypatch.setenv("QK_BUILDER", "device")
    with pytest.raises(engine.QkError, match="injected"):
        M.build_kernel_matrix(SingleComm(), ans, X=X[:12], truncation_error=1e-16)
    monkeypatch.setattr(M._engine.Context, "build_share", real)
    # the bond cap: device and host builders truncate to the same bonds and (to rounding) the same states
    monkeypatch.setenv("QK_MAX_BOND", "6")
    Ks = {}
    for which in ("device", "host"):
        monkeypatch.setenv("QK_BUILDER", which)
        Ks[which] = M.build_kernel_matrix(SingleComm(), ans, X=X[:10], truncation_error=1e-16)
    assert np.abs(Ks["device"] - Ks["host"]).max() < 1e-8
    capped = Q.simulate(circuits[0], 1 - 1e-16, max_bond=6)
    assert capped.max_bond() == 6 and capped.fidelity < 1 - 1e-6 and Q.simulate(circuits[0], 1 - 1e-16).max_bond() > 6


@pytest.mark.parametrize("shape", [None, "2", "1"])
def test_bond_cap_on_the_block_path(gpu_ctx, monkeypatch, shape):
    """(In each workgroup shape of the builder: four 256-thread workgroups per CU -- the default at this cap --, two, one of 512.)
    A bond cap that bites where the factorisations run on the matrix cores (a gate's theta has 2 chi >= 48 columns): the device
    builder cut at chi = 32 against the host builder with the same cap (the chi of pytket-cutensornet's Config, ref
    gpu_backend/kernel_state_ansatz.py:141-144) -- same bonds, same fidelity product, the same states up to the rounding of the
    singular values at the cut; and the capped states are NOT the uncapped ones."""
    import qml_cutensornet_amd as Q
    from oracle import restatement as R

    if shape:
        monkeypatch.setenv("QK_BUILD_WGS", shape)
    n, chi = 18, 32
    X = R.synthetic_features(6, n, 11)
    ans = Q.KernelStateAnsatz(n, 4, 1.0, Q.entanglement_graph(n, 3))
    circuits = [ans.circuit_for_data(x) for x in X]
    host = [Q.simulate(c, 1 - 1e-16, max_bond=chi) for c in circuits]
    free = Q.simulate(circuits[0], 1 - 1e-16)
    assert free.max_bond() > chi and max(m.max_bond() for m in host) == chi and min(m.fidelity for m in host) < 1 - 1e-6
    dev, info = gpu_ctx.build_mps(circuits, max_bond=chi, truncate=True)
    assert not info["dropped"]
    for d, h in zip(dev, host):
        assert np.array_equal(d.bond_dims(), h.bond_dims())
        assert abs(d.fidelity - h.fidelity) < 1e-9 * max(1.0, h.fidelity)
        ov = abs(R.mps_inner(d.tensors, h.tensors)) ** 2 / (abs(R.mps_inner(d.tensors, d.tensors)) * abs(R.mps_inner(h.tensors, h.tensors)))
        assert abs(ov - 1) < 1e-9, ov


def _graded(rng, p, q, decades):
    u, _ = np.linalg.qr(rng.standard_normal((p, q)) + 1j * rng.standard_normal((p, q)))
    v, _ = np.linalg.qr(rng.standard_normal((q, q)) + 1j * rng.standard_normal((q, q)))
    return (u * 10.0 ** (-decades * np.arange(q) / max(1, q - 1))) @ v.conj().T


@pytest.mark.parametrize("kind,p,q,par", [("random", 64, 48, None), ("random", 130, 100, None), ("rank", 96, 64, 20), ("graded", 78, 66, 22), ("graded", 160, 128, 22),
                                          ("graded", 300, 256, 22), ("graded", 512, 272, 14)])
@pytest.mark.parametrize("wide", [False, True])
def test_preconditioned_block_factorisation_against_lapack(gpu_ctx, monkeypatch, kind, p, q, par, wide):
    """The builder's factorisation for matrices beyond its LDS working set (sorted columns, Gram-Schmidt R, block Jacobi of R^H on
    the f64 matrix cores, W = A V) against LAPACK, in both workgroup shapes: singular values of the part the truncation keeps to
    1e-13 of the largest (LAPACK's own accuracy is eps x that), W = A V, the kept part of the decomposition to 1e-12 of the norm,
    V orthonormal to 1e-11.  Graded matrices are what a gate's theta looks like."""
    if wide:
        monkeypatch.setenv("QK_BUILD_WGS", "1")
    rng = np.random.default_rng(p + q)
    if kind == "random":
        a = rng.standard_normal((p, q)) + 1j * rng.standard_normal((p, q))
    elif kind == "rank":
        a = (rng.standard_normal((p, par)) + 1j * rng.standard_normal((p, par))) @ (rng.standard_normal((par, q)) + 1j * rng.standard_normal((par, q)))
    else:
        a = _graded(rng, p, q, par)
    w, v, sig, order, sweeps, _ = gpu_ctx.debug_jacobi_precond(a)
    u_ref, s_ref, vh_ref = np.linalg.svd(a, full_matrices=False)
    tot = (s_ref ** 2).sum()
    keep = int((np.cumsum((s_ref ** 2)[::-1])[::-1] > 1e-16 * tot).sum())
    assert int((sig > 0).sum()) >= keep and sweeps <= 16
    sg = sig[order]
    wk, vk = w[:, order[:keep]], v[:, order[:keep]]
    assert np.abs(sg[:keep] - s_ref[:keep]).max() < 1e-13 * s_ref[0]  # (LAPACK's own values are good to eps x the largest one)
    assert np.abs(a @ vk - wk).max() < 1e-13 * s_ref[0]
    assert np.abs(wk @ vk.conj().T - (u_ref[:, :keep] * s_ref[:keep]) @ vh_ref[:keep]).max() < 1e-12 * s_ref[0]
    assert np.abs(vk.conj().T @ vk - np.eye(keep)).max() < 1e-11


def test_device_builder_at_bonds_up_to_256(gpu_ctx):
    """The heaviest states of the 60-qubit x 6-layer headline set (bonds to 248, thetas of 500 x 500 through the preconditioned block
    factorisation) and a 28-qubit x 7-layer set: device-built against host-built -- |<dev|host>|^2 = 1 to 1e-10, the same bonds,
    the same fidelity."""
    import qml_cutensornet_amd as Q
    from oracle import restatement as R
    from qml_cutensornet_amd.data import synthetic_features

    for n, reps, X, pick in ((28, 7, R.synthetic_features(6, 28, 11), 4), (60, 6, synthetic_features(500, 60, 5), 3)):
        ans = Q.KernelStateAnsatz(n, reps, 1.0, Q.entanglement_graph(n, 2))
        circs = [ans.circuit_for_data(x) for x in X]
        if n == 60:
            w = np.array([float((np.sin(np.pi * np.asarray(c.alpha)[np.asarray(c.op) == 2]) ** 2).sum()) for c in circs])
            circs = [circs[i] for i in np.argsort(-w)[:pick]]
        else:
            circs = circs[:pick]
        dev, info = gpu_ctx.build_mps(circs, max_bond=320)
        host = [Q.simulate(c, 1 - 1e-16) for c in circs]
        assert max(m.max_bond() for m in host) > (150 if n == 28 else 200)
        with gpu_ctx.upload(dev) as xs, gpu_ctx.upload(host) as ys:
            z = np.abs(np.diag(gpu_ctx.overlaps(xs, ys))) ** 2
        assert np.abs(z - 1).max() < 1e-10
        for a, b in zip(dev, host):
            assert np.array_equal(a.bond_dims(), b.bond_dims())
            assert abs(a.fidelity - b.fidelity) < 1e-12
