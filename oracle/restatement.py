"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED by the reference.

Plain-numpy restatement of the algorithm behind the reference's Gram hot path.
Every function cites the reference lines it follows.  Short names:
  G = /root/reference/gpu_backend/kernel_state_ansatz.py
  C = /root/reference/cpu_backend/kernel_state_ansatz.py
  J = /root/reference/KernelPkg/src/KernelPkg.jl
  M = /root/reference/main.py

Nothing here is tuned; clarity over speed.  The product has its own,
separately written, circuit generator and MPS builder -- the tests compare the
two, so neither may import the other.
"""
from __future__ import annotations

import math

import numpy as np

# --------------------------------------------------------------------------
# circuit definition  (M:21-45, G:53-88, J:8-42)
# --------------------------------------------------------------------------


def entanglement_graph(nq: int, nn: int) -> list[tuple[int, int]]:
    """Edge list of the linear "nn nearest neighbours" map (M:21-45).

    For every distance d = 1..nn: first the pairs (i, i+d) whose left qubit is
    not the right end of an earlier pair of this distance, then the pairs whose
    left qubit is.  M:41 iterates a Python ``set``; ascending order is used
    here (all XXPhase gates commute, so the order is immaterial to the state).
    """
    edges: list[tuple[int, int]] = []
    for d in range(1, nn + 1):
        right_ends: set[int] = set()
        for i in range(nq):
            if i not in right_ends and i + d < nq:
                edges.append((i, i + d))
                right_ends.add(i + d)
        for i in sorted(right_ends):
            if i + d < nq:
                edges.append((i, i + d))
    return edges


def ansatz_gates(features, reps: int, gamma: float, edges, hadamard_init: bool = True):
    """Routed gate list of U(x) as (name, qubits, half_turns) tuples (G:53-88, C:113-131).

    H on every qubit; then ``reps`` times: Rz(alpha=(2/pi)*gamma*x_i) on every
    qubit (G:58-60) and XXPhase(alpha=gamma^2 (1-x_a)(1-x_b)) on every edge
    (G:62-66).  A non-adjacent XXPhase is routed by a SWAP chain q0 -> q1-1, the
    gate on (q1-1, q1), and the inverse chain (G:82-88).  Angles are pytket
    half-turns.
    """
    x = np.asarray(features, dtype=float)
    n = x.shape[0]
    gates = []
    if hadamard_init:
        gates += [("H", (q,), None) for q in range(n)]
    for _ in range(reps):
        for q in range(n):
            gates.append(("Rz", (q,), (2.0 / math.pi) * gamma * x[q]))
        for (a, b) in edges:
            alpha = gamma * gamma * (1.0 - x[a]) * (1.0 - x[b])
            lo, hi = min(a, b), max(a, b)
            for q in range(lo, hi - 1):
                gates.append(("SWAP", (q, q + 1), None))
            gates.append(("XXPhase", (hi - 1, hi), alpha))
            for q in reversed(range(lo, hi - 1)):
                gates.append(("SWAP", (q, q + 1), None))
    return gates


def gate_matrix(name: str, alpha) -> np.ndarray:
    """TKET-convention matrices, theta = pi*alpha/2 (J:8-42); H and SWAP standard (J:50,60)."""
    if name == "H":
        return np.array([[1, 1], [1, -1]], dtype=complex) / math.sqrt(2.0)
    if name == "SWAP":
        m = np.zeros((4, 4), dtype=complex)
        m[0, 0] = m[1, 2] = m[2, 1] = m[3, 3] = 1
        return m
    th = math.pi * alpha / 2.0
    c, s = math.cos(th), math.sin(th)
    if name == "Rz":
        return np.array([[np.exp(-1j * th), 0], [0, np.exp(1j * th)]])
    if name == "Rx":
        return np.array([[c, -1j * s], [-1j * s, c]])
    if name == "XXPhase":
        m = np.eye(4, dtype=complex) * c
        for r, cc in ((0, 3), (1, 2), (2, 1), (3, 0)):
            m[r, cc] = -1j * s
        return m
    if name == "ZZPhase":
        return np.diag([np.exp(-1j * th), np.exp(1j * th), np.exp(1j * th), np.exp(-1j * th)])
    raise ValueError(f"unknown gate {name}")


# --------------------------------------------------------------------------
# O1: exact state vector (independent of any MPS code)
# --------------------------------------------------------------------------


def statevector(n: int, gates) -> np.ndarray:
    """U|0...0> as a dense vector; qubit 0 is the most significant axis (J:68 initial state)."""
    psi = np.zeros((2,) * n, dtype=complex)
    psi[(0,) * n] = 1.0
    for name, qs, alpha in gates:
        m = gate_matrix(name, alpha)
        if len(qs) == 1:
            (q,) = qs
            psi = np.moveaxis(np.tensordot(m, psi, axes=([1], [q])), 0, q)
        else:
            q0, q1 = qs
            m4 = m.reshape(2, 2, 2, 2)  # [out0, out1, in0, in1]
            psi = np.moveaxis(np.tensordot(m4, psi, axes=([2, 3], [q0, q1])), [0, 1], [q0, q1])
    return psi.reshape(-1)


def gram_statevector(X, Y, reps, gamma, edges, hadamard_init=True) -> np.ndarray:
    """K[j, i] = |<psi(x_i)|psi(y_j)>|^2, rows = Y, cols = X (G:383-387, J:106)."""
    X = np.asarray(X, dtype=float)
    n = X.shape[1]
    sx = [statevector(n, ansatz_gates(x, reps, gamma, edges, hadamard_init)) for x in X]
    sy = sx if Y is None else [statevector(n, ansatz_gates(y, reps, gamma, edges, hadamard_init)) for y in np.asarray(Y, dtype=float)]
    K = np.empty((len(sy), len(sx)))
    for j, b in enumerate(sy):
        for i, a in enumerate(sx):
            K[j, i] = abs(np.vdot(a, b)) ** 2
    return K


# --------------------------------------------------------------------------
# O2: closed form for an empty entanglement map
# --------------------------------------------------------------------------


def gram_product_closed_form(X, Y, reps, gamma) -> np.ndarray:
    """With no XXPhase gates every qubit is H then Rz(theta=r*gamma*x):  K = prod_i cos^2(r gamma (x_i - y_i))."""
    X = np.asarray(X, dtype=float)
    Y = X if Y is None else np.asarray(Y, dtype=float)
    d = Y[:, None, :] - X[None, :, :]
    return np.prod(np.cos(reps * gamma * d) ** 2, axis=2)


# --------------------------------------------------------------------------
# MPS simulation (semantics of J:45-72 / G:221) -- naive, one SVD per 2-qubit gate
# --------------------------------------------------------------------------


def _truncate(s: np.ndarray, cutoff: float) -> int:
    """Number of singular values kept: discard the smallest ones while their
    summed squared weight stays <= cutoff * total (ITensors ``cutoff``, J:68;
    equivalently kept weight >= 1 - truncation_error, G:142).  The tail is
    summed from the smallest value up, so the test is well conditioned."""
    p = s * s
    total = float(p.sum())
    keep = len(s)
    acc = 0.0
    while keep > 1 and acc + float(p[keep - 1]) <= cutoff * total:
        acc += float(p[keep - 1])
        keep -= 1
    return keep


def mps_simulate(n: int, gates, cutoff: float = 1e-16):
    """MPS of U|0..0>: site tensors [chi_l, 2, chi_r], complex128.

    Keeps the orthogonality centre on the gate (QR moves) so that the SVD
    truncation is the optimal one, like ITensors ``apply`` (J:68).
    """
    A = [np.zeros((1, 2, 1), dtype=complex) for _ in range(n)]
    for t in A:
        t[0, 0, 0] = 1.0
    centre = 0

    def move_centre(to):
        nonlocal centre
        while centre < to:
            l, _, r = A[centre].shape
            q, rr = np.linalg.qr(A[centre].reshape(l * 2, r))
            A[centre] = q.reshape(l, 2, -1)
            A[centre + 1] = np.tensordot(rr, A[centre + 1], axes=(1, 0))
            centre += 1
        while centre > to:
            l, _, r = A[centre].shape
            q, rr = np.linalg.qr(A[centre].reshape(l, 2 * r).T)
            A[centre] = q.T.reshape(-1, 2, r)
            A[centre - 1] = np.tensordot(A[centre - 1], rr.T, axes=(2, 0))
            centre -= 1

    for name, qs, alpha in gates:
        m = gate_matrix(name, alpha)
        if len(qs) == 1:
            (q,) = qs
            A[q] = np.einsum("pq,lqr->lpr", m, A[q])
            continue
        k = qs[0]
        assert qs[1] == k + 1
        move_centre(k)
        theta = np.tensordot(A[k], A[k + 1], axes=(2, 0))  # l p q r
        theta = np.einsum("abpq,lpqr->labr", m.reshape(2, 2, 2, 2), theta)
        l, _, _, r = theta.shape
        u, s, vh = np.linalg.svd(theta.reshape(l * 2, 2 * r), full_matrices=False)
        keep = _truncate(s, cutoff)
        u, s, vh = u[:, :keep], s[:keep], vh[:keep]
        s = s / np.linalg.norm(s)
        A[k] = u.reshape(l, 2, keep)
        A[k + 1] = (s[:, None] * vh).reshape(keep, 2, r)
        centre = k + 1
    return A


# --------------------------------------------------------------------------
# the hot path: overlap sweep and Gram fill  (G:380-387, J:101-109)
# --------------------------------------------------------------------------


def mps_inner(x, y) -> complex:
    """<x|y>: E_0 = 1; E_{k+1}[R,r] = sum_{L,l,p} E_k[L,l] conj(A_k[L,p,R]) B_k[l,p,r]  (G:380; ITensors ``inner``, J:106)."""
    E = np.ones((1, 1), dtype=complex)
    for a, b in zip(x, y):
        T = np.tensordot(E, b, axes=(1, 0))  # L p r
        E = np.tensordot(a.conj(), T, axes=([0, 1], [0, 1]))  # R r
    return complex(E[0, 0])


def gram_from_mps(xs, ys=None) -> np.ndarray:
    """K[j, i] = |<x_i|y_j>|^2 (rows = Y, cols = X; G:383-387).  ``ys=None``: symmetric, lower triangle mirrored (G:390-395)."""
    if ys is None:
        n = len(xs)
        K = np.empty((n, n))
        for j in range(n):
            for i in range(j + 1):
                z = mps_inner(xs[i], xs[j])
                K[j, i] = K[i, j] = (z * z.conjugate()).real
        return K
    K = np.empty((len(ys), len(xs)))
    for j, y in enumerate(ys):
        for i, x in enumerate(xs):
            z = mps_inner(x, y)
            K[j, i] = (z * z.conjugate()).real
    return K


def synthetic_features(n_points: int, n_features: int, seed: int) -> np.ndarray:
    """Synthetic stand-in for M:130-143: standard normal, column-standardised, column MinMax to [0, 2]."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n_points, n_features))
    X = (X - X.mean(axis=0)) / X.std(axis=0)
    lo, hi = X.min(axis=0), X.max(axis=0)
    return 2.0 * (X - lo) / (hi - lo)
