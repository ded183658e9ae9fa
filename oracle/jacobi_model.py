"""TEST INFRASTRUCTURE -- a numpy model of the device MPS builder's factorisation (csrc/qk_build.hip: jacobi_orth).

Not part of the product: only tests/ import it.  It restates, sequentially, what the HIP kernel does in parallel -- the
round-robin pair order, the rotation test with its absolute floor, the column pre-sort, the early exit after second-order
small rotations -- so that the *algorithm* (not the kernel) can be checked on the CPU against LAPACK: convergence on the
numerically rank-deficient matrices a circuit produces, singular values, and the states a builder gets when this
factorisation replaces gesdd / QR (the reference's simulate(..., MPSxGate, ...), gpu_backend/kernel_state_ansatz.py:221).
"""
from __future__ import annotations

import numpy as np

MAX_SWEEPS = 40


def jacobi(a: np.ndarray):
    """One-sided Jacobi of a (p x q): returns (W = A V with mutually orthogonal columns, V unitary, column norms, sweeps)."""
    a = np.array(a, dtype=np.complex128)
    p, q = a.shape
    order = np.argsort(-(np.abs(a) ** 2).sum(0), kind="stable")  # de Rijk: start from decreasing norms
    a = a[:, order]
    v = np.eye(q, dtype=np.complex128)[:, order]
    frob = float((np.abs(a) ** 2).sum())
    tol2 = 1e-29 * max(p, 10)
    sweeps = 0
    if q >= 2:
        qe = q + (q & 1)
        half, nr = qe // 2, qe - 1
        for sweeps in range(1, MAX_SWEEPS + 1):
            rotated, worst = False, 0.0
            for r in range(nr):
                for k in range(half):
                    c1, c2 = (nr, r) if k == 0 else ((r + k) % nr, (r - k) % nr)
                    if c1 >= q or c2 >= q:
                        continue
                    if c1 > c2:
                        c1, c2 = c2, c1
                    x, y = a[:, c1].copy(), a[:, c2].copy()
                    al, be, g = np.vdot(x, x).real, np.vdot(y, y).real, np.vdot(x, y)
                    g2 = abs(g) ** 2
                    scale2 = max(al, be) * max(min(al, be), 1e-3 * frob)
                    if g2 > tol2 * scale2:
                        worst = max(worst, g2 / scale2)
                        iga = 1.0 / np.sqrt(g2)
                        zeta = 0.5 * (be - al) * iga
                        t = np.copysign(1.0, zeta) / (abs(zeta) + np.sqrt(1.0 + zeta * zeta))
                        c = 1.0 / np.sqrt(1.0 + t * t)
                        s = c * t
                        ph = g * iga
                        a[:, c1], a[:, c2] = c * x - s * np.conj(ph) * y, s * ph * x + c * y
                        vx, vy = v[:, c1].copy(), v[:, c2].copy()
                        v[:, c1], v[:, c2] = c * vx - s * np.conj(ph) * vy, s * ph * vx + c * vy
                        rotated = True
            if not rotated or worst <= 1e-20:
                break
        else:
            raise RuntimeError(f"no convergence in {MAX_SWEEPS} sweeps on a {p} x {q} matrix")
    return a, v, np.sqrt((np.abs(a) ** 2).sum(0)), sweeps


def svd_model(a: np.ndarray, **_):
    """Drop-in for scipy.linalg.svd(a, full_matrices=False): (U, s, Vh) through the Jacobi on the smaller side."""
    m, n = a.shape
    w, v, sig, _ = jacobi(a if n <= m else a.T)
    o = np.argsort(-sig, kind="stable")
    s = sig[o]
    wn = w[:, o] / np.where(s > 0, s, 1.0)
    return (wn, s, v[:, o].conj().T) if n <= m else (v[:, o].conj(), s, wn.T)


def qr_model(m_: np.ndarray, **_):
    """Drop-in for scipy.linalg.qr(m, mode="economic") as the centre moves use it: M = (W/s)(s V^H), null columns dropped."""
    w, v, sig, _ = jacobi(m_)
    o = np.argsort(-sig, kind="stable")
    s = sig[o]
    k = max(1, int((s > 1e-15 * s[0]).sum()))
    return w[:, o[:k]] / s[:k], s[:k, None] * v[:, o[:k]].conj().T
