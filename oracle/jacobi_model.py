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


def _rotation(al, be, g):
    """The one-sided Jacobi rotation for a column pair with Gram entries al = <a1,a1>, be = <a2,a2>, g = <a1,a2>:
    (c, s1, s2) with a1' = c a1 + s1 a2, a2' = s2 a1 + c a2 (qk_build.hip: jacobi_orth)."""
    iga = 1.0 / abs(g)
    zeta = 0.5 * (be - al) * iga
    t = np.copysign(1.0, zeta) / (abs(zeta) + np.sqrt(1.0 + zeta * zeta))
    c = 1.0 / np.sqrt(1.0 + t * t)
    s = c * t
    ph = g * iga
    return c, -s * np.conj(ph), s * ph


def jacobi_block(a: np.ndarray, nb_cols: int = 8, inner: int = 1, intra_once: bool = True, deflate: float = 0.0, stats: dict | None = None, floor: float = 1e-3):
    """The BLOCK one-sided Jacobi of the device builder (csrc/qk_build.hip: jacobi_block) restated sequentially: columns in
    blocks of `nb_cols`; a visit of a block pair forms the 2 nb_cols x 2 nb_cols Gram matrix G = P^H P of its panel (on the
    device: f64 matrix cores), runs `inner` cyclic sweeps of two-sided Jacobi rotations on G -- the same rotation and the
    same test as the scalar kernel, taken from G's entries -- accumulating them in J, and applies P <- P J, V <- V J (matrix
    cores again).  Block pairs in round-robin order.  Returns (W, V, column norms, sweeps) like `jacobi`."""
    a = np.array(a, dtype=np.complex128)
    p, q = a.shape
    order = np.argsort(-(np.abs(a) ** 2).sum(0), kind="stable")
    a = a[:, order]
    v = np.eye(q, dtype=np.complex128)[:, order]
    frob = float((np.abs(a) ** 2).sum())
    tol2 = 1e-29 * max(p, 10)
    w = 2 * nb_cols
    qpad = -(-q // w) * w
    a = np.concatenate([a, np.zeros((p, qpad - q), dtype=np.complex128)], axis=1)
    v = np.concatenate([np.concatenate([v, np.zeros((q, qpad - q))], axis=1), np.concatenate([np.zeros((qpad - q, q)), np.eye(qpad - q)], axis=1)], axis=0).astype(np.complex128)
    nb = qpad // nb_cols
    wr, whalf = w - 1, w // 2
    sweeps, visits = 0, 0
    for sweeps in range(1, MAX_SWEEPS + 1):
        rotated, worst = False, 0.0
        # DEFLATION (deflate = the share of ||A||_F^2 that may be frozen, e.g. 1e-3 x the truncation budget): the trailing
        # blocks whose columns together hold less than that leave the sweeps.  A frozen set never gains weight (a rotation
        # moves weight to the longer column) and is truncated afterwards, so its residual inner products with the active
        # columns -- of second order in its norm -- do not matter.  Half of the columns of a gate's theta are such noise.
        if deflate > 0.0 and sweeps > 1:
            n2 = (np.abs(a[:, : nb * nb_cols]) ** 2).sum(0)
            while nb > 2 and n2[(nb - 2) * nb_cols :].sum() <= deflate * frob:
                nb -= 2
        nr, half = nb - 1, nb // 2
        for r in range(nr):
            for k in range(half):
                b1, b2 = (nr, r) if k == 0 else ((r + k) % nr, (r - k) % nr)
                if b1 > b2:
                    b1, b2 = b2, b1
                cols = np.r_[b1 * nb_cols : (b1 + 1) * nb_cols, b2 * nb_cols : (b2 + 1) * nb_cols]
                pnl = a[:, cols]
                g = pnl.conj().T @ pnl
                jm = np.eye(w, dtype=np.complex128)
                any_rot = False
                # the pairs of a visit: in the first round of a sweep every pair of the panel's 16 columns (15 steps of 8
                # disjoint pairs), afterwards only the 64 CROSS pairs (8 steps: column i of the first block with column
                # (i + t) mod 8 of the second) -- every column pair of the matrix then meets exactly once per sweep
                if r == 0 or not intra_once:
                    steps = [[(wr, rr) if kk == 0 else ((rr + kk) % wr, (rr - kk) % wr) for kk in range(whalf)] for rr in range(wr)]
                else:
                    steps = [[(i, nb_cols + (i + t) % nb_cols) for i in range(nb_cols)] for t in range(nb_cols)]
                for _ in range(inner):
                    for step in steps:
                        for c1, c2 in step:
                            if c1 > c2:
                                c1, c2 = c2, c1
                            al, be, gg = g[c1, c1].real, g[c2, c2].real, g[c1, c2]
                            g2 = abs(gg) ** 2
                            scale2 = max(al, be) * max(min(al, be), floor * frob)
                            if not g2 > tol2 * scale2:
                                continue
                            worst = max(worst, g2 / scale2)
                            c, s1, s2 = _rotation(al, be, gg)
                            x, y = g[:, c1].copy(), g[:, c2].copy()
                            g[:, c1], g[:, c2] = c * x + s1 * y, s2 * x + c * y
                            x, y = g[c1, :].copy(), g[c2, :].copy()
                            g[c1, :], g[c2, :] = c * x + np.conj(s1) * y, np.conj(s2) * x + c * y
                            x, y = jm[:, c1].copy(), jm[:, c2].copy()
                            jm[:, c1], jm[:, c2] = c * x + s1 * y, s2 * x + c * y
                            any_rot = True
                visits += 1
                if any_rot:
                    a[:, cols] = pnl @ jm
                    v[:, cols] = v[:, cols] @ jm
                    rotated = True
        if not rotated or worst <= 1e-20:
            break
    else:
        raise RuntimeError(f"block Jacobi: no convergence in {MAX_SWEEPS} sweeps on a {p} x {q} matrix")
    a, v = a[:, :q], v[:q, :q]
    if stats is not None:
        stats["visits"], stats["active_blocks"] = visits, nb
    return a, v, np.sqrt((np.abs(a) ** 2).sum(0)), sweeps


def mgs_r(a: np.ndarray) -> np.ndarray:
    """The triangular factor R of A = Q R by modified Gram-Schmidt, column by column (the device applies the same
    projections panel-wise).  As backward stable for R as Householder QR; Q is not needed and not kept."""
    a = np.array(a, dtype=np.complex128)
    p, q = a.shape
    r = np.zeros((q, q), dtype=np.complex128)
    for k in range(q):
        nrm = np.sqrt(np.vdot(a[:, k], a[:, k]).real)
        r[k, k] = nrm
        if nrm > 0.0:
            a[:, k] /= nrm
            if k + 1 < q:
                r[k, k + 1 :] = a[:, k].conj() @ a[:, k + 1 :]
                a[:, k + 1 :] -= np.outer(a[:, k], r[k, k + 1 :])
    return r


def jacobi_precond(a: np.ndarray, cut: float = 1e-22, stats: dict | None = None, floor: float = 1e-22):
    """The preconditioned factorisation of the device builder (csrc/qk_build.hip: jacobi_precond) -- Drmac / Veselic: a
    gate's theta is a GRADED matrix (singular values falling by twenty orders of magnitude), on which plain one-sided Jacobi
    needs 13-17 sweeps, peeling about a decade and a half per sweep.  So: (1) columns sorted by decreasing norm; (2) the
    triangular factor R of the sorted matrix (Gram-Schmidt; Q is never needed); (3) rows of R whose squared norm is below
    `cut` x ||A||_F^2 are dropped -- six orders of magnitude below the truncation budget; (4) block Jacobi on
    L = R[:r]^H (q x r; 5-6 sweeps: L's columns are nearly orthogonal already), WITHOUT accumulating rotations: (5) the right
    singular vectors of A are the normalised columns of L V_L, and W = A V.  Returns (W, V, sig, sweeps) with W = A V
    (p x r), V (q x r) in the ORIGINAL column order of A."""
    a = np.array(a, dtype=np.complex128)
    p, q = a.shape
    order = np.argsort(-(np.abs(a) ** 2).sum(0), kind="stable")
    frob = float((np.abs(a) ** 2).sum())
    r_ = mgs_r(a[:, order])
    rows2 = (np.abs(r_) ** 2).sum(1)
    above = np.nonzero(rows2 > cut * frob)[0]
    rk = int(above[-1]) + 1 if above.size else 1
    wl, _, sig, sweeps = jacobi_block(r_[:rk].conj().T, stats=stats, floor=floor)
    vs = np.zeros((q, rk), dtype=np.complex128)
    nz = sig > 0
    vs[:, nz] = wl[:, nz] / sig[nz]
    v = np.zeros((q, rk), dtype=np.complex128)
    v[order] = vs  # row `pos` of the sorted matrix is column order[pos] of A
    if stats is not None:
        stats["rank"] = rk
    return a @ v, v, sig, sweeps


def svd_precond_model(a: np.ndarray, **_):
    """Drop-in for scipy.linalg.svd(a, full_matrices=False) through jacobi_precond on the smaller side."""
    m, n = a.shape
    w, v, sig, _ = jacobi_precond(a if n <= m else a.T)
    o = np.argsort(-sig, kind="stable")
    s = sig[o]
    wn = w[:, o] / np.where(s > 0, s, 1.0)
    k = min(m, n)
    if len(s) < k:  # the dropped part: zero singular values (its weight is below 1e-24 of the total)
        s = np.concatenate([s, np.zeros(k - len(s))])
        wn = np.concatenate([wn, np.zeros((wn.shape[0], k - wn.shape[1]))], axis=1)
        vo = np.concatenate([v[:, o], np.zeros((v.shape[0], k - v.shape[1]))], axis=1)
    else:
        vo = v[:, o]
    return (wn, s, vo.conj().T) if n <= m else (vo.conj(), s, wn.T)


def qr_precond_model(m_: np.ndarray, **_):
    """Drop-in for the centre moves' QR through jacobi_precond: M = (W/s)(s V^H), null columns dropped."""
    w, v, sig, _ = jacobi_precond(m_)
    o = np.argsort(-sig, kind="stable")
    s = sig[o]
    k = max(1, int((s > 1e-15 * s[0]).sum()))
    return w[:, o[:k]] / s[:k], s[:k, None] * v[:, o[:k]].conj().T


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
