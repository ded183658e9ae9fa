"""Feature-map ansatz as a compiled gate *program* (no pytket, no sympy).

Host-side mirror of the reference's ``KernelStateAnsatz``
(/root/reference/gpu_backend/kernel_state_ansatz.py:16-103).  The reference keeps
a symbolic pytket circuit and substitutes sympy symbols per data point; here the
circuit is compiled once into flat integer/float arrays and binding a data point
is one vectorised numpy expression.  The gate semantics are those of the
reference:

* H on every qubit when ``hadamard_init``                      (ref :53-55)
* per layer: Rz with half-turn exponent (2/pi)*gamma*f_i       (ref :58-60)
*            XXPhase with exponent gamma^2 (1-f_a)(1-f_b)      (ref :62-66)
* a non-adjacent XXPhase is routed eagerly: SWAP chain up, the gate on the last
  adjacent pair, SWAP chain down                               (ref :68-88)

Angles are pytket half-turns (theta = pi*alpha/2), the convention spelled out in
/root/reference/KernelPkg/src/KernelPkg.jl:8-32.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

OP_H, OP_RZ, OP_XX, OP_SWAP = 0, 1, 2, 3
_OP_NAMES = {OP_H: "H", OP_RZ: "Rz", OP_XX: "XXPhase", OP_SWAP: "SWAP"}


def entanglement_graph(nq: int, nn: int) -> list[tuple[int, int]]:
    """Linear entanglement map with interactions up to distance ``nn``.

    Same edge set and layer structure as /root/reference/main.py:21-45: for each
    distance, a first layer of disjoint pairs, then the pairs that start on a
    right end of the first layer.
    """
    edges = []
    for dist in range(1, nn + 1):
        taken = np.zeros(nq + dist, dtype=bool)
        second = []
        for left in range(nq - dist):
            if taken[left]:
                second.append(left)
            else:
                edges.append((left, left + dist))
                taken[left + dist] = True
        # qubits that were right ends of the first layer and still have a partner
        edges.extend((left, left + dist) for left in second)
    return edges


@dataclass(frozen=True)
class BoundCircuit:
    """A gate program with numeric angles: what ``circuit_for_data`` returns."""

    n_qubits: int
    op: np.ndarray  # int8   [n_gates]
    q0: np.ndarray  # int32  [n_gates]   (for 2-qubit gates the pair is (q0, q0+1))
    alpha: np.ndarray  # float64 [n_gates] half-turns (0 where unused)

    @property
    def n_gates(self) -> int:
        return int(self.op.shape[0])

    def as_tuples(self):
        """(name, qubits, params) triples in the shape of the reference's CPU gate list
        (/root/reference/cpu_backend/kernel_state_ansatz.py:113-131)."""
        out = []
        for o, q, a in zip(self.op.tolist(), self.q0.tolist(), self.alpha.tolist()):
            if o in (OP_H,):
                out.append((_OP_NAMES[o], [q], []))
            elif o == OP_RZ:
                out.append((_OP_NAMES[o], [q], [a]))
            elif o == OP_XX:
                out.append((_OP_NAMES[o], [q, q + 1], [a]))
            else:
                out.append((_OP_NAMES[o], [q, q + 1], []))
        return out


class GateProgram:
    """The symbolic (unbound) circuit; stands in for the reference's ``ansatz_circ``."""

    def __init__(self, n_qubits, op, q0, fa, fb, scale):
        self.n_qubits = int(n_qubits)
        self.op, self.q0, self.fa, self.fb, self.scale = op, q0, fa, fb, scale

    @property
    def n_gates(self) -> int:
        return int(self.op.shape[0])

    def bind(self, x: np.ndarray) -> BoundCircuit:
        alpha = np.zeros(self.op.shape[0])
        rz = self.op == OP_RZ
        xx = self.op == OP_XX
        alpha[rz] = self.scale[rz] * x[self.fa[rz]]
        alpha[xx] = self.scale[xx] * (1.0 - x[self.fa[xx]]) * (1.0 - x[self.fb[xx]])
        return BoundCircuit(self.n_qubits, self.op, self.q0, alpha)


class KernelStateAnsatz:
    """Drop-in for the reference class of the same name (ref :16-103).

    Attributes kept from the reference: ``ansatz_circ`` (needs ``.n_qubits``,
    used at ref :147) and ``feature_symbol_list`` (names ``f_0 .. f_{n-1}``).
    """

    def __init__(self, num_qubits, reps, gamma, entanglement_map, hadamard_init=True):
        n = int(num_qubits)
        self.num_qubits, self.reps, self.gamma = n, int(reps), float(gamma)
        self.entanglement_map = [(int(a), int(b)) for a, b in entanglement_map]
        self.hadamard_init = bool(hadamard_init)
        self.feature_symbol_list = [f"f_{i}" for i in range(n)]
        self.one_q_symbol_list = []
        self.two_q_symbol_list = []

        op, q0, fa, fb, sc = [], [], [], [], []

        def emit(o, q, a=0, b=0, s=0.0):
            op.append(o), q0.append(q), fa.append(a), fb.append(b), sc.append(s)

        if self.hadamard_init:
            for q in range(n):
                emit(OP_H, q)
        rz_scale = (2.0 / np.pi) * self.gamma
        xx_scale = self.gamma * self.gamma
        for _ in range(self.reps):
            for q in range(n):
                emit(OP_RZ, q, q, 0, rz_scale)
            for a, b in self.entanglement_map:
                if not (0 <= a < n and 0 <= b < n) or a == b:
                    raise ValueError(f"bad entanglement pair ({a}, {b}) for {n} qubits")
                lo, hi = (a, b) if a < b else (b, a)
                for q in range(lo, hi - 1):  # bring qubit `lo` next to `hi`
                    emit(OP_SWAP, q)
                emit(OP_XX, hi - 1, a, b, xx_scale)
                for q in range(hi - 2, lo - 1, -1):  # and back
                    emit(OP_SWAP, q)
        self.ansatz_circ = GateProgram(
            n,
            np.asarray(op, dtype=np.int8),
            np.asarray(q0, dtype=np.int32),
            np.asarray(fa, dtype=np.int32),
            np.asarray(fb, dtype=np.int32),
            np.asarray(sc, dtype=np.float64),
        )

    def circuit_for_data(self, feature_values) -> BoundCircuit:
        """Bind one data point.  ``RuntimeError`` on a length mismatch, as ref :96-97."""
        if len(feature_values) != len(self.feature_symbol_list):
            raise RuntimeError("The number of values must match the number of symbols.")
        return self.ansatz_circ.bind(np.asarray(feature_values, dtype=np.float64))
