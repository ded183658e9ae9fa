"""MI355X-native engine for the quantum-kernel Gram hot path of mmetcalf14/qml-cutensornet.

Scope: K[j, i] = |<psi(x_i)|psi(y_j)>|^2 from matrix-product
states, i.e. the reference's ``build_kernel_matrix`` hot region and ``MPS.vdot``
(/root/reference/gpu_backend/kernel_state_ansatz.py:324-405, :380), as hand-written
gfx950 HIP kernels behind a C ABI (``include/qkgram.h``).

The directory is named ``qml-cutensornet_amd`` (not an identifier); import it as
``qml_cutensornet_amd`` through the one-file shim at the repository root.
"""
from .ansatz import KernelStateAnsatz, entanglement_graph  # noqa: F401
from .mps import MPS, random_mps, simulate  # noqa: F401

__all__ = ["KernelStateAnsatz", "entanglement_graph", "MPS", "simulate", "random_mps"]
