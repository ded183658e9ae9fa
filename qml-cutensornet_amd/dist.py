"""Communicator plumbing for the Gram path.

The reference hands an ``mpi4py`` communicator to ``build_kernel_matrix`` and only ever uses
``Get_rank``/``Get_size`` plus pickled point-to-point calls and one ``reduce``
(/root/reference/gpu_backend/kernel_state_ansatz.py:151,348,352,419,428).  Here every rank
keeps all MPS, so the only exchanges left are (1) an all-gather of the MPS built by each rank
(before the hot path) and (2) ONE all-gather of the packed Gram values (the hot path's single
collective).  ``TorchComm`` gives a ``torch.distributed`` process group (RCCL on GPUs, gloo on
CPUs) the few mpi4py-style methods the module surface needs; a real ``mpi4py`` communicator
works as is; ``SingleComm`` is the one-process case.
"""
from __future__ import annotations

import numpy as np


class SingleComm:
    def Get_rank(self) -> int:
        return 0

    def Get_size(self) -> int:
        return 1

    def allgather(self, obj):
        return [obj]

    def barrier(self):
        pass


class TorchComm:
    """mpi4py-flavoured view of an initialised ``torch.distributed`` group."""

    def __init__(self, group=None):
        import torch.distributed as dist

        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._dist, self.group = dist, group

    def Get_rank(self) -> int:
        return self._dist.get_rank(self.group)

    def Get_size(self) -> int:
        return self._dist.get_world_size(self.group)

    def allgather(self, obj):
        out = [None] * self.Get_size()
        self._dist.all_gather_object(out, obj, group=self.group)
        return out

    def barrier(self):
        self._dist.barrier(group=self.group)

    def backend(self) -> str:
        return self._dist.get_backend(self.group)


def comm_allgather(comm, obj):
    """``allgather`` of a picklable object on whatever communicator the caller passed."""
    if comm.Get_size() == 1:
        return [obj]
    if hasattr(comm, "allgather"):
        return list(comm.allgather(obj))
    raise TypeError("the communicator needs an `allgather(obj)` method (mpi4py, TorchComm) when size > 1")


def assemble_gram(ny: int, nx: int, pairs_by_rank, values_by_rank, symmetric: bool) -> np.ndarray:
    """Host-side join of the per-rank shares into the dense matrix: K[j, i] = v for pair (i, j),
    mirrored when symmetric (ref :387, :390-395).  Used when the all-gather ran through a host
    communicator; on the GPU path the same scatter is the ``qk_scatter`` kernel."""
    K = np.zeros((ny, nx), dtype=np.float64)
    for pairs, vals in zip(pairs_by_rank, values_by_rank):
        pairs = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
        vals = np.asarray(vals, dtype=np.float64)[: pairs.shape[0]]
        K[pairs[:, 1], pairs[:, 0]] = vals
        if symmetric:
            K[pairs[:, 0], pairs[:, 1]] = vals
    return K
