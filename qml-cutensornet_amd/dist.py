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

    def allgather_array(self, arr: np.ndarray):
        """All-gather of equally sized host arrays as tensors (no pickling): the packed set images on the host route."""
        import torch

        mine = torch.from_numpy(np.ascontiguousarray(arr))
        parts = [torch.empty_like(mine) for _ in range(self.Get_size())]
        self._dist.all_gather(parts, mine, group=self.group)
        return [p.numpy() for p in parts]

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


def exchange_sets(comm, ctx, local, lo: int, total: int, force_collective: bool = False):
    """Every rank holds the whole data set after this: the replacement of the reference's ring of pickled MPS
    (/root/reference/gpu_backend/kernel_state_ansatz.py:341-352, 415-419).  ``local`` is the device set of the
    states [lo, lo + len(local)) this rank built (or ``None`` for an empty share).  The packed images are exchanged as
    flat buffers -- ONE ``all_gather_into_tensor`` over RCCL when torch.distributed runs on nccl (nothing crosses PCIe),
    the communicator's own ``allgather`` of the host images otherwise -- and assembled with ``qk_mps_set_from_packed``;
    no state is re-packed element by element.  Returns (MpsSet of all ``total`` states, seconds in the exchange)."""
    import time

    size = comm.Get_size()
    if size == 1 and not force_collective:  # (force_collective: the one-rank RCCL smoke test)
        if local is None or len(local) != total:
            raise RuntimeError("exchange_sets: a single rank must hold the whole set")
        return local, 0.0
    t0 = time.perf_counter()
    if local is None:
        n_loc, dims, offs = 0, np.zeros((0, 0), dtype=np.int32), np.zeros((0, 0), dtype=np.int64)
    else:
        n_loc, _, dims, offs = local.image()
    meta = comm_allgather(comm, (int(lo), int(n_loc), dims, offs)) if size > 1 else [(int(lo), int(n_loc), dims, offs)]  # a few KB per rank
    mx = max(1, max(m[1] for m in meta))
    n_sites = max(m[2].shape[1] for m in meta) - 1
    dims_all = np.zeros((total, n_sites + 1), dtype=np.int32)
    offs_all = np.zeros((total, n_sites), dtype=np.int64)
    for r, (lo_r, _, d_r, o_r) in enumerate(meta):
        if d_r.shape[0]:
            dims_all[lo_r : lo_r + d_r.shape[0]] = d_r
            offs_all[lo_r : lo_r + d_r.shape[0]] = o_r + r * mx
    if (dims_all[:, 0] != 1).any():
        raise RuntimeError("exchange_sets left holes; ranks disagree on the data set size")
    use_nccl = False
    try:
        import torch
        import torch.distributed as dist

        use_nccl = dist.is_initialized() and dist.get_world_size() == size and dist.get_backend() == "nccl"
    except ImportError:
        use_nccl = False
    if use_nccl:
        dev = torch.device("cuda", ctx.device_id)
        # torch.empty, not zeros: a fill would run on torch's current stream while copy_image copies on the context's own
        # (non-blocking) stream -- the two are not ordered, so zeros could land on top of the image.  Nothing addresses the
        # bytes behind the image (the offsets end inside it); copy_image returns after its copy has completed.
        send = torch.empty(mx, dtype=torch.float64, device=dev)
        torch.cuda.current_stream(dev).synchronize()  # nothing of torch's is pending on the buffer when the context's stream writes it
        if local is not None:
            local.copy_image(send.data_ptr(), mx)
        recv = torch.empty(size * mx, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(recv, send)
        torch.cuda.synchronize(dev)
        full = ctx.set_from_packed(dims_all, offs_all, recv.data_ptr(), size * mx)
        del recv, send
    else:  # host communicator (mpi4py, gloo): the packed HOST images travel; one upload of the joined buffer
        mine = np.zeros(mx, dtype=np.float64)
        if local is not None:
            local.copy_image(mine.ctypes.data, mx)
        if size == 1:
            parts = [mine]
        elif hasattr(comm, "allgather_array") and comm.backend() != "nccl":
            parts = comm.allgather_array(mine)
        elif hasattr(comm, "Allgather"):  # mpi4py: the buffer interface (no pickling; counts beyond 2^31 bytes are fine as doubles)
            joined_buf = np.empty(size * mx, dtype=np.float64)
            comm.Allgather(mine, joined_buf)
            parts = [joined_buf]
        else:
            parts = comm_allgather(comm, mine)
        joined = parts[0] if len(parts) == 1 else np.concatenate(parts)  # (one part: already the joined buffer -- no second copy of a multi-GB image)
        full = ctx.set_from_packed(dims_all, offs_all, joined.ctypes.data, joined.shape[0])
    return full, time.perf_counter() - t0
