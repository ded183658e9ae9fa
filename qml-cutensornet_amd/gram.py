"""Sharded Gram fill: the device-side replacement of the reference's tile loop, round robin
and final reduce (/root/reference/gpu_backend/kernel_state_ansatz.py:324-428).

Every rank holds all MPS on its GPU (they are small and read-only), computes its share of the
pair list in ONE persistent kernel launch, and the shares are joined by a single RCCL
all-gather of the packed values (the reference's ``reduce(SUM)`` of mostly-zero matrices,
ref :428, only ever gathers).  The dense matrix is then filled on the device by a scatter
kernel that also writes the mirrored half (ref :390-395).
"""
from __future__ import annotations

import numpy as np
import torch

from . import engine


class GramJob:
    """Reusable plan + buffers for one (xset, yset) Gram on this rank's GPU."""

    def __init__(self, ctx: engine.Context, xset: engine.MpsSet, yset: engine.MpsSet | None = None,
                 world_size: int = 1, rank: int = 0, group=None, block: int | None = None, force_collective: bool = False):
        self.ctx, self.xset, self.yset = ctx, xset, yset
        self.world, self.rank, self.group = int(world_size), int(rank), group
        # force_collective: run the all-gathers even in a one-rank group (the RCCL smoke test: tests/test_gpu_nccl.py)
        self.collective = self.world > 1 or bool(force_collective)
        self.symmetric = yset is None
        import os

        if block is None:
            block = int(os.environ.get("QK_PLAN_BLOCK", "0"))  # 0 = no locality tiles: one global cost order
        self.nx = len(xset)
        self.ny = self.nx if self.symmetric else len(yset)
        ydims = None if self.symmetric else yset.dims
        quads = os.environ.get("QK_QUADS", "0") == "1"  # 2x2 pair blocks per workgroup (QK_PLAN_QUADS): tools/ with the lab library only
        self.plan = engine.Plan(xset.dims, ydims, self.world, self.rank, block, quads)
        self.maxp = max(1, self.plan.max_pairs_per_rank)
        dev = torch.device("cuda", ctx.device_id)
        self.dev = dev
        # pair table of ALL ranks, padded with -1: each rank contributes the pairs of its own plan (one small all-gather at
        # set-up; the plans are deterministic, but building P plans on every rank costs P times the planning time)
        pr = self.plan.pairs()
        mine = np.full((self.maxp, 2), -1, dtype=np.int32)
        mine[: pr.shape[0]] = pr
        self.work = {self.rank: self.plan.stats()}  # algorithmic work of THIS rank's share
        my_pairs = torch.from_numpy(mine).to(dev)
        if self.collective:
            import torch.distributed as dist

            self.all_pairs = torch.empty((self.world * self.maxp, 2), dtype=torch.int32, device=dev)
            if dist.get_backend(self.group) == "nccl":
                dist.all_gather_into_tensor(self.all_pairs, my_pairs, group=self.group)
            else:
                dist.all_gather(list(self.all_pairs.view(self.world, self.maxp, 2).unbind(0)), my_pairs, group=self.group)
        else:
            self.all_pairs = my_pairs
        self.my_vals = torch.zeros(self.maxp, dtype=torch.float64, device=dev)
        self.all_vals = torch.zeros(self.world * self.maxp, dtype=torch.float64, device=dev) if self.collective else self.my_vals
        self.K = torch.zeros((self.ny, self.nx), dtype=torch.float64, device=dev)
        self._ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if self.collective else None
        self._recorded = False

    def allgather_ms(self) -> float:
        """Time of the last enqueue's all-gather of the packed values on its stream (it starts when this rank's sweep has ended, so it
        includes waiting for the slowest rank); 0 without a collective.  Synchronises on the second event."""
        if self._ev is None or not self._recorded:
            return 0.0
        self._ev[1].synchronize()
        return float(self._ev[0].elapsed_time(self._ev[1]))

    def enqueue(self) -> torch.Tensor:
        """Enqueue one full Gram on torch's current stream; returns the device matrix (async)."""
        stream = torch.cuda.current_stream(self.dev)
        self.ctx.set_stream(stream.cuda_stream)
        self.ctx.gram_values(self.xset, self.yset, self.plan, self.my_vals.data_ptr())
        if self.collective:
            import torch.distributed as dist

            self._ev[0].record(stream)
            if dist.get_backend(self.group) == "nccl":  # RCCL over xGMI: the one collective of the path
                dist.all_gather_into_tensor(self.all_vals, self.my_vals, group=self.group)
            else:  # gloo (tests, several ranks on one GPU): same result through the list form
                parts = list(self.all_vals.view(self.world, self.maxp).unbind(0))
                dist.all_gather(parts, self.my_vals, group=self.group)
            self._ev[1].record(stream)
            self._recorded = True
        self.ctx.scatter(self.all_pairs.data_ptr(), self.all_vals.data_ptr(), self.all_pairs.shape[0],
                         self.K.data_ptr(), self.nx, self.symmetric)
        return self.K

    def run(self) -> np.ndarray:
        """One full Gram, synchronously, as a host array (rows = Y, cols = X)."""
        return self.enqueue().cpu().numpy()

    def close(self):
        self.plan.close()
