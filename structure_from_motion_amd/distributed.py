"""Multi-GPU host helpers: one process per GPU, point-partitioned, RCCL over xGMI.

Each rank owns a contiguous range of points holding an equal share of the observations
(the same rule libsfmba applies inside sfmba_set_problem when options.nranks > 1).  The only
data exchanged during a solve are per-camera partial sums and scalars, all-reduced by RCCL
inside the library; torch.distributed is used here only to hand the RCCL unique id from rank 0
to the other ranks and for the timing barrier of the benchmark.
"""
from __future__ import annotations

import numpy as np


def point_partition(point_indices, n_points, rank, nranks):
    """[lo, hi) point range of `rank`: boundaries at the prefix sums of observations per point
    closest above rank/nranks of the total."""
    counts = np.bincount(np.asarray(point_indices, dtype=np.int64), minlength=n_points)
    pre = np.concatenate([[0], np.cumsum(counts)])
    n_obs = int(pre[-1])

    def boundary(r):
        if r <= 0:
            return 0
        if r >= nranks:
            return int(n_points)
        target = int(float(n_obs) * r / nranks)
        return int(np.searchsorted(pre, target, side="left"))

    lo = min(boundary(rank), n_points)
    hi = min(max(boundary(rank + 1), lo), n_points)
    return lo, hi


def shard_problem(prob, lo, hi):
    """The sub-problem of points [lo, hi): all cameras, the owned points and their observations
    (point indices re-based to the shard)."""
    pi = np.asarray(prob["point_indices"])
    keep = (pi >= lo) & (pi < hi)
    out = dict(prob)
    out["points_3d"] = np.asarray(prob["points_3d"])[lo:hi]
    out["camera_indices"] = np.asarray(prob["camera_indices"])[keep]
    out["point_indices"] = pi[keep] - lo
    out["points_2d"] = np.asarray(prob["points_2d"])[keep]
    out["n_points"] = hi - lo
    out["n_obs"] = int(keep.sum())
    return out


def init_engine_comm(engine, rank, nranks):
    """Create the RCCL communicator of `engine` (options.rank / options.nranks must already be
    set): rank 0 draws the unique id, torch.distributed broadcasts it."""
    import torch
    import torch.distributed as dist

    if nranks <= 1:
        return
    if rank == 0:
        uid = bytearray(type(engine).comm_unique_id())
    else:
        uid = bytearray(128)
    t = torch.tensor(list(uid), dtype=torch.uint8)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.broadcast(t, src=0)
    engine.comm_init(bytes(t.cpu().tolist()))
