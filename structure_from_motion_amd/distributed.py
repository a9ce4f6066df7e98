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


def number_points_by_camera(prob):
    """The same problem with its points numbered by their smallest observing camera (stable, so
    points of one camera keep their order); also returns `order`, the old number of every new point
    (``points_3d_old[order] = points_3d_new``).  A solve does not depend on the numbering, but a
    rank's contiguous range of points then sees a range of cameras: with the reduced matrix split
    by points (options.matrix_split = 2) its share of the matrix is a band of block rows instead of
    the full pattern."""
    ci = np.asarray(prob["camera_indices"], dtype=np.int64)
    pi = np.asarray(prob["point_indices"], dtype=np.int64)
    n_points = len(prob["points_3d"])
    first = np.full(n_points, np.iinfo(np.int64).max)
    np.minimum.at(first, pi, ci)
    order = np.argsort(first, kind="stable")
    new_of = np.empty_like(order)
    new_of[order] = np.arange(n_points)
    out = dict(prob)
    out["points_3d"] = np.asarray(prob["points_3d"])[order]
    out["point_indices"] = new_of[pi]
    if "true_points_3d" in prob:
        out["true_points_3d"] = np.asarray(prob["true_points_3d"])[order]
    return out, order


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
