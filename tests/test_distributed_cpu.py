"""CPU, world_size 2 over gloo: the multi-GPU decomposition.  Each rank owns a contiguous point
range with an equal share of the observations; the camera-block partial sums (J_c'J_c, J_c'r),
the reduced-matvec partial and the cost partial are all-reduced; everything else is local.
The per-rank arithmetic here is the NumPy model, so this checks the decomposition and the
collective pattern that libsfmba runs over RCCL, not the kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import engine_model as em
from structure_from_motion_amd import distributed as sd
from structure_from_motion_amd import synthetic as syn


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _allreduce(a):
    t = torch.from_numpy(np.ascontiguousarray(a))
    dist.all_reduce(t)
    return t.numpy()


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = syn.make_problem(9, 400, 1500, seed=5, window=9, shuffle=True)
    lo, hi = sd.point_partition(p["point_indices"], p["n_points"], rank, world)
    shard = sd.shard_problem(p, lo, hi)
    m = em.Model(shard)
    r, Jc, Jp, U, gc, V, gp = m.assemble()
    U = _allreduce(U); gc = _allreduce(gc)
    cost = _allreduce(np.array([0.5 * np.sum(r * r)]))[0]
    x = np.random.default_rng(0).normal(size=(9, 6))
    W = np.einsum("nki,nkj->nij", Jc, Jp)
    Vinv = np.linalg.inv(V + 1e-3 * np.eye(3)[None])
    tp = np.zeros((m.np_, 3)); np.add.at(tp, m.pi, np.einsum("nji,nj->ni", W, x[m.ci]))
    sp = np.einsum("nij,nj->ni", Vinv, tp)
    q = np.zeros((9, 6)); np.add.at(q, m.ci, np.einsum("nij,nj->ni", W, sp[m.pi]))
    q = _allreduce(q)
    # the explicit matrix split by points: this rank's blocks B_k = W Vinv W' over its own points and
    # 1 / world of the (all-reduced) camera blocks; the product is the ranks' sum, one all-reduce of
    # 6 numbers per camera, and no block is ever exchanged
    Wd = np.zeros((9 * 6, m.np_ * 3))
    for n in range(m.no):
        Wd[6 * m.ci[n]:6 * m.ci[n] + 6, 3 * m.pi[n]:3 * m.pi[n] + 3] += W[n]
    Vd = np.zeros((m.np_ * 3, m.np_ * 3))
    for k in range(m.np_):
        Vd[3 * k:3 * k + 3, 3 * k:3 * k + 3] = Vinv[k]
    share = -Wd @ Vd @ Wd.T
    for c in range(9):
        share[6 * c:6 * c + 6, 6 * c:6 * c + 6] += U[c] / world
    qs = _allreduce((share @ x.ravel()).reshape(9, 6))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), U=U, gc=gc, cost=cost, q=q, qs=qs, lo=lo, hi=hi,
             n_obs=m.no)
    dist.destroy_process_group()


def test_point_partition_is_balanced_and_covers():
    p = syn.make_problem(9, 400, 1500, seed=5, window=9, shuffle=True)
    edges = [sd.point_partition(p["point_indices"], p["n_points"], r, 4) for r in range(4)]
    assert edges[0][0] == 0 and edges[-1][1] == 400
    for a, b in zip(edges[:-1], edges[1:]):
        assert a[1] == b[0]
    counts = [np.count_nonzero((p["point_indices"] >= lo) & (p["point_indices"] < hi)) for lo, hi in edges]
    assert sum(counts) == 1500 and max(counts) - min(counts) <= 12


def test_two_rank_allreduce_reproduces_the_global_system(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    p = syn.make_problem(9, 400, 1500, seed=5, window=9, shuffle=True)
    m = em.Model(p)
    r, Jc, Jp, U, gc, V, gp = m.assemble()
    x = np.random.default_rng(0).normal(size=(9, 6))
    W = np.einsum("nki,nkj->nij", Jc, Jp)
    Vinv = np.linalg.inv(V + 1e-3 * np.eye(3)[None])
    tp = np.zeros((m.np_, 3)); np.add.at(tp, m.pi, np.einsum("nji,nj->ni", W, x[m.ci]))
    sp = np.einsum("nij,nj->ni", Vinv, tp)
    q = np.zeros((9, 6)); np.add.at(q, m.ci, np.einsum("nij,nj->ni", W, sp[m.pi]))
    outs = [np.load(tmp_path / f"rank{k}.npz") for k in range(2)]
    assert int(outs[0]["n_obs"]) + int(outs[1]["n_obs"]) == 1500
    for z in outs:
        assert np.abs(z["U"] - U).max() <= 1e-12 * np.abs(U).max()
        assert np.abs(z["gc"] - gc).max() <= 1e-12 * np.abs(gc).max()
        assert abs(float(z["cost"]) - 0.5 * np.sum(r * r)) <= 1e-12 * 0.5 * np.sum(r * r)
        assert np.abs(z["q"] - q).max() <= 1e-12 * np.abs(q).max()
        # split explicit matrix: the ranks' shares add up to S = U - W Vinv W'
        Sx = np.einsum("cij,cj->ci", U, x) - q
        assert np.abs(z["qs"] - Sx).max() <= 1e-11 * np.abs(Sx).max()
    assert np.array_equal(outs[0]["U"], outs[1]["U"])     # every rank holds the same reduced data


def test_numbering_points_by_camera_keeps_the_problem_and_bands_the_ranks():
    p = syn.make_problem(60, 1500, 9000, seed=3, window=6)
    q, order = sd.number_points_by_camera(p)
    assert sorted(order.tolist()) == list(range(1500))
    assert np.array_equal(q["points_3d"], p["points_3d"][order])
    # every observation still names the same camera and the same point coordinates
    assert np.array_equal(q["camera_indices"], p["camera_indices"])
    assert np.array_equal(q["points_3d"][q["point_indices"]], p["points_3d"][p["point_indices"]])
    seen = []
    for prob in (p, q):
        lo, hi = sd.point_partition(prob["point_indices"], 1500, 0, 2)
        mine = (prob["point_indices"] >= lo) & (prob["point_indices"] < hi)
        seen.append(len(np.unique(prob["camera_indices"][mine])))
    assert seen[0] == 60 and seen[1] < 45        # a range of points now sees a range of cameras
