"""GPU: the problem structure built on the GPU (options.setup_mode 0, the default) against the one
built on host threads (setup_mode 1): every derived integer array equal element for element, and
therefore bit-identical solves.  Integer work: the bar is exact equality."""
import threading

import numpy as np
import pytest

from helpers import load_golden
from structure_from_motion_amd import engine as E
from structure_from_motion_amd import synthetic as syn

pytestmark = pytest.mark.gpu

ARRAYS = ["pt_ptr", "pm_cam", "pm_pt", "pm_orig", "new2old", "cm_pt", "chunk_cam", "chunk_beg", "cam_chunk_ptr", "wtile_beg",
          "ublk_r", "ublk_c", "ublk_chunk_ptr", "ublk_pos", "ublk_pos_t", "ublk_diag", "pchunk_blk", "pchunk_beg",
          "pair_p", "bsr_ptr", "bsr_col", "bsr_diag"]


def _load(eng, p):
    eng.set_problem(p["camera_params"], p["points_3d"], p["camera_indices"], p["point_indices"], p["points_2d"], p.get("K"))


def compare_paths(p, solve=True, **opts):
    with E.Engine(setup_mode=1, **opts) as host, E.Engine(setup_mode=0, **opts) as dev:
        _load(host, p)
        _load(dev, p)
        for name in ARRAYS:
            a, b = host.structure(name), dev.structure(name)
            assert a.shape == b.shape, (name, a.shape, b.shape)
            assert np.array_equal(a, b), (name, int(np.flatnonzero(a != b)[0]))
        assert np.array_equal(host.residuals(), dev.residuals())
        if solve:
            sh, sd = host.solve(), dev.solve()
            assert (sh.status, sh.iterations, sh.pcg_iterations) == (sd.status, sd.iterations, sd.pcg_iterations)
            assert sh.final_cost == sd.final_cost
            for x, y in zip(host.get_params(), dev.get_params()):
                assert np.array_equal(x, y)


@pytest.mark.parametrize("name", ["tiny_4_50_150", "rho2_11_300_600", "shuffled_9_200_700", "nonortho_6_60_200", "ladybug_49_7776_31843"])
@pytest.mark.parametrize("schur", [1, 2])
def test_golden_problems(name, schur):
    compare_paths(load_golden(name), schur_mode=schur)


def test_duplicates_and_degenerate_geometry():
    # duplicate (camera, point) observations give two entries in a diagonal block
    compare_paths(load_golden("edge_5_13"), solve=False, schur_mode=2)


def test_irregular_structure():
    # long tracks (more than 64 observations: wave tiles of their own), points seen once, points and
    # a camera nobody observes, a camera with several chunks
    from test_gpu_irregular import irregular_problem
    p = irregular_problem(33)
    compare_paths(p, schur_mode=2)
    compare_paths(p, schur_mode=1)


@pytest.mark.parametrize("shuffle", [False, True])
def test_larger_problem_two_level_and_bal(shuffle):
    compare_paths(syn.make_problem(300, 8000, 60000, seed=8, shuffle=shuffle), schur_mode=2, pcg_coarse=64)
    p = syn.make_bal_problem(40, 3000, 15000, seed=9, shuffle=shuffle)
    compare_paths(p, camera_model="bal")


def test_partitions_agree(monkeypatch):
    # every rank's partition: same arrays from both paths (the block pattern covers all ranks' points)
    p = syn.make_problem(30, 2000, 9000, seed=3, shuffle=True)
    for nranks in (2, 3):
        for r in range(nranks):
            compare_paths(p, solve=False, rank=r, nranks=nranks, schur_mode=2)


def test_index_errors_match():
    g = load_golden("tiny_4_50_150")
    for which, val in (("camera_indices", 4), ("point_indices", -1)):
        bad = dict(g)
        bad[which] = g[which].copy()
        bad[which][17] = val
        bad[which][90] = 10 ** 7          # a later one must not be the one reported
        msgs = []
        for mode in (0, 1):
            with E.Engine(setup_mode=mode) as eng, pytest.raises(IndexError) as e:
                _load(eng, bad)
            msgs.append(str(e.value))
        assert msgs[0] == msgs[1] and f"index {val} is out of bounds" in msgs[0]


def test_device_resident_inputs_and_reuse_of_a_handle():
    import torch
    p1, p2 = syn.make_problem(20, 1500, 6000, seed=5), syn.make_problem(9, 200, 700, seed=6, shuffle=True)
    with E.Engine() as eng, E.Engine(setup_mode=1) as ref:
        for p in (p1, p2, p1):           # a larger, a smaller, the larger one again on the same handle
            ci = torch.from_numpy(p["camera_indices"]).cuda()
            pi = torch.from_numpy(p["point_indices"]).cuda()
            uv = torch.from_numpy(p["points_2d"]).cuda()
            torch.cuda.synchronize()
            eng.set_problem_device(p["camera_params"], p["points_3d"], ci, pi, uv, p["K"])
            _load(ref, p)
            for name in ARRAYS:
                assert np.array_equal(eng.structure(name), ref.structure(name)), name
            a, b = eng.solve(), ref.solve()
            assert a.final_cost == b.final_cost and a.iterations == b.iterations


def test_no_observations():
    p = syn.make_problem(4, 50, 150, seed=1, window=4)
    empty = dict(p, camera_indices=p["camera_indices"][:0], point_indices=p["point_indices"][:0], points_2d=p["points_2d"][:0])
    compare_paths(empty, solve=False)


@pytest.mark.parametrize("seed", range(10))
def test_random_structures(seed):
    # arbitrary index arrays over a valid scene: heavy and empty cameras, tracks of one to a hundred
    # observations, duplicates, unobserved points, any order
    rng = np.random.default_rng(100 + seed)
    n_cam, n_pts = int(rng.integers(1, 31)), int(rng.integers(1, 201))
    base = syn.make_problem(max(n_cam, 4), max(n_pts, 8), 2 * max(n_pts, 8), seed=seed, window=3)
    n_obs = int(rng.integers(1, 1501))
    cam_w = rng.random(n_cam) ** 3 + (rng.random(n_cam) < 0.8) * 1e-9
    pt_w = rng.random(n_pts) ** 6 + (rng.random(n_pts) < 0.9) * 1e-9
    p = dict(camera_params=base["camera_params"][:n_cam], points_3d=base["points_3d"][:n_pts], K=base["K"],
             camera_indices=rng.choice(n_cam, n_obs, p=cam_w / cam_w.sum()).astype(np.int64),
             point_indices=rng.choice(n_pts, n_obs, p=pt_w / pt_w.sum()).astype(np.int64),
             points_2d=rng.uniform(0.0, 3000.0, (n_obs, 2)))
    compare_paths(p, solve=False, schur_mode=2)
    compare_paths(p, solve=False, schur_mode=1)
    if seed % 3 == 0:
        for r in range(2):
            compare_paths(p, solve=False, schur_mode=2, rank=r, nranks=2)


def test_host_tensors_are_refused_where_device_pointers_are_expected():
    # a CPU tensor handed to the device-input entry point must come back as an error, not reach a kernel
    import torch
    p = syn.make_problem(9, 200, 700, seed=6)
    ci, pi, uv = (torch.from_numpy(p[k]) for k in ("camera_indices", "point_indices", "points_2d"))
    with E.Engine() as eng:
        with pytest.raises(E.EngineError) as e:
            eng.set_problem_device(p["camera_params"], p["points_3d"], ci, pi.cuda(), uv.cuda(), p["K"])
        assert "camera_indices" in str(e.value) and "device pointer" in str(e.value)
        with pytest.raises(E.EngineError):
            eng.set_problem_device(p["camera_params"], p["points_3d"], ci.cuda(), pi.cuda(), uv, p["K"])
        # pinned host memory is device-accessible and is taken
        eng.set_problem_device(p["camera_params"], p["points_3d"], ci.pin_memory(), pi.cuda(), uv.cuda(), p["K"])
        assert eng.solve().status > 0
