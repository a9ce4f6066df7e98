"""CPU: host-side logic of the drop-in module and the synthetic generator (no GPU calls)."""
import numpy as np
import pandas as pd
import pytest

from oracle import ba_oracle as o
from structure_from_motion_amd import bundle_adjustment as ba
from structure_from_motion_amd import synthetic as syn


class FakeView:
    def __init__(self, vid, R, t):
        self.id, self.rotation, self.translation = vid, R, t.reshape(3, 1)


class FakeWorldPoints:
    def __init__(self, pts, corr):
        self.world_points, self.correspondences = pts, corr


def _reference_view2idx(views, corr):
    """The literal loops of /root/reference/bundle_adjustment.py:98-119."""
    view_idx, camera_params = {}, []
    for view in views:
        if view.id not in view_idx:
            view_idx[view.id] = len(view_idx)
        rot_vec = np.squeeze(view.rotation)
        camera_params.append(np.concatenate((rot_vec, view.translation.reshape((1, 3))), axis=None).tolist())
    points_2d, camera_indices, point_indices = [], [], []
    for i, row in corr.iterrows():
        points_2d.append(row['FeatureIndex'][0]); camera_indices.append(view_idx[row['ViewId'][0]])
        points_2d.append(row['FeatureIndex'][1]); camera_indices.append(view_idx[row['ViewId'][1]])
        point_indices.append(i); point_indices.append(i)
    return (np.array(camera_params), np.array(camera_indices), np.array(point_indices),
            np.array(points_2d), view_idx)


def test_view2idx_matches_reference_loops(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    p = syn.make_problem(5, 80, 160, seed=4, window=5)
    views = [FakeView(f"id{j}", p["camera_params"][j, :9].reshape(3, 3), p["camera_params"][j, 9:]) for j in range(5)]
    ci = p["camera_indices"].reshape(-1, 2)
    uv = p["points_2d"].reshape(-1, 2, 2)
    corr = pd.DataFrame({"ViewId": [[f"id{a}", f"id{b}"] for a, b in ci],
                         "FeatureIndex": [[uv[i, 0], uv[i, 1]] for i in range(len(ci))]})
    b = ba.BundleAdjustment(FakeWorldPoints(p["points_3d"], corr), p["K"], np.zeros((1, 5)), views)
    b.view2idx()
    cp, cidx, pidx, p2d, vidx = _reference_view2idx(views, corr)
    assert np.array_equal(b.camera_params, cp) and np.array_equal(b.camera_indices, cidx)
    assert np.array_equal(b.point_indices, pidx) and np.array_equal(b.points_2d, p2d)
    assert b.view_idx == vidx and b.n_cameras == 5 and b.n_points == 80
    assert b.focal_len == (p["K"][0, 0] + p["K"][1, 1]) / 2
    with np.load(tmp_path / "optimize_data.npz") as z:
        assert np.array_equal(z["points_2d"], p2d)


def test_view2idx_rows_with_more_than_two_entries_and_unknown_ids(tmp_path, monkeypatch):
    # a correspondence that names a third view (the reference's redundant-view rows): the first
    # two entries count, as in the reference's loops; an id no view carries is a KeyError there
    monkeypatch.chdir(tmp_path)
    p = syn.make_problem(5, 40, 80, seed=5, window=5)
    views = [FakeView(f"id{j}", p["camera_params"][j, :9].reshape(3, 3), p["camera_params"][j, 9:]) for j in range(5)]
    ci = p["camera_indices"].reshape(-1, 2)
    uv = p["points_2d"].reshape(-1, 2, 2)
    corr = pd.DataFrame({"ViewId": [[f"id{a}", f"id{b}", "id0"] for a, b in ci],
                         "FeatureIndex": [[uv[i, 0], uv[i, 1], uv[i, 0] + 1.0] for i in range(len(ci))]})
    b = ba.BundleAdjustment(FakeWorldPoints(p["points_3d"], corr), p["K"], np.zeros((1, 5)), views)
    b.view2idx()
    cp, cidx, pidx, p2d, vidx = _reference_view2idx(views, corr)
    assert np.array_equal(b.camera_indices, cidx) and np.array_equal(b.points_2d, p2d) and np.array_equal(b.point_indices, pidx)
    for width in (2, 3):
        bad = pd.DataFrame({"ViewId": [r[:width] for r in corr["ViewId"]], "FeatureIndex": [r[:width] for r in corr["FeatureIndex"]]})
        bad.at[7, "ViewId"] = ["id1", "nobody", "id0"][:width]
        with pytest.raises(KeyError):
            ba.BundleAdjustment(FakeWorldPoints(p["points_3d"], bad), p["K"], np.zeros((1, 5)), views).view2idx()


def test_sparsity_mirror_equals_oracle():
    p = syn.make_problem(4, 30, 90, seed=6, window=4)
    A = ba.bundle_adjustment_sparsity(p["n_cameras"], p["n_points"], p["camera_indices"], p["point_indices"])
    B = o.bundle_adjustment_sparsity(p["n_cameras"], p["n_points"], p["camera_indices"], p["point_indices"])
    assert A.shape == B.shape and (A.tocsr() != B.tocsr()).nnz == 0 and A.dtype == B.dtype


@pytest.mark.parametrize("shape", ["fountain", "ladybug"])
def test_synthetic_shapes(shape):
    p = syn.make_shape(shape, seed=0)
    nc, npts, no = syn.SHAPES[shape]
    assert p["camera_params"].shape == (nc, 12) and p["points_3d"].shape == (npts, 3)
    assert p["camera_indices"].shape == (no,) and p["points_2d"].shape == (no, 2)
    assert np.all(np.diff(p["point_indices"]) >= 0)                       # point-sorted
    counts = np.bincount(p["point_indices"], minlength=npts)
    assert counts.min() >= 2
    pairs = p["camera_indices"] * npts + p["point_indices"]
    assert np.unique(pairs).size == no                                    # no duplicate (cam, pt)
    if shape == "fountain":
        assert np.all(counts == 2)
    R = p["true_camera_params"][:, :9].reshape(-1, 3, 3)
    assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() < 1e-12
    # deterministic in the seed
    q = syn.make_shape(shape, seed=0)
    assert np.array_equal(p["points_2d"], q["points_2d"])


def test_npz_round_trip(tmp_path):
    p = syn.make_problem(4, 20, 50, seed=1, window=4)
    syn.save_npz(tmp_path / "optimize_data.npz", p)
    q = syn.load_npz(tmp_path / "optimize_data.npz")
    for k in ("camera_params", "points_3d", "camera_indices", "point_indices", "points_2d"):
        assert np.array_equal(p[k], q[k])
    assert abs(o.cost(p) - o.cost(q)) == 0


def _growing_table(n_views=6, per_pair=50, seed=3):
    rng = np.random.default_rng(seed)
    views = [FakeView(f"id{j}", np.eye(3) + 0.01 * rng.normal(size=(3, 3)), rng.normal(size=3)) for j in range(n_views)]
    rows = [pd.DataFrame({"ViewId": [[f"id{k - 1}", f"id{k}"] for _ in range(per_pair)],
                          "FeatureIndex": [[rng.uniform(0, 640, 2), rng.uniform(0, 480, 2)] for _ in range(per_pair)]})
            for k in range(1, n_views)]
    return views, rows, rng


def _check_against_reference(b, views, corr):
    cp, cidx, pidx, p2d, vidx = _reference_view2idx(views, corr)
    assert np.array_equal(b.camera_params, cp) and np.array_equal(b.camera_indices, cidx)
    assert np.array_equal(b.point_indices, pidx) and np.array_equal(b.points_2d, p2d) and b.view_idx == vidx


def test_view2idx_reuses_what_it_packed_for_a_growing_table(tmp_path, monkeypatch):
    """The reference's driver packs the whole table again after every image (main.py:103-116); here
    the rows already packed for the same wpSet are taken over once verified, only new rows are
    packed - and the arrays equal the reference's loops every time."""
    monkeypatch.chdir(tmp_path)
    views, rows, rng = _growing_table()
    K, dist = np.eye(3), np.zeros((1, 5))
    wp = FakeWorldPoints(np.empty((0, 3)), pd.DataFrame({"ViewId": [], "FeatureIndex": []}))
    packed = []
    real = ba._pack_rows
    monkeypatch.setattr(ba, "_pack_rows", lambda ids, feats, vi: (packed.append(len(ids)), real(ids, feats, vi))[1])
    for k in range(1, len(views)):
        wp.correspondences = pd.concat((wp.correspondences, rows[k - 1]), ignore_index=True)
        wp.world_points = np.vstack((wp.world_points, rng.normal(size=(len(rows[k - 1]), 3))))
        packed.clear()
        b = ba.BundleAdjustment(wp, K, dist, views[:k + 1])
        b.view2idx()
        _check_against_reference(b, views[:k + 1], wp.correspondences)
        # first call: all 50 rows; later: the sample that verifies the old rows (all of them in a
        # table this small), then the 50 new rows on their own
        assert packed == ([50] if k == 1 else [len(np.unique(np.r_[np.linspace(0, 50 * (k - 1) - 1, min(50 * (k - 1), 192)).astype(int), np.arange(max(0, 50 * (k - 1) - 64), 50 * (k - 1))])), 50])
    # a row re-bound through .at, as WorldPoints.py:54-57 does for redundant views ...
    new_uv = [np.array([1.5, 2.5]), np.array([3.5, 4.5])]
    wp.correspondences.at[70, "FeatureIndex"] = new_uv
    b = ba.BundleAdjustment(wp, K, dist, views); b.view2idx()
    _check_against_reference(b, views, wp.correspondences)
    assert np.array_equal(b.points_2d[140], new_uv[0])
    # ... an entry replaced inside a row's own list, a view id changed in place ...
    wp.correspondences.at[20, "FeatureIndex"][1] = np.array([9.0, 8.0])
    wp.correspondences.at[30, "ViewId"][0] = "id5"
    b = ba.BundleAdjustment(wp, K, dist, views); b.view2idx()
    _check_against_reference(b, views, wp.correspondences)
    assert np.array_equal(b.points_2d[41], [9.0, 8.0]) and b.camera_indices[60] == 5
    # ... a list extended past its second entry (nothing packed changes), rows dropped at the tail,
    # and the views handed over in another order (every camera index changes)
    wp.correspondences.at[10, "FeatureIndex"] += [np.array([0.0, 0.0])]
    wp.correspondences.at[10, "ViewId"] += ["id0"]
    b = ba.BundleAdjustment(wp, K, dist, views); b.view2idx()
    _check_against_reference(b, views, wp.correspondences)
    wp.correspondences = wp.correspondences.iloc[:120].reset_index(drop=True)
    wp.world_points = wp.world_points[:120]
    b = ba.BundleAdjustment(wp, K, dist, views); b.view2idx()
    _check_against_reference(b, views, wp.correspondences)
    b = ba.BundleAdjustment(wp, K, dist, views[::-1]); b.view2idx()
    _check_against_reference(b, views[::-1], wp.correspondences)


def test_fun_cache_compares_contents_at_memory_speed():
    rng = np.random.default_rng(0)
    a = [rng.integers(0, 9, 700_000), rng.random((700_000, 2))]
    b = [x.copy() for x in a]
    assert ba._same_arrays(list(zip(a, b)))
    b[1][-1, 1] += 1e-9                                     # one number in the last chunk
    assert not ba._same_arrays(list(zip(a, b)))
    assert not ba._same_arrays([(a[0], a[0].astype(np.int32))]) and not ba._same_arrays([(a[0], a[0][:-1])])


def test_write_back_tables_equal_the_reference_loops(tmp_path, monkeypatch):
    """write_back cuts every view's [u, v, X, Y, Z] table out of the packed arrays with one sort;
    the reference walks the correspondences per view (view.py:121-132 update_world_points)."""
    monkeypatch.chdir(tmp_path)
    views, rows, rng = _growing_table(n_views=5, per_pair=30, seed=8)
    corr = pd.concat(rows, ignore_index=True)
    corr.at[3, "ViewId"] = ["id2", "id2"]                   # a row naming one view twice counts once, as its first entry
    corr.at[9, "ViewId"] = ["id4", "id0"]
    pts = rng.normal(size=(len(corr), 3))
    wp = FakeWorldPoints(pts, corr)
    b = ba.BundleAdjustment(wp, np.eye(3), np.zeros((1, 5)), views)
    b.view2idx()
    b._poses = rng.normal(size=(5, 12))
    b._points_3d = pts + 0.5
    b.reprojection_error = np.arange(5.0)
    b.write_back(print_error=False)
    assert wp.world_points is b._points_3d
    for j, view in enumerate(views):
        assert np.array_equal(view.rotation, b._poses[j, :9].reshape(3, 3)) and view.translation.shape == (3, 1)
        want = []
        for i, row in corr.iterrows():                      # the reference's loop
            if view.id in row["ViewId"]:
                k = row["ViewId"].index(view.id)
                want.append(np.concatenate((row["FeatureIndex"][k], b._points_3d[i])))
        assert np.array_equal(view.world_points, np.array(want).reshape(-1, 5))
