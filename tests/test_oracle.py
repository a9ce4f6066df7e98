"""CPU: the oracle against the committed golden vectors and against its own literal loop."""
import numpy as np
import pytest

from helpers import golden_names, load_golden, golden_jacobian, rel_err
from oracle import ba_oracle as o
from structure_from_motion_amd import synthetic as syn


@pytest.mark.parametrize("name", golden_names())
def test_fun_matches_golden(name):
    g = load_golden(name)
    x0, args = o.pack(g)
    f = o.fun(x0, *args)
    assert f.shape == (2 * g["n_obs"],)
    assert np.abs(f - g["f0"]).max() <= 1e-12 * max(1.0, np.abs(g["f0"]).max())


@pytest.mark.parametrize("name", ["tiny_4_50_150", "nonortho_6_60_200", "edge_5_13"])
def test_vectorised_projection_equals_reference_loop(name):
    # the per-observation loop is the literal shape of bundle_adjustment.py:22-30
    g = load_golden(name)
    pts = g["points_3d"][g["point_indices"]]
    cams = g["camera_params"][g["camera_indices"]]
    a = o.project(pts, cams, g["K"])
    b = o.project_loop(pts, cams, g["K"])
    assert np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(b).max())


def test_half_turn_rotations_follow_the_literal_loop():
    # fixture made by the per-observation loop: within 1e-5 rad of a half turn cv2.Rodrigues rebuilds
    # the axis from the diagonal, which moves a camera by up to 1e-6 rad (milli-pixels).  acos near
    # -1 amplifies one ulp of the polar factor by 1/sin(angle) ~ 1e6, hence 1e-9 and not 1e-12.
    g = load_golden("nearpi_4_40_120")
    x0, args = o.pack(g)
    f = o.fun(x0, *args)
    assert np.abs(f - g["f0"]).max() <= 1e-9 * np.abs(g["f0"]).max()
    # and the branch matters: the plain polar factor is milli-pixels away
    U, _, Vt = np.linalg.svd(g["camera_params"][:, :9].reshape(-1, 3, 3))
    plain = g["camera_params"].copy()
    plain[:, :9] = (U @ Vt).reshape(-1, 9)
    ci, pi = g["camera_indices"], g["point_indices"]
    Y = np.einsum("nij,nj->ni", (U @ Vt)[ci], g["points_3d"][pi]) + plain[ci, 9:]
    u = g["K"][0, 0] * Y[:, 0] / Y[:, 2] + g["K"][0, 2] - g["points_2d"][:, 0]
    assert np.abs(u - g["f0"][0::2]).max() > 1e-4


def test_rodrigues_round_trip_and_small_angle_snap():
    rng = np.random.default_rng(0)
    for _ in range(20):
        w = rng.normal(size=3)
        R = o.rodrigues_vector_to_matrix(w)
        R2 = o.rodrigues_vector_to_matrix(o.rodrigues_matrix_to_vector(R))
        assert np.abs(R - R2).max() < 1e-12
    # cv2 returns a zero vector below sin(angle) = 1e-5
    R = o.rodrigues_vector_to_matrix(np.array([3e-6, 0, 0]))
    assert np.all(o.rodrigues_matrix_to_vector(R) == 0)
    cp = np.r_[R.ravel(), 0, 0, 0][None]
    assert np.array_equal(o.camera_rotations(cp)[0], np.eye(3))


def test_zero_depth_guard():
    g = load_golden("edge_5_13")
    j = int(np.flatnonzero(g["point_indices"] == 0)[0])
    K = g["K"]
    X = g["points_3d"][0]
    t = g["camera_params"][0, 9:12]
    assert X[2] + t[2] == 0.0
    expect_u = K[0, 0] * (X[0] + t[0]) + K[0, 2] - g["points_2d"][j, 0]   # 1/Z := 1
    assert abs(g["f0"][2 * j] - expect_u) < 1e-9


def test_sparsity_pattern():
    p = syn.make_problem(5, 40, 100, seed=3, window=5)
    A = o.bundle_adjustment_sparsity(p["n_cameras"], p["n_points"], p["camera_indices"], p["point_indices"])
    assert A.shape == (200, 5 * 12 + 40 * 3)
    assert A.nnz == 30 * 100
    B = o.sparsity_csr(p["n_cameras"], p["n_points"], p["camera_indices"], p["point_indices"])
    assert (A.tocsr() != B).nnz == 0
    row = A.tocsr()[2 * 7].indices
    c, q = p["camera_indices"][7], p["point_indices"][7]
    assert set(row) == set(range(12 * c, 12 * c + 12)) | set(range(60 + 3 * q, 60 + 3 * q + 3))


@pytest.mark.parametrize("name", golden_names("jac_data"))
def test_fd_jacobian_golden_is_consistent_with_pattern(name):
    g = load_golden(name)
    J = golden_jacobian(g)
    A = o.sparsity_csr(g["n_cameras"], g["n_points"], g["camera_indices"], g["point_indices"])
    assert ((J != 0).astype(int) - (J != 0).multiply(A != 0).astype(int)).nnz == 0


def test_reference_solve_reproduces_golden():
    g = load_golden("tiny_4_50_150")
    r = o.solve_reference(g)
    assert r["status"] == int(g["ref_status"])
    assert abs(r["cost"] - float(g["ref_cost"])) <= 1e-9 * float(g["ref_cost"])
    assert r["nfev"] == int(g["ref_nfev"])


@pytest.mark.parametrize("name", ["bal_ref_6_40", "bal_ref_12_300"])
def test_vector_to_matrix_rotation_pinned_by_the_reference_script(name):
    """`rotated` in these fixtures is what the reference's own ba_test.rotate (ba_test.py:62-75)
    returned for the gathered points and rotation vectors (oracle/gen_golden_bal_ref.py): the
    same Rodrigues map vector -> rotation that cv2.projectPoints applies inside
    bundle_adjustment.py:28.  The pinhole oracle's restatement of it, applied as a matrix, must
    agree to rounding (the script adds three terms per point, a matrix product five per
    component), including the zero vector of the 12-camera file, where the script's 0/0 is turned
    into the identity (ba_test.py:68-70)."""
    import os
    ref = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    rvec = ref["camera_params"][ref["camera_indices"]][:, :3]
    X = ref["points_3d"][ref["point_indices"]]
    R = np.array([o.rodrigues_vector_to_matrix(r) for r in ref["camera_params"][:, :3]])
    got = np.einsum("nij,nj->ni", R[ref["camera_indices"]], X)
    scale = np.abs(ref["rotated"]).max()
    assert np.abs(got - ref["rotated"]).max() <= 1e-15 * max(1.0, scale)
    if name == "bal_ref_12_300":
        zero = np.flatnonzero(~rvec.any(axis=1))
        assert zero.size and np.array_equal(got[zero], X[zero])
    # every matrix is a rotation, and the matrix -> vector restatement takes it back to the vector
    # (angle below pi): the two halves of the cv2 round trip are consistent with each other
    for r, Rm in zip(ref["camera_params"][:, :3], R):
        assert np.abs(Rm @ Rm.T - np.eye(3)).max() <= 1e-15 * 4
        assert np.abs(o.rodrigues_matrix_to_vector(Rm) - r).max() <= 1e-12 * max(1.0, np.abs(r).max())
