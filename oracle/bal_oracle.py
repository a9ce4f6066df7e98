"""CPU oracle for the second camera model: the 9-number "Bundle Adjustment in the Large" camera
(rotation vector, translation, focal length, two radial distortion coefficients) and its text
file format.  TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the shipped package never does.

PINNED against the reference itself: /root/reference/ba_test.py fetches its input over the network
when imported unless a file of the right name lies in the working directory (ba_test.py:9-14);
oracle/gen_golden_bal_ref.py imports it in the build container on local data files and stores what
the module computed in tests/golden/bal_ref_*.npz (reader outputs, rotate / project / fun, the
pattern, its least_squares result).  tests/test_oracle_bal.py requires this restatement to
reproduce those files bit for bit.  The optimiser is not restated: `solve_reference` drives the
real scipy.optimize.least_squares with the script's keyword arguments (ba_test.py:125-126).

What is restated, and from where:
  read_bal_data()               /root/reference/ba_test.py:17-42   (one readline per number)
  rotate()                      /root/reference/ba_test.py:62-75
  project()                     /root/reference/ba_test.py:78-89
  fun()                         /root/reference/ba_test.py:92-100
  bundle_adjustment_sparsity()  /root/reference/ba_test.py:103-117
  solve_reference()             /root/reference/ba_test.py:120-126
"""
from __future__ import annotations

import bz2
import time

import numpy as np
from scipy.optimize import least_squares
from scipy.sparse import csr_matrix

REFERENCE_KWARGS = dict(x_scale="jac", ftol=1e-4, method="trf")     # xtol, gtol: SciPy's 1e-8


# --------------------------------------------------------------------------- file format
def read_bal_data(file_name):
    """Header "n_cameras n_points n_observations"; one "camera point x y" line per observation;
    then 9 numbers per camera and 3 per point, one number per line.  `.bz2` names are read
    through bz2 (the reference always does), anything else as plain text."""
    opener = bz2.open if str(file_name).endswith(".bz2") else open
    with opener(file_name, "rt") as fh:
        n_cam, n_pts, n_obs = (int(tok) for tok in fh.readline().split())
        cam_idx = np.empty(n_obs, dtype=int)
        pt_idx = np.empty(n_obs, dtype=int)
        uv = np.empty((n_obs, 2))
        for k in range(n_obs):
            c, p, x, y = fh.readline().split()
            cam_idx[k], pt_idx[k] = int(c), int(p)
            uv[k, 0], uv[k, 1] = float(x), float(y)
        cams = np.array([float(fh.readline()) for _ in range(9 * n_cam)]).reshape(n_cam, 9)
        pts = np.array([float(fh.readline()) for _ in range(3 * n_pts)]).reshape(n_pts, 3)
    return cams, pts, cam_idx, pt_idx, uv


# --------------------------------------------------------------------------- model
def rotate(points, rot_vecs):
    """Rodrigues' formula per row: X cos(a) + (k x X) sin(a) + k (k.X)(1 - cos(a)), a = |rvec|,
    k = rvec / a with 0/0 taken as 0 (so a zero vector leaves the point where it is)."""
    angle = np.sqrt(np.sum(rot_vecs * rot_vecs, axis=1))[:, None]
    with np.errstate(invalid="ignore", divide="ignore"):
        axis = np.nan_to_num(rot_vecs / angle)
    along = np.sum(points * axis, axis=1)[:, None]
    c, s = np.cos(angle), np.sin(angle)
    return c * points + s * np.cross(axis, points) + along * (1.0 - c) * axis


def project(points, camera_params):
    """Gathered points (N, 3) and cameras (N, 9) -> image points (N, 2): P = R X + t,
    p = -P[:2] / P[2], scaled by f (1 + k1 |p|^2 + k2 |p|^4)."""
    P = rotate(points, camera_params[:, :3]) + camera_params[:, 3:6]
    with np.errstate(invalid="ignore", divide="ignore"):
        p = -P[:, :2] / P[:, 2:3]
    f, k1, k2 = camera_params[:, 6], camera_params[:, 7], camera_params[:, 8]
    n = np.sum(p * p, axis=1)
    return p * ((1.0 + k1 * n + k2 * (n * n)) * f)[:, None]      # k2 (n^2), then the sum, then f: the reference's rounding


def fun(params, n_cameras, n_points, camera_indices, point_indices, points_2d):
    cams = params[:9 * n_cameras].reshape(n_cameras, 9)
    pts = params[9 * n_cameras:].reshape(n_points, 3)
    return (project(pts[point_indices], cams[camera_indices]) - points_2d).ravel()


def bundle_adjustment_sparsity(n_cameras, n_points, camera_indices, point_indices):
    """Rows 2i, 2i+1 touch the 9 columns of their camera and the 3 of their point; returned as
    CSR (the reference fills a LIL matrix with the same pattern)."""
    n_obs = camera_indices.size
    cols = np.concatenate([9 * camera_indices[:, None] + np.arange(9)[None, :],
                           9 * n_cameras + 3 * point_indices[:, None] + np.arange(3)[None, :]], axis=1)
    cols = np.repeat(cols, 2, axis=0)
    indptr = 12 * np.arange(2 * n_obs + 1)
    A = csr_matrix((np.ones(cols.size, dtype=int), cols.ravel(), indptr),
                   shape=(2 * n_obs, 9 * n_cameras + 3 * n_points))
    A.sum_duplicates()
    return A


# --------------------------------------------------------------------------- solves
def pack(prob):
    x0 = np.hstack([prob["camera_params"].ravel(), prob["points_3d"].ravel()])
    args = (prob["n_cameras"], prob["n_points"], prob["camera_indices"], prob["point_indices"], prob["points_2d"])
    return x0, args


def cost(prob, camera_params=None, points_3d=None):
    cams = prob["camera_params"] if camera_params is None else camera_params
    pts = prob["points_3d"] if points_3d is None else points_3d
    r = project(pts[prob["point_indices"]], cams[prob["camera_indices"]]) - prob["points_2d"]
    return 0.5 * float(np.sum(r * r))


def _solve(prob, verbose=0, **kw):
    x0, args = pack(prob)
    A = bundle_adjustment_sparsity(prob["n_cameras"], prob["n_points"], prob["camera_indices"], prob["point_indices"])
    f0 = fun(x0, *args)
    t0 = time.perf_counter()
    res = least_squares(fun, x0, jac_sparsity=A, verbose=verbose, args=args, **kw)
    dt = time.perf_counter() - t0
    nc = prob["n_cameras"]
    return dict(cost=float(res.cost), status=int(res.status), nfev=int(res.nfev), njev=int(res.njev),
                optimality=float(res.optimality), initial_cost=0.5 * float(f0 @ f0), seconds=dt,
                camera_params=res.x[:9 * nc].reshape(nc, 9).copy(), points_3d=res.x[9 * nc:].reshape(-1, 3).copy())


def solve_reference(prob, verbose=0, max_nfev=None):
    kw = dict(REFERENCE_KWARGS)
    if max_nfev is not None:
        kw["max_nfev"] = max_nfev
    return _solve(prob, verbose=verbose, **kw)


def solve_tight(prob, max_nfev=200):
    return _solve(prob, x_scale="jac", method="trf", ftol=1e-14, xtol=1e-14, gtol=1e-10, max_nfev=max_nfev)
