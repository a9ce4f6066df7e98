"""The reference's second bundle-adjustment driver (/root/reference/ba_test.py) on one MI355X:
the text file format of the "Bundle Adjustment in the Large" data sets and its 9-number camera
(rotation vector, translation, focal length, two radial distortion coefficients).

Same names and argument meaning as the script's functions: ``read_bal_data`` (:17), ``rotate``
(:62), ``project`` (:78), ``fun`` (:92), ``bundle_adjustment_sparsity`` (:103); ``solve`` stands
for the script's ``least_squares`` call (:125-126).  The file is parsed in one vectorised pass
instead of one ``readline`` per number; ``project``/``fun``/``solve`` run in libsfmba.so with
``camera_model='bal'`` (analytic 2x9 / 2x3 Jacobian blocks, Levenberg-Marquardt on the Schur
complement with a matrix-free block-Jacobi PCG).  There is no CPU fallback.
"""
from __future__ import annotations

import bz2

import numpy as np
from scipy.sparse import csr_matrix

from .bundle_adjustment import _ObservationCache, _engine_lock
from .engine import Engine, TERMINATION_MESSAGES

__all__ = ["read_bal_data", "write_bal_data", "rotate", "project", "fun", "bundle_adjustment_sparsity", "solve"]

_fun_cache = _ObservationCache()


def _numbers(text):
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            return np.fromstring(text, dtype=np.float64, sep=" ")
    except Exception:
        return np.array(text.split(), dtype=np.float64)


def read_bal_data(file_name):
    """Reference: ba_test.py:17-42.  Returns ``camera_params`` (n_cameras x 9), ``points_3d``
    (n_points x 3), ``camera_indices``, ``point_indices`` (n_observations, int) and ``points_2d``
    (n_observations x 2).  ``.bz2`` names are decompressed (the script's only case), other names
    are read as plain text.  A file that is too short, has trailing numbers, non-integer or
    out-of-range indices raises ``ValueError``."""
    name = str(file_name)
    opener = bz2.open if name.endswith(".bz2") else open
    with opener(name, "rt") as fh:
        v = _numbers(fh.read())
    if v.size < 3:
        raise ValueError(f"{name}: missing header")
    hdr = v[:3]
    if np.any(hdr != np.floor(hdr)) or np.any(hdr < 0):
        raise ValueError(f"{name}: bad header {hdr.tolist()}")
    n_cam, n_pts, n_obs = (int(x) for x in hdr)
    want = 3 + 4 * n_obs + 9 * n_cam + 3 * n_pts
    if v.size != want:
        raise ValueError(f"{name}: {v.size} numbers, header announces {want}")
    obs = v[3:3 + 4 * n_obs].reshape(n_obs, 4)
    if np.any(obs[:, :2] != np.floor(obs[:, :2])):
        raise ValueError(f"{name}: non-integer camera or point index")
    camera_indices = obs[:, 0].astype(int)
    point_indices = obs[:, 1].astype(int)
    if n_obs and (camera_indices.min() < 0 or camera_indices.max() >= n_cam or
                  point_indices.min() < 0 or point_indices.max() >= n_pts):
        raise ValueError(f"{name}: camera or point index out of range")
    points_2d = np.ascontiguousarray(obs[:, 2:4])
    o = 3 + 4 * n_obs
    camera_params = v[o:o + 9 * n_cam].reshape(n_cam, 9).copy()
    points_3d = v[o + 9 * n_cam:].reshape(n_pts, 3).copy()
    return camera_params, points_3d, camera_indices, point_indices, points_2d


def write_bal_data(file_name, camera_params, points_3d, camera_indices, point_indices, points_2d):
    """The inverse of ``read_bal_data``: numbers are written with 17 significant digits so that a
    file read back gives the same doubles."""
    camera_params = np.asarray(camera_params, dtype=np.float64).reshape(-1, 9)
    points_3d = np.asarray(points_3d, dtype=np.float64).reshape(-1, 3)
    ci = np.asarray(camera_indices).ravel()
    pi = np.asarray(point_indices).ravel()
    uv = np.asarray(points_2d, dtype=np.float64).reshape(-1, 2)
    lines = [f"{camera_params.shape[0]} {points_3d.shape[0]} {ci.size}"]
    lines += [f"{int(c)} {int(p)}     {x:.17g} {y:.17g}" for c, p, (x, y) in zip(ci, pi, uv)]
    lines += [f"{x:.17g}" for x in camera_params.ravel()]
    lines += [f"{x:.17g}" for x in points_3d.ravel()]
    name = str(file_name)
    opener = bz2.open if name.endswith(".bz2") else open
    with opener(name, "wt") as fh:
        fh.write("\n".join(lines) + "\n")


def rotate(points, rot_vecs):
    """Rotate points by given rotation vectors with Rodrigues' formula (reference:
    ba_test.py:62-75); a zero vector is the identity.  Small per-row host arithmetic, kept on
    the host: the GPU path folds it into ``project``."""
    points = np.asarray(points, dtype=np.float64)
    rot_vecs = np.asarray(rot_vecs, dtype=np.float64)
    theta = np.linalg.norm(rot_vecs, axis=1)[:, None]
    k = np.divide(rot_vecs, theta, out=np.zeros_like(rot_vecs), where=theta != 0)
    c, s = np.cos(theta), np.sin(theta)
    return c * points + s * np.cross(k, points) + np.sum(points * k, axis=1)[:, None] * (1 - c) * k


def project(points, camera_params):
    """Convert 3-D points to 2-D by projecting onto images (reference: ba_test.py:78-89);
    ``points`` (N x 3) and ``camera_params`` (N x 9) are per observation.  Returns N x 2."""
    points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    camera_params = np.ascontiguousarray(camera_params, dtype=np.float64).reshape(-1, 9)
    n = camera_params.shape[0]
    if points.shape[0] != n:
        raise ValueError(f"operands could not be broadcast together with shapes {points.shape} {camera_params.shape}")
    if n == 0:
        return np.zeros((0, 2))
    idx = np.arange(n, dtype=np.int64)
    with Engine(camera_model="bal") as eng:
        eng.set_problem(camera_params, points, idx, idx, np.zeros((n, 2)), None)
        return eng.residuals().reshape(n, 2)


def _engine_for(n_cameras, n_points, camera_indices, point_indices, points_2d):
    def make(ci, pi, uv):
        eng = Engine(camera_model="bal", schur_mode=1)      # residuals only
        try:
            eng.set_problem(np.tile(np.r_[np.zeros(5), -1.0, 1.0, 0.0, 0.0], (n_cameras, 1)),
                            np.zeros((n_points, 3)), ci, pi, uv, None)
        except Exception:
            eng.close()
            raise
        return eng

    # (kept between calls by content, as in bundle_adjustment.fun)
    return _fun_cache.get((n_cameras, n_points), (camera_indices, point_indices, points_2d), make)


def fun(params, n_cameras, n_points, camera_indices, point_indices, points_2d):
    """Compute residuals (reference: ba_test.py:92-100).  ``params`` contains camera parameters
    (n_cameras*9) and 3-D coordinates (n_points*3); returns ``(projected - points_2d).ravel()``."""
    params = np.ascontiguousarray(params, dtype=np.float64)
    camera_params = params[:n_cameras * 9].reshape((n_cameras, 9))
    points_3d = params[n_cameras * 9:].reshape((n_points, 3))
    with _engine_lock:
        eng = _engine_for(n_cameras, n_points, camera_indices, point_indices, points_2d)
        eng.set_params(camera_params, points_3d)
        return eng.residuals()


def bundle_adjustment_sparsity(n_cameras, n_points, camera_indices, point_indices):
    """Jacobian sparsity pattern (reference: ba_test.py:103-117), shape (2*n_obs, 9*n_cameras +
    3*n_points): built directly in compressed form and handed back as the ``lil_matrix`` the
    script builds.  The engine does not need it."""
    ci = np.asarray(camera_indices).astype(np.int64)
    pi = np.asarray(point_indices).astype(np.int64)
    cols = np.concatenate([9 * ci[:, None] + np.arange(9)[None, :],
                           9 * n_cameras + 3 * pi[:, None] + np.arange(3)[None, :]], axis=1)
    cols = np.repeat(cols, 2, axis=0)
    A = csr_matrix((np.ones(cols.size, dtype=int), cols.ravel(), 12 * np.arange(2 * ci.size + 1)),
                   shape=(2 * ci.size, 9 * n_cameras + 3 * n_points))
    return A.tolil()


def solve(camera_params, points_3d, camera_indices, point_indices, points_2d,
          ftol=1e-4, xtol=1e-8, gtol=1e-8, max_nfev=None, verbose=0, **engine_options):
    """The script's ``least_squares(fun, x0, jac_sparsity=A, verbose=2, x_scale='jac', ftol=1e-4,
    method='trf')`` (reference: ba_test.py:120-126) on the GPU.  Returns ``(camera_params,
    points_3d, result dict)``; the rotation vectors come back with angles in [0, pi]."""
    opts = dict(ftol=ftol, xtol=xtol, gtol=gtol, verbose=int(verbose), camera_model="bal")
    if max_nfev:
        opts["max_nfev"] = int(max_nfev)
    opts.update(engine_options)
    with Engine(**opts) as eng:
        eng.set_problem(camera_params, points_3d, camera_indices, point_indices, points_2d, None)
        stats = eng.solve()
        cams, pts = eng.get_params()
    out = stats.as_dict()
    out["message"] = TERMINATION_MESSAGES[stats.status]
    out["success"] = stats.status > 0
    return cams, pts, out


def solve_file(file_name, **options):
    """Read a BAL file and solve it: what running the reference script does after its download."""
    cams, pts, ci, pi, uv = read_bal_data(file_name)
    return solve(cams, pts, ci, pi, uv, **options)
