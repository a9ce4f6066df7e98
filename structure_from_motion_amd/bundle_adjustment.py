"""Drop-in for the reference's ``bundle_adjustment`` module on one MI355X.

Same names, arguments, return values and side effects as
/root/reference/bundle_adjustment.py: ``project`` (:10), ``fun`` (:34),
``bundle_adjustment_sparsity`` (:53) and ``BundleAdjustment`` (:77) with ``view2idx`` (:94)
and ``optimize`` (:128).  The arithmetic runs in libsfmba.so (HIP kernels for gfx950, see
include/sfmba.h): ``fun``/``project`` evaluate the reprojection on the GPU and ``optimize``
replaces ``scipy.optimize.least_squares(method='trf')`` over a finite-differenced Jacobian
(:141-143) by a Levenberg-Marquardt / Schur-complement solve with analytic Jacobians.
There is no CPU fallback: without the shared library or without a GPU these functions raise.
"""
from __future__ import annotations

import atexit
import hashlib
import logging
from itertools import chain
import time

import numpy as np
import pandas as pd
from scipy.sparse import csr_matrix

from .engine import Engine, TERMINATION_MESSAGES

__all__ = ["project", "fun", "bundle_adjustment_sparsity", "BundleAdjustment", "solve_arrays", "solve_npz", "release_engines"]

_fun_cache = {"key": None, "engine": None}
# Engines of optimize(), one per set of options, kept between calls: the reference's driver makes a
# new BundleAdjustment after every image (main.py:115-116), and a handle that already holds device
# memory loads the next, slightly larger, problem in milliseconds instead of allocating afresh.
_solve_engines = {}


def _solve_engine(opts):
    key = tuple(sorted(opts.items()))
    eng = _solve_engines.get(key)
    if eng is None:
        if len(_solve_engines) >= 4:                  # a handful of option sets at most
            _solve_engines.pop(next(iter(_solve_engines))).close()
        eng = _solve_engines[key] = Engine(**opts)
    return eng


def release_engines():
    """Free the device memory held between calls by fun() and optimize()."""
    for eng in list(_solve_engines.values()):
        eng.close()
    _solve_engines.clear()
    if _fun_cache["engine"] is not None:
        _fun_cache["engine"].close()
    _fun_cache.update(key=None, engine=None)


atexit.register(release_engines)


def project(points, camera_params, K, dist=np.array([])):
    """
    Project 3D points with the given camera parameters (R matrix and t vector) and intrinsic
    matrix (reference: bundle_adjustment.py:10-31).  ``points`` (N x 3) and ``camera_params``
    (N x 12, [:9] rotation matrix entries, [9:12] translation) are per observation; ``dist``
    is ignored exactly as the reference's empty distortion is.  Returns N x 2.
    """
    points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    camera_params = np.ascontiguousarray(camera_params, dtype=np.float64).reshape(-1, 12)
    n = camera_params.shape[0]
    if points.shape[0] != n:
        raise IndexError(f"index {min(points.shape[0], n)} is out of bounds for axis 0 with size "
                         f"{min(points.shape[0], n)}")
    if n == 0:
        return np.zeros((0,))
    idx = np.arange(n, dtype=np.int64)
    with Engine() as eng:
        eng.set_problem(camera_params, points, idx, idx, np.zeros((n, 2)), K)
        return eng.residuals().reshape(n, 2)


def _digest(a):
    """Content key of an index / observation array: shape, dtype and a hash of the bytes."""
    a = np.ascontiguousarray(a)
    return (a.shape, a.dtype.str, hashlib.blake2b(a.view(np.uint8).reshape(-1), digest_size=16).digest())


def _engine_for(n_cameras, n_points, camera_indices, point_indices, points_2d, K):
    """Engine holding the observations of ``fun``'s arguments.  ``fun`` is called many times with
    the same index and observation arrays and a new ``params``; the engine of the last call is
    kept and reused when the arrays have the same CONTENT (hashing 5 M observations costs a few
    milliseconds, building the problem tens), so editing an array in place between two calls is
    seen, as it is by the reference's pure function."""
    ci = np.asarray(camera_indices)
    pi = np.asarray(point_indices)
    uv = np.asarray(points_2d)
    Ka = np.asarray(K, dtype=np.float64)
    key = (n_cameras, n_points, _digest(ci), _digest(pi), _digest(uv), Ka.tobytes())
    if _fun_cache["key"] != key:
        if _fun_cache["engine"] is not None:
            _fun_cache["engine"].close()
        _fun_cache.update(key=None, engine=None)
        eng = Engine(schur_mode=1)      # residuals only: no reduced-matrix structure is built
        eng.set_problem(np.tile(np.r_[np.eye(3).ravel(), 0, 0, 1.0], (n_cameras, 1)),
                        np.zeros((n_points, 3)), ci, pi, uv, Ka)
        _fun_cache.update(key=key, engine=eng)
    return _fun_cache["engine"]


def fun(params, n_cameras, n_points, camera_indices, point_indices, points_2d, K):
    """
    Reprojection residuals (reference: bundle_adjustment.py:34-50): ``params`` is
    [camera parameters (n_cameras*12); 3D points (n_points*3)]; returns
    ``(projected - points_2d).ravel()`` of length 2*n_obs, evaluated on the GPU.
    """
    params = np.ascontiguousarray(params, dtype=np.float64)
    camera_params = params[:n_cameras * 12].reshape((n_cameras, 12))
    points_3d = params[n_cameras * 12:].reshape((n_points, 3))
    eng = _engine_for(n_cameras, n_points, camera_indices, point_indices, points_2d, K)
    eng.set_params(camera_params, points_3d)
    return eng.residuals()


def bundle_adjustment_sparsity(n_cameras, n_points, camera_indices, point_indices):
    """
    Jacobian sparsity pattern (reference: bundle_adjustment.py:53-74): a ``lil_matrix`` of
    shape (2*n_obs, 12*n_cameras + 3*n_points) with 12 camera and 3 point columns set in both
    rows of every observation.  The engine does not need it (its block structure is
    implicit); it is produced for callers that still hand it to SciPy.
    """
    cam = np.asarray(camera_indices, dtype=np.int64).ravel()
    pt = np.asarray(point_indices, dtype=np.int64).ravel()
    n_obs = cam.size
    # every row holds its camera's 12 columns, then its point's 3: written down as CSR in one go
    # and converted, instead of 30 fancy-index assignments into a LIL matrix
    cols = np.empty((n_obs, 15), dtype=np.int64)
    cols[:, :12] = 12 * cam[:, None] + np.arange(12)
    cols[:, 12:] = 12 * n_cameras + 3 * pt[:, None] + np.arange(3)
    cols = np.repeat(cols, 2, axis=0).ravel()
    A = csr_matrix((np.ones(cols.size, dtype=int), cols, 15 * np.arange(2 * n_obs + 1)),
                   shape=(2 * n_obs, 12 * n_cameras + 3 * n_points))
    return A.tolil()


class BundleAdjustment:
    """Reference: bundle_adjustment.py:77-150.  Same constructor, attributes and results."""

    def __init__(self, wpSet, K, dist, completed_views, ftol=1e-4, xtol=1e-12, gtol=1e-8,
                 max_nfev=None, verbose=2, **engine_options):
        self.completed_views = completed_views
        self.wpSet = wpSet
        self.points_3d = wpSet.world_points
        self.points_2d = []
        self.point_indices = []
        self.camera_indices = []
        self.view_idx = {}
        self.camera_params = []
        self.focal_len = (K[0, 0] + K[1, 1]) / 2
        self.dist = dist[0][:2]
        self.K = K
        self.correspondences = wpSet.correspondences
        self.n_cameras = None
        self.n_points = None
        # solver settings: the literals of bundle_adjustment.py:141 unless overridden
        self.ftol, self.xtol, self.gtol = ftol, xtol, gtol
        self.max_nfev, self.verbose = max_nfev, verbose
        self.engine_options = engine_options
        self.result = None
        self.reprojection_error = None      # per view, filled by optimize()
        self.reprojection_count = None
        self._poses = self._points_3d = None

    def view2idx(self):
        """
        Takes in a list of views and converts them to indices.  For each 2D point a view index
        is assigned (reference: bundle_adjustment.py:94-126, without the per-row Python loop).
        """
        for view in self.completed_views:
            if view.id not in self.view_idx:
                self.view_idx[view.id] = len(self.view_idx)
            rot_vec = np.squeeze(view.rotation)
            params = np.concatenate((rot_vec, view.translation.reshape((1, 3))), axis=None).tolist()
            self.camera_params.append(params)
        self.camera_params = np.array(self.camera_params)

        corr = self.correspondences
        n_rows = len(corr)
        if n_rows:
            # both columns hold one 2-element list per row; the rows are flattened by the
            # interpreter's own iterators, there is no Python-level loop over them
            ids, feats = corr["ViewId"].tolist(), corr["FeatureIndex"].tolist()
            if all(map((2).__eq__, map(len, ids))) and all(map((2).__eq__, map(len, feats))):
                # (an unknown id raises KeyError, as the reference's dictionary look-up does)
                self.camera_indices = np.fromiter(map(self.view_idx.__getitem__, chain.from_iterable(ids)),
                                                  dtype=np.int64, count=2 * n_rows)
                self.points_2d = np.array(list(chain.from_iterable(feats)), dtype=np.float64).reshape(2 * n_rows, -1)
            else:                                                           # longer rows: the first two entries count
                view_ids = np.array(ids, dtype=object).reshape(n_rows, -1)[:, :2].ravel()
                idx = pd.Series(view_ids).map(self.view_idx)
                if idx.isna().any():
                    raise KeyError(view_ids[int(np.flatnonzero(idx.isna().to_numpy())[0])])
                self.camera_indices = idx.to_numpy(dtype=np.int64)
                try:
                    xy = np.asarray(feats, dtype=np.float64)                # (rows, >= 2, 2)
                    self.points_2d = np.ascontiguousarray(xy[:, :2]).reshape(2 * n_rows, -1)
                except ValueError:                                          # ragged rows
                    self.points_2d = np.array([np.asarray(f[k], dtype=np.float64) for f in feats for k in (0, 1)])
            self.point_indices = np.repeat(np.asarray(corr.index), 2)
        else:
            self.camera_indices = np.array([])
            self.point_indices = np.array([])
            self.points_2d = np.array([])
        self.points_3d = np.array(self.points_3d)
        self.n_points = self.points_3d.shape[0]
        self.n_cameras = self.camera_params.shape[0]
        logging.info(f"Number of views processed: {self.n_cameras}.")
        logging.info(f"Number of 3D points processed: {self.n_points}.")
        np.savez('optimize_data', camera_params=self.camera_params, points_3d=self.points_3d,
                 camera_indices=self.camera_indices, point_indices=self.point_indices, points_2d=self.points_2d)

    def optimize(self):
        """
        Performs optimization on the reprojection error by updating poses and 3D points.
        :return poses: (N x 12) array, [:9] rotation matrix entries, [9:12] translation
        :return points_3d: optimized 3D points
        """
        self.view2idx()
        print(len(self.camera_params.ravel()), len(self.points_3d.ravel()))
        opts = dict(ftol=self.ftol, xtol=self.xtol, gtol=self.gtol, verbose=int(self.verbose))
        if self.max_nfev:
            opts["max_nfev"] = int(self.max_nfev)
        opts.update(self.engine_options)
        eng = _solve_engine(opts)
        eng.set_problem(self.camera_params, self.points_3d, self.camera_indices,
                        self.point_indices, self.points_2d, self.K)
        t0 = time.time()
        stats = eng.solve()
        t1 = time.time()
        poses, points_3d = eng.get_params()
        # per-view mean reprojection error at the optimum (what main.py:123 prints per view
        # through ImageView.reproject_view), computed where the data already is
        self.reprojection_error, self.reprojection_count = eng.reprojection_errors()
        logging.info(f"Optimized {self.n_points} in {t1-t0} seconds.")
        self.result = stats.as_dict()
        self.result["message"] = TERMINATION_MESSAGES[stats.status]
        self.result["success"] = stats.status > 0
        self._poses, self._points_3d = poses, points_3d
        return poses, points_3d

    def write_back(self, print_error=True):
        """
        What the reference's driver does with the result of ``optimize()`` (main.py:116-123):
        the refined points go to ``wpSet.world_points``; every view receives its rotation (3 x 3)
        and translation (3 x 1), its ``world_points`` table (n x 5 rows ``[u, v, X, Y, Z]``, one per
        correspondence that names the view, in correspondence order: ``ImageView.update_world_points``,
        view.py:121-132) and, with ``print_error``, the line ``ImageView.reproject_view`` prints
        (view.py:103-119).  The tables are cut out of the packed arrays without a loop over the
        correspondences, and the per-view mean errors are the ones the GPU computed in ``optimize()``.
        Returns the per-view mean reprojection errors.
        """
        if getattr(self, "_poses", None) is None:
            raise RuntimeError("write_back: call optimize() first")
        poses, points_3d = self._poses, self._points_3d
        self.wpSet.world_points = points_3d
        ci2 = np.asarray(self.camera_indices).reshape(-1, 2)
        uv2 = np.asarray(self.points_2d, dtype=np.float64).reshape(-1, 2, 2)
        for j, view in enumerate(self.completed_views):
            view.rotation = poses[j, :9].reshape(3, 3)
            view.translation = poses[j, 9:].reshape(3, 1)
            k = self.view_idx[view.id]
            first = ci2[:, 0] == k
            second = (ci2[:, 1] == k) & ~first           # a row naming the view twice counts once, as its first entry
            rows = np.flatnonzero(first | second)
            uv = np.where(first[rows, None], uv2[rows, 0], uv2[rows, 1])
            view.world_points = np.hstack((uv, points_3d[rows])) if rows.size else np.empty((0, 5))
            if print_error:
                print(f"Reprojection error for image{getattr(view, 'name', view.id)} is {self.reprojection_error[k]}")
        return self.reprojection_error


def solve_npz(path, K, **options):
    """File-level entry point: solve an ``optimize_data.npz`` dump (the five arrays written at
    bundle_adjustment.py:125-126) and return ``(poses, points_3d, stats dict)``."""
    with np.load(path, allow_pickle=False) as z:
        arrays = [z[k] for k in ("camera_params", "points_3d", "camera_indices", "point_indices", "points_2d")]
    return solve_arrays(*arrays, K, **options)


def solve_arrays(camera_params, points_3d, camera_indices, point_indices, points_2d, K, **options):
    """Array-level entry point: the five arrays of ``optimize_data.npz``
    (bundle_adjustment.py:125-126) in, ``(poses, points_3d, stats dict)`` out."""
    with Engine(**options) as eng:
        eng.set_problem(camera_params, points_3d, camera_indices, point_indices, points_2d, K)
        stats = eng.solve()
        poses, pts = eng.get_params()
    out = stats.as_dict()
    out["message"] = TERMINATION_MESSAGES[stats.status]
    out["success"] = stats.status > 0
    return poses, pts, out
