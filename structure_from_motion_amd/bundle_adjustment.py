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
import gc
from concurrent.futures import ThreadPoolExecutor
import logging
from itertools import chain
import operator
import os
import threading
import time
import weakref

import numpy as np
import pandas as pd
from scipy.sparse import csr_matrix

from .engine import Engine, TERMINATION_MESSAGES

__all__ = ["project", "fun", "bundle_adjustment_sparsity", "BundleAdjustment", "solve_arrays", "solve_npz", "release_engines"]

# One lock for the engines kept between calls (below): a handle holds one problem at a time, so
# set_problem .. solve .. get_params of one call must not interleave with another thread's.
_engine_lock = threading.RLock()
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
    with _engine_lock:
        _release_engines()


def _release_engines():
    for eng in list(_solve_engines.values()):
        eng.close()
    _solve_engines.clear()
    for cache in _observation_caches:       # fun() of this module and of bal.py
        cache.close()


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


_compare_pool = None
_observation_caches = []


def _same_arrays(pairs, chunk=1 << 19):
    """True when every (a, b) of ``pairs`` holds equal contents: a full comparison at memory speed
    (a few threads on slices of half a million elements; 160 MB of index and observation arrays
    of 5 M observations take about 10 ms, the arrays of 68 k observations 0.2 ms)."""
    global _compare_pool
    jobs = []
    for a, b in pairs:
        if a.shape != b.shape or a.dtype != b.dtype:
            return False
        a, b = a.reshape(-1), b.reshape(-1)
        jobs.extend((a[s:s + chunk], b[s:s + chunk]) for s in range(0, a.size, chunk))
    if len(jobs) <= 2:
        return all(np.array_equal(x, y) for x, y in jobs)
    if _compare_pool is None:
        _compare_pool = ThreadPoolExecutor(4, thread_name_prefix="sfmba-compare")
    return all(_compare_pool.map(lambda j: np.array_equal(j[0], j[1]), jobs))


class _ObservationCache:
    """The engine of the last ``fun`` call together with PRIVATE COPIES of the index and
    observation arrays it was loaded with.  ``fun`` is called many times with the same arrays and a
    new ``params``; every call compares the arguments with the copies (memory speed, see
    ``_same_arrays``) and reuses the engine when they are equal, so an array edited in place between
    two calls is seen, as it is by the reference's pure function."""

    def __init__(self):
        self.engine = self.meta = self.copies = None
        _observation_caches.append(self)

    def get(self, meta, arrays, make_engine):
        arrays = [np.ascontiguousarray(a) for a in arrays]
        if self.engine is None or self.meta != meta or not _same_arrays(list(zip(arrays, self.copies))):
            self.close()
            eng = make_engine(*arrays)
            self.engine, self.meta, self.copies = eng, meta, [a.copy() for a in arrays]
        return self.engine

    def close(self):
        if self.engine is not None:
            self.engine.close()
        self.engine = self.meta = self.copies = None


_fun_cache = _ObservationCache()


def _engine_for(n_cameras, n_points, camera_indices, point_indices, points_2d, K):
    """Engine holding the observations of ``fun``'s arguments (kept between calls, by content)."""
    Ka = np.asarray(K, dtype=np.float64)

    def make(ci, pi, uv):
        eng = Engine(schur_mode=1)      # residuals only: no reduced-matrix structure is built
        try:
            eng.set_problem(np.tile(np.r_[np.eye(3).ravel(), 0, 0, 1.0], (n_cameras, 1)),
                            np.zeros((n_points, 3)), ci, pi, uv, Ka)
        except Exception:
            eng.close()
            raise
        return eng

    return _fun_cache.get((n_cameras, n_points, Ka.tobytes()), (camera_indices, point_indices, points_2d), make)


def fun(params, n_cameras, n_points, camera_indices, point_indices, points_2d, K):
    """
    Reprojection residuals (reference: bundle_adjustment.py:34-50): ``params`` is
    [camera parameters (n_cameras*12); 3D points (n_points*3)]; returns
    ``(projected - points_2d).ravel()`` of length 2*n_obs, evaluated on the GPU.
    """
    params = np.ascontiguousarray(params, dtype=np.float64)
    camera_params = params[:n_cameras * 12].reshape((n_cameras, 12))
    points_3d = params[n_cameras * 12:].reshape((n_points, 3))
    with _engine_lock:
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


def _pack_rows(ids, feats, view_idx):
    """(camera index, image point) of the first two entries of every correspondence row
    (reference: bundle_adjustment.py:107-115).  An unknown view id raises KeyError, as the
    reference's dictionary look-up does."""
    n_rows = len(ids)
    if all(map((2).__eq__, map(len, ids))) and all(map((2).__eq__, map(len, feats))):
        # rows are flattened by the interpreter's own iterators, no Python-level loop over them
        cam = np.fromiter(map(view_idx.__getitem__, chain.from_iterable(ids)), dtype=np.int64, count=2 * n_rows)
        uv = np.array(list(chain.from_iterable(feats)), dtype=np.float64).reshape(2 * n_rows, -1)
        return cam, uv
    # longer rows: the first two entries count
    view_ids = np.array(ids, dtype=object).reshape(n_rows, -1)[:, :2].ravel() if len(set(map(len, ids))) == 1 else \
        np.array([r[k] for r in ids for k in (0, 1)], dtype=object)
    idx = pd.Series(view_ids).map(view_idx)
    if idx.isna().any():
        raise KeyError(view_ids[int(np.flatnonzero(idx.isna().to_numpy())[0])])
    cam = idx.to_numpy(dtype=np.int64)
    try:
        xy = np.asarray(feats, dtype=np.float64)                # (rows, >= 2, 2)
        uv = np.ascontiguousarray(xy[:, :2]).reshape(2 * n_rows, -1)
    except ValueError:                                          # ragged rows
        uv = np.array([np.asarray(f[k], dtype=np.float64) for f in feats for k in (0, 1)])
    return cam, uv


class _PackedRows:
    __slots__ = ("ids", "feats", "views", "cam", "uv")


# wpSet -> what view2idx packed for it last time
_packed = weakref.WeakKeyDictionary()


def _remember_rows(wpSet, kept, ids, feats, view_idx, cam, uv):
    """Keeps the packed arrays and, per row, a shallow COPY of its two lists (the copies of the rows
    taken over from the last call are taken over as well)."""
    m = 0 if kept is None else len(kept.ids)
    rec = _PackedRows()
    rec.ids = (kept.ids if m else []) + list(map(list, ids[m:]))
    rec.feats = (kept.feats if m else []) + list(map(list, feats[m:]))
    rec.views, rec.cam, rec.uv = tuple(view_idx), cam, uv
    try:
        _packed[wpSet] = rec
    except TypeError:               # an object that cannot be referenced weakly: nothing is kept
        pass


def _first_difference(rows, copies):
    """Index of the first row whose list no longer equals its copy (entries compared by identity
    first, as list comparison does; two different keypoint arrays count as different)."""
    for i, (a, b) in enumerate(zip(rows, copies)):
        try:
            if a != b:
                return i
        except ValueError:          # elementwise comparison of two distinct arrays
            return i
    return len(copies)


def _reusable_rows(wpSet, ids, feats, view_idx):
    """The record of the last call on ``wpSet`` cut down to the rows that may be taken over, or None.

    Verified, not assumed: the views known then are numbered as they were; every old row's lists
    equal the shallow copies taken when they were packed - same length, and entry by entry the very
    same keypoint-array objects and equal view ids (so a row re-bound through ``.at``, an entry
    replaced inside a list, or a list extended as in WorldPoints.py:54-57 is packed afresh, from the
    first such row on); and 192 rows spread over the table plus its last 64 pack to the same
    numbers again.  What this cannot see is a number overwritten INSIDE a keypoint array that was
    packed before, in a row outside that sample; a caller who does that sets ``SFMBA_REPACK=1``."""
    try:
        rec = _packed.get(wpSet)
    except TypeError:
        return None
    if rec is None or os.environ.get("SFMBA_REPACK", "0") != "0":
        return None
    m = min(len(rec.ids), len(ids))
    if m == 0 or rec.uv.ndim != 2 or tuple(view_idx)[:len(rec.views)] != rec.views:
        return None
    try:
        unchanged = ids[:m] == rec.ids[:m] and feats[:m] == rec.feats[:m]      # two list comparisons in C
    except ValueError:
        unchanged = False
    if not unchanged:
        m = min(_first_difference(ids[:m], rec.ids), _first_difference(feats[:m], rec.feats))
        if m == 0:
            return None
    sample = np.unique(np.concatenate((np.linspace(0, m - 1, min(m, 192)).astype(np.int64), np.arange(max(0, m - 64), m))))
    try:
        cam, uv = _pack_rows([ids[i] for i in sample], [feats[i] for i in sample], view_idx)
    except KeyError:
        return None
    pos = np.stack((2 * sample, 2 * sample + 1), axis=1).ravel()
    if uv.shape[1:] != rec.uv.shape[1:] or not (np.array_equal(cam, rec.cam[pos]) and np.array_equal(uv, rec.uv[pos])):
        return None
    out = _PackedRows()
    out.ids, out.feats, out.views, out.cam, out.uv = rec.ids[:m], rec.feats[:m], rec.views, rec.cam[:2 * m], rec.uv[:2 * m]
    return out


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
        self._dump_thread = None
        self.timings = {}                   # seconds per stage of the last optimize() / write_back()

    def view2idx(self, _defer_dump=False):
        """
        Takes in a list of views and converts them to indices.  For each 2D point a view index
        is assigned (reference: bundle_adjustment.py:94-126, without the per-row Python loop).

        The reference's driver calls this after every integrated image on a correspondence table
        that only grew at its tail (main.py:103-116, WorldPoints.py:40-43,62): the packed arrays of
        the last call on the same ``wpSet`` are reused for the rows that are still the very objects
        packed then (see ``_reusable_rows``), and only the new rows are packed.
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
        # (tens of thousands of row lists are walked and a few thousand created: the cyclic garbage
        # collector would traverse all of them several times over, 5 ms per call at 34 k rows, and
        # find nothing)
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            self._pack_correspondences(corr, n_rows)
        finally:
            if gc_was_on:
                gc.enable()
        self.points_3d = np.array(self.points_3d)
        self.n_points = self.points_3d.shape[0]
        self.n_cameras = self.camera_params.shape[0]
        logging.info(f"Number of views processed: {self.n_cameras}.")
        logging.info(f"Number of 3D points processed: {self.n_points}.")
        # (the file name is resolved now; inside optimize() the file is written beside the solve and
        # is complete when optimize() returns)
        dump = dict(camera_params=self.camera_params, points_3d=self.points_3d, camera_indices=self.camera_indices,
                    point_indices=self.point_indices, points_2d=self.points_2d)
        path = os.path.abspath('optimize_data')
        if _defer_dump:
            self._dump_thread = threading.Thread(target=np.savez, args=(path,), kwargs=dump, name="sfmba-dump")
            self._dump_thread.start()
        else:
            np.savez(path, **dump)

    def _pack_correspondences(self, corr, n_rows):
        if n_rows:
            # both columns hold one list per row
            ids, feats = corr["ViewId"].tolist(), corr["FeatureIndex"].tolist()
            kept = _reusable_rows(self.wpSet, ids, feats, self.view_idx)
            m = 0 if kept is None else len(kept.ids)
            cam_new, uv_new = _pack_rows(ids[m:], feats[m:], self.view_idx) if m < n_rows else (kept.cam[:0], kept.uv[:0])
            if m and (uv_new.ndim != 2 or uv_new.shape[1:] != kept.uv.shape[1:]):    # another row shape: everything afresh
                m, (cam_new, uv_new) = 0, _pack_rows(ids, feats, self.view_idx)
            if m:
                self.camera_indices = np.concatenate((kept.cam, cam_new))
                self.points_2d = np.concatenate((kept.uv, uv_new))
            else:
                self.camera_indices, self.points_2d = cam_new, uv_new
            _remember_rows(self.wpSet, kept if m else None, ids, feats, self.view_idx, self.camera_indices, self.points_2d)
            self.point_indices = np.repeat(np.asarray(corr.index), 2)
        else:
            self.camera_indices = np.array([])
            self.point_indices = np.array([])
            self.points_2d = np.array([])
    def optimize(self):
        """
        Performs optimization on the reprojection error by updating poses and 3D points.
        :return poses: (N x 12) array, [:9] rotation matrix entries, [9:12] translation
        :return points_3d: optimized 3D points
        """
        t0 = time.perf_counter()
        self.view2idx(_defer_dump=True)
        self.timings = {"view2idx": time.perf_counter() - t0}
        try:
            return self._optimize_packed()
        finally:
            t0 = time.perf_counter()
            self._dump_thread.join()            # optimize_data.npz is complete when this returns
            self.timings["dump_wait"] = time.perf_counter() - t0

    def _optimize_packed(self):
        print(len(self.camera_params.ravel()), len(self.points_3d.ravel()))
        opts = dict(ftol=self.ftol, xtol=self.xtol, gtol=self.gtol, verbose=int(self.verbose))
        if self.max_nfev:
            opts["max_nfev"] = int(self.max_nfev)
        opts.update(self.engine_options)
        with _engine_lock:
            eng = _solve_engine(opts)
            try:
                ts = time.perf_counter()
                eng.set_problem(self.camera_params, self.points_3d, self.camera_indices,
                                self.point_indices, self.points_2d, self.K)
                t0 = time.time()
                stats = eng.solve()
                t1 = time.time()
                poses, points_3d = eng.get_params()
                # per-view mean reprojection error at the optimum (what main.py:123 prints per view
                # through ImageView.reproject_view), computed where the data already is
                self.reprojection_error, self.reprojection_count = eng.reprojection_errors()
                self.timings.update(set_problem=t0 - time.time() + time.perf_counter() - ts, solve=t1 - t0,
                                    read_back=time.time() - t1)
            except Exception:
                # a handle that failed half-way is not kept for the next call
                _solve_engines.pop(tuple(sorted(opts.items())), None)
                eng.close()
                raise
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
        t_start = time.perf_counter()
        poses, points_3d = self._poses, self._points_3d
        self.wpSet.world_points = points_3d
        ci2 = np.asarray(self.camera_indices, dtype=np.int64).reshape(-1, 2)
        uv_flat = np.asarray(self.points_2d, dtype=np.float64).reshape(-1, 2)        # entry 2 * row + slot
        # every (row, slot) entry grouped by its view with one stable sort (entries of a view then
        # come in correspondence order); a row naming the view twice counts once, as its first entry
        entry_view = ci2.ravel()
        entries = np.arange(entry_view.size)
        if len(ci2):
            keep = np.ones(entry_view.size, dtype=bool)
            keep[1::2] = ci2[:, 1] != ci2[:, 0]
            entries, entry_view = entries[keep], entry_view[keep]
        n_views = len(self.view_idx)
        key = entry_view.astype(np.uint16) if n_views < 65536 else entry_view    # (16-bit keys: a radix sort)
        order = entries[np.argsort(key, kind="stable")]
        bounds = np.concatenate(([0], np.cumsum(np.bincount(entry_view, minlength=n_views))))
        # one table for all views, filled a column at a time (gathers from contiguous vectors are
        # several times faster than row gathers); each view receives its rows of it
        table = np.empty((order.size, 5))
        if order.size:
            row = order >> 1
            for c, column in enumerate(np.ascontiguousarray(uv_flat.T)):
                table[:, c] = column.take(order)
            for c, column in enumerate(np.ascontiguousarray(np.asarray(points_3d, dtype=np.float64).T)):
                table[:, 2 + c] = column.take(row)
        for j, view in enumerate(self.completed_views):
            view.rotation = poses[j, :9].reshape(3, 3)
            view.translation = poses[j, 9:].reshape(3, 1)
            k = self.view_idx[view.id]
            view.world_points = table[bounds[k]:bounds[k + 1]]
            if print_error:
                print(f"Reprojection error for image{getattr(view, 'name', view.id)} is {self.reprojection_error[k]}")
        self.timings["write_back"] = time.perf_counter() - t_start
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
