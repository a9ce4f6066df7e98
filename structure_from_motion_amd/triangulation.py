"""Drop-in for the triangulation helpers of the reference's ``utils`` module on one MI355X.

Same names, arguments, return values and side effects as /root/reference/utils.py:
``triangulate_points`` (:176-211), ``calculate_reprojection_error`` (:337-352) and
``store_3Dpoints_to_views`` (:279-315).  The per-point Python loops and the OpenCV calls
(``cv.triangulatePoints``, ``cv.convertPointsFromHomogeneous``) are replaced by one batched
launch of libsfmba.so (``sfmba_triangulate`` / ``sfmba_point_errors``, see include/sfmba.h).
There is no CPU fallback: without the shared library or without a GPU these functions raise.
"""
from __future__ import annotations

import logging

import numpy as np

from .engine import Engine

__all__ = ["triangulate_points", "calculate_reprojection_error", "store_3Dpoints_to_views"]

_engine = None


def _eng():
    global _engine
    if _engine is None:
        _engine = Engine()
    return _engine


def _projection(K, R, t):
    return np.asarray(K, dtype=np.float64) @ np.hstack((np.asarray(R, dtype=np.float64).reshape(3, 3),
                                                        np.asarray(t, dtype=np.float64).reshape(3, 1)))


def triangulate_points(K, t1, R1, t2, R2, x1, x2, print_error=False):
    """
    Triangulates 3D points from 2D points and projection matrices for two views (reference:
    utils.py:176-211).  Returns the points as an (N x 1 x 3) array, the shape
    ``cv.convertPointsFromHomogeneous`` gives them, or None when fewer than 8 points come in or
    the mean reprojection error in either view is 55 pixels or more.
    """
    logging.info("Triangulating 3D points...")
    if len(x1) < 8:
        return None
    P1, P2 = _projection(K, R1, t1), _projection(K, R2, t2)
    X, error1, error2 = _eng().triangulate(P1, P2, x1, x2)
    if np.asarray(x1).dtype == np.float32:
        # cv.triangulatePoints hands its homogeneous points back in the type of the keypoints
        # (computed in double either way) and cv.convertPointsFromHomogeneous then divides in that
        # type: the unit singular vector (X, 1) / |(X, 1)| rounded to float32, times 1 / w in float32.
        # The error test sees those rounded points, as the reference's does (utils.py:197-208).
        # (Parity with OpenCV itself is unpinned: it is not installed where this was written.)
        h = np.hstack((X, np.ones((len(X), 1)))) / np.sqrt(1.0 + np.sum(X * X, axis=1))[:, None]
        h = np.where(np.isfinite(h), h, 0.0).astype(np.float32)
        w = h[:, 3]
        scale = np.where(w != 0, np.float32(1.0) / np.where(w != 0, w, np.float32(1.0)), np.float32(1.0)).astype(np.float32)
        X = (h[:, :3] * scale[:, None]).astype(np.float32)
        error1 = _eng().point_errors(P1, X, x1)
        error2 = _eng().point_errors(P2, X, x2)
    if print_error:
        print(np.mean(error1), np.mean(error2))
    if np.mean(error1) < 55 and np.mean(error2) < 55:
        return X.reshape(-1, 1, 3)
    return None


def calculate_reprojection_error(point_3D, point_2D, K, R, t):
    """Reprojection error of one 3D point and the reprojected point as a (2 x 1) array
    (reference: utils.py:337-352)."""
    point_2D = np.asarray(point_2D, dtype=np.float64).reshape((2, 1))
    point_3D = np.asarray(point_3D, dtype=np.float64).reshape((3, 1))
    P = _projection(K, R, t)
    error = float(_eng().point_errors(P, point_3D.T, point_2D.T)[0])
    h = P @ np.vstack((point_3D, [[1.0]]))
    scale = 1.0 / h[2, 0] if h[2, 0] != 0 else 1.0       # cv.convertPointsFromHomogeneous
    return error, h[:2] * scale


def store_3Dpoints_to_views(X_initial, view1, view2, K, error_threshold=40.0):
    """
    Stores 3D points to the two views that were used to triangulate them and drops the points
    whose reprojection error in either view reaches ``error_threshold`` (reference:
    utils.py:279-315).  Both error arrays come from one batched launch each.
    """
    view1_points = np.array(view1.tracked_pts[view2.id][0])
    view2_points = np.array(view2.tracked_pts[view1.id][0])
    X_initial = np.asarray(X_initial)
    pts = np.asarray(X_initial, dtype=np.float64).reshape(-1, 3)
    n = pts.shape[0]
    e1 = _eng().point_errors(_projection(K, view1.rotation, view1.translation), pts, view1_points[:n].reshape(n, 2))
    e2 = _eng().point_errors(_projection(K, view2.rotation, view2.translation), pts, view2_points[:n].reshape(n, 2))
    keep = (e1 < error_threshold) & (e2 < error_threshold)
    if keep.any():
        wp1 = np.hstack((view1_points[:n].reshape(n, -1)[keep], pts[keep]))
        wp2 = np.hstack((view2_points[:n].reshape(n, -1)[keep], pts[keep]))
        view1.world_points = np.append(view1.world_points, wp1, axis=0)
        view2.world_points = np.append(view2.world_points, wp2, axis=0)
    rm_indices = np.flatnonzero(~keep).tolist()
    view1.tracked_pts[view2.id] = (np.delete(view1.tracked_pts[view2.id][0], rm_indices, 0),
                                   np.delete(view1.tracked_pts[view2.id][1], rm_indices, 0))
    view2.tracked_pts[view1.id] = (np.delete(view2.tracked_pts[view1.id][0], rm_indices, 0),
                                   np.delete(view2.tracked_pts[view1.id][1], rm_indices, 0))
    return np.delete(X_initial, rm_indices, axis=0)
