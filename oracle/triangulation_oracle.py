"""CPU oracle of the two-view triangulation step.  TEST INFRASTRUCTURE ONLY: imported by tests/,
oracle/gen_golden.py and nothing else; the product path never touches it.

Restates /root/reference/utils.py:176-211 (``triangulate_points``) and :337-352
(``calculate_reprojection_error``) in NumPy.  The reference calls OpenCV 4.5.2.54
(``cv.triangulatePoints``, ``cv.convertPointsFromHomogeneous``), which is not installed here, and
holds no test or fixture for these functions: PARITY UNPINNED at the OpenCV boundary.  What is
restated is OpenCV's published algorithm: per correspondence the 4x4 matrix

    [ x1 P1[2] - P1[0] ;  y1 P1[2] - P1[1] ;  x2 P2[2] - P2[0] ;  y2 P2[2] - P2[1] ]

is decomposed by SVD and the right singular vector of the smallest singular value is the
homogeneous point; ``convertPointsFromHomogeneous`` divides by the last coordinate, or by 1 when
it is exactly zero.
"""
from __future__ import annotations

import numpy as np


def projection(K, R, t):
    """P = K [R | t] as the reference builds it (utils.py:194-195)."""
    return np.asarray(K, float) @ np.hstack((np.asarray(R, float).reshape(3, 3), np.asarray(t, float).reshape(3, 1)))


def dehomogenize(h):
    """cv.convertPointsFromHomogeneous on an (N, k) array: divide by the last column, by 1 where it is 0."""
    h = np.asarray(h, float)
    w = h[:, -1]
    scale = np.where(w != 0, 1.0 / np.where(w != 0, w, 1.0), 1.0)
    return h[:, :-1] * scale[:, None]


def dehomogenize_f32(h):
    """The same for a float32 array, in float32 arithmetic as OpenCV does for CV_32F input:
    scale = 1 / w (or 1), every coordinate times scale."""
    h = np.asarray(h, np.float32)
    w = h[:, -1]
    scale = np.where(w != 0, np.float32(1.0) / np.where(w != 0, w, np.float32(1.0)), np.float32(1.0)).astype(np.float32)
    return (h[:, :-1] * scale[:, None]).astype(np.float32)


def triangulate(P1, P2, x1, x2, single=False):
    """cv.triangulatePoints + convertPointsFromHomogeneous: (N, 3) points.  ``single``: the keypoints
    were float32, so cv.triangulatePoints hands back its (double-computed) homogeneous points rounded
    to float32 and the division happens in float32."""
    P1 = np.asarray(P1, float).reshape(3, 4)
    P2 = np.asarray(P2, float).reshape(3, 4)
    x1 = np.asarray(x1, float).reshape(-1, 2)
    x2 = np.asarray(x2, float).reshape(-1, 2)
    A = np.stack([x1[:, 0:1] * P1[2] - P1[0], x1[:, 1:2] * P1[2] - P1[1],
                  x2[:, 0:1] * P2[2] - P2[0], x2[:, 1:2] * P2[2] - P2[1]], axis=1)       # (N, 4, 4)
    _, _, Vt = np.linalg.svd(A)
    if single:
        return dehomogenize_f32(Vt[:, 3, :].astype(np.float32))
    return dehomogenize(Vt[:, 3, :])


def reprojection_errors(P, X, x):
    """calculate_reprojection_error (utils.py:337-352) for arrays: |x - dehom(P [X; 1])|."""
    P = np.asarray(P, float).reshape(3, 4)
    X = np.asarray(X, float).reshape(-1, 3)
    x = np.asarray(x, float).reshape(-1, 2)
    h = X @ P[:, :3].T + P[:, 3]
    return np.linalg.norm(x - dehomogenize(h), axis=1)


def triangulate_points(K, t1, R1, t2, R2, x1, x2):
    """The reference function's result: (N, 1, 3) points or None (utils.py:176-211)."""
    if len(x1) < 8:
        return None
    P1, P2 = projection(K, R1, t1), projection(K, R2, t2)
    # cv.triangulatePoints creates its output in the type of projPoints1 (the arithmetic is double
    # either way): float32 keypoints give float32 homogeneous points, de-homogenised in float32, and
    # the error test below (utils.py:199-208) sees those rounded points
    X = triangulate(P1, P2, x1, x2, single=np.asarray(x1).dtype == np.float32)
    if reprojection_errors(P1, X, x1).mean() < 55 and reprojection_errors(P2, X, x2).mean() < 55:
        return X.reshape(-1, 1, 3)
    return None


def two_view_problem(n, seed=0, noise_px=0.5, K=None):
    """Seeded two-view scene: points in the unit cube, two ring cameras 0.6 rad apart."""
    from structure_from_motion_amd import synthetic as syn
    rng = np.random.default_rng(seed)
    K = syn.K_FOUNTAIN if K is None else np.asarray(K, float)
    R, t = syn.ring_cameras(10, rng)
    R1, t1, R2, t2 = R[0], t[0], R[1], t[1]
    X = rng.uniform(-1.0, 1.0, (n, 3))
    x1 = syn.project_exact(np.broadcast_to(R1, (n, 3, 3)), np.broadcast_to(t1, (n, 3)), X, K) + rng.normal(0, noise_px, (n, 2))
    x2 = syn.project_exact(np.broadcast_to(R2, (n, 3, 3)), np.broadcast_to(t2, (n, 3)), X, K) + rng.normal(0, noise_px, (n, 2))
    return dict(K=K, R1=R1, t1=t1, R2=R2, t2=t2, x1=x1, x2=x2, X_true=X)


def store_3Dpoints_to_views(X_initial, view1, view2, K, error_threshold=40.0):
    """utils.py:279-315 restated with its per-point loop (small cases only)."""
    view1_points = np.array(view1.tracked_pts[view2.id][0])
    view2_points = np.array(view2.tracked_pts[view1.id][0])
    P1 = projection(K, view1.rotation, view1.translation)
    P2 = projection(K, view2.rotation, view2.translation)
    rm_indices = []
    for i, point_3d in enumerate(X_initial):
        error1 = reprojection_errors(P1, point_3d.reshape(1, 3), view1_points[i].reshape(1, 2))[0]
        error2 = reprojection_errors(P2, point_3d.reshape(1, 3), view2_points[i].reshape(1, 2))[0]
        if error1 < error_threshold and error2 < error_threshold:
            wp1 = np.concatenate((view1_points[i], point_3d.reshape(1, 3)), axis=None)
            view1.world_points = np.append(view1.world_points, [wp1], axis=0)
            wp2 = np.concatenate((view2_points[i], point_3d.reshape(1, 3)), axis=None)
            view2.world_points = np.append(view2.world_points, [wp2], axis=0)
        else:
            rm_indices.append(i)
    view1.tracked_pts[view2.id] = (np.delete(view1.tracked_pts[view2.id][0], rm_indices, 0),
                                   np.delete(view1.tracked_pts[view2.id][1], rm_indices, 0))
    view2.tracked_pts[view1.id] = (np.delete(view2.tracked_pts[view1.id][0], rm_indices, 0),
                                   np.delete(view2.tracked_pts[view1.id][1], rm_indices, 0))
    return np.delete(X_initial, rm_indices, axis=0)
