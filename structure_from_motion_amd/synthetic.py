"""Seeded synthetic bundle-adjustment problems in the reference's array layout.

The arrays produced here are exactly the five arrays the reference's
``BundleAdjustment.view2idx`` builds and dumps to ``optimize_data.npz``
(/root/reference/bundle_adjustment.py:94-126): ``camera_params`` (N_c x 12,
row-major 3x3 rotation followed by the translation), ``points_3d`` (N_p x 3),
``camera_indices`` / ``point_indices`` (N_o, int64) and ``points_2d`` (N_o x 2),
observations sorted by point as the reference's pairs are
(/root/reference/bundle_adjustment.py:107-115).

Scene: points uniform in the cube (-1, 1)^3; cameras on a ring of radius 5 around
the origin (height jitter sigma 0.3) looking at the origin; every point has a
random centre camera and is seen by that camera and its ring neighbours inside a
window; observations are the exact projections plus N(0, noise_px) pixels; the
initial guess is the truth perturbed by a small rotation, translation and point
noise.  Intrinsics default to the reference's K.txt (/root/reference/K.txt).
"""
from __future__ import annotations

import numpy as np

# /root/reference/K.txt (same values as images/datasets/fountain/K.txt)
K_FOUNTAIN = np.array([[2759.48, 0.0, 1520.69],
                       [0.0, 2764.16, 1006.81],
                       [0.0, 0.0, 1.0]])

# Problem shapes named by the benchmark configurations: (cameras, points, observations)
SHAPES = {
    "fountain": (11, 34093, 68186),
    "ladybug": (49, 7776, 31843),
    "venice": (1778, 993923, 5001946),
    "final": (13682, 4456117, 28987644),
    "temple": (47, 1840, 3680),         # the ring of 47 views, two-view tracks (40 per pair of neighbours)
}


def _rodrigues(w):
    """Rotation matrices exp([w]x) for an (N, 3) array of rotation vectors."""
    w = np.asarray(w, dtype=np.float64)
    theta = np.linalg.norm(w, axis=1)
    small = theta < 1e-12
    safe = np.where(small, 1.0, theta)
    k = w / safe[:, None]
    c = np.cos(theta)[:, None, None]
    s = np.sin(theta)[:, None, None]
    Kx = np.zeros((w.shape[0], 3, 3))
    Kx[:, 0, 1] = -k[:, 2]
    Kx[:, 0, 2] = k[:, 1]
    Kx[:, 1, 0] = k[:, 2]
    Kx[:, 1, 2] = -k[:, 0]
    Kx[:, 2, 0] = -k[:, 1]
    Kx[:, 2, 1] = k[:, 0]
    kkT = k[:, :, None] * k[:, None, :]
    R = c * np.eye(3)[None] + (1.0 - c) * kkT + s * Kx
    R[small] = np.eye(3)
    return R


def ring_cameras(n_cameras, rng, radius=5.0, height_sigma=0.3):
    """World->camera rotations (N,3,3) and translations (N,3) of a look-at-origin ring."""
    ang = 2.0 * np.pi * np.arange(n_cameras) / n_cameras
    C = np.stack([radius * np.cos(ang), rng.normal(0.0, height_sigma, n_cameras),
                  radius * np.sin(ang)], axis=1)
    fwd = -C / np.linalg.norm(C, axis=1, keepdims=True)          # camera z axis (towards origin)
    up = np.array([0.0, 1.0, 0.0])
    right = np.cross(np.broadcast_to(up, fwd.shape), fwd)
    right /= np.linalg.norm(right, axis=1, keepdims=True)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], axis=1)                       # rows are camera axes
    t = -np.einsum("nij,nj->ni", R, C)
    return R, t


def _visibility(n_cameras, n_points, n_obs, window, rng):
    """Observation lists sorted by point: (camera_indices, point_indices)."""
    w = int(min(window, n_cameras))
    if n_obs < 2 * n_points or n_obs > w * n_points:
        raise ValueError(f"n_obs={n_obs} must lie in [2*n_points, {w}*n_points]")
    counts = np.full(n_points, 2, dtype=np.int64)
    extra = n_obs - 2 * n_points
    cap = w - 2
    while extra > 0:                                   # spread the remainder, respecting the cap
        room = np.flatnonzero(counts - 2 < cap)
        add = rng.multinomial(extra, np.full(room.size, 1.0 / room.size))
        add = np.minimum(add, cap - (counts[room] - 2))
        counts[room] += add
        extra -= int(add.sum())
    centre = rng.integers(0, n_cameras, n_points)
    # per point: a random ordering of the w in-window offsets; keep the first counts[p]
    order = np.argsort(rng.random((n_points, w)), axis=1)
    keep = np.arange(w)[None, :] < counts[:, None]
    offs = order[keep] - w // 2
    point_indices = np.repeat(np.arange(n_points, dtype=np.int64), counts)
    camera_indices = (centre[point_indices] + offs) % n_cameras
    return camera_indices.astype(np.int64), point_indices


def project_exact(R, t, X, K):
    """Pinhole projection of gathered points X (N,3) with gathered R (N,3,3), t (N,3)."""
    Y = np.einsum("nij,nj->ni", R, X) + t
    u = K[0, 0] * Y[:, 0] / Y[:, 2] + K[0, 2]
    v = K[1, 1] * Y[:, 1] / Y[:, 2] + K[1, 2]
    return np.stack([u, v], axis=1)


def make_problem(n_cameras, n_points, n_obs, seed=0, window=12, K=None, noise_px=0.5,
                 rot_sigma=0.005, trans_sigma=0.01, point_sigma=0.01, shuffle=False):
    """Build one seeded problem; returns a dict with the reference's five arrays plus
    ``K``, the shape counts and the ground truth (``true_camera_params``, ``true_points_3d``)."""
    rng = np.random.default_rng(seed)
    K = K_FOUNTAIN.copy() if K is None else np.asarray(K, dtype=np.float64)
    R, t = ring_cameras(n_cameras, rng)
    X = rng.uniform(-1.0, 1.0, (n_points, 3))
    cam_idx, pt_idx = _visibility(n_cameras, n_points, n_obs, window, rng)
    uv = project_exact(R[cam_idx], t[cam_idx], X[pt_idx], K)
    uv += rng.normal(0.0, noise_px, uv.shape)
    R0 = np.einsum("nij,njk->nik", _rodrigues(rng.normal(0.0, rot_sigma, (n_cameras, 3))), R)
    t0 = t + rng.normal(0.0, trans_sigma, t.shape)
    X0 = X + rng.normal(0.0, point_sigma, X.shape)
    if shuffle:
        perm = rng.permutation(n_obs)
        cam_idx, pt_idx, uv = cam_idx[perm], pt_idx[perm], uv[perm]
    return {
        "camera_params": np.hstack([R0.reshape(n_cameras, 9), t0]),
        "points_3d": X0,
        "camera_indices": cam_idx,
        "point_indices": pt_idx,
        "points_2d": uv,
        "K": K,
        "n_cameras": n_cameras,
        "n_points": n_points,
        "n_obs": n_obs,
        "true_camera_params": np.hstack([R.reshape(n_cameras, 9), t]),
        "true_points_3d": X,
    }


def make_shape(name, seed=0, scale=1.0, **kw):
    """One of the named benchmark shapes, optionally scaled down by ``scale`` (<= 1)."""
    nc, npts, no = SHAPES[name]
    if name == "temple":
        kw.setdefault("window", 2)          # tracks join neighbouring views only
    if scale != 1.0:
        nc = max(4, int(round(nc * scale)))
        npts = max(8, int(round(npts * scale)))
        no = int(round(no * scale))
        w = min(kw.get("window", 12), nc)
        no = min(max(no, 2 * npts), w * npts)
    return make_problem(nc, npts, no, seed=seed, **kw)


def save_npz(path, prob):
    """Write the reference's ``optimize_data.npz`` layout
    (/root/reference/bundle_adjustment.py:125-126)."""
    np.savez(path, camera_params=prob["camera_params"], points_3d=prob["points_3d"],
             camera_indices=prob["camera_indices"], point_indices=prob["point_indices"],
             points_2d=prob["points_2d"])


def load_npz(path, K=None):
    """Read an ``optimize_data.npz`` dump back into a problem dict."""
    with np.load(path, allow_pickle=False) as z:
        prob = {k: z[k] for k in ("camera_params", "points_3d", "camera_indices",
                                  "point_indices", "points_2d")}
    prob["K"] = K_FOUNTAIN.copy() if K is None else np.asarray(K, dtype=np.float64)
    prob["n_cameras"] = prob["camera_params"].shape[0]
    prob["n_points"] = prob["points_3d"].shape[0]
    prob["n_obs"] = prob["camera_indices"].shape[0]
    return prob


def _rotation_vectors(R):
    """Rotation vectors (N, 3) of rotation matrices (N, 3, 3), angles in [0, pi].  The axis comes
    from the symmetric part (k k' = (sym(R) - cos I) / (1 - cos)), which stays well conditioned up
    to pi, and its sign from the skew part."""
    R = np.asarray(R, dtype=np.float64)
    c = np.clip((np.trace(R, axis1=1, axis2=2) - 1.0) * 0.5, -1.0, 1.0)
    skew = np.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], axis=1) * 0.5
    s = np.linalg.norm(skew, axis=1)
    angle = np.arctan2(s, c)
    out = np.zeros((R.shape[0], 3))
    for n in range(R.shape[0]):
        if angle[n] < 1e-6:
            out[n] = skew[n]                      # sin(a) k ~ a k
            continue
        if c[n] > -0.5:
            out[n] = skew[n] * (angle[n] / s[n])
            continue
        S = (0.5 * (R[n] + R[n].T) - c[n] * np.eye(3)) / (1.0 - c[n])
        j = int(np.argmax(np.diag(S)))
        k = S[:, j] / np.sqrt(S[j, j])
        if np.dot(k, skew[n]) < 0:
            k = -k
        out[n] = k * angle[n]
    return out


def make_bal_problem(n_cameras, n_points, n_obs, seed=0, window=12, noise_px=0.5, rot_sigma=0.005,
                     trans_sigma=0.01, point_sigma=0.01, focal_sigma=5.0, k_sigma=0.002, shuffle=False):
    """The same ring scene for the 9-number camera of /root/reference/ba_test.py:78-89
    (rotation vector, translation, f, k1, k2; the camera looks down its negative z axis and the
    image point is -P[:2] / P[2] scaled by f (1 + k1 n + k2 n^2)).  ``camera_params`` is N_c x 9."""
    rng = np.random.default_rng(seed)
    R, t = ring_cameras(n_cameras, rng)
    flip = np.diag([1.0, -1.0, -1.0])                 # points in front of the camera get P[2] < 0
    R = np.einsum("ij,njk->nik", flip, R)
    t = t @ flip.T
    X = rng.uniform(-1.0, 1.0, (n_points, 3))
    f = rng.uniform(900.0, 1500.0, n_cameras)
    k1 = rng.normal(0.0, 0.05, n_cameras)
    k2 = rng.normal(0.0, 0.02, n_cameras)
    cam_idx, pt_idx = _visibility(n_cameras, n_points, n_obs, window, rng)

    def proj(R, t, f, k1, k2, X):
        P = np.einsum("nij,nj->ni", R[cam_idx], X[pt_idx]) + t[cam_idx]
        p = -P[:, :2] / P[:, 2:3]
        n = np.sum(p * p, axis=1)
        return p * (f[cam_idx] * (1.0 + k1[cam_idx] * n + k2[cam_idx] * n * n))[:, None]

    uv = proj(R, t, f, k1, k2, X) + rng.normal(0.0, noise_px, (n_obs, 2))
    R0 = np.einsum("nij,njk->nik", _rodrigues(rng.normal(0.0, rot_sigma, (n_cameras, 3))), R)
    cams0 = np.hstack([_rotation_vectors(R0), t + rng.normal(0.0, trans_sigma, t.shape),
                       (f + rng.normal(0.0, focal_sigma, n_cameras))[:, None],
                       (k1 + rng.normal(0.0, k_sigma, n_cameras))[:, None],
                       (k2 + rng.normal(0.0, k_sigma, n_cameras))[:, None]])
    X0 = X + rng.normal(0.0, point_sigma, X.shape)
    if shuffle:
        perm = rng.permutation(n_obs)
        cam_idx, pt_idx, uv = cam_idx[perm], pt_idx[perm], uv[perm]
    return {"camera_params": cams0, "points_3d": X0, "camera_indices": cam_idx, "point_indices": pt_idx,
            "points_2d": uv, "n_cameras": n_cameras, "n_points": n_points, "n_obs": n_obs,
            "true_camera_params": np.hstack([_rotation_vectors(R), t, f[:, None], k1[:, None], k2[:, None]]),
            "true_points_3d": X}


class SyntheticView:
    """The attributes of the reference's ImageView that bundle adjustment reads and its driver
    writes back (view.py:33,39,46; main.py:117-121)."""

    def __init__(self, vid, rotation, translation):
        self.id, self.name = vid, vid
        self.rotation = rotation
        self.translation = np.asarray(translation, dtype=np.float64).reshape(3, 1)
        self.world_points = None


class SyntheticWorldPoints:
    """world_points + correspondences of the reference's WorldPointSet (WorldPoints.py:12-15)."""

    def __init__(self, world_points, correspondences):
        self.world_points, self.correspondences = world_points, correspondences


def growing_tables(prob):
    """A two-observations-per-point problem as the reference's driver would meet it: the views in
    order and, per integrated view k >= 1, the correspondence rows (one per point, ``ViewId`` /
    ``FeatureIndex`` lists as WorldPoints.py:40-43 builds them) and initial points that become known
    with it, i.e. whose later view is k.  Returns (views, [(rows DataFrame, points)], K)."""
    import pandas as pd
    ci = np.asarray(prob["camera_indices"]).reshape(-1, 2)
    if not np.array_equal(np.asarray(prob["point_indices"]).reshape(-1, 2), np.repeat(np.arange(len(ci)), 2).reshape(-1, 2)):
        raise ValueError("growing_tables needs two observations per point, point-sorted")
    uv = np.asarray(prob["points_2d"], dtype=np.float64).reshape(-1, 2, 2)
    cp = prob["camera_params"]
    views = [SyntheticView(f"view{j:04d}", cp[j, :9].reshape(3, 3).copy(), cp[j, 9:].copy()) for j in range(len(cp))]
    later = ci.max(axis=1)
    steps = []
    for k in range(1, len(cp)):
        rows = np.flatnonzero(later == k)
        frame = pd.DataFrame({"ViewId": [[views[a].id, views[b].id] for a, b in ci[rows]],
                              "FeatureIndex": [[uv[i, 0], uv[i, 1]] for i in rows]})
        steps.append((frame, prob["points_3d"][rows]))
    return views, steps, prob["K"]
