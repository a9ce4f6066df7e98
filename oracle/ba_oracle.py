"""CPU oracle for the bundle-adjustment hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the shipped package never does.

PARITY UNPINNED at the OpenCV boundary: the reference's residual function calls
cv2.Rodrigues / cv2.projectPoints (/root/reference/bundle_adjustment.py:24,28) and cv2 is
not installed here, the reference has no tests or golden vectors for this path, and its
module cannot be imported without cv2.  The projection below is therefore a restatement of
OpenCV 4.5.2's documented algorithm (opencv-python==4.5.2.54, requirements.txt:10), not a
run of it.  One half of it IS pinned by the reference's own code: the vector -> matrix map below
equals the `rotated` outputs of the reference's ba_test.rotate (ba_test.py:62-75, the same
Rodrigues formula) stored in tests/golden/bal_ref_*.npz to 1e-15, zero vector included
(tests/test_oracle.py::test_vector_to_matrix_rotation_pinned_by_the_reference_script).  What stays
unpinned after that is cv2's matrix -> vector branch logic (polar factor, small-angle snap, the
half-turn branch) and the two-line pinhole map.  The optimiser boundary is NOT restated: `solve_reference` drives the real
scipy.optimize.least_squares with the reference's own keyword arguments
(/root/reference/bundle_adjustment.py:141-143); the installed SciPy is 1.15.3, the
reference pins scipy~=1.5.4 (requirements.txt:20).

What is restated, and from where:
  project()   /root/reference/bundle_adjustment.py:10-31
  fun()       /root/reference/bundle_adjustment.py:34-50
  bundle_adjustment_sparsity()  /root/reference/bundle_adjustment.py:53-74
  solve_reference()             /root/reference/bundle_adjustment.py:136-150
  cv2.Rodrigues (matrix -> vector): SVD R = U W Vt, R <- U Vt, axis from the skew part,
      angle acos((trace-1)/2); when sin(angle) < 1e-5 and cos(angle) > 0 the vector is
      set to exactly zero (OpenCV calib3d cvRodrigues2).
  cv2.projectPoints with empty distortion: Y = R(rvec) X + t, z' = (Y2 != 0 ? 1/Y2 : 1),
      u = fx*Y0*z' + cx, v = fy*Y1*z' + cy; the skew entry K[0,1] is ignored.
"""
from __future__ import annotations

import time

import numpy as np
from scipy.optimize import least_squares
from scipy.sparse import csr_matrix, lil_matrix

REFERENCE_KWARGS = dict(x_scale="jac", ftol=1e-4, method="trf", xtol=1e-12)


# --------------------------------------------------------------------------- rotations
def rodrigues_matrix_to_vector(R):
    """cv2.Rodrigues for a 3x3 input (/root/reference/bundle_adjustment.py:24)."""
    U, _, Vt = np.linalg.svd(np.asarray(R, dtype=np.float64).reshape(3, 3))
    R = U @ Vt
    r = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.sqrt(np.dot(r, r) * 0.25)
    c = min(1.0, max(-1.0, (np.trace(R) - 1.0) * 0.5))
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        t = (R[0, 0] + 1) * 0.5
        rx = np.sqrt(max(t, 0.0))
        t = (R[1, 1] + 1) * 0.5
        ry = np.sqrt(max(t, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0)
        t = (R[2, 2] + 1) * 0.5
        rz = np.sqrt(max(t, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)
        if abs(rx) < abs(ry) and abs(rx) < abs(rz) and ((R[1, 2] > 0) != (ry * rz > 0)):
            rz = -rz
        v = np.array([rx, ry, rz])
        return v * (theta / np.linalg.norm(v))
    return r * (theta / (2.0 * s))


def rodrigues_vector_to_matrix(rvec):
    """cv2.Rodrigues for a 3-vector input (used inside cv2.projectPoints)."""
    r = np.asarray(rvec, dtype=np.float64).reshape(3)
    theta = np.linalg.norm(r)
    if theta < np.finfo(np.float64).eps:
        return np.eye(3)
    k = r / theta
    c, s = np.cos(theta), np.sin(theta)
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return c * np.eye(3) + (1 - c) * np.outer(k, k) + s * Kx


def camera_rotations(camera_params):
    """Rotation actually used by the reference for every camera row (N,12) -> (N,3,3).

    Once per camera instead of once per observation: the polar factor U Vt of the raw 3x3,
    snapped to the identity where cv2.Rodrigues returns a zero vector.  The matrix -> vector
    -> matrix round trip is otherwise the identity map on rotations (checked against the
    literal per-observation loop in tests/test_oracle.py).
    """
    R = np.asarray(camera_params, dtype=np.float64)[:, :9].reshape(-1, 3, 3)
    U, _, Vt = np.linalg.svd(R)
    Q = U @ Vt
    rx = Q[:, 2, 1] - Q[:, 1, 2]
    ry = Q[:, 0, 2] - Q[:, 2, 0]
    rz = Q[:, 1, 0] - Q[:, 0, 1]
    s = np.sqrt((rx * rx + ry * ry + rz * rz) * 0.25)
    c = (Q[:, 0, 0] + Q[:, 1, 1] + Q[:, 2, 2] - 1.0) * 0.5
    snap = (s < 1e-5) & (c > 0)
    Q[snap] = np.eye(3)
    # within 1e-5 of a half turn cv2.Rodrigues takes the axis from the diagonal and the angle from
    # the trace: the vector -> matrix round trip is then not the identity (few cameras: slow path)
    for n in np.flatnonzero((s < 1e-5) & ~(c > 0)):
        Q[n] = rodrigues_vector_to_matrix(rodrigues_matrix_to_vector(Q[n]))
    return Q


# --------------------------------------------------------------------------- residual model
def project(points, camera_params, K, dist=np.array([])):
    """Vectorised restatement of /root/reference/bundle_adjustment.py:10-31.

    points (N,3) and camera_params (N,12) are already gathered per observation."""
    points = np.asarray(points, dtype=np.float64)
    camera_params = np.asarray(camera_params, dtype=np.float64)
    Q = camera_rotations(camera_params)
    Y = np.einsum("nij,nj->ni", Q, points) + camera_params[:, 9:12]
    z = Y[:, 2]
    iz = np.where(z != 0.0, 1.0 / np.where(z != 0.0, z, 1.0), 1.0)
    u = K[0, 0] * (Y[:, 0] * iz) + K[0, 2]
    v = K[1, 1] * (Y[:, 1] * iz) + K[1, 2]
    return np.stack([u, v], axis=1)


def project_loop(points, camera_params, K):
    """Literal per-observation loop of /root/reference/bundle_adjustment.py:22-30 with the
    two cv2 calls replaced by their restatements.  Slow; small cases only."""
    out = []
    for idx in range(len(camera_params)):
        R = np.asarray(camera_params[idx][:9], dtype=np.float64).reshape(3, 3)
        rvec = rodrigues_matrix_to_vector(R)
        t = np.asarray(camera_params[idx][9:], dtype=np.float64)
        Y = rodrigues_vector_to_matrix(rvec) @ np.asarray(points[idx], dtype=np.float64) + t
        z = 1.0 / Y[2] if Y[2] != 0.0 else 1.0
        out.append([K[0, 0] * Y[0] * z + K[0, 2], K[1, 1] * Y[1] * z + K[1, 2]])
    return np.array(out)


def fun(params, n_cameras, n_points, camera_indices, point_indices, points_2d, K):
    """/root/reference/bundle_adjustment.py:34-50 with one polar projection per camera."""
    params = np.asarray(params, dtype=np.float64)
    camera_params = params[:n_cameras * 12].reshape((n_cameras, 12))
    points_3d = params[n_cameras * 12:].reshape((n_points, 3))
    Q = camera_rotations(camera_params)
    ci = np.asarray(camera_indices)
    Y = np.einsum("nij,nj->ni", Q[ci], points_3d[np.asarray(point_indices)])
    Y += camera_params[ci, 9:12]
    z = Y[:, 2]
    iz = np.where(z != 0.0, 1.0 / np.where(z != 0.0, z, 1.0), 1.0)
    out = np.empty((ci.size, 2))
    out[:, 0] = K[0, 0] * (Y[:, 0] * iz) + K[0, 2] - points_2d[:, 0]
    out[:, 1] = K[1, 1] * (Y[:, 1] * iz) + K[1, 2] - points_2d[:, 1]
    return out.ravel()


def bundle_adjustment_sparsity(n_cameras, n_points, camera_indices, point_indices):
    """/root/reference/bundle_adjustment.py:53-74 (lil_matrix, dtype int)."""
    m = camera_indices.size * 2
    n = n_cameras * 12 + n_points * 3
    A = lil_matrix((m, n), dtype=int)
    i = np.arange(camera_indices.size)
    for s in range(12):
        A[2 * i, camera_indices * 12 + s] = 1
        A[2 * i + 1, camera_indices * 12 + s] = 1
    for s in range(3):
        A[2 * i, n_cameras * 12 + point_indices * 3 + s] = 1
        A[2 * i + 1, n_cameras * 12 + point_indices * 3 + s] = 1
    return A


def sparsity_csr(n_cameras, n_points, camera_indices, point_indices):
    """Same pattern as bundle_adjustment_sparsity, assembled directly as CSR (the LIL route
    is too slow and too large beyond ~1e5 observations)."""
    ci = np.asarray(camera_indices, dtype=np.int64)
    pi = np.asarray(point_indices, dtype=np.int64)
    n_obs = ci.size
    cols_obs = np.concatenate([ci[:, None] * 12 + np.arange(12)[None, :],
                               n_cameras * 12 + pi[:, None] * 3 + np.arange(3)[None, :]], axis=1)
    cols = np.repeat(cols_obs, 2, axis=0).ravel()
    indptr = np.arange(0, 15 * 2 * n_obs + 1, 15, dtype=np.int64)
    data = np.ones(cols.size, dtype=np.int8)
    return csr_matrix((data, cols, indptr), shape=(2 * n_obs, n_cameras * 12 + n_points * 3))


def pack(prob):
    """x0 and the args tuple exactly as /root/reference/bundle_adjustment.py:136,142-143."""
    x0 = np.hstack((prob["camera_params"].ravel(), prob["points_3d"].ravel()))
    args = (prob["n_cameras"], prob["n_points"], prob["camera_indices"],
            prob["point_indices"], prob["points_2d"], prob["K"])
    return x0, args


def cost(prob, camera_params=None, points_3d=None):
    """0.5 * ||fun||^2 (SciPy's cost convention, least_squares.py loss='linear')."""
    cp = prob["camera_params"] if camera_params is None else camera_params
    pts = prob["points_3d"] if points_3d is None else points_3d
    x = np.hstack((np.asarray(cp).ravel(), np.asarray(pts).ravel()))
    f = fun(x, prob["n_cameras"], prob["n_points"], prob["camera_indices"],
            prob["point_indices"], prob["points_2d"], prob["K"])
    return 0.5 * float(f @ f)


def solve_reference(prob, verbose=0, use_lil=None, max_nfev=None, **overrides):
    """The reference's optimiser call (/root/reference/bundle_adjustment.py:139-150)."""
    x0, args = pack(prob)
    n_obs = prob["camera_indices"].size
    if use_lil is None:
        use_lil = n_obs <= 100_000
    A = (bundle_adjustment_sparsity if use_lil else sparsity_csr)(
        prob["n_cameras"], prob["n_points"], prob["camera_indices"], prob["point_indices"])
    kw = dict(REFERENCE_KWARGS)
    kw.update(overrides)
    t0 = time.time()
    res = least_squares(fun, x0, jac_sparsity=A, verbose=verbose, max_nfev=max_nfev,
                        args=args, **kw)
    t1 = time.time()
    nc, npts = prob["n_cameras"], prob["n_points"]
    return {
        "poses": res.x[:nc * 12].reshape(nc, 12),
        "points_3d": res.x[nc * 12:].reshape(npts, 3),
        "cost": float(res.cost), "initial_cost": 0.5 * float(np.sum(fun(x0, *args) ** 2)),
        "status": int(res.status), "nfev": int(res.nfev), "njev": int(res.njev),
        "optimality": float(res.optimality), "seconds": t1 - t0, "result": res,
    }


def solve_tight(prob, verbose=0, max_nfev=400):
    """Same model and optimiser run to (near) stationarity: the converged-cost oracle."""
    return solve_reference(prob, verbose=verbose, max_nfev=max_nfev,
                           ftol=1e-15, xtol=1e-15, gtol=1e-12)
