"""Shared test helpers: golden fixtures, error measures and the chain rule that maps the
engine's 2x6 / 2x3 Jacobian blocks to the reference's 12+3 columns per observation
(/root/reference/bundle_adjustment.py:53-74)."""
from __future__ import annotations

import glob
import os

import numpy as np
from scipy.sparse import csr_matrix

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names(need=None):
    out = []
    for path in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))):
        if os.path.basename(path).startswith("tri_") and need is None:
            continue                        # two-view triangulation fixtures: need="tri_X"
        if os.path.basename(path).startswith("bal_"):
            continue                        # 9-number camera fixtures: bal_golden_names()
        if os.path.basename(path).startswith("full_"):
            continue                        # results for full-size shapes regenerated from their seed
        if os.path.basename(path).startswith("nearpi_"):
            continue                        # half-turn rotations (cv2's diagonal branch): tested on their own
        if need is None:
            out.append(os.path.basename(path)[:-4])
        else:
            with np.load(path) as z:
                if need in z.files:
                    out.append(os.path.basename(path)[:-4])
    return out


def bal_golden_names():
    """Fixtures of the 9-number camera made by the oracle (oracle/gen_golden.py)."""
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "bal_*.npz"))
                  if not os.path.basename(p).startswith("bal_ref_"))


# Outputs of the reference's own ba_test.py run on a local data file (oracle/gen_golden_bal_ref.py):
# fixture name -> the data file in the published layout it was run on.
BAL_REF_CASES = {"bal_ref_6_40": "bal_problem-6-40-pre.txt.bz2",
                 "bal_ref_12_300": "bal_problem-12-300-pre.txt.bz2",
                 "bal_ref_49_7776": "bal_problem-49-7776-pre.txt.bz2"}


def load_bal_ref(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False) as z:
        g = {k: z[k] for k in z.files}
    g["n_cameras"] = g["camera_params"].shape[0]
    g["n_points"] = g["points_3d"].shape[0]
    g["n_obs"] = g["camera_indices"].size
    return g


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def elementwise_rel(a, b, floor=1.0):
    """max |a-b| / max(|b|, floor): relative per element, absolute below `floor`."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor)).max())


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def blocks_to_reference_jacobian(jc, jp, camera_params, camera_indices, point_indices, n_points):
    """CSR (2*n_obs) x (12*n_cameras + 3*n_points) Jacobian in the reference's parameterisation
    from blocks in the engine's: columns (dw, dt) of the left perturbation Y' = exp([dw]x)Y + dt.

    For the raw rotation entry R_ij: with R = Q P (polar), dQ = Q [w]x where
    (tr(P) I - P) w = axial(Q' E_ij - E_ij' Q); the left rotation is Q w and the translation
    stays fixed, i.e. dt = t x (Q w)."""
    cp = np.asarray(camera_params, dtype=np.float64)
    n_cam = cp.shape[0]
    ci = np.asarray(camera_indices)
    pi = np.asarray(point_indices)
    n_obs = ci.size
    # per camera: 6 x 12 matrix mapping d(raw 12) -> (dw, dt)
    M = np.zeros((n_cam, 6, 12))
    for c in range(n_cam):
        R = cp[c, :9].reshape(3, 3)
        U, _, Vt = np.linalg.svd(R)
        Q = U @ Vt
        P = Q.T @ R
        A = np.trace(P) * np.eye(3) - P
        t = cp[c, 9:12]
        for i in range(3):
            for j in range(3):
                E = np.zeros((3, 3)); E[i, j] = 1.0
                S = Q.T @ E - E.T @ Q
                a = np.array([S[2, 1], S[0, 2], S[1, 0]])
                w = np.linalg.solve(A, a)
                om = Q @ w
                M[c, :3, 3 * i + j] = om
                M[c, 3:, 3 * i + j] = np.cross(t, om)
        M[c, 3:, 9:12] = np.eye(3)
    J12 = np.einsum("nki,nij->nkj", jc, M[ci])          # (n_obs, 2, 12)
    vals = np.concatenate([J12, jp], axis=2)             # (n_obs, 2, 15)
    cols_obs = np.concatenate([ci[:, None] * 12 + np.arange(12)[None, :],
                               n_cam * 12 + pi[:, None] * 3 + np.arange(3)[None, :]], axis=1)
    cols = np.repeat(cols_obs, 2, axis=0).ravel()
    indptr = np.arange(0, 15 * 2 * n_obs + 1, 15)
    J = csr_matrix((vals.reshape(-1), cols, indptr), shape=(2 * n_obs, 12 * n_cam + 3 * n_points))
    J.sum_duplicates()
    J.sort_indices()
    return J


def golden_jacobian(g):
    n = 12 * g["n_cameras"] + 3 * g["n_points"]
    return csr_matrix((g["jac_data"], g["jac_indices"], g["jac_indptr"]), shape=(2 * g["n_obs"], n))


def identity_snapped_cameras(camera_params):
    """Cameras whose rotation cv2.Rodrigues snaps to the identity (sin < 1e-5, cos > 0): the
    reference's finite differences see no change there, so their rotation columns are zero."""
    from oracle import ba_oracle as o
    R = np.asarray(camera_params, dtype=np.float64)[:, :9].reshape(-1, 3, 3)
    U, _, Vt = np.linalg.svd(R)
    Q = U @ Vt
    r = np.stack([Q[:, 2, 1] - Q[:, 1, 2], Q[:, 0, 2] - Q[:, 2, 0], Q[:, 1, 0] - Q[:, 0, 1]], axis=1)
    s = np.sqrt((r * r).sum(axis=1) * 0.25)
    c = (np.trace(Q, axis1=1, axis2=2) - 1) * 0.5
    return np.flatnonzero((s < 1e-5) & (c > 0))


def bal_blocks_to_reference_jacobian(jc, jp, camera_params, camera_indices, point_indices, n_points):
    """CSR (2*n_obs) x (9*n_cameras + 3*n_points) Jacobian in the parameterisation of
    /root/reference/ba_test.py:92-100 (rotation vector, translation, f, k1, k2) from the engine's
    2x9 blocks, whose first six columns are (dw, dt) of the left perturbation
    Y' = exp([dw]x) Y + dt.  A rotation-vector change dr turns the camera by dw = J_l(r) dr (left
    Jacobian of SO(3)) and leaves the translation alone, i.e. dt = t x dw."""
    cp = np.asarray(camera_params, dtype=np.float64)
    n_cam = cp.shape[0]
    ci = np.asarray(camera_indices)
    pi = np.asarray(point_indices)
    n_obs = ci.size
    M = np.zeros((n_cam, 9, 9))
    for c in range(n_cam):
        r = cp[c, :3]
        a = np.linalg.norm(r)
        Kx = _skew(r)
        if a < 1e-8:
            Jl = np.eye(3) + 0.5 * Kx
        else:
            Jl = np.eye(3) + (1 - np.cos(a)) / a ** 2 * Kx + (a - np.sin(a)) / a ** 3 * Kx @ Kx
        M[c, :3, :3] = Jl
        M[c, 3:6, :3] = _skew(cp[c, 3:6]) @ Jl
        M[c, 3:, 3:] = np.eye(6)
    J9 = np.einsum("nki,nij->nkj", jc, M[ci])
    vals = np.concatenate([J9, jp], axis=2)
    cols_obs = np.concatenate([ci[:, None] * 9 + np.arange(9)[None, :],
                               n_cam * 9 + pi[:, None] * 3 + np.arange(3)[None, :]], axis=1)
    cols = np.repeat(cols_obs, 2, axis=0).ravel()
    indptr = np.arange(0, 12 * 2 * n_obs + 1, 12)
    J = csr_matrix((vals.reshape(-1), cols, indptr), shape=(2 * n_obs, 9 * n_cam + 3 * n_points))
    J.sum_duplicates()
    J.sort_indices()
    return J


def bal_golden_jacobian(g):
    n = 9 * g["n_cameras"] + 3 * g["n_points"]
    return csr_matrix((g["jac_data"], g["jac_indices"], g["jac_indptr"]), shape=(2 * g["n_obs"], n))
