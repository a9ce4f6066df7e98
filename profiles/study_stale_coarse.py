"""Study (not part of the product): does the coarse operator of the PREVIOUS LM iteration still
work as the second level of the preconditioner of the current one?  For LM iterations 4 and 5 at
the Venice shape: block-Jacobi PCG, two-level PCG with the current coarse operator, two-level PCG
with Z and E = Z'SZ taken from the iteration before (its iterate, its damping).
Usage: python profiles/study_stale_coarse.py [cluster size]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from structure_from_motion_amd import synthetic as syn
from structure_from_motion_amd.engine import Engine

K = int(sys.argv[1]) if len(sys.argv) > 1 else 48
p = syn.make_shape("venice", seed=0)


def system(n_done):
    """Reduced system of LM iteration n_done + 1 (damping 1e-4 / 3^n_done) and its coarse basis."""
    eng = Engine(max_iterations=n_done)
    eng.set_problem(p["camera_params"], p["points_3d"], p["camera_indices"], p["point_indices"], p["points_2d"], p["K"])
    eng.solve()
    poses, _ = eng.get_params()
    eng.linearize(1e-4 / 3.0 ** n_done)
    b, Minv = eng.get_reduced()
    nc = b.shape[0]
    Q = poses[:, :9].reshape(-1, 3, 3); t = poses[:, 9:]
    V = []
    for k in range(3):
        T = np.zeros(3); T[k] = 1.0
        V.append(np.hstack([np.zeros((nc, 3)), -(Q @ T)]).reshape(-1))
    for k in range(3):
        th = np.zeros(3); th[k] = 1.0
        Qt = Q @ th
        V.append(np.hstack([-Qt, np.cross(Qt, t)]).reshape(-1))
    V.append(np.hstack([np.zeros((nc, 3)), t]).reshape(-1))
    V = np.array(V).T
    ncl = nc // K
    Z = np.zeros((6 * nc, 7 * ncl))
    for c in range(ncl):
        lo, hi = c * nc // ncl, (c + 1) * nc // ncl
        Z[6 * lo:6 * hi, 7 * c:7 * c + 7] = V[6 * lo:6 * hi]
    Z /= np.linalg.norm(Z, axis=0)
    S = lambda x: eng.schur_matvec(x.reshape(nc, 6)).reshape(-1)
    M = lambda r: np.einsum("nij,nj->ni", Minv, r.reshape(nc, 6)).reshape(-1)
    E = Z.T @ np.stack([S(Z[:, j].copy()) for j in range(Z.shape[1])], axis=1)
    return eng, S, M, b.reshape(-1), Z, np.linalg.inv(0.5 * (E + E.T))


def pcg(S, M, b, rtol=0.1, maxit=400):
    x = np.zeros_like(b); r = b.copy(); z = M(r); d = z.copy(); rz = r @ z; rz0 = rz
    for it in range(1, maxit + 1):
        q = S(d); a = rz / (d @ q); x += a * d; r -= a * q; z = M(r); rzn = r @ z
        if rzn <= rtol ** 2 * rz0:
            return it
        d = z + (rzn / rz) * d; rz = rzn
    return maxit


prev = system(2)
for n_done in (3, 4):
    cur = system(n_done)
    _, S, M, b, Z, Einv = cur
    _, _, _, _, Zp, Einvp = prev
    for rtol in (0.1, 0.03):
        plain = pcg(S, M, b, rtol)
        fresh = pcg(S, lambda r: M(r) + Z @ (Einv @ (Z.T @ r)), b, rtol)
        stale = pcg(S, lambda r: M(r) + Zp @ (Einvp @ (Zp.T @ r)), b, rtol)
        mixed = pcg(S, lambda r: M(r) + Z @ (Einvp @ (Z.T @ r)), b, rtol)
        print(f"LM iteration {n_done + 1}, clusters of {K}, rtol {rtol}: block-Jacobi {plain}, two-level {fresh}, "
              f"coarse operator of the iteration before {stale}, its E with the current Z {mixed}", flush=True)
    prev[0].close()
    prev = cur
