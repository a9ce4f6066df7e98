"""Study (not part of the product): why the last linear solves of a run need so many PCG iterations
and what a coarse-space correction would do.  Runs the engine for four LM iterations at the Venice
shape, takes the reduced system of the fifth through the diagnostic entry points
(sfmba_linearize / sfmba_get_reduced / sfmba_schur_matvec) and runs PCG on it in NumPy with
  - the block-Jacobi preconditioner the engine uses,
  - the right-hand side projected off the seven gauge directions (no effect: they carry 1e-7 of it),
  - block-Jacobi plus a coarse space of seven rigid-motion-and-scale vectors per cluster of
    consecutive cameras, M^-1 = M_J^-1 + Z (Z'SZ)^-1 Z'.
Measured on one MI355X: 141 iterations -> 17 / 10 / 6 with clusters of 64 / 32 / 16 cameras
(coarse dimensions 196 / 392 / 784).
Usage: python profiles/study_two_level_pcg.py [shape [cluster sizes ...]]"""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from structure_from_motion_amd import synthetic as syn
from structure_from_motion_amd.engine import Engine
SHAPE = sys.argv[1] if len(sys.argv) > 1 else "venice"
CLUSTERS = [int(a) for a in sys.argv[2:]] or [64, 32, 16]
p = syn.make_shape(SHAPE, seed=0)
with Engine(max_iterations=4) as eng:
    eng.set_problem(p["camera_params"], p["points_3d"], p["camera_indices"], p["point_indices"], p["points_2d"], p["K"])
    st = eng.solve()
    print("after 4 iterations: cost", st.final_cost, "pcg", st.pcg_iterations, flush=True)
    poses, pts = eng.get_params()
    lam = 1.4e-6
    eng.linearize(lam)
    b, Minv = eng.get_reduced()
    nc = b.shape[0]
    def S(x): return eng.schur_matvec(x.reshape(nc, 6)).reshape(-1)
    def M(r): return np.einsum("nij,nj->ni", Minv, r.reshape(nc, 6)).reshape(-1)
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
    print("|S v| / |v| per gauge vector vs typical |S e|:", [float(np.linalg.norm(S(V[:, k])) / np.linalg.norm(V[:, k])) for k in range(7)],
          float(np.linalg.norm(S(b.reshape(-1))) / np.linalg.norm(b)), flush=True)
    def pcg(bvec, rtol=0.1, maxit=400, proj=None):
        x = np.zeros_like(bvec); r = bvec.copy(); z = M(r)
        if proj is not None: z = proj(z)
        pdir = z.copy(); rz = r @ z; rz0 = rz
        for it in range(1, maxit + 1):
            q = S(pdir); a = rz / (pdir @ q); x += a * pdir; r -= a * q; z = M(r)
            if proj is not None: z = proj(z)
            rzn = r @ z
            if rzn <= rtol ** 2 * rz0: return it, x
            pdir = z + (rzn / rz) * pdir; rz = rzn
        return maxit, x
    bb = b.reshape(-1)
    it0, x0 = pcg(bb)
    print("plain PCG iterations:", it0, flush=True)
    Qv, _ = np.linalg.qr(V)
    P = lambda v: v - Qv @ (Qv.T @ v)
    print("gauge share of b:", float(np.linalg.norm(Qv.T @ bb) / np.linalg.norm(bb)), " of x:", float(np.linalg.norm(Qv.T @ x0) / np.linalg.norm(x0)), flush=True)
    def quality(x): return float(np.linalg.norm(S(x) - bb) / np.linalg.norm(bb)), float(0.5 * x @ S(x) - bb @ x)
    print("plain PCG: relative residual, model value:", quality(x0), flush=True)
    for k, cols in [(k, cols) for k in CLUSTERS for cols in ((0, 1, 2, 3, 4, 5, 6), (0, 1, 2, 6), (0, 1, 2))]:
        ncl = (nc + k - 1) // k
        nv = len(cols)
        Z = np.zeros((6 * nc, nv * ncl))
        for c in range(ncl):
            rows = slice(6 * c * k, 6 * min((c + 1) * k, nc))
            Z[rows, nv * c:nv * c + nv] = V[rows][:, cols]
        Z /= np.linalg.norm(Z, axis=0)
        t0 = time.time()
        SZ = np.stack([S(Z[:, j].copy()) for j in range(Z.shape[1])], axis=1)
        E = Z.T @ SZ
        Einv = np.linalg.inv(0.5 * (E + E.T))
        def M2(r, Einv=Einv, Z=Z): return M(r) + Z @ (Einv @ (Z.T @ r))
        def pcg2(bvec, rtol=0.1, maxit=400):
            x = np.zeros_like(bvec); r = bvec.copy(); z = M2(r); pdir = z.copy(); rz = r @ z; rz0 = rz
            for it in range(1, maxit + 1):
                q = S(pdir); a = rz / (pdir @ q); x += a * pdir; r -= a * q; z = M2(r); rzn = r @ z
                if rzn <= rtol ** 2 * rz0: return it, x
                pdir = z + (rzn / rz) * pdir; rz = rzn
            return maxit, x
        it, x = pcg2(bb)
        if nv == 7:
            # the coarse solve as the initial guess only, block-Jacobi PCG on the rest, stopping at
            # the same r'M^-1 r as the plain solve
            x00 = Z @ (Einv @ (Z.T @ bb))
            r00 = bb - S(x00)
            rz_plain = bb @ M(bb)
            rz_start = r00 @ M(r00)
            itg, xg = pcg(r00, rtol=0.1 * np.sqrt(rz_plain / rz_start))
            print(f"   coarse initial guess, then block-Jacobi PCG: {itg} iterations, start r'z ratio {rz_start / rz_plain:.3g}, residual/model {quality(x00 + xg)}", flush=True)
        print(f"two-level, clusters of {k} cameras, vectors {cols} (coarse dimension {Z.shape[1]}): {it} iterations, residual/model {quality(x)} (setup {time.time()-t0:.1f}s)", flush=True)
