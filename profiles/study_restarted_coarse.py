"""Study (not part of the product): can a coarse correction OUTSIDE the single-launch PCG stand in
for one inside it?  Cycles of [coarse correction x += Z E^-1 Z' r; m block-Jacobi CG iterations
restarted from the current residual] against plain block-Jacobi PCG and against the two-level PCG,
on the reduced system of a late LM iteration at the Venice shape (through the diagnostic entry
points, PCG in NumPy).   python profiles/study_restarted_coarse.py [cluster size [m ...]]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from structure_from_motion_amd import synthetic as syn
from structure_from_motion_amd.engine import Engine
K = int(sys.argv[1]) if len(sys.argv) > 1 else 48
MS = [int(a) for a in sys.argv[2:]] or [10, 20, 40]
RTOL = 0.03
p = syn.make_shape("venice", seed=0)
with Engine(max_iterations=2) as eng:
    eng.set_problem(p["camera_params"], p["points_3d"], p["camera_indices"], p["point_indices"], p["points_2d"], p["K"])
    st = eng.solve()
    poses, pts = eng.get_params()
    eng.linearize(1e-7)
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
    ncl = (nc + K - 1) // K
    Z = np.zeros((6 * nc, 7 * ncl))
    for c in range(ncl):
        rows = slice(6 * c * K, 6 * min((c + 1) * K, nc))
        Z[rows, 7 * c:7 * c + 7] = V[rows]
    Z /= np.linalg.norm(Z, axis=0)
    SZ = np.stack([S(Z[:, j].copy()) for j in range(Z.shape[1])], axis=1)
    E = Z.T @ SZ
    Einv = np.linalg.inv(0.5 * (E + E.T))
    bb = b.reshape(-1)
    rz_target = RTOL ** 2 * (bb @ M(bb))

    def cg(rhs, x, maxit, precond):
        """PCG on S e = rhs from e = 0 until the residual of the WHOLE system passes the target."""
        e = np.zeros_like(rhs); r = rhs.copy(); z = precond(r); d = z.copy(); rz = r @ z
        for it in range(1, maxit + 1):
            q = S(d); a = rz / (d @ q); e += a * d; r -= a * q; z = precond(r); rzn = r @ z
            if r @ M(r) <= rz_target: return it, x + e, r, True
            d = z + (rzn / rz) * d; rz = rzn
        return maxit, x + e, r, False

    it, _, _, ok = cg(bb, np.zeros_like(bb), 1000, M)
    print(f"block-Jacobi PCG to {RTOL}: {it} iterations", flush=True)
    it, _, _, ok = cg(bb, np.zeros_like(bb), 1000, lambda r: M(r) + Z @ (Einv @ (Z.T @ r)))
    print(f"two-level PCG (clusters of {K}, coarse dimension {Z.shape[1]}): {it} iterations", flush=True)
    for m in MS:
        x = np.zeros_like(bb); r = bb.copy(); total = 0; cycles = 0
        while cycles < 60:
            ec = Z @ (Einv @ (Z.T @ r)); x = x + ec; r = r - S(ec)      # coarse correction
            n, x, r, ok = cg(r, x, m, M)                                  # m restarted block-Jacobi iterations
            total += n; cycles += 1
            if ok: break
        print(f"cycles of coarse correction + {m} block-Jacobi iterations: {cycles} cycles, {total} iterations, converged {ok}", flush=True)
