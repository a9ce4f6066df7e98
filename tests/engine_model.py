"""NumPy model of the engine's own algorithm (Levenberg-Marquardt, Schur complement on the
points, block-Jacobi preconditioned CG on the reduced camera system).  TEST INFRASTRUCTURE:
it exists so that the GPU tests can compare intermediate quantities (Jacobian blocks, Schur
blocks, the reduced matvec, the step) and so that algorithm changes can be tried on the CPU
first.  The shipped package never imports it.

Conventions shared with the HIP kernels:
  camera state  Q (3x3 rotation, row-major), t;  Y = Q X + t
  residual      r = (fx*Y0/Z + cx - u_obs, fy*Y1/Z + cy - v_obs), 1/Z := 1 when Z == 0
  camera step   left perturbation  Y' = exp([dw]x) Y + dt   =>  Q <- exp([dw]x) Q,
                t <- exp([dw]x) t + dt ;  tangent order (dw0,dw1,dw2,dt0,dt1,dt2)
  with x = Y0/Z, y = Y1/Z, iz = 1/Z:
      Jc = [[-fx*x*y,  fx*(1+x*x), -fx*y,  fx*iz, 0,     -fx*x*iz],
            [-fy*(1+y*y), fy*x*y,   fy*x,  0,     fy*iz, -fy*y*iz]]
      Jp = [[fx*iz*(Q0 - x*Q2)], [fy*iz*(Q1 - y*Q2)]]        (Qk = k-th row of Q)
"""
from __future__ import annotations

import numpy as np


def polar(R):
    U, _, Vt = np.linalg.svd(R)
    return U @ Vt


def expm_so3(w):
    w = np.atleast_2d(w)
    th = np.linalg.norm(w, axis=1)
    th2 = th * th
    small = th < 1e-4
    a = np.where(small, 1.0 - th2 / 6.0, np.sin(np.where(small, 1.0, th)) / np.where(small, 1.0, th))
    b = np.where(small, 0.5 - th2 / 24.0,
                 (1.0 - np.cos(np.where(small, 1.0, th))) / np.where(small, 1.0, th2))
    Kx = np.zeros((w.shape[0], 3, 3))
    Kx[:, 0, 1] = -w[:, 2]; Kx[:, 0, 2] = w[:, 1]
    Kx[:, 1, 0] = w[:, 2]; Kx[:, 1, 2] = -w[:, 0]
    Kx[:, 2, 0] = -w[:, 1]; Kx[:, 2, 1] = w[:, 0]
    return np.eye(3)[None] + a[:, None, None] * Kx + b[:, None, None] * (Kx @ Kx)


class Model:
    def __init__(self, prob, Q=None):
        self.nc, self.np_, self.no = prob["n_cameras"], prob["n_points"], prob["camera_indices"].size
        self.ci = np.asarray(prob["camera_indices"])
        self.pi = np.asarray(prob["point_indices"])
        self.uv = np.asarray(prob["points_2d"], dtype=np.float64)
        K = prob["K"]
        self.fx, self.fy, self.cx, self.cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
        cp = np.asarray(prob["camera_params"], dtype=np.float64)
        self.Q = polar(cp[:, :9].reshape(-1, 3, 3)) if Q is None else Q.copy()
        self.t = cp[:, 9:12].copy()
        self.X = np.asarray(prob["points_3d"], dtype=np.float64).copy()

    # ---- per-observation quantities
    def residuals(self, Q=None, t=None, X=None):
        Q = self.Q if Q is None else Q
        t = self.t if t is None else t
        X = self.X if X is None else X
        Y = np.einsum("nij,nj->ni", Q[self.ci], X[self.pi]) + t[self.ci]
        z = Y[:, 2]
        iz = np.where(z != 0, 1.0 / np.where(z != 0, z, 1.0), 1.0)
        r = np.empty((self.no, 2))
        r[:, 0] = self.fx * Y[:, 0] * iz + self.cx - self.uv[:, 0]
        r[:, 1] = self.fy * Y[:, 1] * iz + self.cy - self.uv[:, 1]
        return r, Y, iz

    def jacobian_blocks(self):
        r, Y, iz = self.residuals()
        x, y = Y[:, 0] * iz, Y[:, 1] * iz
        fx, fy = self.fx, self.fy
        Jc = np.zeros((self.no, 2, 6))
        Jc[:, 0, 0] = -fx * x * y; Jc[:, 0, 1] = fx * (1 + x * x); Jc[:, 0, 2] = -fx * y
        Jc[:, 0, 3] = fx * iz; Jc[:, 0, 5] = -fx * x * iz
        Jc[:, 1, 0] = -fy * (1 + y * y); Jc[:, 1, 1] = fy * x * y; Jc[:, 1, 2] = fy * x
        Jc[:, 1, 4] = fy * iz; Jc[:, 1, 5] = -fy * y * iz
        Qo = self.Q[self.ci]
        Jp = np.empty((self.no, 2, 3))
        Jp[:, 0, :] = (fx * iz)[:, None] * (Qo[:, 0, :] - x[:, None] * Qo[:, 2, :])
        Jp[:, 1, :] = (fy * iz)[:, None] * (Qo[:, 1, :] - y[:, None] * Qo[:, 2, :])
        return r, Jc, Jp

    def cost(self, Q=None, t=None, X=None):
        r, _, _ = self.residuals(Q, t, X)
        return 0.5 * float(np.sum(r * r))

    # ---- normal-equation blocks
    def assemble(self):
        r, Jc, Jp = self.jacobian_blocks()
        U = np.zeros((self.nc, 6, 6)); gc = np.zeros((self.nc, 6))
        V = np.zeros((self.np_, 3, 3)); gp = np.zeros((self.np_, 3))
        np.add.at(U, self.ci, np.einsum("nki,nkj->nij", Jc, Jc))
        np.add.at(gc, self.ci, np.einsum("nki,nk->ni", Jc, r))
        np.add.at(V, self.pi, np.einsum("nki,nkj->nij", Jp, Jp))
        np.add.at(gp, self.pi, np.einsum("nki,nk->ni", Jp, r))
        return r, Jc, Jp, U, gc, V, gp

    def retract(self, dc, dp):
        E = expm_so3(dc[:, :3])
        Q = E @ self.Q
        t = np.einsum("nij,nj->ni", E, self.t) + dc[:, 3:]
        return Q, t, self.X + dp


def lm_solve(prob, ftol=1e-4, xtol=1e-12, gtol=1e-8, max_iter=100, lam0=1e-4, pcg_tol=0.1,
             pcg_max=500, dense=False, verbose=False, log=None, down=1.0 / 30.0):
    """Returns (model, info).  cost convention 0.5*||r||^2."""
    m = Model(prob)
    lam = lam0
    nu = 2.0
    cost = m.cost()
    info = {"initial_cost": cost, "iters": 0, "pcg_iters": 0, "status": 0, "costs": [cost]}
    for it in range(max_iter):
        r, Jc, Jp, U, gc, V, gp = m.assemble()
        gnorm = max(np.abs(gc).max(), np.abs(gp).max())
        if gnorm < gtol:
            info["status"] = 1
            break
        dU = np.clip(np.einsum("nii->ni", U), 1e-6, 1e32)
        dV = np.clip(np.einsum("nii->ni", V), 1e-6, 1e32)
        accepted = False
        while not accepted:
            Us = U.copy(); Vs = V.copy()
            Us[:, range(6), range(6)] += lam * dU
            Vs[:, range(3), range(3)] += lam * dV
            Vinv = np.linalg.inv(Vs)
            # W_o = Jc^T Jp (6x3)
            W = np.einsum("nki,nkj->nij", Jc, Jp)
            WVinv = np.einsum("nij,njk->nik", W, Vinv[m.pi])
            rhs = -gc.copy()
            np.add.at(rhs, m.ci, np.einsum("nij,nj->ni", WVinv, gp[m.pi]))
            # block-Jacobi preconditioner: diagonal blocks of S
            Sd = Us.copy()
            np.add.at(Sd, m.ci, -np.einsum("nij,nkj->nik", WVinv, W))
            Minv = np.linalg.inv(Sd)

            def matvec(x):
                tp = np.zeros((m.np_, 3))
                np.add.at(tp, m.pi, np.einsum("nji,nj->ni", W, x[m.ci]))
                sp = np.einsum("nij,nj->ni", Vinv, tp)
                y = np.einsum("nij,nj->ni", Us, x)
                np.add.at(y, m.ci, -np.einsum("nij,nj->ni", W, sp[m.pi]))
                return y

            if dense:
                n = 6 * m.nc
                S = np.zeros((n, n))
                for c in range(m.nc):
                    S[6 * c:6 * c + 6, 6 * c:6 * c + 6] = Us[c]
                # explicit Schur complement
                order = np.argsort(m.pi, kind="stable")
                pts_sorted = m.pi[order]
                starts = np.flatnonzero(np.r_[True, pts_sorted[1:] != pts_sorted[:-1], True])
                for a, b in zip(starts[:-1], starts[1:]):
                    obs = order[a:b]
                    p = pts_sorted[a]
                    for i in obs:
                        for j in obs:
                            S[6 * m.ci[i]:6 * m.ci[i] + 6, 6 * m.ci[j]:6 * m.ci[j] + 6] -= \
                                W[i] @ Vinv[p] @ W[j].T
                dc = np.linalg.solve(S, rhs.ravel()).reshape(m.nc, 6)
                k = 0
            else:
                x = np.zeros_like(rhs)
                res = rhs.copy()
                z = np.einsum("nij,nj->ni", Minv, res)
                p = z.copy()
                rz = float(np.sum(res * z))
                rz0 = rz
                k = 0
                while k < pcg_max:
                    q = matvec(p)
                    alpha = rz / float(np.sum(p * q))
                    x += alpha * p
                    res -= alpha * q
                    z = np.einsum("nij,nj->ni", Minv, res)
                    rz_new = float(np.sum(res * z))
                    k += 1
                    if rz_new <= pcg_tol * pcg_tol * rz0:
                        break
                    p = z + (rz_new / rz) * p
                    rz = rz_new
                dc = x
            info["pcg_iters"] += k
            tp = np.zeros((m.np_, 3))
            np.add.at(tp, m.pi, np.einsum("nji,nj->ni", W, dc[m.ci]))
            dp = -np.einsum("nij,nj->ni", Vinv, gp + tp)
            lin = r + np.einsum("nki,ni->nk", Jc, dc[m.ci]) + np.einsum("nki,ni->nk", Jp, dp[m.pi])
            model_cost = 0.5 * float(np.sum(lin * lin))
            pred = cost - model_cost
            Qn, tn, Xn = m.retract(dc, dp)
            cost_new = m.cost(Qn, tn, Xn)
            actual = cost - cost_new
            rho = actual / pred if pred > 0 else -1.0
            step_norm = np.sqrt(np.sum(dc * dc) + np.sum(dp * dp))
            x_norm = np.sqrt(3.0 * m.nc + np.sum(m.t * m.t) + np.sum(m.X * m.X))
            if verbose:
                print(f"it {it:3d} cost {cost:.9e} -> {cost_new:.9e} rho {rho:+.3f} lam {lam:.2e} "
                      f"pcg {k:4d} |g| {gnorm:.2e} |dx| {step_norm:.2e}")
            if log is not None:
                log.append(dict(it=it, cost=cost, cost_new=cost_new, rho=rho, lam=lam, pcg=k))
            info["iters"] += 1
            if rho > 1e-3 and np.isfinite(cost_new):
                accepted = True
                m.Q, m.t, m.X = Qn, tn, Xn
                lam *= max(down, 1.0 - (2.0 * rho - 1.0) ** 3)
                lam = max(lam, 1e-12)
                nu = 2.0
                ft = actual < ftol * cost and rho > 0.25
                xt = step_norm < xtol * (xtol + x_norm)
                cost = cost_new
                info["costs"].append(cost)
                if ft and xt:
                    info["status"] = 4
                elif ft:
                    info["status"] = 2
                elif xt:
                    info["status"] = 3
            else:
                lam *= nu
                nu *= 2.0
                if lam > 1e16:
                    info["status"] = 3
                    break
                if step_norm < xtol * (xtol + x_norm):
                    info["status"] = 3
                    break
        if info["status"] != 0:
            break
    info["cost"] = cost
    return m, info
