"""Ad-hoc GPU shake-out (not collected by pytest)."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from structure_from_motion_amd import synthetic as syn
from structure_from_motion_amd.engine import Engine
from oracle import ba_oracle as o
import engine_model as em

def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)

p = syn.make_shape("ladybug", seed=0, shuffle=True)
x0, args = o.pack(p)
f_ref = o.fun(x0, *args)
eng = Engine(verbose=3)
eng.set_problem(p["camera_params"], p["points_3d"], p["camera_indices"], p["point_indices"], p["points_2d"], p["K"])
f = eng.residuals()
print("residual rel err", rel(f, f_ref), flush=True)
m = em.Model(p)
r, Jc, Jp = m.jacobian_blocks()
jc, jp = eng.jacobian_blocks()
print("Jc", rel(jc, Jc), "Jp", rel(jp, Jp), flush=True)
lam = 1e-4
eng.linearize(lam)
U, gc, V, gp = eng.get_blocks()
r_, Jc_, Jp_, U_, gc_, V_, gp_ = m.assemble()
print("U", rel(U, U_), "gc", rel(gc, gc_), "V", rel(V, V_), "gp", rel(gp, gp_), flush=True)
x = np.random.default_rng(0).normal(size=(m.nc, 6))
y = eng.schur_matvec(x)
# model matvec
dU = np.clip(np.einsum("nii->ni", U_), 1e-6, 1e32); dV = np.clip(np.einsum("nii->ni", V_), 1e-6, 1e32)
Us = U_.copy(); Vs = V_.copy()
Us[:, range(6), range(6)] += lam * dU; Vs[:, range(3), range(3)] += lam * dV
Vinv = np.linalg.inv(Vs)
W = np.einsum("nki,nkj->nij", Jc_, Jp_)
tp = np.zeros((m.np_, 3)); np.add.at(tp, m.pi, np.einsum("nji,nj->ni", W, x[m.ci]))
sp = np.einsum("nij,nj->ni", Vinv, tp)
y_ = np.einsum("nij,nj->ni", Us, x); np.add.at(y_, m.ci, -np.einsum("nij,nj->ni", W, sp[m.pi]))
print("matvec", rel(y, y_), flush=True)
rhs, Minv = eng.get_reduced()
WV = np.einsum("nij,njk->nik", W, Vinv[m.pi])
rhs_ = -gc_.copy(); np.add.at(rhs_, m.ci, np.einsum("nij,nj->ni", WV, gp_[m.pi]))
Sd = Us.copy(); np.add.at(Sd, m.ci, -np.einsum("nij,nkj->nik", WV, W))
print("rhs", rel(rhs, rhs_), "Minv", rel(Minv, np.linalg.inv(Sd)), flush=True)
dc, dp, it = eng.solve_step()
print("pcg iters", it, flush=True)
t = time.time(); st = eng.solve(); print("solve", time.time() - t, st.as_dict(), flush=True)
ref = o.solve_reference(p)
print("scipy ref cost", ref["cost"], "engine", st.final_cost, flush=True)
eng.close()

for name, scale in (("venice", 0.05), ("venice", 1.0)):
    t = time.time(); p = syn.make_shape(name, seed=0, scale=scale); print("gen", time.time() - t, p["n_cameras"], p["n_points"], p["n_obs"], flush=True)
    eng = Engine(verbose=3, profile=1)
    t = time.time(); eng.set_problem(p["camera_params"], p["points_3d"], p["camera_indices"], p["point_indices"], p["points_2d"], p["K"]); print("set_problem", time.time() - t, flush=True)
    t = time.time(); st = eng.solve(); print("solve", time.time() - t, st.as_dict(), flush=True)
    eng.close()
