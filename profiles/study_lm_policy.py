"""Globalisation policies of the LM loop on the CPU model (study script, no GPU):

    python profiles/study_lm_policy.py [problem ...]

Exact linear solves (sparse LU of the damped normal equations in the engine's 6-number tangent
space), so that what is compared is the damping / trust-region policy alone.  Prints, per problem,
SciPy's final cost and evaluation count with the reference's settings, and per policy the cost
relative to it and the iteration count under the same ftol rule."""
import sys

sys.path.insert(0, 'tests'); sys.path.insert(0, '.'); sys.path.insert(0, 'profiles')
import numpy as np
import scipy.sparse as sp
from scipy.sparse.linalg import splu

from engine_model import Model
from oracle import ba_oracle as o


def jac(m):
    r, Jc, Jp = m.jacobian_blocks()
    no = m.no
    rows = np.repeat(np.arange(2 * no), 9)
    cols = np.concatenate([6 * m.ci[:, None] + np.arange(6), 6 * m.nc + 3 * m.pi[:, None] + np.arange(3)], axis=1)
    cols = np.repeat(cols, 2, axis=0).ravel()
    vals = np.concatenate([Jc, Jp], axis=2).reshape(-1)
    J = sp.csr_matrix((vals, (rows, cols)), shape=(2 * no, 6 * m.nc + 3 * m.np_))
    return r.ravel(), J


def step_for(J, g, D, lam):
    H = (J.T @ J + sp.diags(lam * D)).tocsc()
    return splu(H).solve(-g)


def run(prob, policy, ftol=1e-4, max_iter=60, lam0=1e-4, verbose=False, **kw):
    m = Model(prob)
    lam, nu = lam0, 2.0
    r, J = jac(m)
    cost = 0.5 * r @ r
    its = 0
    Dmax = None
    hist = []
    delta = None
    while its < max_iter:
        g = J.T @ r
        D = np.clip(np.asarray(J.multiply(J).sum(axis=0)).ravel(), 1e-6, 1e32)
        if policy.get("monotone_scale"):
            Dmax = D if Dmax is None else np.maximum(D, Dmax)
            D = Dmax
        accepted = False
        while not accepted and its < max_iter:
            d = step_for(J, g, D, lam)
            lin = r + J @ d
            pred = cost - 0.5 * lin @ lin
            dc, dp = d[:6 * m.nc].reshape(-1, 6), d[6 * m.nc:].reshape(-1, 3)
            Qn, tn, Xn = m.retract(dc, dp)
            cost_new = m.cost(Qn, tn, Xn)
            actual = cost - cost_new
            rho = actual / pred if pred > 0 else -1.0
            its += 1
            sn = np.sqrt(np.sum(D * d * d))
            if verbose:
                print(f"   it {its:2d} cost {cost:.8e} -> {cost_new:.8e} rho {rho:+.3f} lam {lam:.2e} |Dd| {sn:.3e}")
            if rho > 1e-3 and np.isfinite(cost_new):
                accepted = True
                ft = actual < ftol * cost and rho > 0.25
                m.Q, m.t, m.X = Qn, tn, Xn
                hist.append(cost_new)
                if policy["name"] == "nielsen":
                    lam = max(lam * max(1 / 3, 1 - (2 * rho - 1) ** 3), 1e-15)
                elif policy["name"] == "marquardt":
                    if rho > 0.75:
                        lam = max(lam * policy.get("down", 0.1), policy.get("floor", 1e-15))
                    elif rho < 0.25:
                        lam *= policy.get("up", 4.0)
                elif policy["name"] == "floor":
                    # the damping falls to its floor whenever the step was a good one
                    if rho > policy.get("good", 0.25):
                        lam = max(lam * policy.get("down", 0.01), policy.get("floor", 1e-10))
                    else:
                        lam *= 4.0
                nu = 2.0
                cost = cost_new
                r, J = jac(m)
                if ft:
                    return cost, its, hist
            else:
                lam *= nu
                nu *= 2.0
    return cost, its, hist


def problems():
    from study_forcing_term import temple
    from structure_from_motion_amd import synthetic as syn
    return {"temple9": lambda: temple(9), "temple20": lambda: temple(20), "temple46": lambda: temple(46),
            "ring47": lambda: syn.make_problem(47, 1400, 2800, seed=13, window=3),
            "ladybug": lambda: syn.make_shape("ladybug", seed=0),
            "venice0.02": lambda: syn.make_shape("venice", seed=0, scale=0.02),
            "fountain0.1": lambda: syn.make_shape("fountain", seed=0, scale=0.1)}


if __name__ == "__main__":
    names = sys.argv[1:] or ["temple9", "temple46", "ring47", "ladybug"]
    pols = [dict(name="nielsen"),
            dict(name="marquardt", down=0.1),
            dict(name="floor", down=0.01, floor=1e-10),
            dict(name="floor", down=0.1, floor=1e-10),
            dict(name="nielsen", lam0=1e-8)]
    allp = problems()
    for n in names:
        p = allp[n]()
        ref = o.solve_reference(p, use_lil=False)
        print(f"{n}: scipy cost {ref['cost']:.8f} nfev {ref['nfev']} njev {ref['njev']}", flush=True)
        for pol in pols:
            pol = dict(pol)
            lam0 = pol.pop("lam0", 1e-4)
            c, its, hist = run(p, pol, lam0=lam0, verbose="-v" in sys.argv)
            print(f"   {str(pol):60s} lam0 {lam0:.0e}: rel {c / ref['cost'] - 1:+.2e} its {its}", flush=True)
