"""Sweep of one engine option over small and mid-size problems (study script, GPU box):

    python profiles/study_forcing_term.py [option=v1,v2,...]      (default: pcg_rtol=0,0.01,0.003,0.001)

Per problem: SciPy's final cost with the reference's settings (problems below 300 000 observations;
larger ones are compared with the cost named in the environment as REF_<name>, or not at all), and
for every value of the option the engine's final cost relative to it, LM iterations, PCG iterations
and solve time.  Used for the PCG tolerance, the initial damping (`initial_lambda=1e-4,1e-6`), and,
through the environment of the library, for the cluster size and switch-on rule of the coarse
correction.  The temple-ring chains are built on the data set's calibration (tests/temple_ring.py)."""
import os
import sys

sys.path.insert(0, 'tests')
sys.path.insert(0, '.')
import numpy as np

import temple_ring as tr
from oracle import ba_oracle as o
from structure_from_motion_amd import synthetic as syn
from structure_from_motion_amd.engine import Engine


def temple(k, seed=17):
    """Views 0..k of the temple ring joined by two-view tracks, poses and points perturbed."""
    names, K, R, t, pairs = tr.make_tracks(per_pair=40, seed=5, noise_px=0.3)
    rng = np.random.default_rng(seed)
    n = k + 1
    cp = np.hstack([R[:n].reshape(n, 9), t[:n]]) + rng.normal(0, 0.002, (n, 12))
    X = np.vstack([p['X'] for p in pairs[:k]])
    X = X + rng.normal(0, 0.001, X.shape)
    ci = np.array([[j - 1, j] for j in range(1, n) for _ in range(40)]).ravel()
    pi = np.repeat(np.arange(len(X)), 2)
    uv = np.vstack([np.stack((p['uv_a'], p['uv_b']), 1).reshape(-1, 2) for p in pairs[:k]])
    return dict(camera_params=cp, points_3d=X, camera_indices=ci, point_indices=pi, points_2d=uv, K=K,
                n_cameras=n, n_points=len(X), n_obs=len(ci))


def _num(v):
    try:
        return int(v)
    except ValueError:
        return float(v)


if __name__ == '__main__':
    name, values = "pcg_rtol", [0.0, 0.01, 0.003, 0.001]
    if len(sys.argv) > 1:
        name, vals = sys.argv[1].split("=")
        values = [_num(v) for v in vals.split(",")]
    probs = {"venice": syn.make_shape("venice", seed=0), "venice0.3": syn.make_shape("venice", seed=0, scale=0.3),
             "temple9": temple(9), "temple20": temple(20), "temple46": temple(46),
             "ring47": syn.make_problem(47, 1400, 2800, seed=13, window=3),
             "ladybug": syn.make_shape("ladybug", seed=0), "venice0.1": syn.make_shape("venice", seed=0, scale=0.1),
             "final0.05": syn.make_shape("final", seed=0, scale=0.05), "venice0.02": syn.make_shape("venice", seed=0, scale=0.02),
             "fountain0.1": syn.make_shape("fountain", seed=0, scale=0.1)}
    for pname, p in probs.items():
        if p['n_obs'] < 300000:
            ref = o.solve_reference(p, use_lil=False)
        else:
            ref = dict(cost=float(os.environ.get('REF_' + pname, 'nan')), nfev=0)
        line = f"{pname:12s} scipy {ref['cost']:.6f} nfev {ref['nfev']:3d} |"
        for v in values:
            with Engine(**{name: v}) as e:
                e.set_problem(p["camera_params"], p["points_3d"], p["camera_indices"], p["point_indices"], p["points_2d"], p["K"])
                e.solve()
                e.reset_params()
                st = e.solve()
            line += f" {name} {v}: {st.final_cost / ref['cost'] - 1:+.1e} it {st.iterations} pcg {st.pcg_iterations} {1e3 * st.solve_seconds:.2f}ms |"
        print(line, flush=True)
