"""Sweep (GPU box): the engine against SciPy with the reference's settings on seeded small problems
of many sizes, both camera models - final cost relative to SciPy's, LM iterations against nfev.
   python profiles/study_parity_sweep.py [n_problems]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ba_oracle as o, bal_oracle as b
from structure_from_motion_amd import synthetic as syn
from structure_from_motion_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(12345)
worst = {"pinhole": -1.0, "bal": -1.0}
for k in range(n):
    model = "bal" if k % 3 == 2 else "pinhole"
    if os.environ.get("SFMBA_SWEEP_MODEL"): model = os.environ["SFMBA_SWEEP_MODEL"]
    nc = int(rng.choice([int(a) for a in os.environ["SFMBA_SWEEP_CAMS"].split(",")] if os.environ.get("SFMBA_SWEEP_CAMS") else [3, 5, 8, 12, 20, 33, 47, 49, 64, 90, 130]))
    npts = int(nc * rng.integers(20, 80))
    nobs = int(npts * rng.uniform(2.0, 5.0)) if nc > 3 else 2 * npts
    window = int(min(nc, rng.integers(3, 13)))
    if os.environ.get("SFMBA_SWEEP_WINDOW"): window = int(os.environ["SFMBA_SWEEP_WINDOW"])
    nobs = min(nobs, window * npts)
    hard = float(rng.choice([1.0, 1.0, 3.0, 8.0]))
    seed = 1000 + k
    kw = dict(seed=seed, window=window, rot_sigma=0.005 * hard, trans_sigma=0.01 * hard, point_sigma=0.01 * hard)
    if model == "bal":
        p = syn.make_bal_problem(nc, npts, nobs, **kw); p["K"] = None
        ref = b.solve_reference(p)
    else:
        p = syn.make_problem(nc, npts, nobs, **kw)
        ref = o.solve_reference(p)
    if os.environ.get("SFMBA_SWEEP_OUTLIERS"):
        # gross errors in a fraction of the image points (the reference has no robust loss: they stay in the cost)
        frac = float(os.environ["SFMBA_SWEEP_OUTLIERS"])
        bad = rng.random(p["points_2d"].shape[0]) < frac
        p["points_2d"] = p["points_2d"] + bad[:, None] * rng.normal(0.0, 60.0, p["points_2d"].shape)
        ref = (b if model == "bal" else o).solve_reference(p)
    with Engine(camera_model=model, verbose=3 if os.environ.get("SFMBA_SWEEP_VERBOSE") == str(k) else 0) as e:
        e.set_problem(p["camera_params"], p["points_3d"], p["camera_indices"], p["point_indices"], p["points_2d"], p["K"])
        st = e.solve()
    rel = st.final_cost / ref["cost"] - 1
    worst[model] = max(worst[model], rel)
    flag = "  <-- above SciPy" if rel > 1e-6 else ""
    print(f"{model:7s} cams {nc:3d} pts {npts:5d} obs {nobs:6d} window {window:2d} start x{hard:.0f}: scipy {ref['cost']:.6f} nfev {ref['nfev']:2d} | "
          f"engine {rel:+.1e} it {st.iterations} pcg {st.pcg_iterations} solver {st.linear_solver} status {st.status} {1e3 * st.solve_seconds:.2f} ms{flag}", flush=True)
print("worst relative excess over SciPy:", worst, flush=True)
