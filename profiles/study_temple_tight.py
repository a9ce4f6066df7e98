"""Temple-ring chains run to convergence by SciPy (ftol = xtol = 1e-15, 400 evaluations) and by the engine
(study script, GPU box): python profiles/study_temple_tight.py"""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.'); sys.path.insert(0, 'profiles')
from study_forcing_term import temple
from oracle import ba_oracle as o
from structure_from_motion_amd.engine import Engine
for k in (9, 20, 46):
    p = temple(k)
    t0 = time.time(); r = o.solve_tight(p); ts = time.time() - t0
    with Engine(ftol=1e-15, xtol=1e-15, gtol=1e-12, max_iterations=200, pcg_rtol=1e-3) as e:
        e.set_problem(p["camera_params"], p["points_3d"], p["camera_indices"], p["point_indices"], p["points_2d"], p["K"])
        st = e.solve()
    print(k, "scipy", r["cost"], r["nfev"], r["status"], f"{ts:.1f}s", "| engine", st.final_cost, st.iterations, st.status, f"{1e3*st.solve_seconds:.1f}ms", "rel", st.final_cost / r["cost"] - 1, flush=True)
