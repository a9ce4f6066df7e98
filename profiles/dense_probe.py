"""Dense-solver probe (GPU box): python profiles/dense_probe.py problem [k=v ...] - a warm solve with the dense solver,
phase times from the engine's own event timers."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
sys.path.insert(0, 'tests'); sys.path.insert(0, '.'); sys.path.insert(0, 'profiles')
from study_grid import PROBLEMS
from study_forcing_term import _num
from structure_from_motion_amd.engine import Engine
p = PROBLEMS[sys.argv[1]]()
opts = {a.split("=")[0]: _num(a.split("=")[1]) for a in sys.argv[2:]}
with Engine(profile=1, **opts) as e:
    e.set_problem(p["camera_params"], p["points_3d"], p["camera_indices"], p["point_indices"], p["points_2d"], p["K"])
    e.solve(); e.reset_params()
    st = e.solve()
d = st.as_dict()
print(sys.argv[1], opts, "it", st.iterations, "pcg", st.pcg_iterations, f"{1e3*st.solve_seconds:.3f} ms", {k: round(v, 3) for k, v in d.items() if k.startswith("ms_") and v})
