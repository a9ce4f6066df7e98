"""LM trajectory of one study problem (GPU box): python profiles/study_trajectory.py problem [k=v ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
sys.path.insert(0, 'tests'); sys.path.insert(0, '.'); sys.path.insert(0, 'profiles')
from study_grid import PROBLEMS
from study_forcing_term import _num
from structure_from_motion_amd.engine import Engine
p = PROBLEMS[sys.argv[1]]()
opts = {a.split("=")[0]: _num(a.split("=")[1]) for a in sys.argv[2:]}
os.environ["SFMBA_PCG_TRACE"] = "1"
with Engine(verbose=3, **opts) as e:
    e.set_problem(p["camera_params"], p["points_3d"], p["camera_indices"], p["point_indices"], p["points_2d"], p["K"])
    st = e.solve()
print(sys.argv[1], opts, "final", st.final_cost, "it", st.iterations, "pcg", st.pcg_iterations, flush=True)
