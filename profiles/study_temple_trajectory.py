"""LM trajectory on the temple-ring chain (study script): python profiles/study_temple_trajectory.py <views-1> [engine options k=v ...]"""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.'); sys.path.insert(0, 'profiles')
from study_forcing_term import temple
from oracle import ba_oracle as o
from structure_from_motion_amd.engine import Engine
k = int(sys.argv[1])
def _num(v):
    try:
        return int(v)
    except ValueError:
        return float(v)
opts = {a.split("=")[0]: _num(a.split("=")[1]) for a in sys.argv[2:]}
p = temple(k)
r = o.solve_reference(p, use_lil=False)
print("scipy", r['cost'], r['nfev'])
with Engine(verbose=3, **opts) as e:
    e.set_problem(p["camera_params"], p["points_3d"], p["camera_indices"], p["point_indices"], p["points_2d"], p["K"])
    st = e.solve()
print(st.final_cost, st.iterations, st.pcg_iterations)
