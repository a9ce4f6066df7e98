"""Totals of the solve of one benchmark shape over a sweep of one option (study script):
python profiles/lm_trajectory.py venice pcg_rtol 0.1 0.03 0.01"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from structure_from_motion_amd import synthetic as syn
from structure_from_motion_amd.engine import Engine

shape, option = sys.argv[1], sys.argv[2]
p = syn.make_shape(shape, seed=0)
for v in sys.argv[3:]:
    eng = Engine(**{option: float(v)}, verbose=int(os.environ.get("VERBOSE", "0")))
    for rep in range(2):        # the second solve is the warm one
        eng.set_problem(p["camera_params"], p["points_3d"], p["camera_indices"], p["point_indices"], p["points_2d"], p["K"])
        st = eng.solve()
    print(option, v, "iterations", st.iterations, "pcg", st.pcg_iterations, "cost", repr(st.final_cost),
          "status", st.status, "optimality %.3g" % st.optimality, "ms %.3f" % (st.solve_seconds * 1e3), flush=True)
