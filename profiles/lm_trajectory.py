"""Totals of the solve of benchmark shapes over a sweep of one option (study script):
python profiles/lm_trajectory.py pcg_rtol 0.1,0.03,0.01 venice:1.0:0 ladybug:1.0:3   (shape:scale:seed)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from structure_from_motion_amd import synthetic as syn
from structure_from_motion_amd.engine import Engine

option, values = sys.argv[1], sys.argv[2].split(",")
for case in sys.argv[3:]:
    shape, scale, seed = case.split(":")
    p = syn.make_shape(shape, seed=int(seed), scale=float(scale))
    for v in values:
        eng = Engine(**{option: float(v) if '.' in v or 'e' in v else int(v)}, verbose=int(os.environ.get("VERBOSE", "0")))
        for rep in range(2):        # the second solve is the warm one
            eng.set_problem(p["camera_params"], p["points_3d"], p["camera_indices"], p["point_indices"], p["points_2d"], p["K"])
            st = eng.solve()
        print(case, option, v, "iterations", st.iterations, "pcg", st.pcg_iterations, "coarse solves", st.pcg_coarse_solves,
              "dim", st.pcg_coarse_dim, "cost", repr(st.final_cost),
              "status", st.status, "optimality %.3g" % st.optimality, "ms %.3f" % (st.solve_seconds * 1e3), flush=True)
