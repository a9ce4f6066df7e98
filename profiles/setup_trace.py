"""Phase times of sfmba_set_problem on a warm handle (study script): python profiles/setup_trace.py venice"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SFMBA_TRACE_SETUP"] = "1"
from structure_from_motion_amd import synthetic as syn
from structure_from_motion_amd.engine import Engine

p = syn.make_shape(sys.argv[1] if len(sys.argv) > 1 else "venice", seed=0)
eng = Engine()
for rep in range(3):
    print(f"--- set_problem #{rep}", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    eng.set_problem(p["camera_params"], p["points_3d"], p["camera_indices"], p["point_indices"], p["points_2d"], p["K"])
    print(f"--- total {1e3 * (time.perf_counter() - t0):.2f} ms", file=sys.stderr, flush=True)
