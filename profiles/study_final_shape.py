"""Trajectory of the solve at the Final shape (13 682 cameras): LM iterations, damping, PCG
iterations per linear solve and wall time, under the options given in the environment
(SFMBA_LM_DOWN, SFMBA_PCG_COARSE_FROM, ...).  Run on the GPU box:
    python profiles/study_final_shape.py [verbose]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from structure_from_motion_amd import synthetic as syn
from structure_from_motion_amd.engine import Engine

verbose = int(sys.argv[1]) if len(sys.argv) > 1 else 3
p = syn.make_shape(os.environ.get("SFMBA_STUDY_SHAPE", "final"), seed=0)
for v in (verbose, 0):
    eng = Engine(device=0, verbose=v)
    eng.set_problem(p["camera_params"], p["points_3d"], p["camera_indices"], p["point_indices"], p["points_2d"], p["K"])
    for k in range(1 if v else 4):
        eng.reset_params()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s = eng.solve()
        torch.cuda.synchronize()
        print(f"solve {1e3 * (time.perf_counter() - t0):.2f} ms  lm {s.iterations} nfev {s.nfev} pcg {s.pcg_iterations} "
              f"cost {s.final_cost:.6f} status {s.status}", flush=True)
    del eng
