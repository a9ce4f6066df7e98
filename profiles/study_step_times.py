"""Wall time of each of the first solves of a process at the Venice shape (the bench's default
is 2 warm-ups and 5 timed solves): how long the start-up effects last.
   python profiles/study_step_times.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from structure_from_motion_amd import synthetic as syn
from structure_from_motion_amd.engine import Engine
p = syn.make_shape("venice", seed=0)
eng = Engine(device=0)
eng.set_problem(p["camera_params"], p["points_3d"], p["camera_indices"], p["point_indices"], p["points_2d"], p["K"])
ts = []
for _ in range(16):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.reset_params(); eng.solve(); torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t0))
print("ms per solve, in order:", " ".join(f"{t:.3f}" for t in ts))
