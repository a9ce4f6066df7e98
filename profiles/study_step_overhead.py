"""What a bench step costs outside sfmba_solve (the Python call, reset_params): the loop's wall time
per step against the solve time the library reports.   python profiles/study_step_overhead.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from structure_from_motion_amd import synthetic as syn
from structure_from_motion_amd.engine import Engine
p = syn.make_shape("venice", seed=0)
eng = Engine(device=0)
eng.set_problem(p["camera_params"], p["points_3d"], p["camera_indices"], p["point_indices"], p["points_2d"], p["K"])
for _ in range(3):
    eng.reset_params(); eng.solve()
torch.cuda.synchronize()
t0 = time.perf_counter(); inner = 0.0; n = 30
for _ in range(n):
    eng.reset_params(); st = eng.solve(); inner += st.solve_seconds
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"per step: python loop {1e3*el/n:.4f} ms, inside sfmba_solve {1e3*inner/n:.4f} ms, difference {1e6*(el-inner)/n:.1f} us")
