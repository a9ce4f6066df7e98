"""Single-launch PCG with and without the coarse correction (study script): iteration counts per
linear solve (SFMBA_PCG_TRACE) and totals.  python profiles/fused_coarse_check.py [shape scale]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys
sys.path.insert(0, %r)
from structure_from_motion_amd import synthetic as syn
from structure_from_motion_amd.engine import Engine
p = syn.make_shape(sys.argv[1], seed=0, scale=float(sys.argv[2]))
eng = Engine()
for rep in range(2):
    eng.set_problem(p["camera_params"], p["points_3d"], p["camera_indices"], p["point_indices"], p["points_2d"], p["K"])
    st = eng.solve()
print("RESULT iterations", st.iterations, "pcg", st.pcg_iterations, "coarse solves", st.pcg_coarse_solves, "single", st.pcg_single_launch,
      "fallbacks", st.pcg_single_fallbacks, "cost", repr(st.final_cost), "status", st.status, "ms %%.3f" %% (st.solve_seconds * 1e3))
''' % ROOT
shape, scale = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("venice", "1.0")
for setting in ("0", "1"):
    env = dict(os.environ, SFMBA_PCG_FUSED_COARSE=setting, SFMBA_PCG_TRACE="1")
    r = subprocess.run([sys.executable, "-c", CHILD, shape, scale], env=env, capture_output=True, text=True, timeout=300)
    lines = [l for l in r.stderr.splitlines() if "pcg single launch" in l]
    print("SFMBA_PCG_FUSED_COARSE=" + setting, "rc", r.returncode)
    for l in [x for x in r.stderr.splitlines() if "pcg single launch" in x or "phases" in x or "us/iter" in x][-9:]:
        print("   ", l)
    print("   ", r.stdout.strip()[-400:])
    if r.returncode:
        print(r.stderr[-2000:])
