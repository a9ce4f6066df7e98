"""Wall time and Python profile of a whole BundleAdjustment.optimize() on the 47-view temple-ring chain
(study script, GPU box): python profiles/study_optimize_overheads.py"""
import sys, time, cProfile, pstats, os, tempfile
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, pandas as pd
import temple_ring as tr
from structure_from_motion_amd import bundle_adjustment as ba
from test_gpu_dropin import FakeView, FakeWorldPoints
os.chdir(tempfile.mkdtemp())
names, K, R, t, pairs = tr.make_tracks(per_pair=40, seed=5, noise_px=0.3)
rng = np.random.default_rng(1)
views = [FakeView(names[j], R[j], t[j] + rng.normal(0, 0.002, 3)) for j in range(47)]
rows = pd.concat([pd.DataFrame({"ViewId": [[names[k - 1], names[k]]] * 40, "FeatureIndex": [[a, b] for a, b in zip(pairs[k-1]["uv_a"], pairs[k-1]["uv_b"])]}) for k in range(1, 47)], ignore_index=True)
X = np.vstack([p["X"] for p in pairs]) + rng.normal(0, 0.001, (1840, 3))
def run():
    wp = FakeWorldPoints(X.copy(), rows)
    b = ba.BundleAdjustment(wp, K, np.zeros((1, 5)), views, verbose=0)
    b.optimize()
    return b
run(); run()
t0 = time.perf_counter()
for _ in range(5): b = run()
print("optimize() per call ms:", 1e3 * (time.perf_counter() - t0) / 5, "solve ms", 1e3 * b.result["solve_seconds"], "iters", b.result["iterations"])
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
