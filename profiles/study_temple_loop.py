"""The temple-ring loop of tests/test_gpu_temple_ring.py with engine options from the command line
(study script, GPU box):   python profiles/study_temple_loop.py [k=v ...] [--every N]

Prints, at the stages the test checks (or every N-th view), SciPy's final cost and evaluation count
beside the engine's cost relative to it, LM iterations, PCG iterations and solve time."""
import os
import sys
import tempfile

sys.path.insert(0, 'tests'); sys.path.insert(0, '.'); sys.path.insert(0, 'profiles')
import numpy as np
import pandas as pd

import temple_ring as tr
from oracle import ba_oracle as o
from structure_from_motion_amd import bundle_adjustment as ba
from study_forcing_term import _num
from test_gpu_dropin import FakeView, FakeWorldPoints
from test_gpu_temple_ring import _rot

if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if "=" in a]
    every = int(sys.argv[sys.argv.index("--every") + 1]) if "--every" in sys.argv else 0
    opts = {a.split("=")[0]: _num(a.split("=")[1]) for a in args}
    os.chdir(tempfile.mkdtemp())
    names, K, R, t, pairs = tr.make_tracks(per_pair=40, seed=5, noise_px=0.3)
    rng = np.random.default_rng(17)
    dist = np.zeros((1, 5))

    def new_view(j):
        return FakeView(names[j], _rot(rng.normal(0, 0.003, 3)) @ R[j], t[j] + rng.normal(0, 0.002, 3))

    views = [new_view(0)]
    wp = FakeWorldPoints(np.empty((0, 3)), pd.DataFrame({"ViewId": [], "FeatureIndex": []}))
    total_it = total_ms = 0
    for k in range(1, 47):
        views.append(new_view(k))
        pr = pairs[k - 1]
        rows = pd.DataFrame({"ViewId": [[names[k - 1], names[k]]] * len(pr["X"]),
                             "FeatureIndex": [[a, b] for a, b in zip(pr["uv_a"], pr["uv_b"])]})
        wp.correspondences = pd.concat((wp.correspondences, rows), ignore_index=True)
        wp.world_points = np.vstack((wp.world_points, pr["X"] + rng.normal(0, 0.001, pr["X"].shape)))
        b = ba.BundleAdjustment(wp, K, dist, views, verbose=0, **opts)
        b.optimize()
        res = b.result
        total_it += res["iterations"]; total_ms += 1e3 * res["solve_seconds"]
        if k in (1, 9, 46) or (every and k % every == 0):
            sub = dict(camera_params=b.camera_params, points_3d=b.points_3d, camera_indices=b.camera_indices,
                       point_indices=b.point_indices, points_2d=b.points_2d, K=K,
                       n_cameras=b.n_cameras, n_points=b.n_points, n_obs=len(b.camera_indices))
            ref = o.solve_reference(sub)
            print(f"k {k:2d}: scipy {ref['cost']:.6f} nfev {ref['nfev']:2d} | engine {res['final_cost'] / ref['cost'] - 1:+.2e} "
                  f"it {res['iterations']} pcg {res['pcg_iterations']} status {res['status']} {1e3 * res['solve_seconds']:.2f} ms", flush=True)
        b.write_back(print_error=False)
    print(f"whole loop: {total_it} LM iterations, {total_ms:.1f} ms in solves")
