"""Times the second camera model (9-number camera of /root/reference/ba_test.py) on the GPU at
the Ladybug shape the script downloads (49 / 7776 / 31843) and at the Venice shape, and SciPy with
the script's keyword arguments beside it at the Ladybug shape (the oracle, CPU).
Usage: python profiles/bal_timing.py [--no-cpu] > profiles/r01_bal_timing.json"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from structure_from_motion_amd import engine as E  # noqa: E402
from structure_from_motion_amd import synthetic as syn  # noqa: E402


def gpu_solve(p, repeats=5, **opts):
    with E.Engine(camera_model="bal", **opts) as eng:
        t0 = time.perf_counter()
        eng.set_problem(p["camera_params"], p["points_3d"], p["camera_indices"], p["point_indices"], p["points_2d"], None)
        setup = time.perf_counter() - t0
        best, st = None, None
        for _ in range(repeats):
            eng.reset_params()
            t0 = time.perf_counter()
            st = eng.solve()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        prof = None
        if opts.get("profile"):
            prof = {k: getattr(st, k) for k in ("ms_linearize", "ms_damp", "ms_schur", "ms_pairs", "ms_bsr", "ms_spmv_pm", "ms_spmv_cm", "ms_pcg_vec", "ms_back", "ms_dense")}
    return dict(setup_seconds=setup, solve_seconds=best, iterations=st.iterations, pcg_iterations=st.pcg_iterations,
                initial_cost=st.initial_cost, final_cost=st.final_cost, status=st.status,
                obs_iter_per_s=p["camera_indices"].size * st.iterations / best, family_ms=prof)


def main():
    out = {}
    lady = syn.make_bal_problem(49, 7776, 31843, seed=0)
    out["ladybug_shape_gpu"] = gpu_solve(lady)
    if "--no-cpu" not in sys.argv:
        from oracle import bal_oracle as b
        r = b.solve_reference(lady)
        out["ladybug_shape_scipy"] = dict(solve_seconds=r["seconds"], nfev=r["nfev"], njev=r["njev"], final_cost=r["cost"],
                                          status=r["status"], initial_cost=r["initial_cost"], cores=1)
    venice = syn.make_bal_problem(1778, 993923, 5001946, seed=0)
    out["venice_shape_gpu"] = gpu_solve(venice, repeats=3)
    out["venice_shape_gpu_profiled"] = gpu_solve(venice, repeats=1, profile=1)
    out["venice_shape_gpu_matrix_free"] = gpu_solve(venice, repeats=2, schur_mode=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
