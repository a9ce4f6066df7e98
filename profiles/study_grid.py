"""Grid over engine options on the study problems (GPU box):

    python profiles/study_grid.py "initial_lambda=1e-4,1e-8;pcg_rtol=0,1e-3" [problem ...]

Every combination of the listed values; per problem SciPy's cost with the reference's settings
(below 300 000 observations) and per combination the engine's final cost relative to it, LM
iterations, PCG iterations and the time of a warm solve."""
import itertools
import sys

sys.path.insert(0, 'tests'); sys.path.insert(0, '.'); sys.path.insert(0, 'profiles')
from study_forcing_term import temple, _num
from oracle import ba_oracle as o
from structure_from_motion_amd import synthetic as syn
from structure_from_motion_amd.engine import Engine

PROBLEMS = {"venice": lambda: syn.make_shape("venice", seed=0),
            "venice_s1": lambda: syn.make_shape("venice", seed=1), "venice_s2": lambda: syn.make_shape("venice", seed=2),
            "final0.1": lambda: syn.make_shape("final", seed=0, scale=0.1),
            "final0.1_s1": lambda: syn.make_shape("final", seed=1, scale=0.1),
            "venice0.3_s1": lambda: syn.make_shape("venice", seed=1, scale=0.3),
            "venice0.3_s3": lambda: syn.make_shape("venice", seed=3, scale=0.3), "venice0.3_s4": lambda: syn.make_shape("venice", seed=4, scale=0.3),
            "venice0.3_hard3": lambda: syn.make_shape("venice", seed=0, scale=0.3, rot_sigma=0.015, trans_sigma=0.03, point_sigma=0.03),
            "venice0.3_hard10": lambda: syn.make_shape("venice", seed=0, scale=0.3, rot_sigma=0.05, trans_sigma=0.1, point_sigma=0.1),
            "venice_hard3": lambda: syn.make_shape("venice", seed=0, rot_sigma=0.015, trans_sigma=0.03, point_sigma=0.03),
            "venice_hard10": lambda: syn.make_shape("venice", seed=0, rot_sigma=0.05, trans_sigma=0.1, point_sigma=0.1),
            "final0.2_hard3": lambda: syn.make_shape("final", seed=0, scale=0.2, rot_sigma=0.015, trans_sigma=0.03, point_sigma=0.03),
            "ladybug_hard10": lambda: syn.make_shape("ladybug", seed=0, rot_sigma=0.05, trans_sigma=0.1, point_sigma=0.1),
            "fountain_hard10": lambda: syn.make_shape("fountain", seed=0, rot_sigma=0.05, trans_sigma=0.1, point_sigma=0.1),
            "venice0.3": lambda: syn.make_shape("venice", seed=0, scale=0.3),
            "venice0.1": lambda: syn.make_shape("venice", seed=0, scale=0.1),
            "temple1": lambda: temple(1), "temple3": lambda: temple(3),
            "temple9": lambda: temple(9), "temple20": lambda: temple(20), "temple46": lambda: temple(46),
            "ring47": lambda: syn.make_problem(47, 1400, 2800, seed=13, window=3),
            "ladybug": lambda: syn.make_shape("ladybug", seed=0),
            "fountain": lambda: syn.make_shape("fountain", seed=0),
            "final": lambda: syn.make_shape("final", seed=0),
            "final0.05": lambda: syn.make_shape("final", seed=0, scale=0.05),
            "final0.2": lambda: syn.make_shape("final", seed=0, scale=0.2),
            "venice0.02": lambda: syn.make_shape("venice", seed=0, scale=0.02),
            "fountain0.1": lambda: syn.make_shape("fountain", seed=0, scale=0.1)}

if __name__ == "__main__":
    grid = {}
    for part in sys.argv[1].split(";"):
        k, v = part.split("=")
        grid[k] = [_num(x) for x in v.split(",")]
    names = sys.argv[2:] or ["temple9", "temple20", "temple46", "ring47", "ladybug", "venice0.1", "venice"]
    for pname in names:
        p = PROBLEMS[pname]()
        if p['n_obs'] < 300000:
            ref = o.solve_reference(p, use_lil=False)
            print(f"{pname}: scipy {ref['cost']:.6f} nfev {ref['nfev']}", flush=True)
        else:
            ref = None
            print(f"{pname}:", flush=True)
        for combo in itertools.product(*grid.values()):
            opts = dict(zip(grid.keys(), combo))
            with Engine(**opts) as e:
                e.set_problem(p["camera_params"], p["points_3d"], p["camera_indices"], p["point_indices"], p["points_2d"], p["K"])
                e.solve()
                e.reset_params()
                st = e.solve()
            rel = f"{st.final_cost / ref['cost'] - 1:+.1e}" if ref else f"{st.final_cost:.6f}"
            print(f"   {str(opts):60s} {rel} it {st.iterations} pcg {st.pcg_iterations} {1e3 * st.solve_seconds:.2f} ms", flush=True)
