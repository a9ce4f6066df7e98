"""The PCG model-gain rule with the 9-number camera (GPU box): final cost against the converged
cost, LM and PCG iterations and solve time with the rule on (SFMBA_PCG_GAIN unset) and off (-1),
on ordinary and poor starts.   python profiles/study_bal_gain.py [scale ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT); sys.path.insert(0, ROOT)
from structure_from_motion_amd import synthetic as syn
from structure_from_motion_amd.engine import Engine

def problem(scale, hard, seed):
    nc, npt, no = syn.SHAPES["venice"]
    nc, npt, no = max(8, int(nc * scale)), int(npt * scale), int(no * scale)
    return syn.make_bal_problem(nc, npt, no, seed=seed, rot_sigma=0.005 * hard, trans_sigma=0.01 * hard, point_sigma=0.01 * hard)

# values of SFMBA_PCG_GAIN to compare (None = the library's default, -1 = rule off): SFMBA_STUDY_GAINS="0.05,0.2,-1"
GAINS = [None if g == "default" else g for g in os.environ.get("SFMBA_STUDY_GAINS", "default,-1").split(",")]
scales = [float(a) for a in sys.argv[1:]] or [0.3, 1.0]
for scale in scales:
    for hard in (1, 3, 10):
        for seed in (0, 1):
            p = problem(scale, hard, seed)
            args = (p["camera_params"], p["points_3d"], p["camera_indices"], p["point_indices"], p["points_2d"], None)
            with Engine(camera_model="bal", ftol=1e-12, max_nfev=60) as e:
                e.set_problem(*args)
                conv = e.solve().final_cost
            line = f"scale {scale} sigma x{hard} seed {seed}: converged {conv:.4f} |"
            for gain in GAINS:
                if gain is None: os.environ.pop("SFMBA_PCG_GAIN", None)
                else: os.environ["SFMBA_PCG_GAIN"] = gain
                with Engine(camera_model="bal") as e:
                    e.set_problem(*args)
                    e.solve(); e.reset_params()
                    st = e.solve()
                line += f" gain {'default' if gain is None else gain}: {st.final_cost / conv - 1:+.1e} it {st.iterations} pcg {st.pcg_iterations} {1e3 * st.solve_seconds:.2f} ms |"
            os.environ.pop("SFMBA_PCG_GAIN", None)
            print(line, flush=True)
