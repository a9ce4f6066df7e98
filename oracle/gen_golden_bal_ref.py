"""Golden vectors for the 9-number camera produced BY THE REFERENCE'S OWN CODE.

TEST INFRASTRUCTURE ONLY, build container only: this script imports /root/reference/ba_test.py,
which does not exist on the GPU box; only the .npz / .bz2 files it writes under tests/golden/
travel.  Run as  python oracle/gen_golden_bal_ref.py .

/root/reference/ba_test.py is a script: importing it downloads its input unless a file named
problem-49-7776-pre.txt.bz2 lies in the working directory (ba_test.py:9-14), reads that file
(ba_test.py:17-45), and runs least_squares on it (ba_test.py:120-126).  It never looks at the
sizes in the file, so any problem in the published layout can be handed to it under that name.
For every case below a child process
  1. changes into a scratch directory holding the case's data file under that name,
  2. replaces urllib.request.urlretrieve by a function that raises (no network is ever tried),
  3. imports ba_test from /root/reference,
  4. stores what the module computed: the five arrays read_bal_data returned, f0 = fun(x0),
     rotate() and project() on the gathered arrays, the pattern's shape / nnz / column checksum,
     and res.x / cost / status / nfev / njev / optimality of its least_squares call.
The stored files pin oracle/bal_oracle.py (tests/test_oracle_bal.py) and the engine
(tests/test_gpu_bal.py) against the reference itself, not against a restatement.
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference"
DATA_NAME = "problem-49-7776-pre.txt.bz2"       # ba_test.py:10


def child(out_npz, full):
    """Runs inside the scratch directory that holds DATA_NAME.  `full`: also store the arrays the
    reader returned and the per-observation rotate() / project() outputs (small cases; the large
    case keeps its data file as the input fixture and f0, which is project() minus points_2d)."""
    import urllib.request

    def no_network(*a, **k):
        raise RuntimeError("ba_test.py tried to download its input: the local data file is missing")

    urllib.request.urlretrieve = no_network
    sys.path.insert(0, REFERENCE)
    import ba_test as ref                       # reads the file and runs the solve (module level)

    gathered_pts = ref.points_3d[ref.point_indices]
    gathered_cams = ref.camera_params[ref.camera_indices]
    A = ref.A.tocsr()
    A.sort_indices()
    extra = {}
    if full:
        extra = dict(camera_params=ref.camera_params, points_3d=ref.points_3d, camera_indices=ref.camera_indices,
                     point_indices=ref.point_indices, points_2d=ref.points_2d,
                     rotated=ref.rotate(gathered_pts, gathered_cams[:, :3]), projected=ref.project(gathered_pts, gathered_cams))
    np.savez_compressed(
        out_npz, **extra,
        n_cameras=np.array(ref.n_cameras), n_points=np.array(ref.n_points), n_observations=np.array(ref.camera_indices.size),
        f0=ref.f0,
        pattern_shape=np.array(A.shape), pattern_nnz=np.array(A.nnz),
        pattern_colsum=np.array(int(A.indices.astype(np.int64).sum())), pattern_indptr_last=np.array(int(A.indptr[-1])),
        ref_x=ref.res.x, ref_cost=np.array(ref.res.cost), ref_status=np.array(ref.res.status),
        ref_nfev=np.array(ref.res.nfev), ref_njev=np.array(-1 if ref.res.njev is None else ref.res.njev),
        ref_optimality=np.array(ref.res.optimality), ref_initial_cost=np.array(0.5 * float(ref.f0 @ ref.f0)))


def run_case(name, data_file, full=True):
    """Hand `data_file` to the reference script under the name it looks for."""
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copy(data_file, os.path.join(tmp, DATA_NAME))
        out = os.path.join(OUT, f"bal_ref_{name}.npz")
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", out, "full" if full else "lean"], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL)
    with np.load(out) as z:
        print(f"bal_ref_{name}: {int(z['n_cameras'])} cameras / {int(z['n_points'])} points / "
              f"{int(z['n_observations'])} observations; cost {float(z['ref_initial_cost']):.6e} -> {float(z['ref_cost']):.9e}, "
              f"status {int(z['ref_status'])}, nfev {int(z['ref_nfev'])}")


def main():
    sys.path.insert(0, ROOT)
    from oracle.gen_golden import write_bal_text
    from structure_from_motion_amd import synthetic as syn

    # the small published-layout file the reader tests already hold
    run_case("6_40", os.path.join(OUT, "bal_problem-6-40-pre.txt.bz2"))
    # a zero rotation vector (0/0 in rotate(), ba_test.py:68-70) and shuffled observations
    p = syn.make_bal_problem(12, 300, 900, seed=46, shuffle=True)
    p["camera_params"][3, :3] = 0.0
    path = os.path.join(OUT, "bal_problem-12-300-pre.txt.bz2")
    write_bal_text(path, p)
    run_case("12_300", path)
    # the shape the script was written for: 49 cameras / 7 776 points / 31 843 observations
    p = syn.make_bal_problem(*syn.SHAPES["ladybug"], seed=47)
    path = os.path.join(OUT, "bal_problem-49-7776-pre.txt.bz2")
    write_bal_text(path, p)
    run_case("49_7776", path, full=False)


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3] == "full")
    else:
        main()
