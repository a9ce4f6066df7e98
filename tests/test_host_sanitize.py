"""GPU box: the library's HOST code under AddressSanitizer + UndefinedBehaviorSanitizer.

tests/host_sanitize/Makefile builds libsfmba with every -fsanitize= behind -Xarch_host (the gfx950
code objects are untouched) and the plain-C caller of tests/c_client/client.c against it.  The caller
loads a problem, evaluates residuals, solves, reads the result back and provokes the index-range
error; here it runs on an irregular problem (long tracks, single observations, duplicates, an
unobserved camera), on a problem without observations and on a small regular one, through both
set-up paths - the host path (counting sorts, pair lists, chunks and wave tiles on host threads) and
the device path, whose host side is the argument checks, read-backs and the coarse-space set-up."""
import os
import struct
import subprocess

import numpy as np
import pytest

from helpers import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "host_sanitize", "build", "client_hostsan")

pytestmark = pytest.mark.gpu


def _write(path, g):
    with open(path, "wb") as fh:
        fh.write(struct.pack("<3q", g["n_cameras"], g["n_points"], g["n_obs"]))
        for a, dt in ((g["K"], "<f8"), (g["camera_params"], "<f8"), (g["points_3d"], "<f8"),
                      (g["camera_indices"], "<i8"), (g["point_indices"], "<i8"), (g["points_2d"], "<f8")):
            fh.write(np.ascontiguousarray(a, dtype=dt).tobytes())


def _problems():
    from test_gpu_irregular import irregular_problem
    g = load_golden("shuffled_9_200_700")
    empty = dict(g, camera_indices=np.zeros(0, np.int64), point_indices=np.zeros(0, np.int64),
                 points_2d=np.zeros((0, 2)), n_obs=0)
    return {"irregular": irregular_problem(31), "irregular_small": irregular_problem(7, n_cam=12, n_pts=90),
            "regular": g, "empty": empty}


@pytest.mark.parametrize("setup", ["host", "device"])
@pytest.mark.parametrize("name", ["irregular", "irregular_small", "regular", "empty"])
def test_host_code_is_clean_under_the_sanitizers(tmp_path, name, setup):
    if not os.path.exists(EXE):
        pytest.skip("tests/host_sanitize/build/client_hostsan not built (make -C tests/host_sanitize)")
    g = _problems()[name]
    path = tmp_path / "problem.bin"
    _write(path, g)
    env = dict(os.environ, SFMBA_SETUP=setup,
               # (the HIP runtime keeps its allocations until the process ends: no leak report)
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1:protect_shadow_gap=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([EXE, str(path)], capture_output=True, text=True, timeout=300, env=env)
    report = r.stderr
    assert "AddressSanitizer" not in report and "runtime error" not in report, report[-4000:]
    if name == "empty":
        # nothing to solve: either a clean refusal or a solve that ends at once - never a crash
        assert r.returncode in (0, 1), report[-2000:]
        return
    assert r.returncode == 0, report[-2000:]
    out = dict(line.split(" ", 1) for line in r.stdout.strip().split("\n"))
    assert int(out["status"].split()[0]) > 0 and float(out["final_cost"]) <= float(out["initial_cost"])
    assert out["bad_index_rc"] == "-2" and "out of bounds" in out["bad_index_msg"]
