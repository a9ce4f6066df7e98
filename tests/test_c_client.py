"""The boundary is a plain C ABI: include/sfmba.h compiles as C99 and a caller written in C
(tests/c_client/client.c) solves a problem through libsfmba.so without any Python in the data path."""
import os
import struct
import subprocess

import numpy as np
import pytest

from helpers import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_client", "client.c")
BIN = os.path.join(ROOT, "tests", "c_client", "client")
LIBDIR = os.path.join(ROOT, "structure_from_motion_amd")


def build_client(out=BIN):
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), SRC, "-o", out,
           "-L" + LIBDIR, "-lsfmba", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath-link,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)


def test_header_compiles_as_c99(tmp_path):
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
                        "-c", SRC, "-o", str(tmp_path / "client.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_client_links_against_the_library(tmp_path):
    if not os.path.exists(os.path.join(LIBDIR, "libsfmba.so")):
        pytest.skip("libsfmba.so not built")
    build_client(str(tmp_path / "client"))


@pytest.mark.gpu
def test_c_caller_gets_the_same_solve_as_the_python_binding(tmp_path):
    from structure_from_motion_amd.engine import Engine
    g = load_golden("shuffled_9_200_700")
    path = tmp_path / "problem.bin"
    with open(path, "wb") as fh:
        fh.write(struct.pack("<3q", g["n_cameras"], g["n_points"], g["n_obs"]))
        for a, dt in ((g["K"], "<f8"), (g["camera_params"], "<f8"), (g["points_3d"], "<f8"),
                      (g["camera_indices"], "<i8"), (g["point_indices"], "<i8"), (g["points_2d"], "<f8")):
            fh.write(np.ascontiguousarray(a, dtype=dt).tobytes())
    exe = BIN
    if not os.path.exists(exe):                       # a host-only C file: seconds with gcc
        exe = str(tmp_path / "client")
        build_client(exe)
    r = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = dict(line.split(" ", 1) for line in r.stdout.strip().split("\n"))
    with Engine() as eng:
        eng.set_problem(g["camera_params"], g["points_3d"], g["camera_indices"], g["point_indices"], g["points_2d"], g["K"])
        st = eng.solve()
        poses, pts = eng.get_params()
    assert out["status"].split()[0] == str(st.status) and int(out["status"].split()[2]) == st.iterations
    assert float(out["initial_cost"]) == st.initial_cost and float(out["final_cost"]) == st.final_cost
    assert abs(float(out["cost_from_residuals"]) - st.initial_cost) <= 1e-12 * st.initial_cost
    assert [float(x) for x in out["camera0"].split()] == list(poses[0, 9:12])
    assert [float(x) for x in out["point0"].split()] == list(pts[0])
    assert out["bad_index_rc"] == "-2" and "out of bounds" in out["bad_index_msg"]
