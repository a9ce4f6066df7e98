"""CPU: the triangulation oracle against its committed fixtures and against properties of the
DLT solution (no GPU)."""
import os

import numpy as np
import pytest

from helpers import GOLDEN_DIR, golden_names
from oracle import triangulation_oracle as to


def load_tri(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False) as z:
        return {k[4:]: z[k] for k in z.files}


@pytest.mark.parametrize("name", golden_names("tri_X"))
def test_oracle_reproduces_fixture(name):
    g = load_tri(name)
    P1, P2 = to.projection(g["K"], g["R1"], g["t1"]), to.projection(g["K"], g["R2"], g["t2"])
    X = to.triangulate(P1, P2, g["x1"], g["x2"])
    assert np.abs(X - g["X"]).max() <= 1e-12
    assert np.abs(to.reprojection_errors(P1, X, g["x1"]) - g["err1"]).max() <= 1e-9


def test_noise_free_points_are_recovered():
    g = load_tri("tri_exact_64")
    assert np.abs(g["X"] - g["X_true"]).max() <= 1e-9
    assert g["err1"].max() <= 1e-9 and g["err2"].max() <= 1e-9


def test_solution_is_the_smallest_right_singular_vector():
    g = load_tri("tri_two_view_400")
    P1, P2 = to.projection(g["K"], g["R1"], g["t1"]), to.projection(g["K"], g["R2"], g["t2"])
    for i in range(0, 400, 37):
        A = np.array([g["x1"][i, 0] * P1[2] - P1[0], g["x1"][i, 1] * P1[2] - P1[1],
                      g["x2"][i, 0] * P2[2] - P2[0], g["x2"][i, 1] * P2[2] - P2[1]])
        w, V = np.linalg.eigh(A.T @ A)
        h = V[:, 0]
        assert np.abs(h[:3] / h[3] - g["X"][i]).max() <= 1e-7


def test_reference_return_conventions():
    p = to.two_view_problem(30, seed=5)
    X = to.triangulate_points(p["K"], p["t1"], p["R1"], p["t2"], p["R2"], p["x1"], p["x2"])
    assert X.shape == (30, 1, 3)
    assert to.triangulate_points(p["K"], p["t1"], p["R1"], p["t2"], p["R2"], p["x1"][:7], p["x2"][:7]) is None
    bad = p["x2"] + 500.0                   # mean reprojection error far above 55 pixels
    assert to.triangulate_points(p["K"], p["t1"], p["R1"], p["t2"], p["R2"], p["x1"], bad) is None


def test_dehomogenize_guards_a_zero_last_coordinate():
    h = np.array([[2.0, 4.0, 2.0], [3.0, 5.0, 0.0]])
    assert np.array_equal(to.dehomogenize(h), np.array([[1.0, 2.0], [3.0, 5.0]]))
