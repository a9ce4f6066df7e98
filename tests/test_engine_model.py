"""CPU: the NumPy model of the engine's algorithm against the oracle's golden vectors.
This validates the mathematics (analytic Jacobian, chain rule to the 12-number camera,
Schur/PCG Levenberg-Marquardt) that the GPU tests then check kernel by kernel."""
import numpy as np
import pytest

import engine_model as em
from helpers import (golden_names, load_golden, golden_jacobian, rel_err,
                     blocks_to_reference_jacobian, identity_snapped_cameras)


@pytest.mark.parametrize("name", golden_names())
def test_model_residuals(name):
    g = load_golden(name)
    m = em.Model(g)
    # the model does not apply the cv2 small-angle snap; the fixtures' identity cameras are exact
    r, _, _ = m.residuals()
    assert np.abs(r.ravel() - g["f0"]).max() <= 1e-10 * max(1.0, np.abs(g["f0"]).max())


@pytest.mark.parametrize("name", golden_names("jac_data"))
def test_analytic_jacobian_matches_scipy_finite_differences(name):
    g = load_golden(name)
    m = em.Model(g)
    _, Jc, Jp = m.jacobian_blocks()
    J = blocks_to_reference_jacobian(Jc, Jp, g["camera_params"], g["camera_indices"],
                                     g["point_indices"], g["n_points"]).toarray()
    Jfd = golden_jacobian(g).toarray()
    snapped = identity_snapped_cameras(g["camera_params"])
    for c in snapped:   # cv2 quirk: no finite-difference response in these rotation columns
        assert np.all(Jfd[:, 12 * c:12 * c + 9] == 0)
        J[:, 12 * c:12 * c + 9] = 0
    assert np.abs(J - Jfd).max() <= 1e-5 * np.abs(Jfd).max()


@pytest.mark.parametrize("name", golden_names("tight_cost"))
def test_model_lm_reaches_oracle_cost(name):
    g = load_golden(name)
    if g["n_obs"] > 5000:
        pytest.skip("NumPy model is slow at this size; covered on the GPU")
    _, info = em.lm_solve(g, ftol=1e-4)
    assert info["cost"] <= float(g["ref_cost"]) * (1 + 1e-6)
    _, info = em.lm_solve(g, ftol=1e-15, xtol=1e-15, max_iter=60, pcg_tol=1e-3)
    assert abs(info["cost"] - float(g["tight_cost"])) <= 1e-6 * float(g["tight_cost"])


def test_dense_and_pcg_steps_agree():
    g = load_golden("tiny_4_50_150")
    _, a = em.lm_solve(g, ftol=1e-12, max_iter=8, dense=True)
    _, b = em.lm_solve(g, ftol=1e-12, max_iter=8, pcg_tol=1e-10)
    assert abs(a["cost"] - b["cost"]) <= 1e-9 * a["cost"]
