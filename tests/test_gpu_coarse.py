"""GPU: the two-level preconditioner of the launch-per-iteration PCG (options.pcg_coarse) and of
the single-launch PCG.

No counterpart in the reference (its LSMR runs unpreconditioned, scipy/optimize/_lsq/trf.py:479);
the checks are against NumPy algebra on the matrices the engine hands out, and against the
block-Jacobi solver on the same problems.  Tolerances: 1e-8 relative for the preconditioner
(two inversions of a 7 n_clusters matrix in fp64), 1e-6 relative for converged costs."""
import numpy as np
import pytest

from structure_from_motion_amd import synthetic as syn
from structure_from_motion_amd import engine as E

pytestmark = pytest.mark.gpu


def _rotations(rvec):
    return syn._rodrigues(np.asarray(rvec, float))


def _problem(model, n_cam, n_pts, n_obs, seed):
    if model == "bal":
        p = syn.make_bal_problem(n_cam, n_pts, n_obs, seed=seed)
        p["K"] = None
    else:
        p = syn.make_problem(n_cam, n_pts, n_obs, seed=seed)
    return p


def _engine(p, model, **opts):
    eng = E.Engine(camera_model=model, linear_solver=1, schur_mode=2, **opts)
    eng.set_problem(p["camera_params"], p["points_3d"], p["camera_indices"], p["point_indices"], p["points_2d"], p["K"])
    return eng


def _coarse_basis(eng, model, cluster):
    """Z (cam_dim n_cam x 7 n_cl) as the header documents it, from the engine's current cameras."""
    cams, _ = eng.get_params()
    nc, nd = eng.n_cameras, eng.cam_dim
    if model == "bal":
        Q, t = _rotations(cams[:, :3]), cams[:, 3:6]
    else:
        Q, t = cams[:, :9].reshape(-1, 3, 3), cams[:, 9:12]
    V = np.zeros((nc, nd, 7))
    for a in range(3):
        u = Q[:, :, a]                      # Q e_a
        V[:, 3:6, a] = -u
        V[:, 0:3, 3 + a] = -u
        V[:, 3:6, 3 + a] = np.cross(u, t)
    V[:, 3:6, 6] = t
    n_cl = max(1, nc // cluster)
    Z = np.zeros((nc * nd, 7 * n_cl))
    for i in range(n_cl):
        lo, hi = i * nc // n_cl, (i + 1) * nc // n_cl
        Z[nd * lo:nd * hi, 7 * i:7 * i + 7] = V[lo:hi].reshape(-1, 7)
    return Z


@pytest.mark.parametrize("model,cluster", [("pinhole", 64), ("pinhole", 48), ("bal", 64)])
def test_preconditioner_is_block_jacobi_plus_coarse_correction(model, cluster):
    p = _problem(model, 330, 6000, 40000, seed=5)
    with _engine(p, model, pcg_coarse=cluster) as eng:
        eng.linearize(1e-5)
        nc, nd = eng.n_cameras, eng.cam_dim
        _, Minv = eng.get_reduced()
        Z = _coarse_basis(eng, model, cluster)
        SZ = np.stack([eng.schur_matvec(Z[:, j]).reshape(-1) for j in range(Z.shape[1])], axis=1)
        Em = Z.T @ SZ
        assert np.abs(Em - Em.T).max() <= 1e-9 * np.abs(Em).max()
        rng = np.random.default_rng(0)
        for _ in range(2):
            r = rng.normal(size=(nc, nd))
            z, dim = eng.precondition(r)
            assert dim == Z.shape[1] == 7 * (nc // cluster)
            want = np.einsum("nij,nj->ni", Minv, r).reshape(-1) + Z @ np.linalg.solve(Em, Z.T @ r.reshape(-1))
            assert np.abs(z.reshape(-1) - want).max() <= 1e-8 * np.abs(want).max()
        # the r'z the solver continues with is symmetric in its arguments
        a, b = rng.normal(size=(2, nc, nd))
        za, _ = eng.precondition(a)
        zb, _ = eng.precondition(b)
        assert abs(float((b * za).sum() - (a * zb).sum())) <= 1e-9 * abs(float((b * za).sum()))


def test_without_coarse_space_the_preconditioner_is_block_jacobi():
    p = _problem("pinhole", 60, 1500, 9000, seed=6)
    with _engine(p, "pinhole", pcg_mode=1) as eng:       # below 256 cameras no coarse space is built
        eng.linearize(1e-4)
        _, Minv = eng.get_reduced()
        r = np.random.default_rng(1).normal(size=(60, 6))
        z, dim = eng.precondition(r)
    assert dim == 0
    np.testing.assert_allclose(z, np.einsum("nij,nj->ni", Minv, r), rtol=1e-13, atol=0)


@pytest.mark.parametrize("model", ["pinhole", "bal"])
def test_two_level_solve_same_cost_fewer_iterations(model):
    p = _problem(model, 400, 30000, 200000, seed=7)
    out = {}
    for name, coarse in (("block-jacobi", -1), ("two-level", 64), ("auto", 0)):
        # (tolerances well below the reference's ftol = 1e-4: the three runs take different inexact
        # steps and are compared where they end, not where a loose test happens to stop them)
        with _engine(p, model, pcg_mode=1, pcg_coarse=coarse, ftol=1e-9, gtol=1e-12) as eng:
            out[name] = eng.solve()
            again = eng.residuals()
        assert np.isfinite(again).all()
        print(name, out[name].final_cost, out[name].iterations, out[name].pcg_iterations)
    ref = out["block-jacobi"]
    assert ref.pcg_coarse_solves == 0 and ref.pcg_coarse_dim == 0
    assert out["two-level"].pcg_coarse_solves == out["two-level"].iterations and out["two-level"].pcg_coarse_dim == 7 * (400 // 64)
    assert 0 < out["auto"].pcg_coarse_solves < out["auto"].iterations and out["auto"].pcg_coarse_dim == 7 * (400 // 48)
    for name in ("two-level", "auto"):
        st = out[name]
        assert abs(st.final_cost - ref.final_cost) <= 1e-6 * ref.final_cost
        assert st.pcg_iterations <= ref.pcg_iterations
    assert out["two-level"].pcg_iterations < 0.6 * ref.pcg_iterations


@pytest.mark.parametrize("model", ["pinhole", "bal"])
def test_restriction_formed_by_the_update_kernel_gives_the_same_solve(monkeypatch, model):
    # the tuning switch that folds the restriction Z'r into k_pcg_update (per workgroup and cluster
    # slot, put together by the prolongation): another order of additions, the same solve
    p = _problem(model, 400, 30000, 200000, seed=7)
    out = []
    for fused in ("0", "1"):
        monkeypatch.setenv("SFMBA_PCG_FUSED_RESTRICT", fused)
        with _engine(p, model, pcg_mode=1, pcg_coarse=64, ftol=1e-9, gtol=1e-12) as eng:
            out.append(eng.solve())
    a, b = out
    assert a.pcg_coarse_solves == a.iterations and b.pcg_coarse_solves == b.iterations
    assert a.iterations == b.iterations
    assert abs(a.pcg_iterations - b.pcg_iterations) <= 2
    assert abs(a.final_cost - b.final_cost) <= 1e-9 * a.final_cost


def test_two_level_solve_is_reproducible():
    p = _problem("pinhole", 300, 8000, 60000, seed=8)
    costs = []
    for _ in range(2):
        with _engine(p, "pinhole", pcg_coarse=64) as eng:
            st = eng.solve()
            costs.append((st.final_cost, st.pcg_iterations, st.iterations))
    assert costs[0] == costs[1]


def test_a_two_level_solve_that_does_not_converge_is_redone_without_the_coarse_space():
    # an iteration cap the later solves cannot meet: the first of them is redone with the
    # block-Jacobi blocks alone and the coarse space stays off for the rest of the run
    p = _problem("pinhole", 300, 8000, 60000, seed=8)
    with _engine(p, "pinhole", pcg_coarse=64, pcg_max_iter=3) as eng:
        capped = eng.solve()
    with _engine(p, "pinhole", pcg_coarse=-1, pcg_mode=1, pcg_max_iter=3) as eng:
        plain = eng.solve()
    assert 0 < capped.pcg_coarse_solves < capped.iterations
    assert np.isfinite(capped.final_cost) and capped.final_cost < 0.01 * capped.initial_cost
    assert capped.final_cost <= plain.final_cost * 1.05


@pytest.mark.parametrize("model", ["pinhole", "bal"])
def test_unobserved_camera_stays_where_it_is(model):
    # the reference's step for a camera nobody observes is exactly zero (its Jacobian columns are
    # empty); the coarse vectors of its cluster must not drag it along
    p = _problem(model, 130, 3000, 20000, seed=9)
    keep = p["camera_indices"] != 70
    q = dict(p, camera_indices=p["camera_indices"][keep], point_indices=p["point_indices"][keep], points_2d=p["points_2d"][keep])
    with _engine(q, model, pcg_coarse=48) as eng:
        st = eng.solve()
        cams, _ = eng.get_params()
    assert st.pcg_coarse_dim == 14 and st.pcg_coarse_solves == st.iterations
    assert st.final_cost < 0.05 * st.initial_cost
    if model == "bal":
        assert np.array_equal(cams[70, 3:], p["camera_params"][70, 3:])      # (the rotation vector goes through a matrix)
        assert np.abs(cams[70, :3] - p["camera_params"][70, :3]).max() <= 1e-12
    else:
        assert np.array_equal(cams[70, 9:], p["camera_params"][70, 9:])
        assert np.abs(cams[70, :9] - p["camera_params"][70, :9]).max() <= 1e-12
    assert np.abs(cams[69] - p["camera_params"][69]).max() > 1e-6            # its neighbours do move


def test_a_single_cluster_is_not_built():
    p = _problem("pinhole", 80, 2000, 12000, seed=10)
    with _engine(p, "pinhole", pcg_coarse=48) as eng:      # 80 // 48 = 1 cluster: the global gauge only
        st = eng.solve()
    assert st.pcg_coarse_dim == 0 and st.pcg_coarse_solves == 0


# ---- the same coarse space inside the single-launch PCG ------------------------------------

def _solve_single_launch(p, monkeypatch, coarse, model="pinhole", **opts):
    monkeypatch.setenv("SFMBA_PCG_FUSED_COARSE", "1" if coarse else "0")
    with _engine(p, model, pcg_mode=2, **opts) as eng:
        first = eng.solve()
        params = eng.get_params()
        eng.reset_params()
        second = eng.solve()
    return first, second, params


@pytest.mark.parametrize("n_cam", [100, 300, 700])
def test_single_launch_pcg_with_the_coarse_correction(monkeypatch, n_cam):
    # long camera chains: block-Jacobi needs many iterations late in the run; the coarse operator is
    # set up after the first long solve and used by the solves after it, inside the one launch
    p = syn.make_problem(n_cam, 25 * n_cam, 130 * n_cam, seed=21)
    plain, _, (cams0, pts0) = _solve_single_launch(p, monkeypatch, False, ftol=1e-10, pcg_rtol=0.03)
    st, again, (cams1, pts1) = _solve_single_launch(p, monkeypatch, True, ftol=1e-10, pcg_rtol=0.03)
    assert plain.pcg_coarse_solves == 0 and st.pcg_coarse_solves > 0
    assert st.pcg_single_launch == st.iterations and st.pcg_single_fallbacks == 0 and st.pcg_pipelined == 1
    # clusters as small as the camera graph lets the kernel hold (at most four among the columns of
    # a workgroup's rows), at least four and at most 36 of them, seven coarse unknowns each
    assert st.pcg_coarse_dim % 7 == 0 and 4 * 7 <= st.pcg_coarse_dim <= 36 * 7
    assert st.pcg_iterations < 0.7 * plain.pcg_iterations
    # the same minimum (a preconditioner changes the path of the inexact steps, not where they end)
    assert st.status > 0 and abs(st.final_cost - plain.final_cost) <= 1e-6 * plain.final_cost
    # (the points agree up to the free gauge, along which the two runs drift a little differently)
    assert np.abs(pts1 - pts0).max() <= 1e-2
    # every sum inside the kernel runs in a fixed order and the operator belongs to one iterate
    assert again.final_cost == st.final_cost and again.pcg_iterations == st.pcg_iterations and again.iterations == st.iterations


def test_single_launch_coarse_correction_is_skipped_where_its_limits_do_not_hold(monkeypatch):
    # too few cameras for four clusters of eight: no coarse space at all
    small = syn.make_problem(28, 700, 3700, seed=22)
    st, _, _ = _solve_single_launch(small, monkeypatch, True, ftol=1e-10)
    assert st.pcg_coarse_solves == 0 and st.pcg_coarse_dim == 0 and st.pcg_single_launch == st.iterations


@pytest.mark.parametrize("n_cam", [300, 700])
def test_single_launch_pcg_with_the_coarse_correction_and_the_nine_number_camera(monkeypatch, n_cam):
    """The same kernel with 9 x 9 blocks (the camera of /root/reference/ba_test.py:78-89; its
    intrinsics take no part in the coarse space): fewer iterations, the same minimum, repeatable."""
    b = _problem("bal", n_cam, 25 * n_cam, 130 * n_cam, seed=23)
    plain, _, (cams0, pts0) = _solve_single_launch(b, monkeypatch, False, model="bal", ftol=1e-10, pcg_rtol=0.03)
    st, again, (cams1, pts1) = _solve_single_launch(b, monkeypatch, True, model="bal", ftol=1e-10, pcg_rtol=0.03)
    assert plain.pcg_coarse_solves == 0 and st.pcg_coarse_solves > 0
    assert st.pcg_single_launch == st.iterations and st.pcg_single_fallbacks == 0 and st.pcg_pipelined == 1
    assert st.pcg_coarse_dim % 7 == 0 and 4 * 7 <= st.pcg_coarse_dim <= 36 * 7
    assert st.pcg_iterations < 0.7 * plain.pcg_iterations
    assert st.status > 0 and abs(st.final_cost - plain.final_cost) <= 1e-6 * plain.final_cost
    assert np.abs(pts1 - pts0).max() <= 1e-2
    assert again.final_cost == st.final_cost and again.pcg_iterations == st.pcg_iterations and again.iterations == st.iterations
