"""CPU: the temple-ring calibration file and the synthetic tracks built on it (tests/temple_ring.py)."""
import numpy as np

import temple_ring as tr


def test_calibration_file_reads_as_documented():
    names, K, R, t = tr.read_calibration()
    assert len(names) == 47 and names[0] == "templeR0001.png" and names[-1] == "templeR0047.png"
    assert K[0, 0] == 1520.4 and K[1, 1] == 1525.9 and K[0, 2] == 302.32 and K[1, 2] == 246.87
    assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() < 1e-12 and np.all(np.linalg.det(R) > 0.999)
    # the box lies in front of every camera
    for j in range(47):
        assert tr.project(K, R[j], t[j], np.array([0.5 * (tr.BOX_LO + tr.BOX_HI)]))[1][0] > 0.3


def test_tracks_join_neighbouring_views_inside_the_image():
    names, K, R, t, pairs = tr.make_tracks(per_pair=12, seed=3, noise_px=0.0)
    assert len(pairs) == 46
    for j, pr in enumerate(pairs):
        assert pr["X"].shape == (12, 3) and np.all(pr["X"] >= tr.BOX_LO) and np.all(pr["X"] <= tr.BOX_HI)
        for v, uv in ((j, pr["uv_a"]), (j + 1, pr["uv_b"])):
            want, z = tr.project(K, R[v], t[v], pr["X"])
            assert np.all(z > 0) and np.array_equal(uv, want)
            assert np.all(uv >= 0) and np.all(uv[:, 0] < tr.WIDTH) and np.all(uv[:, 1] < tr.HEIGHT)
