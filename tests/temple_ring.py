"""The Middlebury temple ring as a bundle-adjustment workload (test helper).

tests/golden/templeR_par.txt is the calibration file that ships with the data set the reference
reconstructs (/root/reference/images/datasets/templeRing/templeR_par.txt: one line per image,
"name k11..k33 r11..r33 t1 t2 t3", projection K [R t]); the images themselves are not needed here.
Points are drawn inside the model's published bounding box, every track joins two neighbouring
views of the ring (the reference matches each new image against the one before it and triangulates
one world point per match: main.py:84-113, WorldPoints.py:22-43), observations are exact
projections plus pixel noise."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BOX_LO = np.array([-0.023121, -0.038009, -0.091940])      # README of the data set
BOX_HI = np.array([0.078626, 0.121636, -0.017395])
WIDTH, HEIGHT = 640, 480


def read_calibration(path=os.path.join(HERE, "golden", "templeR_par.txt")):
    with open(path) as f:
        n = int(f.readline())
        rows = [f.readline().split() for _ in range(n)]
    vals = np.array([[float(v) for v in r[1:]] for r in rows])
    assert vals.shape == (n, 21)
    K = vals[:, :9].reshape(n, 3, 3)
    assert np.all(K == K[0])                              # one camera, 47 poses
    return [r[0] for r in rows], K[0], vals[:, 9:18].reshape(n, 3, 3), vals[:, 18:21]


def project(K, R, t, X):
    Y = X @ R.T + t
    return (Y[:, :2] / Y[:, 2:3]) * np.array([K[0, 0], K[1, 1]]) + np.array([K[0, 2], K[1, 2]]), Y[:, 2]


def make_tracks(per_pair=40, seed=5, noise_px=0.3):
    """Tracks between neighbouring views: for every pair (j, j + 1) `per_pair` points of the box that
    both views see.  Returns names, K, R, t and a list (one entry per pair) of dicts with the true
    points (n x 3) and their noisy pixels in both views (n x 2 each)."""
    names, K, R, t = read_calibration()
    rng = np.random.default_rng(seed)
    pairs = []
    for j in range(len(names) - 1):
        X = np.empty((0, 3))
        while len(X) < per_pair:
            c = rng.uniform(BOX_LO, BOX_HI, size=(4 * per_pair, 3))
            ok = np.ones(len(c), bool)
            for v in (j, j + 1):
                uv, z = project(K, R[v], t[v], c)
                ok &= (z > 0) & (uv[:, 0] >= 0) & (uv[:, 0] < WIDTH) & (uv[:, 1] >= 0) & (uv[:, 1] < HEIGHT)
            X = np.vstack((X, c[ok]))
        X = X[:per_pair]
        uv = [project(K, R[v], t[v], X)[0] + rng.normal(0, noise_px, (per_pair, 2)) for v in (j, j + 1)]
        pairs.append(dict(X=X, uv_a=uv[0], uv_b=uv[1]))
    return names, K, R, t, pairs
