"""GPU: the triangulation entry point that allocates nothing (sfmba_triangulate_noalloc) against
sfmba_triangulate through the C ABI.  Same kernel, same inputs: results must be bit-identical.
Two calls in a row, the second with a smaller input, so that whatever the first call left in the
library's fixed device array cannot leak into the second."""
import numpy as np
import pytest

from oracle import triangulation_oracle as to
from structure_from_motion_amd.engine import Engine, EngineError, TRIANGULATE_NOALLOC_MAX

pytestmark = pytest.mark.gpu


def _problem(n, seed):
    p = to.two_view_problem(n, seed=seed, noise_px=0.4)
    return (to.projection(p["K"], p["R1"], p["t1"]), to.projection(p["K"], p["R2"], p["t2"]), p["x1"], p["x2"])


def test_noalloc_equals_allocating_entry_point_on_two_calls_in_a_row():
    big, small = _problem(50_000, seed=21), _problem(777, seed=22)
    with Engine() as eng:
        want_big = eng.triangulate(*big)
        want_small = eng.triangulate(*small)
        got_big = eng.triangulate(*big, noalloc=True)
        got_small = eng.triangulate(*small, noalloc=True)       # the array still holds the first call's data
    for got, want in ((got_big, want_big), (got_small, want_small)):
        for g, w in zip(got, want):
            assert g.shape == w.shape and np.array_equal(g, w)
    # and against the CPU oracle, to the tolerance of the allocating entry point's own test
    X = to.triangulate(small[0], small[1], small[2], small[3])
    assert np.abs(got_small[0] - X).max() <= 1e-9


def test_noalloc_limit_and_empty_input():
    P = np.eye(3, 4)
    with Engine() as eng:
        X, e1, e2 = eng.triangulate(P, P, np.zeros((0, 2)), np.zeros((0, 2)), noalloc=True)
        assert X.shape == (0, 3) and e1.shape == (0,) and e2.shape == (0,)
        full = _problem(TRIANGULATE_NOALLOC_MAX, seed=23)
        got = eng.triangulate(*full, noalloc=True)             # exactly the limit is accepted
        want = eng.triangulate(*full)
        assert all(np.array_equal(g, w) for g, w in zip(got, want))
        n = TRIANGULATE_NOALLOC_MAX + 1
        with pytest.raises(EngineError, match="exceed the limit of 65536"):
            eng.triangulate(P, P, np.zeros((n, 2)), np.zeros((n, 2)), noalloc=True)


def test_noalloc_on_a_second_handle_of_the_same_gpu():
    small = _problem(300, seed=24)
    with Engine() as a, Engine() as b:
        ra = a.triangulate(*small, noalloc=True)
        rb = b.triangulate(*small, noalloc=True)                # same device, same array, one call after the other
        want = a.triangulate(*small)
    assert all(np.array_equal(x, w) and np.array_equal(y, w) for x, y, w in zip(ra, rb, want))
