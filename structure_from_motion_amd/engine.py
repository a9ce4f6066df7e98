"""ctypes binding of libsfmba.so (declared in include/sfmba.h).

`Engine` owns one `sfmba_handle`, i.e. one bundle-adjustment problem resident on one MI355X.
Every method is a thin call through the C ABI; there is no CPU fallback: if the shared
library cannot be loaded, or no GPU is present, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

# SFMBA_LIBRARY selects another build of the same library (A/B measurements of kernel variants)
_LIB_PATH = os.environ.get("SFMBA_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsfmba.so")

# every symbol include/sfmba.h declares
EXPORTS = (
    "sfmba_default_options", "sfmba_create", "sfmba_destroy", "sfmba_last_error", "sfmba_set_print",
    "sfmba_set_problem", "sfmba_set_problem_device", "sfmba_set_params", "sfmba_reset_params", "sfmba_residuals", "sfmba_jacobian_blocks",
    "sfmba_solve", "sfmba_get_params", "sfmba_reprojection_errors", "sfmba_triangulate", "sfmba_triangulate_noalloc", "sfmba_point_errors",
    "sfmba_linearize", "sfmba_get_blocks",
    "sfmba_schur_matvec", "sfmba_get_reduced", "sfmba_precondition", "sfmba_solve_step", "sfmba_get_structure",
    "sfmba_comm_unique_id", "sfmba_comm_init",
    "sfmba_local_group_create", "sfmba_local_group_destroy", "sfmba_comm_init_local",
)

SOLVER_AUTO, SOLVER_PCG, SOLVER_DENSE = 0, 1, 2
SCHUR_AUTO, SCHUR_IMPLICIT, SCHUR_EXPLICIT = 0, 1, 2
CAMERA_PINHOLE, CAMERA_BAL = 0, 1
TRIANGULATE_NOALLOC_MAX = 65536     # SFMBA_TRIANGULATE_NOALLOC_MAX of include/sfmba.h
# per camera model: numbers per camera in camera_params, tangent numbers per camera
CAMERA_SHAPES = {CAMERA_PINHOLE: (12, 6), CAMERA_BAL: (9, 9)}
CAMERA_NAMES = {"pinhole": CAMERA_PINHOLE, "bal": CAMERA_BAL}

TERMINATION_MESSAGES = {   # scipy/optimize/_lsq/least_squares.py:18-25
    -1: "Improper input parameters status returned from `leastsq`",
    0: "The maximum number of function evaluations is exceeded.",
    1: "`gtol` termination condition is satisfied.",
    2: "`ftol` termination condition is satisfied.",
    3: "`xtol` termination condition is satisfied.",
    4: "Both `ftol` and `xtol` termination conditions are satisfied.",
}


PRINT_SINK = C.CFUNCTYPE(None, C.c_char_p, C.c_void_p)


@PRINT_SINK
def _print_to_sys_stdout(text, _user):
    # the iteration table goes where SciPy's print() would put it: the current sys.stdout
    sys.stdout.write(text.decode("utf-8", "replace"))


class Options(C.Structure):
    _fields_ = [
        ("ftol", C.c_double), ("xtol", C.c_double), ("gtol", C.c_double),
        ("max_nfev", C.c_int32), ("max_iterations", C.c_int32),
        ("initial_lambda", C.c_double), ("pcg_rtol", C.c_double),
        ("pcg_max_iter", C.c_int32), ("pcg_batch", C.c_int32), ("linear_solver", C.c_int32), ("schur_mode", C.c_int32),
        ("dense_max_dim", C.c_int32), ("verbose", C.c_int32), ("profile", C.c_int32),
        ("device", C.c_int32), ("rank", C.c_int32), ("nranks", C.c_int32),
        ("pcg_mode", C.c_int32), ("pcg_blocks", C.c_int32), ("camera_model", C.c_int32), ("pcg_coarse", C.c_int32),
        ("setup_mode", C.c_int32), ("matrix_split", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("iterations", C.c_int32), ("accepted", C.c_int32),
        ("nfev", C.c_int32), ("njev", C.c_int32), ("pcg_iterations", C.c_int32),
        ("linear_solver", C.c_int32), ("schur_mode", C.c_int32),
        ("initial_cost", C.c_double), ("final_cost", C.c_double),
        ("optimality", C.c_double), ("solve_seconds", C.c_double),
        ("ms_linearize", C.c_double), ("n_linearize", C.c_int64),
        ("ms_damp", C.c_double), ("n_damp", C.c_int64),
        ("ms_spmv_pm", C.c_double), ("n_spmv_pm", C.c_int64),
        ("ms_spmv_cm", C.c_double), ("n_spmv_cm", C.c_int64),
        ("ms_pcg_vec", C.c_double), ("n_pcg_vec", C.c_int64),
        ("ms_back", C.c_double), ("n_back", C.c_int64),
        ("ms_cost", C.c_double), ("n_cost", C.c_int64),
        ("ms_dense", C.c_double), ("n_dense", C.c_int64),
        ("ms_schur", C.c_double), ("n_schur", C.c_int64),
        ("ms_pairs", C.c_double), ("n_pairs", C.c_int64),
        ("ms_bsr", C.c_double), ("n_bsr", C.c_int64),
        ("schur_pairs", C.c_int64), ("schur_blocks", C.c_int64), ("schur_chunks", C.c_int64),
        ("pcg_single_launch", C.c_int64), ("pcg_single_fallbacks", C.c_int64),
        ("pcg_coarse_solves", C.c_int64), ("pcg_coarse_dim", C.c_int64),
        ("ms_comm", C.c_double), ("n_comm", C.c_int64),
        ("pcg_pipelined", C.c_int64),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_ }


class EngineError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libsfmba error {code}: {message}")
        self.code = code


_lib = None


def load_library(path=None):
    """Load libsfmba.so and declare its prototypes.  Raises OSError when it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    lib = C.CDLL(path or _LIB_PATH)
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_void_p
    lib.sfmba_default_options.argtypes = [C.POINTER(Options)]
    lib.sfmba_default_options.restype = None
    lib.sfmba_create.argtypes = [C.POINTER(Options), C.POINTER(vp)]
    lib.sfmba_destroy.argtypes = [vp]
    lib.sfmba_last_error.argtypes = [vp]
    lib.sfmba_last_error.restype = C.c_char_p
    lib.sfmba_set_print.argtypes = [vp, PRINT_SINK, vp]
    lib.sfmba_set_problem.argtypes = [vp, C.c_int64, C.c_int64, C.c_int64, dp, dp, ip, ip, dp, dp]
    lib.sfmba_set_problem_device.argtypes = [vp, C.c_int64, C.c_int64, C.c_int64, dp, dp, vp, vp, vp, dp]
    lib.sfmba_set_params.argtypes = [vp, dp, dp]
    lib.sfmba_reset_params.argtypes = [vp]
    lib.sfmba_residuals.argtypes = [vp, dp]
    lib.sfmba_jacobian_blocks.argtypes = [vp, dp, dp]
    lib.sfmba_solve.argtypes = [vp, C.POINTER(Stats)]
    lib.sfmba_get_params.argtypes = [vp, dp, dp]
    lib.sfmba_reprojection_errors.argtypes = [vp, dp, ip]
    lib.sfmba_triangulate.argtypes = [vp, C.c_int64, dp, dp, dp, dp, dp, dp, dp]
    lib.sfmba_triangulate_noalloc.argtypes = [vp, C.c_int64, dp, dp, dp, dp, dp, dp, dp]
    lib.sfmba_point_errors.argtypes = [vp, C.c_int64, dp, dp, dp, dp]
    lib.sfmba_linearize.argtypes = [vp, C.c_double]
    lib.sfmba_get_blocks.argtypes = [vp, dp, dp, dp, dp]
    lib.sfmba_schur_matvec.argtypes = [vp, dp, dp]
    lib.sfmba_get_reduced.argtypes = [vp, dp, dp]
    lib.sfmba_precondition.argtypes = [vp, dp, dp, C.POINTER(C.c_int32)]
    lib.sfmba_solve_step.argtypes = [vp, dp, dp, C.POINTER(C.c_int32)]
    lib.sfmba_get_structure.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int32), C.c_int64, C.POINTER(C.c_int64)]
    lib.sfmba_comm_unique_id.argtypes = [vp]
    lib.sfmba_comm_init.argtypes = [vp, vp]
    lib.sfmba_local_group_create.argtypes = [C.c_int32, C.POINTER(vp)]
    lib.sfmba_local_group_destroy.argtypes = [vp]
    lib.sfmba_comm_init_local.argtypes = [vp, vp]
    for name in EXPORTS:
        if name not in ("sfmba_default_options", "sfmba_last_error"):
            getattr(lib, name).restype = C.c_int
    if path is None:
        _lib = lib
    return lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def default_options(**kw):
    lib = load_library()
    o = Options()
    lib.sfmba_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError(f"unknown option {k!r}")
        setattr(o, k, v)
    return o


class Engine:
    """One problem on one GPU.  Array arguments use the reference's layouts
    (/root/reference/bundle_adjustment.py:94-126)."""

    def __init__(self, **options):
        self._lib = load_library()
        self._h = C.c_void_p()
        model = options.get("camera_model", CAMERA_PINHOLE)
        if isinstance(model, str):
            if model not in CAMERA_NAMES:
                raise ValueError(f"unknown camera model {model!r}")
            options = dict(options, camera_model=CAMERA_NAMES[model])
        self.options = default_options(**options)
        if self.options.camera_model not in CAMERA_SHAPES:
            raise ValueError(f"unknown camera model {self.options.camera_model}")
        # numbers per camera in camera_params / in a camera's tangent space
        self.cam_params, self.cam_dim = CAMERA_SHAPES[self.options.camera_model]
        rc = self._lib.sfmba_create(C.byref(self.options), C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise EngineError(rc, {-6: "no usable GPU (libsfmba needs a gfx950 device)"}.get(rc, "sfmba_create failed"))
        self.n_cameras = self.n_points = self.n_obs = 0
        self._lib.sfmba_set_print(self._h, _print_to_sys_stdout, None)

    # -- plumbing
    def _check(self, rc):
        if rc != 0:
            msg = self._lib.sfmba_last_error(self._h).decode("utf-8", "replace")
            if rc == -2:
                raise IndexError(msg)          # NumPy's behaviour at bundle_adjustment.py:49
            if rc == -3:
                raise ValueError(msg)          # SciPy's behaviour, least_squares.py:844-845
            raise EngineError(rc, msg)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.sfmba_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- problem
    def set_problem(self, camera_params, points_3d, camera_indices, point_indices, points_2d, K):
        cp, pts = _f64(camera_params), _f64(points_3d)
        ci, pi, uv = _i64(camera_indices), _i64(point_indices), _f64(points_2d)
        if K is None and self.options.camera_model == CAMERA_PINHOLE:
            raise ValueError("K must be 3x3")
        K = _f64(K) if K is not None else None
        if cp.ndim != 2 or cp.shape[1] != self.cam_params:
            raise ValueError(f"camera_params must be (n_cameras, {self.cam_params})")
        if pts.ndim != 2 or pts.shape[1] != 3:
            raise ValueError("points_3d must be (n_points, 3)")
        if ci.shape != pi.shape or ci.ndim != 1 or uv.shape != (ci.size, 2):
            raise ValueError("camera_indices, point_indices (n_obs,) and points_2d (n_obs, 2) disagree")
        if K is not None and K.shape != (3, 3):
            raise ValueError("K must be 3x3")
        self.n_cameras, self.n_points, self.n_obs = cp.shape[0], pts.shape[0], ci.size
        ipp = C.POINTER(C.c_int64)
        self._check(self._lib.sfmba_set_problem(
            self._h, cp.shape[0], pts.shape[0], ci.size, _dp(cp), _dp(pts),
            ci.ctypes.data_as(ipp), pi.ctypes.data_as(ipp), _dp(uv), _dp(K)))

    def set_problem_device(self, camera_params, points_3d, camera_indices, point_indices, points_2d, K):
        """set_problem with the observations already on this engine's GPU: camera_indices,
        point_indices (int64, n_obs) and points_2d (float64, n_obs x 2) are contiguous device
        tensors (anything with data_ptr(), shape and dtype, e.g. torch.Tensor); the rest as in
        set_problem."""
        cp, pts = _f64(camera_params), _f64(points_3d)
        K = _f64(K) if K is not None else None
        if cp.ndim != 2 or cp.shape[1] != self.cam_params:
            raise ValueError(f"camera_params must be (n_cameras, {self.cam_params})")
        if pts.ndim != 2 or pts.shape[1] != 3:
            raise ValueError("points_3d must be (n_points, 3)")
        if K is None and self.options.camera_model == CAMERA_PINHOLE or (K is not None and K.shape != (3, 3)):
            raise ValueError("K must be 3x3")
        n = int(camera_indices.shape[0])
        if tuple(point_indices.shape) != (n,) or tuple(points_2d.shape) != (n, 2) or len(camera_indices.shape) != 1:
            raise ValueError("camera_indices, point_indices (n_obs,) and points_2d (n_obs, 2) disagree")
        for t, want in ((camera_indices, "int64"), (point_indices, "int64"), (points_2d, "float64")):
            if want not in str(t.dtype) or not t.is_contiguous():
                raise ValueError(f"device arrays must be contiguous {want}")
        self.n_cameras, self.n_points, self.n_obs = cp.shape[0], pts.shape[0], n
        self._check(self._lib.sfmba_set_problem_device(
            self._h, cp.shape[0], pts.shape[0], n, _dp(cp), _dp(pts),
            C.c_void_p(camera_indices.data_ptr()), C.c_void_p(point_indices.data_ptr()),
            C.c_void_p(points_2d.data_ptr()), _dp(K)))

    def set_params(self, camera_params, points_3d):
        cp, pts = _f64(camera_params), _f64(points_3d)
        if cp.shape != (self.n_cameras, self.cam_params) or pts.shape != (self.n_points, 3):
            raise ValueError("parameter shapes do not match the loaded problem")
        self._check(self._lib.sfmba_set_params(self._h, _dp(cp), _dp(pts)))

    def reset_params(self):
        self._check(self._lib.sfmba_reset_params(self._h))

    def get_params(self):
        cp = np.empty((self.n_cameras, self.cam_params))
        pts = np.zeros((self.n_points, 3))
        self._check(self._lib.sfmba_get_params(self._h, _dp(cp), _dp(pts)))
        return cp, pts

    # -- evaluation
    def residuals(self):
        out = np.zeros(2 * self.n_obs)
        self._check(self._lib.sfmba_residuals(self._h, _dp(out)))
        return out

    def jacobian_blocks(self):
        jc = np.zeros((self.n_obs, 2, self.cam_dim))
        jp = np.zeros((self.n_obs, 2, 3))
        self._check(self._lib.sfmba_jacobian_blocks(self._h, _dp(jc), _dp(jp)))
        return jc, jp

    def solve(self):
        st = Stats()
        self._check(self._lib.sfmba_solve(self._h, C.byref(st)))
        return st

    def reprojection_errors(self):
        """(mean reprojection error per camera, observation count per camera) at the current estimate."""
        mean = np.empty(self.n_cameras)
        cnt = np.zeros(self.n_cameras, dtype=np.int64)
        self._check(self._lib.sfmba_reprojection_errors(self._h, _dp(mean), cnt.ctypes.data_as(C.POINTER(C.c_int64))))
        return mean, cnt

    def triangulate(self, P1, P2, x1, x2, noalloc=False):
        """Batched two-view linear triangulation: (X n x 3, err1, err2) for 3x4 projection matrices
        P1, P2 and n x 2 pixel coordinates x1, x2 (see sfmba_triangulate).  `noalloc` takes the
        entry point that stages through the library's fixed device array instead of allocating
        (at most TRIANGULATE_NOALLOC_MAX correspondences per call).  That array is one per GPU,
        shared by every engine on it and not locked by the library: calls with ``noalloc`` that
        overlap on one GPU (two threads, two engines) race - the caller serialises them."""
        P1 = _f64(P1).reshape(3, 4); P2 = _f64(P2).reshape(3, 4)
        x1 = _f64(x1).reshape(-1, 2); x2 = _f64(x2).reshape(-1, 2)
        n = x1.shape[0]
        if x2.shape[0] != n:
            raise ValueError("x1 and x2 must hold the same number of points")
        X = np.empty((n, 3)); e1 = np.empty(n); e2 = np.empty(n)
        call = self._lib.sfmba_triangulate_noalloc if noalloc else self._lib.sfmba_triangulate
        self._check(call(self._h, n, _dp(P1), _dp(P2), _dp(x1), _dp(x2), _dp(X), _dp(e1), _dp(e2)))
        return X, e1, e2

    def point_errors(self, P, X, x):
        """Reprojection error of every point X[i] against x[i] under the 3x4 projection matrix P."""
        P = _f64(P).reshape(3, 4)
        X = _f64(X).reshape(-1, 3); x = _f64(x).reshape(-1, 2)
        n = X.shape[0]
        if x.shape[0] != n:
            raise ValueError("X and x must hold the same number of points")
        e = np.empty(n)
        self._check(self._lib.sfmba_point_errors(self._h, n, _dp(P), _dp(X), _dp(x), _dp(e)))
        return e

    # -- diagnostics
    def linearize(self, lam):
        self._check(self._lib.sfmba_linearize(self._h, float(lam)))

    def get_blocks(self):
        U = np.zeros((self.n_cameras, self.cam_dim, self.cam_dim)); gc = np.zeros((self.n_cameras, self.cam_dim))
        V = np.zeros((self.n_points, 3, 3)); gp = np.zeros((self.n_points, 3))
        self._check(self._lib.sfmba_get_blocks(self._h, _dp(U), _dp(gc), _dp(V), _dp(gp)))
        return U, gc, V, gp

    def schur_matvec(self, x):
        x = _f64(x).reshape(self.n_cameras, self.cam_dim)
        y = np.empty_like(x)
        self._check(self._lib.sfmba_schur_matvec(self._h, _dp(x), _dp(y)))
        return y

    def get_reduced(self):
        rhs = np.empty((self.n_cameras, self.cam_dim)); Minv = np.empty((self.n_cameras, self.cam_dim, self.cam_dim))
        self._check(self._lib.sfmba_get_reduced(self._h, _dp(rhs), _dp(Minv)))
        return rhs, Minv

    def precondition(self, r):
        """z = M^-1 r of the launch-per-iteration PCG and the coarse dimension (0: block-Jacobi only)."""
        r = _f64(r).reshape(self.n_cameras, self.cam_dim)
        z = np.empty_like(r)
        n = C.c_int32(0)
        self._check(self._lib.sfmba_precondition(self._h, _dp(r), _dp(z), C.byref(n)))
        return z, n.value

    def solve_step(self):
        dc = np.zeros((self.n_cameras, self.cam_dim)); dp = np.zeros((self.n_points, 3))
        it = C.c_int32(0)
        self._check(self._lib.sfmba_solve_step(self._h, _dp(dc), _dp(dp), C.byref(it)))
        return dc, dp, it.value

    def structure(self, name):
        """One of the integer arrays the set-up derived from the index arrays (sfmba_get_structure)."""
        n = C.c_int64(0)
        self._check(self._lib.sfmba_get_structure(self._h, name.encode(), None, 0, C.byref(n)))
        out = np.empty(n.value, dtype=np.int32)
        self._check(self._lib.sfmba_get_structure(self._h, name.encode(), out.ctypes.data_as(C.POINTER(C.c_int32)), n.value, C.byref(n)))
        return out

    # -- multi-GPU
    @staticmethod
    def comm_unique_id():
        lib = load_library()
        buf = C.create_string_buffer(128)
        rc = lib.sfmba_comm_unique_id(buf)
        if rc != 0:
            raise EngineError(rc, "could not obtain an RCCL unique id")
        return buf.raw

    def comm_init_local(self, group):
        """Join an in-process group made by `local_group_create` (call from this rank's thread)."""
        self._check(self._lib.sfmba_comm_init_local(self._h, group))

    def comm_init(self, unique_id):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._check(self._lib.sfmba_comm_init(self._h, buf))


def local_group_create(nranks):
    """In-process communicator for `nranks` engines driven by `nranks` host threads."""
    lib = load_library()
    g = C.c_void_p()
    rc = lib.sfmba_local_group_create(int(nranks), C.byref(g))
    if rc != 0:
        raise EngineError(rc, "sfmba_local_group_create failed")
    return g


def local_group_destroy(group):
    load_library().sfmba_local_group_destroy(group)
