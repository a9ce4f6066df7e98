/*
 * sfmba.h - C ABI of libsfmba.so, the MI355X (gfx950) bundle-adjustment engine.
 *
 * The library replaces the arithmetic behind BundleAdjustment.optimize() of the reference
 * (/root/reference/bundle_adjustment.py:128-150): the residual function fun()
 * (bundle_adjustment.py:34-50, cv2.Rodrigues + cv2.projectPoints per observation), the
 * finite-differenced sparse Jacobian requested through jac_sparsity (bundle_adjustment.py:53-74,
 * 139) and scipy.optimize.least_squares(method='trf') (bundle_adjustment.py:141-143).
 *
 * All arrays are plain host pointers in the reference's own layout (the five arrays of
 * optimize_data.npz, bundle_adjustment.py:125-126).  Nothing is retained after a call returns.
 * Every function returns 0 on success and a negative sfmba_error otherwise; the message of the
 * most recent failure on a handle is available from sfmba_last_error().  No function throws or
 * aborts across this boundary.  A handle is not re-entrant; distinct handles are independent.
 */
#ifndef SFMBA_H
#define SFMBA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sfmba_handle sfmba_handle;

enum sfmba_error {
    SFMBA_OK = 0,
    SFMBA_ERR_INVALID_ARGUMENT = -1, /* null pointer, negative size, no problem set */
    SFMBA_ERR_INDEX_RANGE = -2,      /* camera/point index outside [0,n): NumPy IndexError at bundle_adjustment.py:49 */
    SFMBA_ERR_NONFINITE = -3,        /* non-finite initial residuals: SciPy ValueError, least_squares.py:844-845 */
    SFMBA_ERR_HIP = -4,              /* a HIP runtime call failed */
    SFMBA_ERR_COMM = -5,             /* RCCL missing or a collective failed */
    SFMBA_ERR_NO_DEVICE = -6         /* no usable gfx950 device */
};

/* Termination codes: the ones scipy.optimize.least_squares reports
 * (scipy/optimize/_lsq/least_squares.py:18-25). */
enum sfmba_status {
    SFMBA_STATUS_MAX_NFEV = 0,
    SFMBA_STATUS_GTOL = 1,
    SFMBA_STATUS_FTOL = 2,
    SFMBA_STATUS_XTOL = 3,
    SFMBA_STATUS_FTOL_XTOL = 4
};

enum sfmba_linear_solver {
    SFMBA_SOLVER_AUTO = 0, /* the dense Cholesky for up to 48 cameras (288 unknowns) and, up to 1536 unknowns,
                              for two-view tracks (at most 2.5 observations per point); PCG otherwise, and a
                              step on which PCG runs out of iterations is re-solved by the dense Cholesky
                              when the unknowns number <= dense_max_dim (sfmba_stats.n_dense counts them) */
    SFMBA_SOLVER_PCG = 1,  /* preconditioned CG on the reduced camera system (explicit or matrix-free) */
    SFMBA_SOLVER_DENSE = 2 /* explicit Schur complement + fp64 MFMA blocked Cholesky */
};

typedef struct sfmba_options {
    /* Same meaning and defaults as the reference's least_squares call
     * (bundle_adjustment.py:141: ftol=1e-4, xtol=1e-12; SciPy default gtol=1e-8,
     * max_nfev = 100*n when 0). */
    double ftol;
    double xtol;
    double gtol;
    int32_t max_nfev;
    /* Levenberg-Marquardt / linear solver controls (no counterpart in the reference). */
    int32_t max_iterations;   /* cap on LM iterations (accepted + rejected); 0 = 200 */
    double initial_lambda;    /* initial damping; 0 = chosen by the library: 1e-4, or 1e-10 where the camera system is factored exactly */
    double pcg_rtol;          /* > 0: PCG stops at sqrt(r'M^-1 r / r0'M^-1 r0) <= pcg_rtol.  0 = choose: inside
                                 sfmba_solve a linear solve stops once that ratio has passed 0.03 (single-launch
                                 solver; 0.1 with a launch per iteration) AND its last two iterations each lowered
                                 the quadratic model by at most 0.05 ftol cost, or at a thirtieth of that ratio */
    int32_t pcg_max_iter;     /* 0 = 1000 */
    int32_t pcg_batch;        /* PCG iterations queued per host check; 0 = 4 (pipelined), 1 = check every iteration */
    int32_t linear_solver;    /* enum sfmba_linear_solver */
    int32_t schur_mode;       /* 0 = auto, 1 = implicit (matrix-free matvec over the observations),
                                 2 = explicit block-sparse camera-camera matrix */
    int32_t dense_max_dim;    /* 0 = 1536 */
    int32_t verbose;          /* 2 prints one line per LM iteration (SciPy verbose=2 columns) */
    int32_t profile;          /* 1 brackets every kernel family with HIP events (fills sfmba_stats.ms_*) */
    int32_t device;           /* HIP device ordinal, -1 = current device */
    /* Point partition of this handle: it owns the rank-th of nranks contiguous point ranges
     * (equal observation counts).  nranks <= 1 means the whole problem. */
    int32_t rank;
    int32_t nranks;
    /* PCG on the explicit reduced matrix: 0 = choose (single launch when the device can give
     * every camera row a wavefront of its own), 1 = two launches per iteration, 2 = the whole
     * solve in one launch whose workgroups synchronise through memory (falls back to 1 per solve
     * on a breakdown of its recurrence or when its workgroups do not all become resident).  The
     * single launch runs pipelined conjugate gradients (the global sums of an iteration travel
     * while its matrix-vector product is formed) when every wavefront owns at most one row and no
     * row has more than 64 blocks, the single-reduction form of Chronopoulos and Gear otherwise;
     * same iterates in exact arithmetic. */
    int32_t pcg_mode;
    int32_t pcg_blocks;       /* workgroups of the single-launch solver; 0 = one per compute unit */
    int32_t camera_model;     /* enum sfmba_camera_model; default SFMBA_CAMERA_PINHOLE */
    /* Two-level preconditioner of PCG on the explicit matrix: next to the block-Jacobi blocks, a
     * coarse space of seven vectors (translation, rotation, scale of the world) per cluster of
     * consecutive cameras, M^-1 = M_J^-1 + Z (Z'SZ)^-1 Z'.
     *   0 = choose: built for 256 cameras and more, clusters of k = n_cameras / 100 cameras kept
     *       within 48..128.  After the first solve of an sfmba_solve that takes 12 iterations or
     *       more the operator is set up from that solve's matrix on a second stream while the next
     *       linearisation runs; the solves after it use it, and it is renewed when one of them
     *       takes 48 iterations or more.  Launch-per-iteration solver: two more launches per
     *       iteration apply it.  Single-launch solver (either camera model, pipelined form, at
     *       most 288 coarse unknowns, rows that reach at most four clusters): applied inside the
     *       launch, and with the pinhole camera also below 256 cameras, with
     *       clusters as small as eight cameras (the smallest size from n_cameras / 36 up with
     *       which the rows of each of the kernel's workgroups reach at most four clusters; at
     *       least four clusters), switched on after a solve of 12 iterations or more that took
     *       2.5 times the iterations of the solve before it, or of 40 or more;
     *  -1 = never;
     *   k > 0 = clusters of k cameras (at least 48; raised where 7 n_cameras / k would exceed 4096),
     *       used from the first solve on, the operator set up from the matrix of every solve in
     *       front of it; selects pcg_mode 1.
     * n_cl = n_cameras / k clusters, cluster i = cameras [i n_cameras / n_cl, (i+1) n_cameras / n_cl);
     * with fewer than two clusters no coarse space is built (one cluster would hold the gauge
     * directions of the whole problem, on which S vanishes up to the damping).  Cameras without
     * observations take no part in it and do not move. */
    int32_t pcg_coarse;
    /* Where sfmba_set_problem builds the structure of the problem (orderings, chunks, observation
     * pair lists, block pattern): 0 = on the GPU (sorts and scans; milliseconds), 1 = on host
     * threads (the first implementation, kept as the layout the GPU path is tested against; both
     * produce the same arrays, so solves are bit-identical). */
    int32_t setup_mode;
    /* How a partitioned solve (nranks > 1) holds the explicit reduced matrix:
     *   1 = replicated: the upper blocks of all ranks are summed by one all-reduce per LM iteration
     *       (36 doubles per block) and every rank runs the same PCG, with no collective inside it;
     *   2 = split by points: a rank keeps only the blocks its own points produce (S = sum over the
     *       ranks of S_k; the diagonal blocks, 63 doubles per camera with the right-hand side, are
     *       still summed, and every rank holds 1/nranks of them), the product q = S p of every PCG
     *       iteration is summed over the ranks (6 doubles per camera), and the coarse matrix of
     *       the two-level preconditioner is summed once per set-up.  Launch-per-iteration PCG only;
     *       no dense fall-back.  Same iterates as the replicated form up to rounding;
     *   0 = choose: replicated (the split form has only run in the in-process group on one GPU; it
     *       pays where the 36 doubles per block of the replicated form cost more than a collective
     *       per PCG iteration, and its product shrinks only as far as a rank's points touch fewer
     *       cameras than all points do).
     * Ignored with nranks <= 1, with the matrix-free product and with the dense solver. */
    int32_t matrix_split;
} sfmba_options;

/* Camera models.  D = tangent numbers per camera (width of a Jacobian block and of every
 * camera-sized vector of the diagnostics below), P = numbers per camera in camera_params.
 *   SFMBA_CAMERA_PINHOLE  P = 12, D = 6: row-major 3x3 rotation block + translation, one fixed K for
 *       all cameras, no distortion (bundle_adjustment.py:10-31); tangent (dw, dt).
 *   SFMBA_CAMERA_BAL      P = 9, D = 9: rotation vector, translation, f, k1, k2 of the "Bundle
 *       Adjustment in the Large" files; pixel = f (1 + k1 n + k2 n^2) p with p = -(R X + t)[:2] /
 *       (R X + t)[2], n = |p|^2 (ba_test.py:62-100); tangent (dw, dt, df, dk1, dk2); K is not
 *       read and may be NULL.  Every linear solver is available; of the single-launch PCG only the
 *       pipelined form exists for this model, so where that does not apply (see pcg_mode) the
 *       explicit matrix is solved with a launch per iteration. */
enum sfmba_camera_model { SFMBA_CAMERA_PINHOLE = 0, SFMBA_CAMERA_BAL = 1 };

typedef struct sfmba_stats {
    int32_t status;           /* enum sfmba_status */
    int32_t iterations;       /* LM iterations, accepted + rejected */
    int32_t accepted;         /* accepted steps */
    int32_t nfev;             /* cost evaluations (initial + one per trial step) */
    int32_t njev;             /* linearisations */
    int32_t pcg_iterations;   /* total PCG iterations (0 for the dense solver) */
    int32_t linear_solver;    /* solver actually used */
    int32_t schur_mode;       /* 1 implicit, 2 explicit */
    double initial_cost;      /* 0.5*||f||^2, SciPy's convention */
    double final_cost;
    double optimality;        /* inf-norm of the gradient at the final iterate */
    double solve_seconds;     /* wall time of sfmba_solve */
    /* GPU time per kernel family inside sfmba_solve, HIP-event measured, milliseconds, and
     * the number of launches of each. */
    double ms_linearize;  int64_t n_linearize;
    double ms_damp;       int64_t n_damp;
    double ms_spmv_pm;    int64_t n_spmv_pm;   /* point-major half of the Schur matvec */
    double ms_spmv_cm;    int64_t n_spmv_cm;   /* camera-major half of the Schur matvec */
    double ms_pcg_vec;    int64_t n_pcg_vec;   /* camera-vector kernels of PCG */
    double ms_back;       int64_t n_back;
    double ms_cost;       int64_t n_cost;
    double ms_dense;      int64_t n_dense;     /* dense Cholesky factorisation + solves */
    double ms_schur;      int64_t n_schur;     /* explicit Schur assembly (pair kernel + reductions + finish) */
    double ms_pairs;      int64_t n_pairs;     /* the pair kernel k_schur_pairs alone */
    double ms_bsr;        int64_t n_bsr;       /* block-CSR matvec of the explicit reduced system */
    int64_t schur_pairs;                       /* observation pairs per assembly (explicit mode) */
    int64_t schur_blocks;                      /* 6x6 blocks of the reduced matrix (explicit mode) */
    int64_t schur_chunks;                      /* wavefront-sized pair chunks per assembly */
    int64_t pcg_single_launch;                 /* linear solves done by the single-launch PCG */
    int64_t pcg_single_fallbacks;              /* single-launch solves redone launch by launch */
    int64_t pcg_coarse_solves;                 /* linear solves preconditioned with the coarse space */
    int64_t pcg_coarse_dim;                    /* coarse unknowns (7 x clusters), 0 = no coarse space built */
    double ms_comm;       int64_t n_comm;      /* all-reduces between the ranks (stream time: the wait for the slowest rank included) */
    int64_t pcg_pipelined;                     /* 1: the single-launch solves ran the pipelined-CG kernel (every camera row on a
                                                  wavefront of its own, rows of at most 64 blocks), 0: the plain one */
} sfmba_stats;

/* Fill *opts with the defaults listed above. */
void sfmba_default_options(sfmba_options* opts);

/* Create / destroy an engine instance bound to one GPU. */
int sfmba_create(const sfmba_options* opts, sfmba_handle** out);
int sfmba_destroy(sfmba_handle* h);
const char* sfmba_last_error(const sfmba_handle* h);

/* Where the progress text of sfmba_solve goes (options.verbose >= 1: the termination message;
 * >= 2: one row per iteration with the columns of SciPy's verbose=2 table,
 * scipy/optimize/_lsq/common.py print_header_nonlinear / print_iteration_nonlinear, which the
 * reference asks for at bundle_adjustment.py:141).  sink(text, user) receives whole lines; NULL
 * restores the default, the process's stdout, flushed after every line.  The sink is called on the
 * thread that called sfmba_solve. */
int sfmba_set_print(sfmba_handle* h, void (*sink)(const char* text, void* user), void* user);

/* Load one problem: the arrays BundleAdjustment.view2idx() produces
 * (bundle_adjustment.py:94-126) plus the 3x3 intrinsic matrix (row-major; only K[0][0],
 * K[1][1], K[0][2], K[1][2] are used, as cv2.projectPoints does with empty distortion).
 * camera_params is n_cameras x 12 (row-major 3x3 rotation block, then translation); the 3x3
 * block is replaced by its polar orthogonal factor exactly as cv2.Rodrigues does at
 * bundle_adjustment.py:24.  Observations may come in any order.
 * With SFMBA_CAMERA_BAL camera_params is n_cameras x 9 (the arrays read_bal_data returns,
 * ba_test.py:17-42) and K is ignored. */
int sfmba_set_problem(sfmba_handle* h, int64_t n_cameras, int64_t n_points, int64_t n_obs,
                      const double* camera_params, const double* points_3d,
                      const int64_t* camera_indices, const int64_t* point_indices,
                      const double* points_2d, const double* K);

/* sfmba_set_problem for callers whose observations already live in device memory (a pipeline that
 * keeps correspondences on the GPU): camera_indices, point_indices (int64) and points_2d
 * (n_obs x 2 doubles) are DEVICE pointers on the handle's GPU in the same layout; camera_params,
 * points_3d and K stay host pointers (they are small).  The arrays are only read, and only until
 * the call returns.  A pointer the GPU cannot dereference (ordinary host memory, memory of another
 * GPU) is refused with SFMBA_ERR_INVALID_ARGUMENT; pinned host memory and managed memory are
 * taken.  No counterpart in the reference, whose arrays are NumPy arrays
 * (bundle_adjustment.py:117-122). */
int sfmba_set_problem_device(sfmba_handle* h, int64_t n_cameras, int64_t n_points, int64_t n_obs,
                             const double* camera_params, const double* points_3d,
                             const int64_t* camera_indices_dev, const int64_t* point_indices_dev,
                             const double* points_2d_dev, const double* K);

/* Overwrite the current parameter estimate without re-sending the observations (used to
 * re-run a solve from the same start, and to evaluate fun() at another x). */
int sfmba_set_params(sfmba_handle* h, const double* camera_params, const double* points_3d);

/* Restore the estimate last given to sfmba_set_problem / sfmba_set_params from its copy in
 * device memory (no host transfer): lets a caller repeat a solve from the same start. */
int sfmba_reset_params(sfmba_handle* h);

/* fun(params) of bundle_adjustment.py:34-50 at the current estimate: out has 2*n_obs entries,
 * [du0, dv0, du1, dv1, ...] in the caller's observation order.  On a partitioned handle only
 * the entries of locally owned observations are written (the rest are left untouched). */
int sfmba_residuals(sfmba_handle* h, double* out);

/* Analytic Jacobian blocks at the current estimate, caller's observation order:
 * jac_cam is n_obs x 2 x 6 with columns (dw0, dw1, dw2, dt0, dt1, dt2) of the left
 * perturbation Y' = exp([dw]x) Y + dt of the camera-frame point Y = R X + t;
 * jac_pt is n_obs x 2 x 3 (d residual / d X).  They take the place of the 2-point finite
 * differences SciPy forms through jac_sparsity (bundle_adjustment.py:139-141).
 * With SFMBA_CAMERA_BAL jac_cam is n_obs x 2 x 9: the same six columns, then (df, dk1, dk2). */
int sfmba_jacobian_blocks(sfmba_handle* h, double* jac_cam, double* jac_pt);

/* Run the Levenberg-Marquardt solve from the current estimate: the replacement for the
 * least_squares call at bundle_adjustment.py:141-143. */
int sfmba_solve(sfmba_handle* h, sfmba_stats* stats);

/* Current estimate in the reference's output layout (bundle_adjustment.py:147-150):
 * poses n_cameras x 12 (orthonormal rotation block + translation), points n_points x 3.
 * With SFMBA_CAMERA_BAL: n_cameras x 9, the rotation vector with its angle in [0, pi]. */
int sfmba_get_params(sfmba_handle* h, double* camera_params, double* points_3d);

/* Mean reprojection error |projection - observation| (pixels) of every camera at the current
 * estimate and, optionally, its observation count: what ImageView.reproject_view() prints per
 * view after each bundle adjustment (/root/reference/view.py:103-119 with
 * utils.calculate_reprojection_error, utils.py:337-352; main.py:123).  A camera without
 * observations gets NaN, as np.mean([]) does. */
int sfmba_reprojection_errors(sfmba_handle* h, double* mean_error, int64_t* n_observations);

/* ---- the step that feeds bundle adjustment its initial points ---- */

/* Two-view linear triangulation of n correspondences, batched: the replacement for
 * cv.triangulatePoints + cv.convertPointsFromHomogeneous + the per-point error loop of
 * triangulate_points (/root/reference/utils.py:176-211).  P1, P2 are the 3x4 projection
 * matrices K [R | t] (row-major), x1, x2 are n x 2 pixel coordinates.  X receives n x 3
 * points (right singular vector of the smallest singular value of the 4x4 DLT matrix,
 * de-homogenised; a zero last coordinate divides by 1 as OpenCV does), err1 / err2 the
 * reprojection error |x - dehom(P X)| of every point in either view
 * (calculate_reprojection_error, utils.py:337-352).  err1 / err2 may be NULL.  Needs no problem. */
int sfmba_triangulate(sfmba_handle* h, int64_t n, const double* P1, const double* P2,
                      const double* x1, const double* x2, double* X, double* err1, double* err2);

/* sfmba_triangulate for callers that cannot allocate device memory (tight per-image-pair loops,
 * harnesses that forbid hipMalloc/hipFree, which also synchronise the device).  It has no
 * counterpart in the reference.  Same arguments, same kernel and bit-identical results; the
 * device staging sfmba_triangulate allocates and frees per call lives in a fixed array inside the
 * library's code object instead (72 bytes per correspondence, 4.5 MiB on every GPU the library
 * runs on, resolved per device in sfmba_create).  At most SFMBA_TRIANGULATE_NOALLOC_MAX
 * correspondences per call: more returns SFMBA_ERR_INVALID_ARGUMENT and a message naming the
 * limit.  The call makes no hipMalloc / hipFree.  All handles on one GPU share that GPU's array:
 * calls that overlap in time on the same GPU (from several threads or handles) race and must be
 * serialised by the caller. */
#define SFMBA_TRIANGULATE_NOALLOC_MAX 65536
int sfmba_triangulate_noalloc(sfmba_handle* h, int64_t n, const double* P1, const double* P2,
                              const double* x1, const double* x2, double* X, double* err1, double* err2);

/* Reprojection errors of n given points in one view with projection matrix P (3x4, row-major):
 * calculate_reprojection_error (utils.py:337-352) for a whole array, as the filters of
 * store_3Dpoints_to_views (utils.py:279-315) and reproject_view (view.py:103-119) use it. */
int sfmba_point_errors(sfmba_handle* h, int64_t n, const double* P, const double* X,
                       const double* x, double* err);

/* ---- diagnostics used by the parity tests: one linearisation at the current estimate ---- */

/* Linearise and damp with lambda.  Afterwards the blocks below can be read. */
int sfmba_linearize(sfmba_handle* h, double lambda);
/* U: n_cameras x 36 (J_c'J_c per camera), g_cam: n_cameras x 6, V: n_points x 9 (J_p'J_p per
 * point, owned points only on a partitioned handle), g_pt: n_points x 3.  Any may be NULL.
 * Here and below 6 and 36 read D and D*D for a camera model with D tangent numbers. */
int sfmba_get_blocks(sfmba_handle* h, double* U, double* g_cam, double* V, double* g_pt);
/* y = S x with S = (U + lambda D_c) - sum_p W_p (V_p + lambda D_p)^-1 W_p', x and y n_cameras x 6. */
int sfmba_schur_matvec(sfmba_handle* h, const double* x, double* y);
/* Reduced right-hand side b = -g_cam + sum W V*^-1 g_pt and inverse of the block-Jacobi
 * preconditioner blocks (n_cameras x 36).  Either may be NULL. */
int sfmba_get_reduced(sfmba_handle* h, double* rhs, double* precond_inv);
/* z = M^-1 r with the preconditioner the launch-per-iteration PCG would use at the current
 * linearisation, r and z n_cameras x 6: the block-Jacobi blocks, plus the coarse correction
 * Z (Z'SZ)^-1 Z' r when the handle has a coarse space (options.pcg_coarse); *coarse_dim then
 * receives 7 x clusters, otherwise 0.  (No counterpart in the reference: LSMR there runs
 * unpreconditioned on the column-scaled Jacobian.) */
int sfmba_precondition(sfmba_handle* h, const double* r, double* z, int32_t* coarse_dim);
/* Solve the damped normal equations for the step at the current linearisation:
 * step_cam n_cameras x 6, step_pt n_points x 3; *pcg_iters receives the iteration count. */
int sfmba_solve_step(sfmba_handle* h, double* step_cam, double* step_pt, int32_t* pcg_iters);

/* One of the integer arrays sfmba_set_problem derived from the index arrays, by name: the
 * point-major order (pt_ptr, pm_cam, pm_pt, pm_orig, new2old), the camera-major order (cm_pt,
 * chunk_cam, chunk_beg, cam_chunk_ptr), wtile_beg, and for the explicit reduced matrix ublk_r,
 * ublk_c, ublk_chunk_ptr, ublk_pos, ublk_pos_t, ublk_diag, pchunk_blk, pchunk_beg, pair_p,
 * bsr_ptr, bsr_col, bsr_diag.  *count receives the length; out may be NULL to ask for it.
 * Lets the tests compare the two set-up paths (options.setup_mode) array by array. */
int sfmba_get_structure(sfmba_handle* h, const char* name, int32_t* out, int64_t capacity, int64_t* count);

/* ---- multi-GPU: one handle (one process) per GPU, RCCL over xGMI ---- */

/* Rank 0 obtains a 128-byte RCCL unique id and distributes it (torch.distributed broadcast);
 * every rank then joins.  After sfmba_comm_init the camera-block partial sums, the reduced
 * matvec partials and the cost partials of each iteration are all-reduced across ranks. */
int sfmba_comm_unique_id(void* out128);
int sfmba_comm_init(sfmba_handle* h, const void* unique_id128);

/* Same decomposition inside ONE process: one handle per host thread (each on its own GPU, or
 * several on one GPU), joined to an in-process group that all-reduces through device copies in
 * rank order.  Every member must call sfmba_comm_init_local, and later sfmba_solve /
 * sfmba_linearize / sfmba_get_params, concurrently from its own thread. */
int sfmba_local_group_create(int32_t nranks, void** group);
int sfmba_local_group_destroy(void* group);
int sfmba_comm_init_local(sfmba_handle* h, void* group);

#ifdef __cplusplus
}
#endif
#endif /* SFMBA_H */
