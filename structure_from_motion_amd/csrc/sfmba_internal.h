// Host/device shared declarations of libsfmba: the HBM layout of one problem partition and the
// launch wrappers of every kernel family.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sfmba_device.h"

namespace sfmba {

constexpr int kBlock = 256;        // threads per workgroup (4 wavefronts) unless a kernel says otherwise
// Kernels in which every wavefront owns one camera chunk of up to 512 observations: the chunks
// differ in length and a workgroup's registers and LDS return to the compute unit only when its
// last wavefront has finished, so two wavefronts per workgroup keep more of them resident than
// four (fused camera-major pass at the Venice shape: 138 -> 117 us per launch).
#ifndef SFMBA_CM_BLOCK
#define SFMBA_CM_BLOCK 128
#endif
constexpr int kCmBlock = SFMBA_CM_BLOCK;
constexpr int kChunk = 512;        // observations per camera-major chunk (one wavefront each)
constexpr int kMaxScalBlocks = 1 << 18;

// Scalar reduction slots (block partials -> k_scal_finalize -> scalars[slot]).
enum Slot {
    SLOT_COST = 0,      // sum r^2 at the linearisation point
    SLOT_GMAX_PT,       // max |g_p|
    SLOT_MODEL,         // sum (r + J step)^2
    SLOT_STEP2_PT,      // sum |dp|^2
    SLOT_X2_PT,         // sum |X|^2
    SLOT_TRIAL,         // sum r^2 at the trial point
    SLOT_NONFINITE,     // count of non-finite trial residuals
    SLOT_PCGFLAG,       // non-zero: the single-launch PCG of this step must be redone (any rank)
    SLOT_STEP2_CAM,     // sum |dc|^2
    SLOT_X2_CAM,        // sum |t|^2 (+3 per camera for the rotation block)
    SLOT_GMAX_CAM,      // max |g_c|
    SLOT_COUNT
};

// PCG state kept in device memory so that the host can queue iterations without reading back.
struct PcgState {
    double rz;        // r'z of the current iterate
    double rz0;       // r'z at iteration 0
    double tol2;      // stop when rz <= tol2 * rz0 (on return: the tolerance the solve was stopped under)
    double tol2_loose;   // ... or when rz <= tol2_loose * rz0 and the last two iterations lowered the
    double q_abs;        // quadratic model by at most q_abs each (q_abs < 0: this rule is off)
    double dq1, dq2;     // model decrease of the last and the last but one iteration (alpha r'z / 2)
    int iter;         // matvecs completed
    int max_iter;
    int done;
    int pad;
    long long ticks;  // single-launch solver: duration in ticks of the 100 MHz wall clock
};

// Arguments of the single-launch PCG (k_pcg_fused, k_pcg_pipe): the symmetrically scaled system
// Shat = F S F', bhat = F rhs with F = L^-1 of the diagonal blocks (S_cc = L L'), so that the
// block-Jacobi preconditioner becomes the identity and x = F' xhat.  nd = tangent numbers per
// camera: 6, or 9 (k_pcg_pipe only).
struct FusedPcg {
    int n_cam;
    int nd;
    const int* bsr_ptr; const int* bsr_col; const int* bsr_diag;
    const double* Shat;       // [nnzb][nd nd]
    const double* F;          // [n_cam][nd nd] lower triangular, row-major
    const double* bhat;       // [n_cam][nd]
    double* vec;              // [4][n_cam][18] (r, w, s) records in rotating generation slots
                              // (k_pcg_pipe: [4][n_cam][nd], w only)
    double* p; double* xh;    // [n_cam][6], touched by the owning wavefront only (k_pcg_fused)
    double* x;                // [n_cam][nd] solution of the unscaled system
    double2* dots;            // [4][gridDim.x] per-workgroup (r'r, w'r) records in rotating slots,
                              // then one abort word; all bits set = not written / no abort
    PcgState* out;            // final state; pad = 1 on breakdown, 2 on exchange timeout
    double* flag_slot;        // receives pad as a double (a scalar slot the ranks sum), may be null
    double* flag_host;        // the same slot in pinned host memory, may be null
    double tol2;              // stop when r'r <= tol2 r0'r0 ...
    double tol2_loose;        // ... or when r'r <= tol2_loose r0'r0 and each of the last two iterations lowered
    double q_abs;             // the quadratic model by at most q_abs (< 0: this rule is off)
    int max_iter;
    long long timeout_ticks;  // of the 100 MHz wall clock: any later poll
    long long admit_ticks;    // the first poll (all workgroups resident and started)
    int blocks;               // workgroups of the launch
    int pipelined;            // 1: k_pcg_pipe (reduction overlapped with the product), 0: k_pcg_fused
    // two-level preconditioner inside the launch (k_pcg_pipe2): the coarse operator of sfmba_coarse.hip
    int coarse;               // 1: use it
    int c_n, c_npad;          // coarse unknowns, padded
    const int* cam_cl;        // [n_cam] cluster of each camera
    const int* cl_ptr;        // [clusters + 1] camera range of each cluster
    const int* wg_cl;         // [blocks][5] per workgroup: count, then the clusters among the columns of its rows
    const double* c_z;        // [n_cam][7][nd] coarse basis
    const double* c_zhat;     // [n_cam][7][nd] work array: F^-T Z of every camera (9 x 9 blocks), or null
    const double* c_einv;     // [c_npad][c_npad] (D E D)^-1
    const double* c_scale;    // [c_npad] D
};
constexpr int kPcgSlots = 4;      // rotating generation slots of the single-launch PCG
constexpr int kPcgVecRec = 18;    // r[6] w[6] s[6] of one camera and generation
constexpr int kPcgRecWords = 16;  // words reserved per workgroup and generation for its record of global
                                  // sums (k_pcg_fused and k_pcg_pipe use the first two)

// Everything a kernel needs, passed by value.  "pm" arrays are in point-major observation
// order (observations of one point contiguous), "cm" arrays in camera-major order.
struct Dev {
    int n_cam;            // all cameras (replicated on every rank)
    int n_pts;            // points owned by this partition
    int n_obs;            // observations of the owned points
    int n_chunks;         // camera-major chunks
    int model;            // CameraModel: record layout of `cams`, tangent size of a camera
    Intr K;               // pinhole model only
    // static structure
    int n_wtiles;             // wave tiles of the point-major order
    const int* wtile_beg;     // [n_wtiles+1] observation range of each tile: whole points, at most
                              // 64 observations unless the tile is a single point with more
    const int* pt_ptr;        // [n_pts+1] CSR offsets into pm arrays
    const int* pm_cam;        // [n_obs]
    const int* pm_pt;         // [n_obs] local point id
    const int* pm_orig;       // [n_obs] caller's observation index
    const double2* pm_uv;     // [n_obs]
    const int* cm_pt;         // [n_obs] local point id
    const double2* cm_uv;     // [n_obs]
    const int* chunk_cam;     // [n_chunks]
    const int* chunk_beg;     // [n_chunks+1] offsets into cm arrays (chunk q = [beg[q], beg[q+1]))
    const int* cam_chunk_ptr; // [n_cam+1] chunk range of each camera
    // parameters (current and trial)
    const double2* cams;      // [n_cam][12] (pinhole) or [n_cam][16] (BAL)
    const double4* X;         // [n_pts] (x, y, z, pad)
    double2* cams_trial;
    double4* X_trial;
    // linearisation products
    double4* pm_seed;   // [n_obs] (x, y, iz, ru): Jacobian seeds and the u residual, point-major
    double* pm_rv;      // [n_obs] v residual, point-major
    double* cm_sx; double* cm_sy; double* cm_siz;   // seeds, camera-major
    double* V;          // [n_pts][6] packed upper J_p'J_p
    double* gp;         // [n_pts][4]
    double* Vinv;       // [n_pts][6] (V + lambda D)^-1
    double4* s;         // [n_pts] work vector of the matvec / V*^-1 g_p
    double4* s0;        // [n_pts] V*^-1 g_p
    // camera-sized arrays: ND = 6 (pinhole) or 9 (BAL) tangent numbers, NU = ND (ND + 1) / 2
    double* chunk_part; // [n_chunks][NU + ND]
    double* UG;         // [n_cam][NU + ND]: packed upper J_c'J_c | g_c (all ranks' sum)
    double* T;          // [n_cam][NU + ND]: sum W V*^-1 W' | sum W V*^-1 g_p
    double* Minv;       // [n_cam][ND * ND] inverse of the block-Jacobi blocks
    double* rhs;        // [n_cam][ND]
    // PCG vectors, [n_cam][ND]
    double* px; double* pr; double* pz; double* pq; double* qS;
    double* pp[2];      // search direction, double-buffered by iteration parity
    PcgState* pcg;      // [2], double-buffered by iteration parity
    unsigned char* cam_seen; // [n_cam] 1 where the camera has observations on any rank (U_cc not zero), written by the finishing pass of every damping
    PcgState* pcg_mirror;   // pinned host memory: the launch-per-iteration solver writes the state of every iteration there too
    double* dot_part;   // block partials of p'q at [0, dot_off), of r'z at [dot_off, ...)
    int dot_off;        // offset of the r'z partials inside dot_part (p'q partials start at 0)
    // explicit block-sparse Schur complement (sfmba_schur.hip)
    int n_ublk;               // upper blocks (row camera <= column camera), diagonal always present
    int n_pchunks;            // pair chunks
    int nnzb;                 // blocks of the full block-CSR matrix
    const int* ublk_r; const int* ublk_c;         // [n_ublk]
    const int* ublk_chunk_ptr;                    // [n_ublk+1] pair-chunk range of each upper block
    const int* ublk_diag;                         // [n_cam] upper block (c, c) of every camera
    const int* ublk_pos; const int* ublk_pos_t;   // [n_ublk] slots of (r,c) and (c,r) in Sfull
    const int* pchunk_blk; const int* pchunk_beg; // [n_pchunks], [n_pchunks+1]
    const int4* pchunk_rec;   // [n_pchunks] (first pair, end, row camera, column camera): what a wavefront of the pair kernel needs, one load
    const int* pair_p;        // [n_pairs] the point of every pair, by block and, inside a block, by point
    double* pair_part;        // [n_pchunks][ND * ND]
    double* Bsum;             // [n_ublk][ND * ND] sum W_a Vinv W_b' per upper block (all ranks' sum)
    const int* bsr_ptr; const int* bsr_col; const int* bsr_diag;   // [n_cam+1], [nnzb], [n_cam]
    double* Sfull;            // [nnzb][ND * ND] row-major blocks
    // scalars
    double* scal_part;  // [SLOT_COUNT][kMaxScalBlocks]
    double* scalars;    // [SLOT_COUNT]
    double* scalars_host;   // the same slots in pinned host memory (written beside `scalars`, so that the
                            // host needs no copy after its wait), or null
};

// Coarse space of the two-level PCG preconditioner (sfmba_coarse.hip): seven vectors per cluster
// of consecutive cameras, E = Z'SZ dense.
constexpr int kCoarseVecs = 7;
struct Coarse {
    int n_cl;                 // clusters
    int n;                    // 7 n_cl coarse unknowns
    int n_pad;                // n rounded up to a multiple of 32 (identity on the padding)
    int n_cpairs;             // cluster pairs (I, J) that share at least one block of S
    const int* cl_ptr;        // [n_cl+1] camera range of each cluster
    const int* cam_cl;        // [n_cam] cluster of each camera
    const int* cpair_i; const int* cpair_j;   // [n_cpairs]
    int n_cchunks;            // chunks the blocks of the pairs are cut into
    const int* cpair_cptr;    // [n_cpairs+1] chunk range of each pair
    const int* cchunk_beg;    // [n_cchunks+1] range of each chunk in cpair_blk / cpair_row
    double* cpart;            // [n_cchunks][49] chunk partials of E
    const int* cpair_blk;     // [nnzb] blocks of Sfull grouped by cluster pair
    const int* cpair_row;     // [nnzb] row camera of each of those blocks
    double* Z;                // [n_cam][7][ND]
    double* E[2];             // [n_pad][n_pad] each: assembly, scaling and the sweep alternate between them
    double* scale;            // [n_pad] D = diag(E)^-1/2
    double* rc;               // [n_pad] restricted residual (zero on the padding)
    double* rpart;            // [workgroups of k_pcg_update][2][7] the restriction as that kernel leaves it: per workgroup and
                              // cluster slot (a workgroup's cameras lie in at most two clusters) the sums Z'r over its cameras
    const double* Einv;       // whichever of E[] holds (D E D)^-1
};

inline int cam_dim(const Dev& d) { return d.model == MODEL_BAL ? Bal::ND : Pinhole::ND; }
inline int cam_dim(int model) { return model == MODEL_BAL ? Bal::ND : Pinhole::ND; }
inline int cam_sums(int model) { return model == MODEL_BAL ? Bal::NV : Pinhole::NV; }      // packed upper + gradient
inline int cam_words(int model) { return model == MODEL_BAL ? Bal::CW : Pinhole::CW; }     // double2 per camera record

// ---- launch wrappers (sfmba_kernels.hip) -------------------------------------------------
void launch_residuals(const Dev& d, double* out_dev, hipStream_t st);
void launch_jacobian_blocks(const Dev& d, double* jc_dev, double* jp_dev, hipStream_t st);
// grad_only: cost and gradient maxima alone (the point a solve ends on); V, U, seeds are not produced
// keep_v: also store V_p and g_p when the points are damped in the same pass (lambda > 0)
// defer_scalars: the cost and gradient-maximum slots are left as block partials for a later
// launch_back_pm(..., with_lin_scalars) or launch_lin_scalars (they are only read by the host)
void launch_lin_pm(const Dev& d, bool store_seeds, double lambda, hipStream_t st, bool grad_only = false, bool keep_v = true, bool defer_scalars = false);
void launch_lin_scalars(const Dev& d, hipStream_t st);   // lambda > 0: damp the points in the same pass;   // seeds are only stored for the matrix-free Schur product
void launch_lin_cm(const Dev& d, hipStream_t st, bool grad_only = false);               // chunk partials (NU + ND values)
void launch_chunk_reduce(const Dev& d, int nvals, double* out, hipStream_t st);
// launch_lin_cm + launch_sdiag_cm + both reductions in one pass (damping known, explicit matrix):
// UG and tpart receive what k_finish_diag needs of them; chunk rows of lin_sdiag_values() sums
int lin_sdiag_values(int model);
void launch_lin_sdiag_cm(const Dev& d, double* tpart, hipStream_t st);
void launch_damp_pm(const Dev& d, double lambda, hipStream_t st);
void launch_sdiag_cm(const Dev& d, hipStream_t st);             // chunk partials (NU + ND values)
void launch_precond(const Dev& d, double lambda, hipStream_t st);
void launch_spmv_pm(const Dev& d, const double* x, const PcgState* state, hipStream_t st);
void launch_spmv_cm(const Dev& d, const double4* svec, const PcgState* state, hipStream_t st);   // chunk partials (6 values)
// sfmba_pcg.hip: one PCG iteration is  [head + direction (+ block-CSR matvec)]  [implicit
// matvec kernels]  [update]; `cur` is the iteration parity (state and direction are double-buffered)
int pcg_update_blocks(const Dev& d);
int pcg_spmv_blocks(const Dev& d);
void launch_pcg_init(const Dev& d, double tol, int max_iter, hipStream_t st, double tol_loose = 0.0, double q_abs = -1.0);
void launch_pcg_spmv_bsr(const Dev& d, int cur, hipStream_t st);            // explicit mode
void launch_pcg_dir(const Dev& d, int cur, hipStream_t st);                 // implicit mode: direction only
void launch_pcg_q(const Dev& d, double lambda, int cur, hipStream_t st);    // implicit mode: q = U* p - qS
void launch_pcg_dot(const Dev& d, int cur, hipStream_t st);
// coarse != nullptr: also leaves the restriction Z'r of the new residual in coarse->rpart (two-level preconditioner)
void launch_pcg_update(const Dev& d, int cur, int n_pq_parts, hipStream_t st, const Coarse* coarse = nullptr);
// whole PCG solve in one launch (explicit mode); returns the launch error
int pcg_fused_max_blocks(int device);
int pcg_fused_rows_per_block();
size_t pcg_fused_record_words(int blocks);
size_t pcg_fused_vector_words(int n_cam);
// `armed`: the records already hold the sentinel (k_finish_diag did it)
hipError_t launch_pcg_fused(const FusedPcg& f, bool armed, hipStream_t st);
void launch_schur_apply(const Dev& d, const double* x, double lambda, double* y, hipStream_t st);
void launch_back_pm(const Dev& d, const double* dc, double* dp_out, hipStream_t st, bool with_lin_scalars = false);
void launch_cam_update(const Dev& d, const double* dc, hipStream_t st);
void launch_gmax_cam(const Dev& d, hipStream_t st);
void launch_reproj_cm(const Dev& d, double* out2, hipStream_t st);   // [n_cam][2] = (sum of errors, count)
int scal_blocks_pt(const Dev& d);
void launch_vec_add(double* a, const double* b, size_t n, hipStream_t st);
// caller's point numbering <-> internal order: X[p] = in[new2old[p]] and back (n x 3 doubles)
void launch_points_in(const double* in, const int* new2old, int n, double4* X, hipStream_t st);
void launch_points_out(const double4* X, const int* new2old, int n, double* out, hipStream_t st);
int blocks_cam(const Dev& d);
// sfmba_schur.hip
int bsr_blocks(const Dev& d);
void launch_schur_pairs(const Dev& d, hipStream_t st);
void launch_pchunk_records(const Dev& d, int4* rec, hipStream_t st);
void launch_block_reduce(const Dev& d, hipStream_t st);
// full matrix, preconditioner, right-hand side; with `scaled` also F, bhat, Shat of the single-launch PCG
void launch_schur_finish(const Dev& d, double lambda, const double* tpart, const FusedPcg* scaled, hipStream_t st, double share = 1.0);
size_t diag_collect_words(const Dev& d, int nv);
void launch_diag_collect(const Dev& d, int nv, const double* tpart, double* out, hipStream_t st);
void launch_diag_spread(const Dev& d, int nv, const double* in, double* tpart, hipStream_t st);
void launch_bsr_spmv(const Dev& d, const double* x, double* y, hipStream_t st);
void launch_bsr_to_dense(const Dev& d, double* A, int n, hipStream_t st);
// sfmba_coarse.hip
const double* launch_coarse_setup(const Dev& d, const Coarse& k, hipStream_t st, hipEvent_t assembled = nullptr,
                                  void (*sum_over_ranks)(void*, double*, size_t) = nullptr, void* ctx = nullptr);    // returns the buffer holding the inverse
// from_parts: the restriction is taken from k.rpart (left by launch_pcg_update) instead of being formed here
void launch_coarse_apply(const Dev& d, const Coarse& k, int state, hipStream_t st, bool from_parts = false);  // pz += Z E^-1 Z' pr, r'z partials
// sfmba_triangulate.hip (projection matrices are host arrays of 12 doubles, row-major 3x4)
void launch_triangulate(const double* P1, const double* P2, int n, const double2* x1, const double2* x2,
                        double* X, double* err1, double* err2, hipStream_t st);
void launch_reprojection_errors(const double* P, int n, const double* X, const double2* x, double* err, hipStream_t st);
// sfmba_dense.hip
int dense_padded(int n_real);
void launch_dense_cholesky_solve(double* A, int n, int n_real, double* Linv_all, double* work, int* fail,
                                 const double* b, double* x, hipStream_t st);
// sfmba_chol.hip: the same solve in one launch (block rows exchange tiles by data flow)
size_t chol_flow_words(int n);
hipError_t launch_chol_flow(const Dev& d, double* A, int n, int n_real, const double* b, double* x, double* buf, int* fail,
                            double* fail_slot, double* fail_host, hipStream_t st, long long* trace = nullptr);

}  // namespace sfmba
