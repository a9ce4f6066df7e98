// The engine instance behind the C ABI (struct sfmba_handle) and the small host-side helpers its
// translation units share: error types, device buffers, the in-process group, timing families.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/sfmba.h"
#include "sfmba_internal.h"

namespace sfmba {

// ---------------------------------------------------------------------------------- errors
struct HipError { hipError_t e; const char* what; };
#define HIP_TRY(expr)                                              \
    do {                                                           \
        hipError_t e_ = (expr);                                    \
        if (e_ != hipSuccess) throw HipError{e_, #expr};           \
    } while (0)

struct ApiError { int code; std::string msg; };

[[noreturn]] inline void fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw ApiError{code, buf};
}

// ---------------------------------------------------------------------------------- buffers
// Device array that only ever grows: loading another problem into a handle reuses the memory of
// the last one (hipMalloc and hipFree cost far more than the set-up kernels, and hipFree waits for
// the device).  `n` is the size in use, `cap` what is allocated.
template <class T>
struct DBuf {
    T* p = nullptr;
    size_t n = 0, cap = 0;
    void alloc(size_t count) {
        if (count > cap) {
            release();
            HIP_TRY(hipMalloc((void**)&p, count * sizeof(T)));
            cap = count;
        }
        n = count;
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; } n = 0; cap = 0; }
    void upload(const std::vector<T>& h, hipStream_t st) {
        if (h.size() != n) alloc(h.size());
        if (n) HIP_TRY(hipMemcpyAsync(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice, st));
    }
    void zero(hipStream_t st) { if (n) HIP_TRY(hipMemsetAsync(p, 0, n * sizeof(T), st)); }
    ~DBuf() { release(); }
    DBuf() = default;
    DBuf(const DBuf&) = delete;
    DBuf& operator=(const DBuf&) = delete;
};

// In-process group: the handles of several host threads of one process (one per GPU, or
// several on one GPU) all-reduce through device copies instead of RCCL.
struct LocalGroup {
    int n = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    long generation = 0;
    int next_rank_check = 0;
    std::vector<double*> bufs;
    std::vector<size_t> counts;
    std::vector<int> devices;
    bool failed = false;      // written by rank 0 between two barriers, read by the others after the second
    bool aborted = false;     // a member failed outside a collective: nobody waits for it any more
    // Returns false when the group was aborted (before or while waiting).
    bool barrier() {
        std::unique_lock<std::mutex> lk(m);
        if (aborted) return false;
        const long gen = generation;
        if (++arrived == n) { arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen || aborted; });
        return !aborted;
    }
    void abort() {
        std::lock_guard<std::mutex> lk(m);
        aborted = true;
        cv.notify_all();
    }
};

enum Family { F_LIN = 0, F_DAMP, F_SPMV_PM, F_SPMV_CM, F_PCGVEC, F_BACK, F_COST, F_DENSE, F_SCHUR, F_BSR, F_PAIRS, F_COMM, F_COUNT };

// ---------------------------------------------------------------------------------- handle
struct HandleData {
    sfmba_options opt;
    std::string err;
    int device = 0;
    hipStream_t stream = nullptr;
    void (*print_cb)(const char*, void*) = nullptr;   // sink of the iteration table (sfmba_set_print)
    void* print_user = nullptr;
    unsigned char* scratch = nullptr;     // this device's address of the library's scratch array
    bool has_problem = false;
    bool linearized = false;
    bool lin_owed = false;         // the scalar slots of the last linearisation are still block partials
    double lin_lambda = 0.0;
    double damped_lambda = -1.0;   // damping for which Vinv and s0 are current
    bool have_v = false;           // V_p and g_p of the current linearisation are in memory
    bool fuse_cm = false;          // explicit matrix: one camera-major pass instead of two when the damping is known
    double cm_lambda = -1.0;       // damping for which the one-pass camera sums (UG, T) are current; < 0: two passes

    // global shape and this partition
    int64_t n_cam = 0, n_pts_all = 0, n_obs_all = 0;
    int64_t pt_lo = 0, pt_hi = 0;          // owned point range
    int n_pts = 0, n_obs = 0, n_chunks = 0, nb_cam = 0;
    int chunk = kChunk;
    int model = MODEL_PINHOLE;             // camera model of the loaded problem
    int cd = 6, cv = 27, cw = 6;           // per camera: tangent numbers, packed upper + gradient, 16-byte words
    std::vector<int> new2old;              // internal (renumbered) local point id -> caller's local point id
                                           // (host copy; fetched from new2old_dev when the structure was built on the GPU)
    DBuf<int> new2old_dev;                 // the same on the device
    bool device_built = false;             // structure built by device_setup
    // staging of the caller's arrays and temporaries of device_setup (kept between problems)
    DBuf<int64_t> in_cam, in_pt;
    DBuf<double2> in_uv;
    DBuf<double> in_pts;
    DBuf<unsigned char> ws[12];
    const int* setup_pm2cm = nullptr;      // point-major -> camera-major position (inside ws[0])
    hipStream_t copy_stream = nullptr;     // uploads that overlap the set-up kernels
    // block pattern on the host (coarse-space set-up), filled by either set-up path
    std::vector<int> bsr_ptr_host, bsr_col_host;

    // device memory
    DBuf<int> pt_ptr, pm_cam, pm_pt, pm_orig, cm_pt, chunk_cam, chunk_beg, cam_chunk_ptr;
    DBuf<double2> pm_uv, cm_uv;
    DBuf<double2> cams[2], cams0;
    DBuf<double4> X[2], X0;
    DBuf<double4> pm_seed;
    DBuf<int> wtile_beg;
    int n_wtiles = 0;
    DBuf<double> pm_rv, cm_sx, cm_sy, cm_siz;
    DBuf<double> V, gp, Vinv, chunk_part, UG, T, Minv, rhs;
    DBuf<unsigned char> cam_seen;          // [n_cam] cameras with observations (on any rank), from the last finishing pass
    DBuf<double4> s, s0;
    DBuf<double> px, pr, pz, pp[2], pq, qS;
    DBuf<PcgState> pcg;
    PcgState* pcg_host = nullptr;          // pinned, [2]; [0] is also written directly by the single-launch PCG
    double* scalars_host = nullptr;        // pinned, [SLOT_COUNT]: the scalar slots as the kernels write them
    hipEvent_t pcg_ev[2] = {nullptr, nullptr};
    DBuf<double> dot_part, scal_part, scalars;
    DBuf<double> tmp_a, tmp_b;             // API staging (residuals, Jacobian blocks, vectors)
    // explicit block-sparse Schur complement
    bool explicit_schur = false;
    int n_ublk = 0, n_pchunks = 0, nnzb = 0, nb_bsr = 0, dot_off = 0;
    int64_t n_pairs = 0;
    DBuf<int> ublk_r, ublk_c, ublk_chunk_ptr, ublk_pos, ublk_pos_t, ublk_diag, pchunk_blk, pchunk_beg, pchunk_rec;
    DBuf<int> pair_p, bsr_ptr, bsr_col, bsr_diag;
    DBuf<double> pair_part, Bsum, Sfull;
    // single-launch PCG on the scaled system
    bool fused_pcg = false, fused_armed = false, fused_refused = false;
    int fused_blocks = 0;
    int64_t fused_solves = 0, fused_fallbacks = 0;
    DBuf<double> Shat, Fscale, bhat, fvec;    // fvec: generation records, p, xh
    DBuf<double2> fdots;
    bool trace_pcg = false;
    FusedPcg fused{};
    // two-level preconditioner of the launch-per-iteration PCG on the explicit matrix
    bool coarse_built = false;   // cluster structure and buffers present
    bool coarse_forced = false;  // asked for: in use from the first solve on
    bool coarse_on = false;      // the next solve uses it
    bool coarse_failed = false;  // a solve with it did not converge: not used again in this sfmba_solve
    bool coarse_valid = false;   // E^-1 belongs to the matrix in Sfull
    int64_t coarse_solves = 0;
    int coarse_from = 0;         // switched on after a solve of this many iterations
    DBuf<int> cl_ptr, cam_cl, cpair_i, cpair_j, cpair_cptr, cchunk_beg, cpair_blk, cpair_row;
    DBuf<double> cZ, cE[2], cscale, crc, cpart, crpart;
    bool fused_restrict = false;           // restriction formed by k_pcg_update (tuning switch, off: measured slower)
    Coarse coarse{};
    // the same coarse space inside the single-launch PCG (k_pcg_pipe2): the operator is set up on
    // copy_stream after a solve that took fused_coarse_from iterations or more and used by the
    // solves after it
    bool coarse_async = false;        // launch-per-iteration solver: operator from an earlier solve, set up on copy_stream
    bool fused_coarse_ok = false;     // structure and sizes admit it
    bool fused_coarse_ready = false;  // an operator has been set up (or is being set up) in this sfmba_solve
    int fused_coarse_from = 12;
    int fused_prev_iter = 0;           // iterations of the single-launch solve before the last one (0: none yet in this run)
    const double* fused_einv = nullptr;
    bool fused_coarse_wanted = false; // set up an operator beside the next linearisation
    bool coarse_start_pending = false; // ... or right behind the trial step that follows the solve which asked for it
    double solve_rtol = 0.0;          // PCG tolerance of the sfmba_solve in progress (0 outside one)
    double model_gain_floor = -1.0;   // inside a solve: a PCG iteration that lowers the model by less than this is not worth its time (< 0: rule off)
    DBuf<int> wg_cl;
    hipEvent_t ev_c_assembled = nullptr, ev_c_done = nullptr;
    // dense Cholesky solver
    bool use_dense = false;      // dense Cholesky asked for
    bool dense_allowed = false;  // small enough for the dense fallback
    bool dense_auto = false;     // dense Cholesky chosen by the library (small system): exact steps, damping from its floor
    int dense_n = 0;
    DBuf<double> denseA, denseLinv, denseWork;
    DBuf<double> denseFlow;      // tiles and exchange words of the single-launch factorisation
    bool dense_flow = false;     // the single-launch factorisation is used (small systems)
    DBuf<int> dense_fail;
    int dense_fallbacks = 0;
    bool split = false;           // partitioned solve with the reduced matrix split by points (options.matrix_split)
    DBuf<double> czhat;           // F^-T Z of every camera for the single-launch PCG with 9 x 9 blocks
    DBuf<double> dsum;            // its per-camera collective: diagonal blocks | T_c and right-hand-side part
    int cur = 0;                           // index of the current parameter buffers
    bool at_start = false;                 // the current parameters are the loaded ones (cams0, X0: read only), a trial step goes into buffer 0
    Dev dev{};

    // communication
    ncclComm_t comm = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;   // ncclCommDestroy of the library `comm` came from
    LocalGroup* group = nullptr;
    bool has_comm() const { return comm != nullptr || group != nullptr; }

    // timing
    bool profile = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    struct Span { int fam; hipEvent_t a, b; };
    std::vector<Span> spans;
    double fam_ms[F_COUNT] = {};
    int64_t fam_n[F_COUNT] = {};

    ~HandleData() {
        for (auto e : ev_pool) (void)hipEventDestroy(e);
        for (auto e : pcg_ev) if (e) (void)hipEventDestroy(e);
        if (pcg_host) (void)hipHostFree(pcg_host);
        if (scalars_host) (void)hipHostFree(scalars_host);
        if (ev_c_assembled) (void)hipEventDestroy(ev_c_assembled);
        if (ev_c_done) (void)hipEventDestroy(ev_c_done);
        if (comm && comm_destroy) comm_destroy(comm);
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

}  // namespace sfmba

struct sfmba_handle : sfmba::HandleData {};

namespace sfmba {

using H = sfmba_handle;

// Phase times of sfmba_set_problem on stderr when SFMBA_TRACE_SETUP is set.
struct SetupTrace {
    bool on = getenv("SFMBA_TRACE_SETUP") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "  setup %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

// sfmba_setup.hip: the problem structure (orderings, chunks, wave tiles, pair lists, block
// pattern) built on the GPU from the caller's index arrays already in device memory.
// device_setup needs the two index arrays only; device_place_uv, called after it, puts the image
// points into both orders.
void device_setup(H* h, const int64_t* cam_idx, const int64_t* pt_idx);
void device_place_uv(H* h, const double2* uv);

// sfmba_host.cpp: what the options ask of the reduced system (and the reset of the handle's
// explicit-matrix state), and everything that follows from the structure once its counts, device
// arrays and the block pattern on the host are in place; shared by both set-up paths.
struct SchurPlan { int mode; bool want_dense; };
SchurPlan plan_schur(H* h);
void finish_schur_setup(H* h, bool want_dense);

}  // namespace sfmba
