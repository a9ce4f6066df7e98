// Host side of libsfmba: problem packing (polar projection of the rotation blocks, point-major
// and camera-major orderings, chunking, partitioning over ranks), the Levenberg-Marquardt loop
// that takes the place of scipy.optimize.least_squares(method='trf')
// (/root/reference/bundle_adjustment.py:141-143; scipy/optimize/_lsq/trf.py:401-560), and the
// C ABI declared in include/sfmba.h.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "sfmba_handle.h"

using namespace sfmba;

namespace {

// ---------------------------------------------------------------------------------- RCCL
// Resolved at run time so that a single-GPU process never needs the library.
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string& err) {
        if (lib) return true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        lib = dlopen(names[0], RTLD_NOW | RTLD_NOLOAD);   // reuse a copy the process already holds
        for (int i = 0; !lib && i < 3; ++i) lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (!lib) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
        GetUniqueId = (decltype(GetUniqueId))dlsym(lib, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(lib, "ncclCommInitRank");
        AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!GetUniqueId || !CommInitRank || !AllReduce || !CommDestroy || !GetErrorString) {
            err = "librccl lacks a required symbol";
            return false;
        }
        return true;
    }
};
Rccl g_rccl;

// ---------------------------------------------------------------------------------- polar
// Orthogonal polar factor U V' of a 3x3 (what cv2.Rodrigues keeps of its input,
// /root/reference/bundle_adjustment.py:24) by one-sided Jacobi SVD, then the cv2 rule that a
// rotation with sin(angle) < 1e-5 and cos(angle) > 0 is the identity.
void polar3(const double* R, double* Q) {
    double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[i][j] = R[3 * i + j];
    const double eps = std::numeric_limits<double>::epsilon();
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double al = 0, be = 0, ga = 0;
                for (int i = 0; i < 3; ++i) { al += A[i][p] * A[i][p]; be += A[i][q] * A[i][q]; ga += A[i][p] * A[i][q]; }
                if (std::fabs(ga) <= eps * std::sqrt(al * be) || ga == 0.0) continue;
                rotated = true;
                double zeta = (be - al) / (2.0 * ga);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < 3; ++i) {
                    double a = A[i][p], b = A[i][q];
                    A[i][p] = c * a - s * b; A[i][q] = s * a + c * b;
                    a = V[i][p]; b = V[i][q];
                    V[i][p] = c * a - s * b; V[i][q] = s * a + c * b;
                }
            }
        if (!rotated) break;
    }
    double U[3][3], sig[3];
    for (int j = 0; j < 3; ++j) {
        sig[j] = std::sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
    }
    const double smax = std::max(sig[0], std::max(sig[1], sig[2]));
    bool good[3];
    int ngood = 0;
    for (int j = 0; j < 3; ++j) {
        good[j] = sig[j] > 1e-14 * smax && sig[j] > 0.0;
        if (good[j]) { ++ngood; for (int i = 0; i < 3; ++i) U[i][j] = A[i][j] / sig[j]; }
    }
    if (ngood == 3) {
        // nothing to complete
    } else if (ngood == 2) {
        int z = !good[0] ? 0 : (!good[1] ? 1 : 2), a = (z + 1) % 3, b = (z + 2) % 3;
        U[0][z] = U[1][a] * U[2][b] - U[2][a] * U[1][b];
        U[1][z] = U[2][a] * U[0][b] - U[0][a] * U[2][b];
        U[2][z] = U[0][a] * U[1][b] - U[1][a] * U[0][b];
    } else {   // rank <= 1: no meaningful rotation; keep whatever direction exists, complete arbitrarily
        double e[3] = {1, 0, 0};
        int g = good[0] ? 0 : (good[1] ? 1 : (good[2] ? 2 : -1));
        if (g < 0) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) U[i][j] = (i == j); }
        else {
            if (std::fabs(U[0][g]) > 0.9) { e[0] = 0; e[1] = 1; }
            int a = (g + 1) % 3, b = (g + 2) % 3;
            double n[3] = {U[1][g] * e[2] - U[2][g] * e[1], U[2][g] * e[0] - U[0][g] * e[2], U[0][g] * e[1] - U[1][g] * e[0]};
            double nn = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            for (int i = 0; i < 3; ++i) U[i][a] = n[i] / nn;
            U[0][b] = U[1][g] * U[2][a] - U[2][g] * U[1][a];
            U[1][b] = U[2][g] * U[0][a] - U[0][g] * U[2][a];
            U[2][b] = U[0][g] * U[1][a] - U[1][g] * U[0][a];
        }
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Q[3 * i + j] = U[i][0] * V[j][0] + U[i][1] * V[j][1] + U[i][2] * V[j][2];
    double rx = Q[7] - Q[5], ry = Q[2] - Q[6], rz = Q[3] - Q[1];
    double s = std::sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (Q[0] + Q[4] + Q[8] - 1.0) * 0.5;
    if (s < 1e-5 && c > 0) {
        for (int i = 0; i < 9; ++i) Q[i] = (i % 4 == 0) ? 1.0 : 0.0;
    } else if (s < 1e-5) {
        // Within 1e-5 of a half turn cv2.Rodrigues takes the axis from the diagonal of the matrix
        // (signs from the first row, one more sign rule for the smallest component) and the angle
        // acos(c) from the trace; cv2.projectPoints then rotates by exactly that vector.
        const double cc = std::min(1.0, std::max(-1.0, c));
        const double theta = std::acos(cc);
        double vx = std::sqrt(std::max((Q[0] + 1.0) * 0.5, 0.0));
        double vy = std::sqrt(std::max((Q[4] + 1.0) * 0.5, 0.0)) * (Q[1] < 0 ? -1.0 : 1.0);
        double vz = std::sqrt(std::max((Q[8] + 1.0) * 0.5, 0.0)) * (Q[2] < 0 ? -1.0 : 1.0);
        if (std::fabs(vx) < std::fabs(vy) && std::fabs(vx) < std::fabs(vz) && ((Q[5] > 0) != (vy * vz > 0))) vz = -vz;
        const double n = std::sqrt(vx * vx + vy * vy + vz * vz);
        const double k[3] = {vx / n, vy / n, vz / n};
        const double ct = std::cos(theta), st = std::sin(theta), v1 = 1.0 - ct;
        Q[0] = ct + v1 * k[0] * k[0];         Q[1] = v1 * k[0] * k[1] - st * k[2];  Q[2] = v1 * k[0] * k[2] + st * k[1];
        Q[3] = v1 * k[1] * k[0] + st * k[2];  Q[4] = ct + v1 * k[1] * k[1];         Q[5] = v1 * k[1] * k[2] - st * k[0];
        Q[6] = v1 * k[2] * k[0] - st * k[1];  Q[7] = v1 * k[2] * k[1] + st * k[0];  Q[8] = ct + v1 * k[2] * k[2];
    }
}

// ---------------------------------------------------------------------------------- scratch
// Device memory of sfmba_triangulate_noalloc: the staging that sfmba_triangulate allocates per
// call (pixel coordinates in, points and errors out), here a fixed array in the code object of
// the library.  Every GPU the library runs on holds its own copy at its own address until the
// process ends; `scratch` itself is only a host-side stand-in for the symbol, the address on a
// device comes from hipGetSymbolAddress with that device current (scratch_ptr below) and is what
// copies and kernels are given.
constexpr int64_t kNoallocMax = SFMBA_TRIANGULATE_NOALLOC_MAX;
constexpr size_t kScrX1 = 0;                                   // [max] double2
constexpr size_t kScrX2 = kScrX1 + 16 * (size_t)kNoallocMax;   // [max] double2
constexpr size_t kScrX = kScrX2 + 16 * (size_t)kNoallocMax;    // [max][3] double
constexpr size_t kScrErr = kScrX + 24 * (size_t)kNoallocMax;   // [2][max] double
constexpr size_t kScrBytes = kScrErr + 16 * (size_t)kNoallocMax;
static __device__ __align__(256) unsigned char scratch[kScrBytes];

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned char* g_scratch_ptr[kMaxDevices] = {};

// Address of the scratch array on `device`, which must be the current device; resolved once.
hipError_t scratch_ptr(int device, unsigned char** out) {
    *out = nullptr;
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_scratch_ptr[device] = static_cast<unsigned char*>(p);
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}


}  // namespace

namespace {

void refresh_dev(H* h) {
    Dev& d = h->dev;
    d.n_cam = (int)h->n_cam; d.n_pts = h->n_pts; d.n_obs = h->n_obs; d.n_chunks = h->n_chunks;
    d.model = h->model;
    d.pt_ptr = h->pt_ptr.p; d.pm_cam = h->pm_cam.p; d.pm_pt = h->pm_pt.p; d.pm_orig = h->pm_orig.p;
    d.pm_uv = h->pm_uv.p; d.cm_pt = h->cm_pt.p; d.cm_uv = h->cm_uv.p;
    d.chunk_cam = h->chunk_cam.p; d.chunk_beg = h->chunk_beg.p; d.cam_chunk_ptr = h->cam_chunk_ptr.p;
    // (after sfmba_reset_params the current point is the loaded one, which is kept read-only: no copy)
    d.cams = h->at_start ? h->cams0.p : h->cams[h->cur].p; d.X = h->at_start ? h->X0.p : h->X[h->cur].p;
    d.cams_trial = h->cams[h->at_start ? 0 : h->cur ^ 1].p; d.X_trial = h->X[h->at_start ? 0 : h->cur ^ 1].p;
    d.pm_seed = h->pm_seed.p; d.pm_rv = h->pm_rv.p;
    d.n_wtiles = h->n_wtiles; d.wtile_beg = h->wtile_beg.p;
    d.cm_sx = h->cm_sx.p; d.cm_sy = h->cm_sy.p; d.cm_siz = h->cm_siz.p;
    d.V = h->V.p; d.gp = h->gp.p; d.Vinv = h->Vinv.p; d.s = h->s.p; d.s0 = h->s0.p;
    d.chunk_part = h->chunk_part.p; d.UG = h->UG.p; d.T = h->T.p; d.Minv = h->Minv.p; d.rhs = h->rhs.p; d.cam_seen = h->cam_seen.p;
    d.px = h->px.p; d.pr = h->pr.p; d.pz = h->pz.p; d.pp[0] = h->pp[0].p; d.pp[1] = h->pp[1].p; d.pq = h->pq.p; d.qS = h->qS.p;
    d.pcg = h->pcg.p; d.pcg_mirror = h->pcg_host; d.dot_part = h->dot_part.p; d.scal_part = h->scal_part.p; d.scalars = h->scalars.p;
    // (with a communicator the slots are summed over the ranks on the device and then copied)
    d.scalars_host = h->has_comm() ? nullptr : h->scalars_host;
    d.dot_off = h->dot_off;
    d.n_ublk = h->n_ublk; d.n_pchunks = h->n_pchunks; d.nnzb = h->nnzb;
    d.ublk_r = h->ublk_r.p; d.ublk_c = h->ublk_c.p; d.ublk_chunk_ptr = h->ublk_chunk_ptr.p;
    d.ublk_pos = h->ublk_pos.p; d.ublk_pos_t = h->ublk_pos_t.p; d.ublk_diag = h->ublk_diag.p;
    d.pchunk_blk = h->pchunk_blk.p; d.pchunk_beg = h->pchunk_beg.p;
    d.pchunk_rec = reinterpret_cast<const int4*>(h->pchunk_rec.p);
    d.pair_p = h->pair_p.p;
    d.pair_part = h->pair_part.p; d.Bsum = h->Bsum.p;
    d.bsr_ptr = h->bsr_ptr.p; d.bsr_col = h->bsr_col.p; d.bsr_diag = h->bsr_diag.p; d.Sfull = h->Sfull.p;
}

// ---- timing spans -------------------------------------------------------------------------
hipEvent_t next_event(H* h) {
    if (h->ev_used == h->ev_pool.size()) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        h->ev_pool.push_back(e);
    }
    return h->ev_pool[h->ev_used++];
}
struct Scope {
    H* h; int fam; hipEvent_t a = nullptr;
    Scope(H* h_, int fam_) : h(h_), fam(fam_) {
        if (h->profile) { a = next_event(h); HIP_TRY(hipEventRecord(a, h->stream)); }
        h->fam_n[fam]++;
    }
    ~Scope() {
        if (h->profile && a) {
            hipEvent_t b = nullptr;
            if (h->ev_used < h->ev_pool.size()) b = h->ev_pool[h->ev_used++];
            else if (hipEventCreate(&b) == hipSuccess) { h->ev_pool.push_back(b); h->ev_used++; }
            if (b && hipEventRecord(b, h->stream) == hipSuccess) h->spans.push_back({fam, a, b});
        }
    }
};
void collect_spans(H* h) {
    for (auto& s : h->spans) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, s.a, s.b));
        h->fam_ms[s.fam] += ms;
    }
    h->spans.clear();
    h->ev_used = 0;
}

// ---- collectives --------------------------------------------------------------------------
void allreduce(H* h, double* buf, size_t count) {
    if (count == 0 || !h->has_comm()) return;
    Scope t(h, F_COMM);
    if (h->group) {
        LocalGroup* g = h->group;
        const int rk = h->opt.rank;
        HIP_TRY(hipStreamSynchronize(h->stream));
        g->bufs[rk] = buf;
        g->counts[rk] = count;
        if (!g->barrier()) fail(SFMBA_ERR_COMM, "in-process group aborted: another member failed");
        if (rk == 0) {                       // fixed rank order: bitwise reproducible
            hipError_t e = hipSuccess;
            for (int r = 1; r < g->n; ++r)   // ranks out of step: refuse before touching memory
                if (g->counts[r] != count) e = hipErrorInvalidValue;
            for (int r = 1; r < g->n && e == hipSuccess; ++r) {
                launch_vec_add(buf, g->bufs[r], count, h->stream);
                e = hipGetLastError();
            }
            // (device-to-device copies need not have landed when a blocking hipMemcpy returns: queue
            // them behind the sums and wait for the stream before the other ranks are released)
            for (int r = 1; r < g->n && e == hipSuccess; ++r)
                e = hipMemcpyAsync(g->bufs[r], buf, count * sizeof(double), hipMemcpyDeviceToDevice, h->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            g->failed = e != hipSuccess;
        }
        if (!g->barrier()) fail(SFMBA_ERR_COMM, "in-process group aborted: another member failed");
        if (g->failed) fail(SFMBA_ERR_COMM, "in-process all-reduce failed");
        return;
    }
    if (!h->comm) return;
    ncclResult_t r = g_rccl.AllReduce(buf, buf, count, ncclFloat64, ncclSum, h->comm, h->stream);
    if (r != ncclSuccess) fail(SFMBA_ERR_COMM, "ncclAllReduce: %s", g_rccl.GetErrorString(r));
}

// The coarse matrix of a split reduced matrix is the ranks' sum (launch_coarse_setup calls this).
void sum_coarse_over_ranks(void* ctx, double* E, size_t n) { allreduce(static_cast<H*>(ctx), E, n); }
const double* coarse_setup_here(H* h) {
    const bool split = h->split && h->has_comm();
    return launch_coarse_setup(h->dev, h->coarse, h->stream, nullptr, split ? sum_coarse_over_ranks : nullptr, h);
}

void read_scalars(H* h, double* out) {
    if (!h->has_comm()) {
        // every kernel that fills a slot has also written it to pinned host memory
        HIP_TRY(hipStreamSynchronize(h->stream));
        memcpy(out, h->scalars_host, SLOT_COUNT * sizeof(double));
        return;
    }
    HIP_TRY(hipMemcpyAsync(out, h->scalars.p, SLOT_COUNT * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
}

// Sum the partition-local scalar slots over ranks (the camera slots are replicated already).
void reduce_scalars(H* h, int first, int count) {
    if (!h->has_comm()) return;
    allreduce(h, h->scalars.p + first, (size_t)count);
}

// ---- solver phases ------------------------------------------------------------------------
// Linearise at the current estimate: seeds, residuals, V_p, g_p, U_c, g_c.  Leaves the cost sum
// in scalars[SLOT_COST] (this partition's part), max|g_p| in scalars[SLOT_GMAX_PT] and max|g_c|
// in scalars[SLOT_GMAX_CAM].  (Running the two halves on two streams was measured: no gain.)
// `lambda` is the damping of the step that will follow (0 = not known): the points are then damped
// in the same pass and phase_damp(lambda) finds that done.
// `grad_only`: the solve ends at this point and only its gradient norm is still wanted.
// `defer`: nobody on the device needs the scalar slots of this linearisation and the host will read
// them only with those of the trial step that follows: they are finished by that step's launch
// (phase_trial) instead of two small launches of their own.
// The coarse operator of the single-launch PCG is set up on copy_stream beside the kernels of the
// next linearisation.  It reads the reduced matrix up to the event ev_c_assembled: the finishing
// passes that overwrite it on the main stream wait for the event first, and the PCG launch that uses
// the operator waits for ev_c_done.  (Its other inputs are not rewritten while it runs: the cameras of
// the current point - a trial step goes into the other buffer - and the flags of the observed cameras.)
// (Giving the second stream the device's highest priority was measured and changes nothing: a sweep
// of the coarse operator's chain that is launched while k_schur_pairs runs - one-wavefront workgroups
// at three per SIMD, refilled one at a time - finds its four wavefront slots on one compute unit only
// when that kernel's grid is exhausted, 150 instead of 20 us, whatever the priority.)
void ensure_copy_stream(H* h) {
    if (!h->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
}
// What rewrites the reduced matrix (the finishing passes of phase_damp) waits for the set-up's
// assembly, which at 13 682 cameras takes as long as the point-major pass beside it.
void wait_coarse_matrix(H* h) {
    if (h->fused_coarse_ready) HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_c_assembled, 0));
}
void start_fused_coarse(H* h) {
    if (!h->ev_c_done) {
        HIP_TRY(hipEventCreateWithFlags(&h->ev_c_assembled, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_c_done, hipEventDisableTiming));
    }
    ensure_copy_stream(h);
    // (the caller has waited for the main stream: the matrix of the solve just finished is in
    // place and no single-launch solve that could still read the last operator is running)
    h->fused_einv = launch_coarse_setup(h->dev, h->coarse, h->copy_stream, h->ev_c_assembled);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev_c_done, h->copy_stream));
    h->fused_coarse_ready = true;
}
// Fills the coarse fields of the launch arguments; with an operator at hand the launch waits for it.
void arm_fused_coarse(H* h, FusedPcg& f) {
    f.coarse = 0;
    if (!(h->fused_coarse_ok && h->fused_coarse_ready)) return;
    HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_c_done, 0));
    f.coarse = 1;
    f.c_n = h->coarse.n; f.c_npad = h->coarse.n_pad;
    f.cam_cl = h->coarse.cam_cl; f.cl_ptr = h->coarse.cl_ptr;
    f.c_z = h->coarse.Z; f.c_einv = h->fused_einv; f.c_scale = h->coarse.scale;
    f.c_zhat = h->cd != 6 ? h->czhat.p : nullptr;
    f.wg_cl = h->wg_cl.p;
}

void phase_linearize(H* h, double lambda = 0.0, bool grad_only = false, bool defer = false) {
    Scope t(h, F_LIN);
    refresh_dev(h);
    defer = defer && !h->has_comm();
    launch_lin_pm(h->dev, !h->explicit_schur, lambda, h->stream, grad_only, /*keep_v=*/false, defer);
    h->lin_owed = defer;
    h->damped_lambda = lambda > 0.0 && !grad_only ? lambda : -1.0;
    h->have_v = !grad_only && !(lambda > 0.0);
    if (h->fuse_cm && h->explicit_schur && lambda > 0.0 && !grad_only) {
        // the points are damped already: U_c - T_c in one camera-major pass (phase_damp finds it done)
        launch_lin_sdiag_cm(h->dev, h->Bsum.p + (size_t)h->n_ublk * h->cd * h->cd, h->stream);
        h->cm_lambda = lambda;
    } else {
        launch_lin_cm(h->dev, h->stream, grad_only);
        launch_chunk_reduce(h->dev, h->cv, h->UG.p, h->stream);
        h->cm_lambda = -1.0;
    }
    if (h->has_comm()) {
        // one collective for the camera blocks and the gradient maximum over the points: the
        // maximum travels as one slot per rank behind the blocks (a sum all-reduce of slots of
        // which only one rank fills each)
        const int nr = h->opt.nranks;
        double* slots = h->UG.p + (size_t)h->n_cam * h->cv;
        HIP_TRY(hipMemsetAsync(slots, 0, nr * sizeof(double), h->stream));
        HIP_TRY(hipMemcpyAsync(slots + h->opt.rank, h->scalars.p + SLOT_GMAX_PT, sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        allreduce(h, h->UG.p, (size_t)h->n_cam * h->cv + nr);
    }
    if (!defer) launch_gmax_cam(h->dev, h->stream);
    // The coarse operator's set-up is queued on its stream once both passes of the linearisation are
    // on their way.  It reads nothing a linearisation writes: the cameras of the point it belongs to,
    // the reduced matrix, and the flags of the cameras that have observations, which the last
    // finishing pass left.  (Its launches take the host 45 us at 1 778 cameras and 70 us at 13 682 -
    // SFMBA_TRACE_COARSE_HOST=1 prints them - and several times that under the kernel tracer, whose
    // time lines therefore show the main stream waiting for the host around this point; without the
    // tracer the order of these calls makes no measurable difference.)
    if (h->fused_coarse_wanted && !grad_only) start_fused_coarse(h);
    h->fused_coarse_wanted = false;
    h->linearized = false;
}

// Vinv and V*^-1 g_p for another damping of the same linearisation: from V_p and g_p when they were
// stored, otherwise by the point-major linearisation pass again (this time keeping them).
void damp_points(H* h, double lambda) {
    if (h->have_v) { launch_damp_pm(h->dev, lambda, h->stream); return; }
    launch_lin_pm(h->dev, !h->explicit_schur, lambda, h->stream, false, /*keep_v=*/true);
    h->have_v = true;
}

// Damp with lambda: Vinv, V*^-1 g_p, block-Jacobi blocks, reduced right-hand side; in explicit
// mode also the whole block-sparse reduced matrix S.
void phase_damp(H* h, double lambda) {
    if (h->fused_coarse_wanted) { start_fused_coarse(h); h->fused_coarse_wanted = false; }   // (after a rejected step)
    if (h->explicit_schur) {
        // per camera, behind the blocks: T_c = sum W Vinv W' (its own observations: the diagonal
        // block's share) and sum W Vinv g_p (the point part of the right-hand side)
        double* tpart = h->Bsum.p + (size_t)h->n_ublk * h->cd * h->cd;
        if (h->damped_lambda != lambda) {
            Scope t(h, F_DAMP);
            damp_points(h, lambda);
        }
        if (h->cm_lambda < 0.0) {
            Scope t(h, F_DAMP);
            launch_sdiag_cm(h->dev, h->stream);
            launch_chunk_reduce(h->dev, h->cv, tpart, h->stream);
        } else if (h->cm_lambda != lambda) {
            // another damping of a linearisation whose camera sums were formed in one pass (a
            // rejected step): the pass again, with the new Vinv and s0
            Scope t(h, F_DAMP);
            launch_lin_sdiag_cm(h->dev, tpart, h->stream);
            h->cm_lambda = lambda;
            if (h->has_comm()) allreduce(h, h->UG.p, (size_t)h->n_cam * h->cv);
        }
        {
            Scope t(h, F_SCHUR);
            {
                Scope t2(h, F_PAIRS);
                launch_schur_pairs(h->dev, h->stream);
            }
            launch_block_reduce(h->dev, h->stream);
        }
        if (h->split && h->has_comm()) {
            // matrix split by points: the off-diagonal blocks stay this rank's own; what is summed
            // over the ranks is per camera - the diagonal block and the camera-major sums
            const size_t words = diag_collect_words(h->dev, h->cv);
            h->dsum.alloc(words);
            launch_diag_collect(h->dev, h->cv, tpart, h->dsum.p, h->stream);
            allreduce(h, h->dsum.p, words);
            launch_diag_spread(h->dev, h->cv, h->dsum.p, tpart, h->stream);
        } else {
            allreduce(h, h->Bsum.p, (size_t)h->n_ublk * h->cd * h->cd + (size_t)h->n_cam * h->cv);
        }
        {
            Scope t(h, F_SCHUR);
            wait_coarse_matrix(h);
            launch_schur_finish(h->dev, lambda, tpart, h->fused_pcg ? &h->fused : nullptr, h->stream,
                                h->split && h->has_comm() ? 1.0 / h->opt.nranks : 1.0);
            h->fused_armed = h->fused_pcg;
        }
        h->coarse_valid = false;
    } else {
        Scope t(h, F_DAMP);
        if (h->damped_lambda != lambda) damp_points(h, lambda);
        launch_sdiag_cm(h->dev, h->stream);
        launch_chunk_reduce(h->dev, h->cv, h->T.p, h->stream);
        allreduce(h, h->T.p, (size_t)h->n_cam * h->cv);
        launch_precond(h->dev, lambda, h->stream);
    }
    h->linearized = true;
    h->lin_lambda = lambda;
    h->damped_lambda = lambda;
}

// qS = sum over observations of W Vinv W' x (this partition's part, then all ranks').
void schur_product(H* h, const double* x, const PcgState* state) {
    {
        Scope t(h, F_SPMV_PM);
        launch_spmv_pm(h->dev, x, state, h->stream);
    }
    {
        Scope t(h, F_SPMV_CM);
        launch_spmv_cm(h->dev, h->s.p, state, h->stream);
    }
    {
        Scope t(h, F_PCGVEC);
        launch_chunk_reduce(h->dev, h->cd, h->qS.p, h->stream);
    }
    allreduce(h, h->qS.p, (size_t)h->n_cam * h->cd);
}

// The launch-per-iteration PCG switches the coarse space on once a solve has taken this many
// iterations: below that the set-up (assembly and inversion of E) costs more than it saves.
constexpr int kCoarseFromIterations = 24;

// Block-Jacobi PCG on S dc = rhs; result in px.  Returns the iteration count.
// The forcing term of the inexact step when the caller leaves it open.  A PCG iteration of the
// single-launch solver costs a hundredth of an LM iteration (3 to 5 us against the passes over the
// observations), so the linear systems are solved more accurately there: the outer iteration then
// converges in one step less on most problems (4 instead of 5 at the Venice shape).  With a launch
// (or two) per PCG iteration the extra iterations cost what the saved step gains.
// Chosen once per sfmba_solve (a single launch that is refused half-way does not change it).
double pcg_rtol_of(const H* h) {
    if (h->opt.pcg_rtol > 0) return h->opt.pcg_rtol;
    if (h->solve_rtol > 0) return h->solve_rtol;
    return h->explicit_schur && h->fused_pcg && !h->use_dense ? 0.03 : 0.1;
}

// The stopping rules of a linear solve.  With the tolerance given in the options: the residual rule
// alone.  Left open, inside a solve: the residual must pass the tolerance of pcg_rtol_of (`loose`)
// and the iteration goes on while it still lowers the quadratic model by more than a twentieth of
// what the ftol rule of the outer loop can see (ftol x cost) per iteration, down to `tight`.  Far from
// the solution that ends a solve early in relative terms, near it the steps become accurate, and
// the outer loop is not sent into another iteration by the linear solver alone.
struct PcgStop { double tight, loose, q_abs; };
PcgStop pcg_stop_of(const H* h) {
    const double tol = pcg_rtol_of(h);
    if (h->opt.pcg_rtol > 0 || !(h->model_gain_floor >= 0.0)) return {tol, tol, -1.0};
    return {tol / 30.0, tol, h->model_gain_floor};
}

PcgState phase_pcg(H* h, double lambda, bool allow_fused = true) {
    const PcgStop stop = pcg_stop_of(h);
    const double tol = stop.tight;
    const int max_iter = h->opt.pcg_max_iter > 0 ? h->opt.pcg_max_iter : 1000;
    if (allow_fused && h->explicit_schur && h->fused_pcg) {
        // the whole solve in one launch; a breakdown of the single-reduction
        // recurrence or a barrier timeout falls through to the launch-per-iteration solver
        FusedPcg& f = h->fused;
        f.x = h->px.p; f.out = &h->pcg_host[0]; f.tol2 = tol * tol; f.max_iter = max_iter; f.flag_slot = nullptr; f.flag_host = nullptr;
        f.tol2_loose = stop.loose * stop.loose; f.q_abs = stop.q_abs;
        arm_fused_coarse(h, f);
        hipError_t e = hipErrorNotSupported;
        if (!h->fused_refused) {
            Scope t(h, F_BSR);
            // (plain launch: grids of several handles that meet on one GPU can keep each other from
            // becoming resident; the admission timeout then sends those solves to the fallback)
            e = launch_pcg_fused(f, h->fused_armed, h->stream);
            h->fused_armed = false;
        }
        PcgState st{};
        st.pad = 4;                             // launch refused
        if (e == hipSuccess) {
            HIP_TRY(hipStreamSynchronize(h->stream));        // the kernel wrote its state to pinned host memory
            st = h->pcg_host[0];
        } else if (!h->fused_refused) {
            if (h->trace_pcg) fprintf(stderr, "pcg single launch refused: %s\n", hipGetErrorString(e));
            (void)hipGetLastError();
            h->fused_refused = true;            // no further launches; with a communicator this rank
            if (!h->has_comm()) h->fused_pcg = false;   // still takes part in the ranks' agreement below
        }
        if (h->has_comm()) {
            // every rank solves the same replicated system, but a timeout or a refused launch can
            // hit one of them: either all keep the single-launch result or all redo the solve
            double flag = st.pad != 0 ? 1.0 : 0.0;
            HIP_TRY(hipMemcpyAsync(h->tmp_b.p, &flag, sizeof(double), hipMemcpyHostToDevice, h->stream));
            allreduce(h, h->tmp_b.p, 1);
            HIP_TRY(hipMemcpyAsync(&flag, h->tmp_b.p, sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            if (flag != 0.0 && st.pad == 0) st.pad = 3;
        }
        if (h->trace_pcg && e == hipSuccess)
            fprintf(stderr, "pcg single launch: %d iterations, %.2f us in kernel, flag %d\n", st.iter, st.ticks * 0.01, st.pad);
        if (st.pad == 0) { h->fused_solves++; if (f.coarse) h->coarse_solves++; return st; }
        h->fused_fallbacks++;
    }
    // two-level preconditioner: the coarse matrix belongs to one damping of one linearisation
    // two-level preconditioner: either the operator of this very matrix, set up here (asked for
    // with pcg_coarse > 0, and the old policy), or the one set up on the second stream from the
    // matrix of an earlier solve of the run (the default: see start_fused_coarse)
    const bool async_op = h->coarse_async && h->fused_coarse_ready && !h->coarse_failed && !h->fused_pcg;
    const bool two_level = h->explicit_schur && h->coarse_built && (h->coarse_on || async_op);
    if (two_level && async_op && !h->coarse_on) {
        HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_c_done, 0));
        h->coarse.Einv = h->fused_einv;
    } else if (two_level && !h->coarse_valid) {
        Scope t(h, F_SCHUR);
        h->coarse.Einv = coarse_setup_here(h);
        h->coarse_valid = true;
    }
    {
        Scope t(h, F_PCGVEC);
        launch_pcg_init(h->dev, tol, max_iter, h->stream, stop.loose, stop.q_abs);
        if (two_level) launch_coarse_apply(h->dev, h->coarse, 0, h->stream);
    }
    // Iterations are queued in batches; the head of every iteration writes its state into pinned
    // host memory as well (all fields, a fence, then `done`), an event marks the end of batch k and
    // the host looks at the state only after batch k+1 has been queued, so the GPU never waits for
    // the host - and no copy sits between the batches (a device-to-host copy in the stream is a
    // kernel of its own: 6 us and a 4 us bubble per batch).  What the host reads may already
    // belong to an iteration of batch k+1: it only acts on `done`, after which the record no longer
    // changes.  Kernels queued after convergence return at once (PcgState.done).
    int batch = h->opt.pcg_batch > 0 ? h->opt.pcg_batch : 4;
    int cur = 0, queued = 0;
    auto queue_batch = [&](int slot) {
        for (int i = 0; i < batch; ++i) {
            int n_pq;
            if (h->explicit_schur && h->split && h->has_comm()) {
                // this rank's share of q = S p; the ranks' sum; p'q from the complete product
                {
                    Scope t(h, F_BSR);
                    launch_pcg_spmv_bsr(h->dev, cur, h->stream);
                }
                allreduce(h, h->pq.p, (size_t)h->n_cam * h->cd);
                Scope t(h, F_PCGVEC);
                launch_pcg_dot(h->dev, cur, h->stream);
                n_pq = pcg_update_blocks(h->dev);
            } else if (h->explicit_schur) {
                Scope t(h, F_BSR);
                launch_pcg_spmv_bsr(h->dev, cur, h->stream);
                n_pq = pcg_spmv_blocks(h->dev);
            } else {
                {
                    Scope t(h, F_PCGVEC);
                    launch_pcg_dir(h->dev, cur, h->stream);
                }
                schur_product(h, h->pp[cur ^ 1].p, h->pcg.p + (cur ^ 1));
                Scope t(h, F_PCGVEC);
                launch_pcg_q(h->dev, lambda, cur, h->stream);
                n_pq = h->nb_cam;
            }
            Scope t(h, F_PCGVEC);
            // (the restriction of the two-level preconditioner stays a launch of its own: formed inside
            // k_pcg_update per workgroup and cluster and put together by the prolongation it was measured
            // 4 us per iteration slower at 13 682 cameras - the seven wavefront sums per cluster slot and
            // the prolongation's two extra dependent loads cost more than the launch they replace;
            // SFMBA_PCG_FUSED_RESTRICT=1 selects that form)
            launch_pcg_update(h->dev, cur, n_pq, h->stream, two_level && h->fused_restrict ? &h->coarse : nullptr);
            if (two_level) launch_coarse_apply(h->dev, h->coarse, cur ^ 1, h->stream, h->fused_restrict);
            cur ^= 1;
            ++queued;
        }
        HIP_TRY(hipEventRecord(h->pcg_ev[slot], h->stream));
    };
    auto read_state = [&]() {
        const volatile PcgState* m = h->pcg_host;
        PcgState s{};
        s.done = m->done;
        std::atomic_thread_fence(std::memory_order_acquire);
        s.rz = m->rz; s.rz0 = m->rz0; s.tol2 = m->tol2; s.tol2_loose = m->tol2_loose; s.q_abs = m->q_abs;
        s.iter = m->iter; s.max_iter = m->max_iter;
        return s;
    };
    const bool pipelined = batch > 1;
    int slot = 0;
    queue_batch(slot);
    PcgState st{};
    for (;;) {
        if (pipelined) queue_batch(slot ^ 1);
        HIP_TRY(hipEventSynchronize(h->pcg_ev[slot]));
        st = read_state();
        if (st.done) break;
        if (queued >= max_iter + 2 * batch) {
            // (cannot happen while max_iter is what the kernels stop at; the record of the last
            // iteration queued, at rest)
            HIP_TRY(hipStreamSynchronize(h->stream));
            st = read_state();
            break;
        }
        if (!pipelined) queue_batch(slot);
        else slot ^= 1;
    }
    if (two_level) {
        h->coarse_solves++;
        if (!(st.rz0 >= 0.0 && st.rz >= 0.0 && st.rz <= st.tol2 * st.rz0)) {
            // did not converge, produced a NaN, or r'M^-1 r came out negative (a preconditioner that
            // is not positive definite): the coarse matrix was too ill-conditioned to invert; this
            // solve and the rest of the run use the block-Jacobi blocks alone
            h->coarse_on = false;
            h->coarse_failed = true;
            return phase_pcg(h, lambda, allow_fused);
        }
    } else if (h->explicit_schur && h->coarse_built && !h->coarse_failed && !h->coarse_async && st.iter >= h->coarse_from) {
        h->coarse_on = true;        // solves only get longer as the damping falls
    }
    // With the operator set up on the second stream: the set-up starts here, as soon as the solve
    // that asks for it is known to be over, beside the trial step and the next linearisation.  (It
    // used to start with the linearisation: at 13 682 cameras its 24 sweeps then ran into
    // k_schur_pairs, beside which a sweep starves - 1 ms for one of them - and the next solve waited
    // 0.45 ms for the operator.)  The solve has converged when the host sees `done`: what is still
    // queued of it returns at once and reads nothing of the coarse operator, the matrix stays as it
    // is until the finishing passes of the next damping, which wait for the set-up's assembly.
    if (h->explicit_schur && h->coarse_built && h->coarse_async && !h->coarse_failed && !h->coarse_on && !h->fused_pcg &&
        h->solve_rtol > 0 && st.iter >= (two_level ? 4 * h->coarse_from : h->coarse_from)) {
        if (st.done) h->coarse_start_pending = true;
        else h->fused_coarse_wanted = true;
    }
    return st;
}

// Dense path: scatter S into a dense array, factor it on the matrix cores, solve for the camera
// step.  Returns false when a pivot was not positive (caller falls back to PCG).
// The single-launch factorisation (sfmba_chol.hip) serves systems up to this many unknowns: one
// workgroup per 32 rows, all of them resident at once.
constexpr int kDenseFlowMax = 1536;

// Buffers of the dense solver for the problem at hand (a handle outlives its problems, and the
// next one may have more cameras).
void ensure_dense(H* h) {
    const int need = dense_padded(h->cd * (int)h->n_cam);
    h->dense_n = need;
    h->denseA.alloc((size_t)need * need);
    h->denseLinv.alloc((size_t)need * 32);
    h->denseWork.alloc((size_t)need * 2);
    h->dense_fail.alloc(1);
    h->dense_flow = need <= kDenseFlowMax;
    if (const char* e = getenv("SFMBA_DENSE_FLOW")) h->dense_flow = h->dense_flow && atoi(e) != 0;
    if (h->dense_flow) h->denseFlow.alloc(chol_flow_words(need));
}

bool phase_dense(H* h) {
    Scope t(h, F_DENSE);
    const int n = h->dense_n, n_real = h->cd * (int)h->n_cam;
    int failed = 0;
    if (h->dense_flow) {
        HIP_TRY(launch_chol_flow(h->dev, h->denseA.p, n, n_real, h->rhs.p, h->px.p, h->denseFlow.p, h->dense_fail.p, nullptr, nullptr, h->stream));
        HIP_TRY(hipMemcpyAsync(&failed, h->dense_fail.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (failed != 2) return failed == 0;
        // given up (time limit: the workgroups were not resident together): the chain of launches
        h->dense_flow = false;
    }
    h->denseA.zero(h->stream);
    launch_bsr_to_dense(h->dev, h->denseA.p, n, h->stream);
    h->dense_fail.zero(h->stream);
    launch_dense_cholesky_solve(h->denseA.p, n, n_real, h->denseLinv.p, h->denseWork.p, h->dense_fail.p, h->rhs.p, h->px.p, h->stream);
    HIP_TRY(hipMemcpyAsync(&failed, h->dense_fail.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return failed == 0;
}

// Camera step of the damped normal equations into px; returns the PCG iteration count (0 = dense).
// PCG is the default; the dense Cholesky runs when it was asked for, and as the fallback when PCG
// ran out of iterations on a system small enough to factor (6 n_cameras <= dense_max_dim).
int phase_linear_solve(H* h, double lambda, bool allow_fused = true) {
    if (h->use_dense) {
        if (phase_dense(h)) return 0;
        h->dense_fallbacks++;
        // (chosen by the library: the matrix is not positive definite to working precision at this
        // damping, which the caller raises)
        if (h->dense_auto) return -1;
    }
    const PcgState st = phase_pcg(h, lambda, allow_fused);
    const bool converged = !(st.rz > st.tol2 * st.rz0);
    if (!converged && !h->use_dense && h->dense_allowed && h->explicit_schur) {
        if (h->dense_n != dense_padded(h->cd * (int)h->n_cam)) ensure_dense(h);
        h->dense_fallbacks++;
        if (phase_dense(h)) return st.iter;
        // factorisation failed too: keep the PCG iterate (it is a descent direction)
        (void)phase_pcg(h, lambda, allow_fused);
    }
    return st.iter;
}

// What a rank whose single launch was refused puts into the flag slot: larger than any sum of the
// kernel's own flags (1 breakdown, 2 timeout) over 64 ranks.
constexpr double kPcgRefused = 1024.0;

// Single-launch PCG without waiting for its outcome: its flag goes into a scalar slot that
// travels (and, with a communicator, is summed over the ranks) with the scalars of the trial step,
// its state into pinned host memory; both are looked at after the trial step, which saves a host
// synchronisation and, between ranks, a collective per LM iteration.  Returns false when the solve
// has to take the synchronous route (other solver, launch refused earlier).
bool launch_linear_solve_async(H* h) {
    if (h->explicit_schur && h->use_dense && h->dense_flow && h->dense_auto) {
        // single-launch factorisation: its failure word (1: not positive definite, 2: given up)
        // travels in the same slot
        Scope t(h, F_DENSE);
        const int n = h->dense_n, n_real = h->cd * (int)h->n_cam;
        HIP_TRY(launch_chol_flow(h->dev, h->denseA.p, n, n_real, h->rhs.p, h->px.p, h->denseFlow.p, h->dense_fail.p,
                                 h->scalars.p + SLOT_PCGFLAG, h->has_comm() ? nullptr : h->scalars_host + SLOT_PCGFLAG, h->stream));
        return true;
    }
    if (!(h->explicit_schur && h->fused_pcg && !h->use_dense)) return false;
    if (h->fused_refused && !h->has_comm()) { h->fused_pcg = false; return false; }
    const PcgStop stop = pcg_stop_of(h);
    const double tol = stop.tight;
    FusedPcg& f = h->fused;
    f.x = h->px.p; f.out = &h->pcg_host[0]; f.tol2 = tol * tol; f.max_iter = h->opt.pcg_max_iter > 0 ? h->opt.pcg_max_iter : 1000;
    f.tol2_loose = stop.loose * stop.loose; f.q_abs = stop.q_abs;
    f.flag_slot = h->scalars.p + SLOT_PCGFLAG;
    f.flag_host = h->has_comm() ? nullptr : h->scalars_host + SLOT_PCGFLAG;
    arm_fused_coarse(h, f);
    hipError_t e = hipErrorNotSupported;
    if (!h->fused_refused) {
        Scope t(h, F_BSR);
        e = launch_pcg_fused(f, h->fused_armed, h->stream);
        h->fused_armed = false;
    }
    if (e != hipSuccess) {
        // Refused on this rank.  Which route a solve takes must be the same decision on every rank
        // (the routes issue different collectives), so this rank stays on this one and reports
        // kPcgRefused through the flag slot, which is summed over the ranks with the scalars of the
        // trial step: every rank then redoes this solve launch by launch and gives the single
        // launch up for good (do_solve).
        (void)hipGetLastError();
        if (!h->has_comm()) { h->fused_refused = true; h->fused_pcg = false; return false; }
        h->fused_refused = true;
        const double v = kPcgRefused;
        HIP_TRY(hipMemcpyAsync(h->scalars.p + SLOT_PCGFLAG, &v, sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return true;        // (the kernel writes its state straight into pcg_host[0])
}

// Step evaluation: back-substitution, retraction into the trial buffers, trial cost.
void phase_trial(H* h, double* dp_out) {
    {
        Scope t(h, F_BACK);
        launch_cam_update(h->dev, h->px.p, h->stream);          // the trial cameras first:
        launch_back_pm(h->dev, h->px.p, dp_out, h->stream, h->lin_owed);     // this pass also evaluates the trial cost
        h->lin_owed = false;
    }
    reduce_scalars(h, SLOT_MODEL, SLOT_PCGFLAG - SLOT_MODEL + 1);
}

// Scalars of the linearisation: the slots, and the largest |g_p| over all partitions.  They stay
// in place until the next linearisation, so they can also be read together with the scalars of
// the trial step that follows.
double read_lin_scalars(H* h, double* sc) {
    if (h->lin_owed) { launch_lin_scalars(h->dev, h->stream); h->lin_owed = false; }
    std::vector<double> slots;
    if (h->has_comm()) {
        slots.resize(h->opt.nranks);
        HIP_TRY(hipMemcpyAsync(slots.data(), h->UG.p + (size_t)h->n_cam * h->cv, slots.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    }
    read_scalars(h, sc);
    double m = sc[SLOT_GMAX_PT];
    for (double v : slots) m = std::max(m, v);
    return std::max(sc[SLOT_GMAX_CAM], m);
}

// Progress text goes to the caller's sink when one is set (sfmba_set_print: the Python binding
// routes it to sys.stdout, where SciPy's verbose=2 table goes), otherwise to stdout, flushed line
// by line so that it stays in order with whatever the caller prints itself.
void emit(H* h, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h->print_cb) { h->print_cb(buf, h->print_user); return; }
    fputs(buf, stdout);
    fflush(stdout);
}
void print_header(H* h) {
    emit(h, "%s\n", "   Iteration     Total nfev        Cost      Cost reduction    Step norm     Optimality   ");
}
void print_row(H* h, int it, int nfev, double cost, double red, double step, double opt, bool first) {
    if (first) emit(h, "%8d     %10d      %13.4e                                 %10.2e    \n", it, nfev, cost, opt);
    else emit(h, "%8d     %10d      %13.4e    %10.2e     %10.2e     %10.2e    \n", it, nfev, cost, red, step, opt);
}

void do_solve(H* h, sfmba_stats* out) {
    if (!h->has_problem) fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_solve: no problem set");
    HIP_TRY(hipSetDevice(h->device));
    const sfmba_options& o = h->opt;
    const double ftol = o.ftol, xtol = o.xtol, gtol = o.gtol;
    const int64_t n_par = (h->model == MODEL_BAL ? 9 : 12) * h->n_cam + 3 * h->n_pts_all;
    const int64_t max_nfev = o.max_nfev > 0 ? o.max_nfev : 100 * n_par;
    const int max_it = o.max_iterations > 0 ? o.max_iterations : 200;
    // Damping: 1e-4 of the scaling to begin with where the camera system is solved by PCG; where it is
    // factored (exact steps) it starts at its floor, i.e. with Gauss-Newton steps, and a step that
    // fails sends it to the 1e-4 of the other case.
    const bool exact_steps = h->use_dense && h->dense_auto;
    const double lambda_floor = h->use_dense ? 1e-12 : 1e-15;
    double lambda = o.initial_lambda > 0 ? o.initial_lambda : exact_steps ? 1e-10 : 1e-4;
    double nu = 2.0;
    sfmba_stats st{};
    st.linear_solver = h->use_dense ? SFMBA_SOLVER_DENSE : SFMBA_SOLVER_PCG;
    h->dense_fallbacks = 0;
    h->fused_solves = h->fused_fallbacks = 0;
    h->fused_prev_iter = 0;
    for (int f = 0; f < F_COUNT; ++f) { h->fam_ms[f] = 0; h->fam_n[f] = 0; }
    h->coarse_on = h->coarse_forced; h->coarse_failed = false; h->coarse_solves = 0;
    h->fused_coarse_ready = false; h->fused_coarse_wanted = false; h->coarse_start_pending = false;
    h->solve_rtol = 0.0;
    h->solve_rtol = pcg_rtol_of(h);
    struct ClearRtol { H* h; ~ClearRtol() { h->solve_rtol = 0.0; h->model_gain_floor = -1.0; } } clear_rtol{h};
    // largest cut of the damping after a step: a step the model predicted to 1 % cuts it thirtyfold
    // (Nielsen's rule with a floor of 1/30 instead of 1/3; tuning: SFMBA_LM_DOWN)
    // - where the linear solver copes with little damping: the factorisation, or PCG with the coarse
    // correction at hand; block-Jacobi PCG alone pays for every cut with more iterations (measured with
    // the 9-number camera at the Venice shape: twice the PCG iterations, no outer iteration saved)
    // ... or PCG on the explicit matrix of fewer than 256 cameras, whose iterations cost microseconds: there
    // the slow cut left the 9-number camera above SciPy's final cost in 5 of 14 seeded problems of 40 to
    // 130 cameras (by up to 4e-5), the fast one in none, with fewer iterations of either kind
    const bool low_damping_ok = h->use_dense || (h->explicit_schur && (h->n_cam < 256 || (h->fused_pcg ? h->fused_coarse_ok : h->coarse_built)));
    // (the camera with its own intrinsics is the less linear model: a tenth there - on starts ten times
    // as far off as usual the thirtyfold cut took 13 iterations and stopped 3e-4 above the converged
    // cost, the tenfold one 6 iterations to within 1e-6, and both take 3 on ordinary starts)
    double lm_down = low_damping_ok ? (h->model == MODEL_BAL ? 1.0 / 10.0 : 1.0 / 30.0) : 1.0 / 3.0;
    if (const char* e = getenv("SFMBA_LM_DOWN")) lm_down = atof(e);
    double gain_fraction = 0.05;             // of ftol x cost, per PCG iteration (tuning: SFMBA_PCG_GAIN, negative = rule off)
    if (const char* e = getenv("SFMBA_PCG_GAIN")) gain_fraction = atof(e);
    // tuning hook: SFMBA_RTOL_SCHEDULE="0.03,0.001" gives the PCG tolerance of the first, second, ...
    // linear solve (the last entry for all later ones) where the options leave it open
    std::vector<double> rtol_schedule;
    if (const char* e = getenv("SFMBA_RTOL_SCHEDULE"))
        for (const char* q = e; *q;) {
            char* end = nullptr;
            const double v = strtod(q, &end);
            if (end == q) break;
            if (v > 0) rtol_schedule.push_back(v);
            q = *end == ',' ? end + 1 : end;
        }
    // test hook: SFMBA_CHOL_TEST_FAIL="k:v" makes the k-th single-launch factorisation of a solve
    // count as failed on every rank - v = 1 not positive definite, v = 2 given up - so that the two
    // recoveries below run without a matrix that provokes them
    int chol_fail_at = -1, chol_fail_kind = 0, chol_launches = 0;
    if (const char* e = getenv("SFMBA_CHOL_TEST_FAIL")) {
        if (sscanf(e, "%d:%d", &chol_fail_at, &chol_fail_kind) != 2) chol_fail_at = -1;
    }
    h->profile = o.profile != 0;
    if (const char* e = getenv("SFMBA_PROFILE")) h->profile = atoi(e) != 0;
    auto t0 = std::chrono::steady_clock::now();
    double sc[SLOT_COUNT];

    phase_linearize(h, lambda);
    st.njev = 1;
    reduce_scalars(h, SLOT_COST, 1);
    double g_norm = read_lin_scalars(h, sc);
    double cost = 0.5 * sc[SLOT_COST];
    st.nfev = 1;
    st.initial_cost = cost;
    if (!std::isfinite(cost)) fail(SFMBA_ERR_NONFINITE, "Residuals are not finite in the initial point.");
    int status = -1;
    double step_norm = 0.0, actual = 0.0;
    const bool talk = o.verbose >= 2 && o.rank <= 0;
    if (talk) { print_header(h); print_row(h, 0, st.nfev, cost, actual, step_norm, g_norm, true); }
    int accepted = 0, held_back_steps = 0;
    double lambda_rejected = 0.0;
    if (g_norm < gtol) status = SFMBA_STATUS_GTOL;
    // Unless every iteration is printed, the gradient norm of a new linearisation is not waited
    // for: it is read with the scalars of the next trial step (one host synchronisation less per
    // iteration), and a step computed from a point that turns out to satisfy gtol is dropped.
    bool lin_pending = false;
    for (int it = 0; it < max_it && status < 0; ++it) {
        const int pcg_before = st.pcg_iterations;
        const int64_t coarse_before = h->coarse_solves;
        if (!rtol_schedule.empty() && !(o.pcg_rtol > 0))
            h->solve_rtol = rtol_schedule[std::min<size_t>(st.iterations, rtol_schedule.size() - 1)];
        h->model_gain_floor = gain_fraction >= 0.0 ? gain_fraction * ftol * cost : -1.0;
        phase_damp(h, lambda);
        const bool pending_dense = h->use_dense && h->dense_flow && h->dense_auto;
        const bool pending = launch_linear_solve_async(h);
        if (!pending) {
            const int its = phase_linear_solve(h, lambda);
            if (its < 0) {
                lambda = std::max(16.0 * lambda, 1e-8);
                if (lambda > 1e32) status = SFMBA_STATUS_XTOL;
                continue;
            }
            st.pcg_iterations += its;
            HIP_TRY(hipMemsetAsync(h->scalars.p + SLOT_PCGFLAG, 0, sizeof(double), h->stream));
        }
        phase_trial(h, nullptr);
        if (h->coarse_start_pending) {
            // (queued behind the trial step: the thirty launches of the set-up take the host 0.4 ms,
            // which the trial step's kernel covers)
            h->coarse_start_pending = false;
            if (getenv("SFMBA_COARSE_START_LATE")) h->fused_coarse_wanted = true;
            else start_fused_coarse(h);
        }
        if (lin_pending) {
            g_norm = read_lin_scalars(h, sc);
            lin_pending = false;
            if (g_norm < gtol) { status = SFMBA_STATUS_GTOL; break; }
        } else {
            read_scalars(h, sc);
        }
        if (pending && pending_dense) {
            // the factorisation's failure word, summed over the ranks (which all factor the same matrix)
            const double all_ranks = h->has_comm() ? (double)o.nranks : 1.0;
            double flag = sc[SLOT_PCGFLAG];
            if (chol_launches++ == chol_fail_at) flag = chol_fail_kind == 1 ? all_ranks : all_ranks + 1.0;
            if (flag != 0.0) {
                if (flag != all_ranks) {
                    // given up somewhere (time limit): every rank takes the chain of launches from now on
                    h->dense_flow = false;
                    const int its = phase_linear_solve(h, lambda, false);
                    if (its >= 0) {
                        HIP_TRY(hipMemsetAsync(h->scalars.p + SLOT_PCGFLAG, 0, sizeof(double), h->stream));
                        phase_trial(h, nullptr);
                        read_scalars(h, sc);
                    }
                    if (its < 0) { lambda = std::max(16.0 * lambda, 1e-8); if (lambda > 1e32) status = SFMBA_STATUS_XTOL; continue; }
                } else {
                    // not positive definite to working precision at this damping
                    h->dense_fallbacks++;
                    lambda = std::max(16.0 * lambda, 1e-8);
                    if (lambda > 1e32) status = SFMBA_STATUS_XTOL;
                    continue;
                }
            }
        } else if (pending) {
            // identical on every rank: the flag is the ranks' sum, the state that of the same
            // replicated solve
            const PcgState ps = h->pcg_host[0];
            if (h->trace_pcg)
                fprintf(stderr, "pcg single launch: %d iterations, %.2f us in kernel, flag %d (all ranks %g)\n", ps.iter, ps.ticks * 0.01, ps.pad, sc[SLOT_PCGFLAG]);
            const bool failed = sc[SLOT_PCGFLAG] != 0.0;
            if (sc[SLOT_PCGFLAG] >= kPcgRefused) h->fused_pcg = false;     // on every rank alike
            const bool converged = !(ps.rz > ps.tol2 * ps.rz0);
            if (!failed && (converged || !h->dense_allowed)) {
                h->fused_solves++;
                if (h->fused.coarse) h->coarse_solves++;
                st.pcg_iterations += ps.iter;
                // a long solve: the ones after it get the coarse correction, set up from this
                // solve's matrix while the next linearisation runs.  The operator stays in use for
                // the rest of the run (the reduced matrix of the later iterations differs little
                // from this one: measured, the operator of the iteration before preconditions as
                // well as the current one) and is renewed when a solve with it grows long again.
                // A launch with the correction costs about 40 us before its first iteration and 5
                // instead of 3 us per iteration for a third to a quarter of the iterations: it pays
                // from about 40 block-Jacobi iterations, or sooner where the solves of a run grow
                // (they triple from one to the next while the damping falls; a chain of views with
                // two-view tracks stays at 20 to 30 and is left alone).
                if (h->fused_coarse_ok && h->fused_pcg) {
                    const bool growing = ps.iter >= 2.5 * std::max(h->fused_prev_iter, 1);
                    if (h->fused.coarse ? ps.iter >= 4 * h->fused_coarse_from
                                        : ps.iter >= h->fused_coarse_from && (growing || ps.iter >= 40))
                        h->fused_coarse_wanted = true;
                }
                h->fused_prev_iter = ps.iter;
            } else {
                // breakdown, timeout, or out of iterations with the factorisation at hand: solve
                // again launch by launch (which knows the dense fallback) and redo the trial step
                if (failed) h->fused_fallbacks++;
                st.pcg_iterations += phase_linear_solve(h, lambda, false);
                HIP_TRY(hipMemsetAsync(h->scalars.p + SLOT_PCGFLAG, 0, sizeof(double), h->stream));
                phase_trial(h, nullptr);
                read_scalars(h, sc);
            }
        }
        st.nfev++;
        st.iterations++;
        const double model_cost = 0.5 * sc[SLOT_MODEL];
        const double cost_new = sc[SLOT_NONFINITE] > 0 ? std::numeric_limits<double>::infinity() : 0.5 * sc[SLOT_TRIAL];
        const double pred = cost - model_cost;
        actual = cost - cost_new;
        const double rho = (pred > 0 && std::isfinite(cost_new)) ? actual / pred : -1.0;
        step_norm = std::sqrt(sc[SLOT_STEP2_PT] + sc[SLOT_STEP2_CAM]);
        const double x_norm = std::sqrt(sc[SLOT_X2_PT] + sc[SLOT_X2_CAM]);
        // A step taken at a damping that a rejection has just raised is short because it was held
        // back, not because the iterate has converged: its small reduction does not end the run (SciPy's
        // trust region shrinks by quarters; the damping here may jump a hundredfold).  At most four
        // such steps in a row are passed over.
        const bool held_back = exact_steps && nu > 2.0 && held_back_steps < 4;
        const bool ft = actual < ftol * cost && rho > 0.25 && !held_back;
        const bool xt = step_norm < xtol * (xtol + x_norm);
        if (o.verbose >= 3 && o.rank <= 0)
            emit(h, "  lm %3d cost %.9e -> %.9e rho %+.3f lambda %.2e |dx| %.2e pcg %d%s\n", it, cost, cost_new, rho, lambda, step_norm,
                 st.pcg_iterations - pcg_before, h->coarse_solves > coarse_before ? " (two-level)" : "");
        if (rho > 1e-3) {
            if (h->at_start) { h->at_start = false; h->cur = 0; } else h->cur ^= 1;
            ++accepted;
            held_back_steps = nu > 2.0 ? held_back_steps + 1 : 0;
            {
                // (after a rejection the damping is not cut back to where that step failed: it stays
                // above three times the rejected value, a bound that halves with every accepted step;
                // without it a well-predicted step at 1e-5 sent the damping to 1e-6, where the next step
                // failed, up again, and round in a cycle of three)
                double next = lambda * std::max(lm_down, 1.0 - std::pow(2.0 * rho - 1.0, 3));
                if (lambda_rejected > 0.0) next = std::max(next, std::min(lambda, 3.0 * lambda_rejected));
                lambda_rejected *= 0.5;
                lambda = std::max(next, lambda_floor);
            }
            nu = 2.0;
            cost = cost_new;
            // also the gradient norm at the new iterate; damps for the next step, unless this was the last
            phase_linearize(h, lambda, ft || xt, /*defer=*/!talk && !(ft || xt));
            st.njev++;
            lin_pending = !talk;
            if (!lin_pending) g_norm = read_lin_scalars(h, sc);
            if (talk) print_row(h, accepted, st.nfev, cost, actual, step_norm, g_norm, false);
            if (ft && xt) status = SFMBA_STATUS_FTOL_XTOL;
            else if (ft) status = SFMBA_STATUS_FTOL;
            else if (xt) status = SFMBA_STATUS_XTOL;
            else if (!lin_pending && g_norm < gtol) status = SFMBA_STATUS_GTOL;
        } else {
            lambda_rejected = lambda;
            // (exact steps start at the damping floor: a failure there goes straight to the 1e-4 of the
            // other case instead of doubling twenty times)
            lambda = exact_steps && lambda < 1e-8 ? 1e-4 : lambda * std::max(nu, exact_steps ? 4.0 : 2.0);
            nu *= 2.0;
            if (xt || lambda > 1e32) status = SFMBA_STATUS_XTOL;
        }
        if (status < 0 && st.nfev >= max_nfev) status = SFMBA_STATUS_MAX_NFEV;
    }
    if (status < 0) status = SFMBA_STATUS_MAX_NFEV;
    if (lin_pending) {
        g_norm = read_lin_scalars(h, sc);
        if (status == SFMBA_STATUS_MAX_NFEV && g_norm < gtol) status = SFMBA_STATUS_GTOL;
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    // (a coarse set-up started beside the last linearisation of a run that then ran out of
    // iterations may still be at work on buffers the next call reuses)
    if (h->fused_coarse_ready) HIP_TRY(hipStreamSynchronize(h->copy_stream));
    auto t1 = std::chrono::steady_clock::now();
    if (h->profile) collect_spans(h);
    st.status = status;
    st.accepted = accepted;
    st.final_cost = cost;
    st.optimality = g_norm;
    st.solve_seconds = std::chrono::duration<double>(t1 - t0).count();
    st.ms_linearize = h->fam_ms[F_LIN]; st.n_linearize = h->fam_n[F_LIN];
    st.ms_damp = h->fam_ms[F_DAMP]; st.n_damp = h->fam_n[F_DAMP];
    st.ms_spmv_pm = h->fam_ms[F_SPMV_PM]; st.n_spmv_pm = h->fam_n[F_SPMV_PM];
    st.ms_spmv_cm = h->fam_ms[F_SPMV_CM]; st.n_spmv_cm = h->fam_n[F_SPMV_CM];
    st.ms_pcg_vec = h->fam_ms[F_PCGVEC]; st.n_pcg_vec = h->fam_n[F_PCGVEC];
    st.ms_back = h->fam_ms[F_BACK]; st.n_back = h->fam_n[F_BACK];
    st.ms_cost = h->fam_ms[F_COST]; st.n_cost = h->fam_n[F_COST];
    st.ms_dense = h->fam_ms[F_DENSE]; st.n_dense = h->fam_n[F_DENSE];
    st.ms_schur = h->fam_ms[F_SCHUR]; st.n_schur = h->fam_n[F_SCHUR];
    st.ms_bsr = h->fam_ms[F_BSR]; st.n_bsr = h->fam_n[F_BSR];
    st.schur_mode = h->explicit_schur ? 2 : 1;
    st.schur_pairs = h->explicit_schur ? h->n_pairs : 0;
    st.schur_blocks = h->explicit_schur ? h->nnzb : 0;
    st.schur_chunks = h->explicit_schur ? h->n_pchunks : 0;
    st.ms_pairs = h->fam_ms[F_PAIRS]; st.n_pairs = h->fam_n[F_PAIRS];
    st.ms_comm = h->fam_ms[F_COMM]; st.n_comm = h->fam_n[F_COMM];
    st.pcg_pipelined = h->fused_pcg && h->fused.pipelined ? 1 : 0;
    st.pcg_single_launch = h->fused_solves; st.pcg_single_fallbacks = h->fused_fallbacks;
    st.pcg_coarse_solves = h->coarse_solves;
    st.pcg_coarse_dim = h->explicit_schur && h->coarse_built ? h->coarse.n : 0;
    if (o.verbose >= 1 && o.rank <= 0) {
        static const char* msg[] = {"The maximum number of function evaluations is exceeded.",
                                    "`gtol` termination condition is satisfied.",
                                    "`ftol` termination condition is satisfied.",
                                    "`xtol` termination condition is satisfied.",
                                    "Both `ftol` and `xtol` termination conditions are satisfied."};
        emit(h, "%s\nFunction evaluations %d, initial cost %.4e, final cost %.4e, first-order optimality %.2e.\n",
             msg[status], st.nfev, st.initial_cost, st.final_cost, st.optimality);
    }
    if (out) *out = st;
}

// ---- problem packing ----------------------------------------------------------------------
// Rotation matrix of a rotation vector by Rodrigues' formula, as /root/reference/ba_test.py:62-75
// applies it to a point: a zero vector is the identity.
void rotation_of_vector(const double* r, double* Q) {
    const double a = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    double k[3] = {0.0, 0.0, 0.0};
    if (a > 0.0) for (int i = 0; i < 3; ++i) k[i] = r[i] / a;
    const double c = std::cos(a), s = std::sin(a), v = 1.0 - c;
    Q[0] = c + v * k[0] * k[0];         Q[1] = v * k[0] * k[1] - s * k[2];  Q[2] = v * k[0] * k[2] + s * k[1];
    Q[3] = v * k[1] * k[0] + s * k[2];  Q[4] = c + v * k[1] * k[1];         Q[5] = v * k[1] * k[2] - s * k[0];
    Q[6] = v * k[2] * k[0] - s * k[1];  Q[7] = v * k[2] * k[1] + s * k[0];  Q[8] = c + v * k[2] * k[2];
}

// Rotation vector (angle in [0, pi]) of a rotation matrix.  Away from pi the axis is the skew part;
// near pi, where that vanishes, it comes from the symmetric part (k k' = (sym(Q) - c I) / (1 - c))
// with the sign the skew part indicates.
void vector_of_rotation(const double* Q, double* r) {
    const double sx = 0.5 * (Q[7] - Q[5]), sy = 0.5 * (Q[2] - Q[6]), sz = 0.5 * (Q[3] - Q[1]);
    const double s = std::sqrt(sx * sx + sy * sy + sz * sz);
    const double c = 0.5 * (Q[0] + Q[4] + Q[8] - 1.0);
    const double a = std::atan2(s, c);
    if (a < 1e-7) { r[0] = sx; r[1] = sy; r[2] = sz; return; }
    if (c > -0.5) { const double f = a / s; r[0] = sx * f; r[1] = sy * f; r[2] = sz * f; return; }
    double S[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) S[i][j] = (0.5 * (Q[3 * i + j] + Q[3 * j + i]) - (i == j ? c : 0.0)) / (1.0 - c);
    int j = 0;
    if (S[1][1] > S[j][j]) j = 1;
    if (S[2][2] > S[j][j]) j = 2;
    const double n = std::sqrt(S[j][j]);
    double k[3] = {S[0][j] / n, S[1][j] / n, S[2][j] / n};
    if (k[0] * sx + k[1] * sy + k[2] * sz < 0.0) for (int i = 0; i < 3; ++i) k[i] = -k[i];
    for (int i = 0; i < 3; ++i) r[i] = k[i] * a;
}

void pack_cameras(H* h, const double* camera_params, std::vector<double2>& out) {
    if (h->model == MODEL_BAL) {
        out.resize((size_t)h->n_cam * 8);
        for (int64_t c = 0; c < h->n_cam; ++c) {
            const double* cp = camera_params + 9 * c;
            for (int i = 0; i < 9; ++i)
                if (!std::isfinite(cp[i])) fail(SFMBA_ERR_NONFINITE, "camera_params[%lld] is not finite", (long long)c);
            double Q[9];
            rotation_of_vector(cp, Q);
            const double v[16] = {Q[0], Q[1], Q[2], Q[3], Q[4], Q[5], Q[6], Q[7], Q[8], cp[3], cp[4], cp[5], cp[6], cp[7], cp[8], 0.0};
            for (int i = 0; i < 8; ++i) out[8 * c + i] = make_double2(v[2 * i], v[2 * i + 1]);
        }
        return;
    }
    out.resize((size_t)h->n_cam * 6);
    for (int64_t c = 0; c < h->n_cam; ++c) {
        double Q[9];
        const double* cp = camera_params + 12 * c;
        for (int i = 0; i < 12; ++i)
            if (!std::isfinite(cp[i])) fail(SFMBA_ERR_NONFINITE, "camera_params[%lld] is not finite", (long long)c);
        polar3(cp, Q);
        double v[12] = {Q[0], Q[1], Q[2], Q[3], Q[4], Q[5], Q[6], Q[7], Q[8], cp[9], cp[10], cp[11]};
        for (int i = 0; i < 6; ++i) out[6 * c + i] = make_double2(v[2 * i], v[2 * i + 1]);
    }
}

// `points_on_device`: in_pts already holds ALL points in the caller's numbering (sfmba_set_problem
// uploaded them beside the set-up kernels); otherwise the owned range is uploaded here.
void do_set_params(H* h, const double* camera_params, const double* points_3d, bool points_on_device = false) {
    HIP_TRY(hipSetDevice(h->device));
    std::vector<double2> cams;
    pack_cameras(h, camera_params, cams);
    // the owned points as the caller numbers them, permuted into the internal order on the device
    const size_t np = (size_t)h->n_pts;
    if (!points_on_device) h->in_pts.alloc(3 * np);
    h->cur = 0;
    h->at_start = false;
    h->cams[0].upload(cams, h->stream);
    h->cams[1].alloc(cams.size());
    h->X[0].alloc(np); h->X[1].alloc(np); h->X0.alloc(np);
    h->cams0.alloc(cams.size());
    if (np) {
        if (!points_on_device)
            HIP_TRY(hipMemcpyAsync(h->in_pts.p, points_3d + 3 * h->pt_lo, 3 * np * sizeof(double), hipMemcpyHostToDevice, h->stream));
        launch_points_in(h->in_pts.p + (points_on_device ? 3 * (size_t)h->pt_lo : 0), h->new2old_dev.p, (int)np, h->X[0].p, h->stream);
        HIP_TRY(hipMemcpyAsync(h->X[1].p, h->X[0].p, np * sizeof(double4), hipMemcpyDeviceToDevice, h->stream));   // points without observations are never rewritten
        HIP_TRY(hipMemcpyAsync(h->X0.p, h->X[0].p, np * sizeof(double4), hipMemcpyDeviceToDevice, h->stream));
    }
    HIP_TRY(hipMemcpyAsync(h->cams0.p, h->cams[0].p, cams.size() * sizeof(double2), hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->linearized = false;
    refresh_dev(h);
}


int bsr_blocks_host(int n_cam, int model) { Dev d{}; d.n_cam = n_cam; d.model = model; return bsr_blocks(d); }

// Open-addressing set of 64-bit keys (camera pairs), with a small direct-mapped filter in
// front: consecutive points mostly repeat the pairs of their neighbours.
struct KeySet {
    std::vector<uint64_t> tab, filt;
    size_t mask, count = 0;
    static constexpr uint64_t kEmpty = ~0ull;
    KeySet() : tab(1u << 16, kEmpty), filt(1u << 14, kEmpty), mask((1u << 16) - 1) {}
    static uint64_t mix(uint64_t k) { k *= 0x9E3779B97F4A7C15ull; return k ^ (k >> 29); }
    void grow() {
        std::vector<uint64_t> old;
        old.swap(tab);
        tab.assign(old.size() * 4, kEmpty);
        mask = tab.size() - 1;
        count = 0;
        for (uint64_t k : old) if (k != kEmpty) put(k);
    }
    void put(uint64_t k) {
        size_t i = mix(k) & mask;
        while (tab[i] != kEmpty) { if (tab[i] == k) return; i = (i + 1) & mask; }
        tab[i] = k;
        if (++count * 2 > tab.size()) grow();
    }
    void insert(uint64_t k) {
        uint64_t& f = filt[mix(k) & (filt.size() - 1)];
        if (f == k) return;
        f = k;
        put(k);
    }
};

// Structure of the explicit reduced matrix: upper blocks = camera pairs sharing a point (of ANY
// rank, so that all ranks agree on the layout) plus every diagonal; per upper block the list of
// this partition's observation pairs; the full block-CSR pattern and where each upper block and
// its transpose land in it.
}  // namespace

// What the options and the environment ask of the reduced system.
SchurPlan sfmba::plan_schur(H* h) {
    int mode = h->opt.schur_mode;
    if (const char* e = getenv("SFMBA_SCHUR_MODE")) mode = atoi(e);
    int lin = h->opt.linear_solver;
    if (const char* e = getenv("SFMBA_LINEAR_SOLVER")) lin = atoi(e);
    const int dense_max = h->opt.dense_max_dim > 0 ? h->opt.dense_max_dim : 1536;
    const int64_t n_red = (int64_t)h->cd * h->n_cam;       // unknowns of the reduced system
    // Left to choose, a small camera system is factored (one launch, sfmba_chol.hip) rather than
    // iterated on: the reference's own workloads (11 views, a ring of 47) have 66 to 282 unknowns,
    // an exact step lets the damping fall to its floor, and the Gauss-Newton steps that follow are
    // what SciPy's trust-region solver takes on them (scipy/optimize/_lsq/trf.py:473-499).
    int auto_dense_dim = 288;          // up to 48 six-number cameras: nine block rows of the single-launch factorisation
    // Two-view tracks - what the reference's own correspondence table holds, two observations per
    // point (WorldPoints.py:40-43) - leave the scale of every camera pair free: the cost has a flat
    // valley along which damped, inexact steps crawl (on seeded chains of 49 to 200 cameras PCG ended
    // up to 4e-2 above SciPy's cost after 4 to 6 times SciPy's evaluations).  Such problems are
    // factored as far as the single-launch factorisation reaches.
    if (h->n_obs_all <= (int64_t)(2.5 * (double)h->n_pts_all)) auto_dense_dim = std::min(dense_max, 1536);
    if (const char* e = getenv("SFMBA_DENSE_AUTO_DIM")) auto_dense_dim = atoi(e);
    h->dense_auto = lin == SFMBA_SOLVER_AUTO && mode != 1 && dense_padded((int)std::min<int64_t>(n_red, 1 << 20)) <= auto_dense_dim;
    h->dense_allowed = lin == SFMBA_SOLVER_AUTO && n_red <= dense_max;
    if (n_red > 16384 && lin == SFMBA_SOLVER_DENSE)
        fail(SFMBA_ERR_INVALID_ARGUMENT, "dense solver: reduced system of %lld unknowns is too large", (long long)n_red);
    h->use_dense = false;
    // A partitioned solve keeps the reduced matrix replicated (all upper blocks summed over the
    // ranks, PCG the same on every rank) or split by points (every rank the blocks of its own
    // points, the product summed over the ranks per PCG iteration): options.matrix_split.
    int split = h->opt.matrix_split;
    if (const char* e = getenv("SFMBA_MATRIX_SPLIT")) split = atoi(e);
    h->split = h->opt.nranks > 1 && mode != 1 && lin != SFMBA_SOLVER_DENSE && split == 2;
    if (h->split) { h->dense_auto = false; h->dense_allowed = false; }
    const bool want_dense = lin == SFMBA_SOLVER_DENSE || h->dense_auto;
    if (want_dense) mode = 2;          // the dense matrix is filled from the explicit blocks
    h->explicit_schur = false;
    h->n_ublk = h->n_pchunks = h->nnzb = 0;
    h->n_pairs = 0;
    return {mode, want_dense};
}

namespace {

void build_schur_structure(H* h, int64_t n_obs_all, const int64_t* cam_idx, const int64_t* pt_idx,
                           const std::vector<int64_t>& pre, const std::vector<int>& pt_ptr,
                           const std::vector<int>& pm_cam) {
    const int n_cam = (int)h->n_cam, np = h->n_pts, no = h->n_obs;
    SetupTrace tr;
    const SchurPlan plan = plan_schur(h);
    const int mode = plan.mode;
    const bool want_dense = plan.want_dense;
    if (mode == 1) return;
    // pairs of this partition
    int64_t n_pairs = 0;
    {
        int ntc = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
        if (const char* e = getenv("SFMBA_THREADS")) ntc = std::max(1, atoi(e));
        if (no < 200000) ntc = 1;
        std::vector<int64_t> part((size_t)ntc, 0);
        auto count = [&](int t) {
            int64_t n = 0;
            for (int p = (int)((int64_t)np * t / ntc); p < (int)((int64_t)np * (t + 1) / ntc); ++p) {
                const int64_t k = pt_ptr[p + 1] - pt_ptr[p];      // (an observation with itself is not a pair: the
                n += k * (k - 1) / 2;                               // diagonal blocks come from a camera-major pass)
                for (int a = pt_ptr[p]; a < pt_ptr[p + 1]; ++a)     // same-camera duplicates inside a point add one entry each
                    for (int b = a + 1; b < pt_ptr[p + 1]; ++b)
                        if (pm_cam[a] == pm_cam[b]) ++n;
            }
            part[t] = n;
        };
        std::vector<std::thread> th;
        for (int t = 1; t < ntc; ++t) th.emplace_back(count, t);
        count(0);
        for (auto& x : th) x.join();
        for (int64_t n : part) n_pairs += n;
    }
    if (mode == 0 && (n_pairs > 24 * (int64_t)std::max(no, 1) || n_pairs > 1500000000LL)) return;
    if (n_pairs > 2000000000LL) fail(SFMBA_ERR_INVALID_ARGUMENT, "explicit Schur complement: too many observation pairs");

    tr.lap("schur: count pairs");
    // distinct camera pairs over all points of all ranks
    KeySet keys;
    auto key_of = [](int r, int c) { return ((uint64_t)(uint32_t)r << 32) | (uint32_t)c; };
    if (h->n_obs == h->n_obs_all || h->split) {      // (split by points: the blocks of this rank's own points)
        for (int p = 0; p < np; ++p)
            for (int a = pt_ptr[p]; a < pt_ptr[p + 1]; ++a)
                for (int b = a + 1; b < pt_ptr[p + 1]; ++b) {
                    const int ca = pm_cam[a], cb = pm_cam[b];
                    if (ca != cb) keys.insert(key_of(std::min(ca, cb), std::max(ca, cb)));
                }
    } else {
        std::vector<int> cams_by_pt((size_t)n_obs_all);
        std::vector<int64_t> fillp(pre.begin(), pre.end() - 1);
        for (int64_t i = 0; i < n_obs_all; ++i) cams_by_pt[fillp[pt_idx[i]]++] = (int)cam_idx[i];
        for (int64_t p = 0; p < h->n_pts_all; ++p)
            for (int64_t a = pre[p]; a < pre[p + 1]; ++a)
                for (int64_t b = a + 1; b < pre[p + 1]; ++b) {
                    const int ca = cams_by_pt[a], cb = cams_by_pt[b];
                    if (ca != cb) keys.insert(key_of(std::min(ca, cb), std::max(ca, cb)));
                }
    }
    tr.lap("schur: camera-pair set");
    std::vector<uint64_t> ukeys;
    ukeys.reserve(keys.count + n_cam);
    for (uint64_t k : keys.tab) if (k != KeySet::kEmpty) ukeys.push_back(k);
    for (int c = 0; c < n_cam; ++c) ukeys.push_back(key_of(c, c));
    std::sort(ukeys.begin(), ukeys.end());
    const int n_ublk = (int)ukeys.size();
    if (mode == 0 && (int64_t)n_ublk * 2 * 288 > (8LL << 30)) return;
    std::vector<int> ur(n_ublk), uc(n_ublk), urow_ptr((size_t)n_cam + 1, 0);
    for (int u = 0; u < n_ublk; ++u) { ur[u] = (int)(ukeys[u] >> 32); uc[u] = (int)(ukeys[u] & 0xffffffffu); urow_ptr[ur[u] + 1]++; }
    for (int c = 0; c < n_cam; ++c) urow_ptr[c + 1] += urow_ptr[c];
    auto find_blk = [&](int r, int c) {
        const int* b = uc.data() + urow_ptr[r];
        const int* e = uc.data() + urow_ptr[r + 1];
        return (int)(std::lower_bound(b, e, c) - uc.data());
    };
    tr.lap("schur: sort block keys");
    // Count, then fill, the pair list of each upper block (points in increasing order), on several
    // host threads: thread t owns a contiguous range of points with an equal share of the pairs,
    // counts its pairs per block, and after a prefix sum over (block, thread) fills its own
    // disjoint slice of every block's list, so the result does not depend on the thread count.
    // Block lookups go through a small per-thread direct-mapped cache (consecutive points mostly
    // repeat the camera pairs of their neighbours) before the binary search in the block row.
    std::vector<int> bcount((size_t)n_ublk + 1, 0);
    int nt = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* e = getenv("SFMBA_THREADS")) nt = std::max(1, atoi(e));
    if (n_pairs < 200000) nt = 1;
    std::vector<int> pbeg((size_t)nt + 1, np);
    {
        pbeg[0] = 0;
        int64_t acc = 0;
        int t = 1;
        for (int p = 0; p < np && t < nt; ++p) {
            const int64_t k = pt_ptr[p + 1] - pt_ptr[p];
            acc += k * (k - 1) / 2;
            if (acc * nt >= (int64_t)t * n_pairs) pbeg[t++] = p + 1;
        }
    }
    struct BlkCache {
        std::vector<uint64_t> key; std::vector<int> val;
        BlkCache() : key(4096, ~0ull), val(4096, 0) {}
    };
    auto for_each_pair = [&](int p0, int p1, BlkCache& bc, auto&& emit) {
        auto blk = [&](int r, int c) {
            const uint64_t k = ((uint64_t)(uint32_t)r << 32) | (uint32_t)c;
            const size_t slot = (size_t)((k * 0x9E3779B97F4A7C15ull) >> 52);
            if (bc.key[slot] != k) { bc.key[slot] = k; bc.val[slot] = find_blk(r, c); }
            return bc.val[slot];
        };
        for (int p = p0; p < p1; ++p)
            for (int a = pt_ptr[p]; a < pt_ptr[p + 1]; ++a) {
                const int ca = pm_cam[a];
                for (int b = a + 1; b < pt_ptr[p + 1]; ++b) {
                    const int cb = pm_cam[b];
                    if (ca == cb) { const int u = blk(ca, ca); emit(u, a, b, p); emit(u, b, a, p); }
                    else if (ca < cb) emit(blk(ca, cb), a, b, p);
                    else emit(blk(cb, ca), b, a, p);
                }
            }
    };
    std::vector<std::vector<int>> tcount((size_t)nt);
    auto run_threads = [&](auto&& body) {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(body, t);
        body(0);
        for (auto& x : th) x.join();
    };
    run_threads([&](int t) {
        BlkCache bc;
        std::vector<int>& cnt = tcount[t];
        cnt.assign((size_t)n_ublk, 0);
        for_each_pair(pbeg[t], pbeg[t + 1], bc, [&](int u, int, int, int) { cnt[u]++; });
    });
    for (int u = 0; u < n_ublk; ++u) {
        int tot = 0;
        for (int t = 0; t < nt; ++t) { const int c = tcount[t][u]; tcount[t][u] = tot; tot += c; }   // offset inside the block
        bcount[u + 1] = bcount[u] + tot;
    }
    std::vector<int> pp((size_t)n_pairs);
    run_threads([&](int t) {
        BlkCache bc;
        std::vector<int>& off = tcount[t];
        for_each_pair(pbeg[t], pbeg[t + 1], bc, [&](int u, int, int, int p) { pp[bcount[u] + off[u]++] = p; });
    });
    // (a pair is listed by its point alone: the pair kernel rebuilds both observations from it)
    tr.lap("schur: pair lists");
    // pair chunks
    int pchunk = 512;
    if (const char* e = getenv("SFMBA_PAIR_CHUNK")) { int v = atoi(e); if (v >= 64) pchunk = v; }
    std::vector<int> pchunk_blk, pchunk_beg, ublk_chunk_ptr((size_t)n_ublk + 1, 0);
    for (int u = 0; u < n_ublk; ++u) {
        ublk_chunk_ptr[u] = (int)pchunk_blk.size();
        for (int b = bcount[u]; b < bcount[u + 1]; b += pchunk) { pchunk_blk.push_back(u); pchunk_beg.push_back(b); }
    }
    ublk_chunk_ptr[n_ublk] = (int)pchunk_blk.size();
    pchunk_beg.push_back((int)n_pairs);
    // full block-CSR pattern
    std::vector<int> bsr_ptr((size_t)n_cam + 1, 0);
    for (int u = 0; u < n_ublk; ++u) { bsr_ptr[ur[u] + 1]++; if (ur[u] != uc[u]) bsr_ptr[uc[u] + 1]++; }
    for (int c = 0; c < n_cam; ++c) bsr_ptr[c + 1] += bsr_ptr[c];
    const int nnzb = bsr_ptr[n_cam];
    std::vector<int> bsr_col((size_t)nnzb), bsr_diag((size_t)n_cam), upos(n_ublk), upos_t(n_ublk);
    {
        std::vector<int> rf(bsr_ptr.begin(), bsr_ptr.end() - 1);
        for (int u = 0; u < n_ublk; ++u) {
            const int r = ur[u], c = uc[u];
            upos[u] = rf[r]++;
            bsr_col[upos[u]] = c;
            if (r == c) { upos_t[u] = upos[u]; bsr_diag[r] = upos[u]; }
            else { upos_t[u] = rf[c]++; bsr_col[upos_t[u]] = r; }
        }
    }
    tr.lap("schur: chunks + block-CSR");
    hipStream_t st = h->stream;
    h->ublk_r.upload(ur, st); h->ublk_c.upload(uc, st); h->ublk_chunk_ptr.upload(ublk_chunk_ptr, st);
    h->ublk_pos.upload(upos, st); h->ublk_pos_t.upload(upos_t, st);
    {
        std::vector<int> udiag(urow_ptr.begin(), urow_ptr.end() - 1);      // (c, c) is the first upper block of row c
        h->ublk_diag.upload(udiag, st);
    }
    h->pchunk_blk.upload(pchunk_blk, st); h->pchunk_beg.upload(pchunk_beg, st);
    h->pair_p.upload(pp, st);
    h->bsr_ptr.upload(bsr_ptr, st); h->bsr_col.upload(bsr_col, st); h->bsr_diag.upload(bsr_diag, st);
    h->n_ublk = n_ublk; h->n_pchunks = (int)pchunk_blk.size(); h->nnzb = nnzb; h->n_pairs = n_pairs;
    h->bsr_ptr_host = bsr_ptr; h->bsr_col_host = bsr_col;
    finish_schur_setup(h, want_dense);
    tr.lap("schur: upload + solver set-up");
}

// Everything of the explicit reduced system that depends on its structure only through the counts
// (n_ublk, n_pchunks, nnzb, n_pairs) and the block pattern on the host: value arrays, the
// single-launch PCG, the coarse space, the dense solver.  Shared by both set-up paths.
}  // namespace

void sfmba::finish_schur_setup(H* h, bool want_dense) {
    const int n_cam = (int)h->n_cam;
    const int n_ublk = h->n_ublk, nnzb = h->nnzb;
    const size_t nn = (size_t)h->cd * h->cd;
    const std::vector<int>& bsr_ptr = h->bsr_ptr_host;
    const std::vector<int>& bsr_col = h->bsr_col_host;
    SetupTrace tr;
    hipStream_t st = h->stream;
    h->pair_part.alloc((size_t)std::max<size_t>((size_t)h->n_pchunks, 1) * nn);
    h->Bsum.alloc((size_t)n_ublk * nn + (size_t)n_cam * h->cv);     // + the per-camera sums T_c, one collective
    h->Sfull.alloc((size_t)nnzb * nn);
    h->Bsum.zero(st); h->Sfull.zero(st);
    h->pchunk_rec.alloc(4 * (size_t)std::max(h->n_pchunks, 1));
    if (h->n_pchunks > 0) {
        Dev d{};
        d.n_pchunks = h->n_pchunks; d.pchunk_blk = h->pchunk_blk.p; d.pchunk_beg = h->pchunk_beg.p;
        d.ublk_r = h->ublk_r.p; d.ublk_c = h->ublk_c.p;
        launch_pchunk_records(d, reinterpret_cast<int4*>(h->pchunk_rec.p), st);
    }
    HIP_TRY(hipStreamSynchronize(st));
    h->explicit_schur = true;
    // one camera-major pass per linearisation instead of two (k_lin_sdiag_cm): 2 % of a solve with
    // the 6-number camera and, its 72 running sums per lane notwithstanding, with the 9-number one
    h->fuse_cm = true;
    if (const char* e = getenv("SFMBA_FUSE_CM")) h->fuse_cm = atoi(e) != 0;
    h->cm_lambda = -1.0;
    // single-launch PCG: one workgroup per compute unit at most, every workgroup co-resident
    int pcg_mode = h->opt.pcg_mode;
    if (const char* e = getenv("SFMBA_PCG_MODE")) pcg_mode = atoi(e);
    int coarse = h->opt.pcg_coarse;
    if (const char* e = getenv("SFMBA_PCG_COARSE")) coarse = atoi(e);
    if (coarse > 0) pcg_mode = 1;           // the coarse space belongs to the launch-per-iteration solver
    if (h->split) pcg_mode = 1;             // so does the matrix split over the ranks (a collective per iteration)
    h->fused_pcg = false;
    if (pcg_mode != 1 && !want_dense) {
        const int cap = pcg_fused_max_blocks(h->device);
        const int rows = pcg_fused_rows_per_block();
        int blocks = std::min(cap, (n_cam + rows - 1) / rows);
        if (h->opt.pcg_blocks > 0) blocks = std::min(blocks, h->opt.pcg_blocks);
        if (const char* e = getenv("SFMBA_PCG_BLOCKS")) blocks = std::min(blocks, std::max(1, atoi(e)));
        // left to choose, take the single launch only while every wavefront holds its one row in
        // registers: beyond that an iteration streams the matrix and is bound by bandwidth, not by
        // launches (Final shape, 13 682 cameras: 48 us per iteration either way)
        if (pcg_mode == 0 && n_cam > cap * rows) blocks = 0;
        // the pipelined variant wants one row per wavefront at most and rows of at most 64 blocks;
        // it is the only one written for blocks other than 6 x 6
        int longest = 0;
        for (int c = 0; c < n_cam; ++c) longest = std::max(longest, bsr_ptr[c + 1] - bsr_ptr[c]);
        bool pipelined = n_cam <= blocks * rows && longest <= 64;
        if (const char* e = getenv("SFMBA_PCG_PIPE")) pipelined = pipelined && atoi(e) != 0;
        if (h->cd != 6 && !pipelined) blocks = 0;
        if (blocks > 0) {
            const size_t v = (size_t)n_cam * h->cd;
            h->Shat.alloc((size_t)nnzb * nn);
            h->Fscale.alloc((size_t)n_cam * nn);
            h->bhat.alloc(v);
            h->fvec.alloc(pcg_fused_vector_words(n_cam) + 2 * v);
            h->fdots.alloc((pcg_fused_record_words(blocks) + 1) / 2);
            FusedPcg& f = h->fused;
            f.n_cam = n_cam;
            f.nd = h->cd;
            f.bsr_ptr = h->bsr_ptr.p; f.bsr_col = h->bsr_col.p; f.bsr_diag = h->bsr_diag.p;
            f.Shat = h->Shat.p; f.F = h->Fscale.p; f.bhat = h->bhat.p;
            f.vec = h->fvec.p;
            f.p = f.vec + pcg_fused_vector_words(n_cam); f.xh = f.p + v;
            f.dots = h->fdots.p;
            h->trace_pcg = getenv("SFMBA_PCG_TRACE") != nullptr;
            f.timeout_ticks = 20000000LL;         // 0.2 s of the 100 MHz wall clock
            f.admit_ticks = 200000LL;             // 2 ms
            f.blocks = blocks;
            f.pipelined = pipelined ? 1 : 0;
            h->fused_blocks = blocks;
            h->fused_pcg = true;
            h->fused_refused = false;
            // test hook: the launch counts as refused on this rank of a partitioned solve
            if (const char* e = getenv("SFMBA_PCG_REFUSE_RANK")) h->fused_refused = h->opt.nranks > 1 && atoi(e) == h->opt.rank;
        }
    }
    // Coarse space of the two-level preconditioner: clusters of consecutive cameras, never smaller
    // than the cameras one workgroup of the vector kernels handles (k_coarse_prolong relies on it)
    // and large enough that the dense coarse matrix stays below 4096 x 4096.
    h->coarse_built = false; h->coarse_forced = false; h->coarse_on = false; h->coarse_valid = false; h->coarse_async = false;
    // (left to choose: about a hundred clusters, of 48 to 128 cameras: the set-up grows with the
    // cube of their number, the iteration count with their size; measured best at the Venice shape
    // with 48 and at the Final shape with 128)
    int coarse_k = coarse > 0 ? std::max(coarse, 48) : std::min(128, std::max(48, n_cam / 100));
    if (const char* e = getenv("SFMBA_COARSE_K")) coarse_k = std::max(48, atoi(e));      // tuning: cameras per cluster
    coarse_k = std::max(coarse_k, (int)((7LL * n_cam + 4095) / 4096));
    // Where the single-launch solver will apply the correction itself (k_pcg_pipe2), the vector
    // kernels' limit does not hold: clusters go down to the eight camera rows of one of its
    // workgroups, as many of them as its 288 coarse unknowns admit (about 36).  A chain of 47 views
    // then gets 5 clusters instead of none.
    // (6 x 6 blocks only: with 9 x 9 blocks a launch with the correction costs more, and at 49 cameras
    // it was measured slower than block-Jacobi alone, 0.68 against 0.54 ms)
    const bool fused_small = coarse == 0 && h->fused_pcg && h->cd == 6 && h->fused.pipelined && n_cam < 36 * 48;
    if (fused_small) {
        // the smallest size, from the eight rows of a workgroup up, with which the columns of every
        // workgroup's rows reach at most four clusters (what k_pcg_pipe2 holds of the coarse operator)
        const int rows = pcg_fused_rows_per_block();
        coarse_k = std::max(rows, (n_cam + 35) / 36);
        for (; n_cam / coarse_k >= 2; coarse_k += (coarse_k + 3) / 4) {
            const int n_cl = n_cam / coarse_k;
            bool fits = true;
            for (int b = 0; fits && b * rows < n_cam; ++b) {
                int list[4], n = 0;
                for (int r = b * rows; fits && r < std::min(n_cam, (b + 1) * rows); ++r)
                    for (int e = bsr_ptr[r]; fits && e < bsr_ptr[r + 1]; ++e) {
                        // cluster i holds cameras [i n_cam / n_cl, (i + 1) n_cam / n_cl)
                        int J = std::min((int)((int64_t)bsr_col[e] * n_cl / n_cam), n_cl - 1);
                        while (J > 0 && (int64_t)J * n_cam / n_cl > bsr_col[e]) --J;
                        while (J + 1 < n_cl && (int64_t)(J + 1) * n_cam / n_cl <= bsr_col[e]) ++J;
                        bool seen = false;
                        for (int q = 0; q < n; ++q) seen = seen || list[q] == J;
                        if (!seen) { if (n < 4) list[n++] = J; else fits = false; }
                    }
            }
            if (fits) break;
        }
    }
    // (a single cluster would span the gauge directions of the whole problem and nothing else: S
    // vanishes on them up to the damping, the coarse matrix is rounding noise and the iteration
    // gains nothing; at least two clusters are required)
    h->fused_coarse_ok = false;
    h->fused_coarse_ready = false;
    h->fused_coarse_wanted = false;
    // (with two or three such small clusters the correction removes a third of the iterations and
    // costs more than that: measured on chains of 21 and 47 views)
    int coarse_min_cams = 256;
    if (const char* e = getenv("SFMBA_COARSE_MIN_CAMS")) coarse_min_cams = atoi(e);       // tuning
    if (coarse >= 0 && !want_dense && (coarse > 0 || n_cam >= coarse_min_cams || fused_small) && n_cam / coarse_k >= (fused_small ? 4 : 2)) {
        const int k = coarse_k;
        const int n_cl = n_cam / k;
        std::vector<int> cl_ptr((size_t)n_cl + 1), cam_cl((size_t)n_cam);
        for (int i = 0; i <= n_cl; ++i) cl_ptr[i] = (int)((int64_t)i * n_cam / n_cl);
        for (int i = 0; i < n_cl; ++i)
            for (int c = cl_ptr[i]; c < cl_ptr[i + 1]; ++c) cam_cl[c] = i;
        // blocks of the matrix grouped by cluster pair (stable counting sort)
        std::vector<int> brow((size_t)nnzb), count((size_t)n_cl * n_cl + 1, 0);
        for (int r = 0; r < n_cam; ++r)
            for (int b = bsr_ptr[r]; b < bsr_ptr[r + 1]; ++b) {
                brow[b] = r;
                count[(size_t)cam_cl[r] * n_cl + cam_cl[bsr_col[b]] + 1]++;
            }
        std::vector<int> cp_i, cp_j, cp_ptr, slot_of((size_t)n_cl * n_cl, -1);
        for (size_t key = 0; key < (size_t)n_cl * n_cl; ++key)
            if (count[key + 1] > 0) {
                slot_of[key] = (int)cp_i.size();
                cp_i.push_back((int)(key / n_cl)); cp_j.push_back((int)(key % n_cl));
                cp_ptr.push_back(0);
            }
        cp_ptr.push_back(0);
        for (size_t key = 0; key < (size_t)n_cl * n_cl; ++key)
            if (slot_of[key] >= 0) cp_ptr[slot_of[key] + 1] = count[key + 1];
        for (size_t i = 1; i < cp_ptr.size(); ++i) cp_ptr[i] += cp_ptr[i - 1];
        std::vector<int> fill(cp_ptr.begin(), cp_ptr.end() - 1), cp_blk((size_t)nnzb), cp_row((size_t)nnzb);
        for (int b = 0; b < nnzb; ++b) {
            const int sl = slot_of[(size_t)cam_cl[brow[b]] * n_cl + cam_cl[bsr_col[b]]];
            cp_blk[fill[sl]] = b; cp_row[fill[sl]] = brow[b]; fill[sl]++;
        }
        // chunks of the pairs' block lists: 8 blocks per group of 7 threads
        const int cchunk = 288;
        std::vector<int> cp_cptr, cc_beg;
        for (size_t q = 0; q + 1 < cp_ptr.size(); ++q) {
            cp_cptr.push_back((int)cc_beg.size());
            for (int b = cp_ptr[q]; b < cp_ptr[q + 1]; b += cchunk) cc_beg.push_back(b);
        }
        cp_cptr.push_back((int)cc_beg.size());
        cc_beg.push_back(nnzb);
        Coarse& k7 = h->coarse;
        k7.n_cchunks = (int)cc_beg.size() - 1;
        h->cpair_cptr.upload(cp_cptr, st); h->cchunk_beg.upload(cc_beg, st);
        h->cpart.alloc((size_t)k7.n_cchunks * kCoarseVecs * kCoarseVecs);
        k7.n_cl = n_cl; k7.n = kCoarseVecs * n_cl; k7.n_pad = (k7.n + 31) / 32 * 32; k7.n_cpairs = (int)cp_i.size();
        h->cl_ptr.upload(cl_ptr, st); h->cam_cl.upload(cam_cl, st);
        h->cpair_i.upload(cp_i, st); h->cpair_j.upload(cp_j, st);
        h->cpair_blk.upload(cp_blk, st); h->cpair_row.upload(cp_row, st);
        h->cZ.alloc((size_t)n_cam * kCoarseVecs * h->cd);
        for (auto& e : h->cE) e.alloc((size_t)k7.n_pad * k7.n_pad);
        h->cscale.alloc((size_t)k7.n_pad); h->crc.alloc((size_t)k7.n_pad);
        h->crc.zero(st);
        h->crpart.alloc((size_t)((n_cam + kBlock / h->cd - 1) / (kBlock / h->cd)) * 2 * kCoarseVecs);
        HIP_TRY(hipStreamSynchronize(st));
        k7.cl_ptr = h->cl_ptr.p; k7.cam_cl = h->cam_cl.p; k7.cpair_i = h->cpair_i.p; k7.cpair_j = h->cpair_j.p;
        k7.cpair_cptr = h->cpair_cptr.p; k7.cchunk_beg = h->cchunk_beg.p; k7.cpart = h->cpart.p; k7.cpair_blk = h->cpair_blk.p; k7.cpair_row = h->cpair_row.p;
        k7.Z = h->cZ.p; k7.E[0] = h->cE[0].p; k7.E[1] = h->cE[1].p; k7.scale = h->cscale.p; k7.rc = h->crc.p; k7.rpart = h->crpart.p;
        k7.Einv = k7.E[0];
        h->coarse_built = true;
        h->fused_restrict = false;
        if (const char* e = getenv("SFMBA_PCG_FUSED_RESTRICT")) h->fused_restrict = atoi(e) != 0;
        h->coarse_from = kCoarseFromIterations;
        h->coarse_forced = coarse > 0;
        // left to choose, the operator comes from an earlier solve's matrix and is set up beside the
        // next linearisation instead of in front of the solve that uses it
        // (split matrix: the set-up contains a collective and stays on the main stream)
        h->coarse_async = coarse == 0 && !h->split;
        if (const char* e = getenv("SFMBA_COARSE_ASYNC")) h->coarse_async = h->coarse_async && atoi(e) != 0;
        // (its set-up costs the solves nothing, so it is switched on earlier than one set up in
        // front of every solve: measured at 0.2, 0.5 and 1 x the Final shape)
        if (h->coarse_async) h->coarse_from = 12;
        if (h->fused_pcg) {
            // the same space inside the single launch (k_pcg_pipe2), where its limits hold:
            // the pipelined form, at most 256 workgroups of 8 rows, clusters of at
            // least 8 cameras spread over at most 7 workgroups, at most 288 coarse unknowns after
            // padding, at most 4 clusters among the columns of the rows of one workgroup
            const int rows = pcg_fused_rows_per_block();
            bool ok = (h->cd == 6 || h->cd == 9) && h->fused.pipelined && h->fused_blocks <= 256 && k7.n_pad <= 288 && k >= rows;
            for (int i = 0; ok && i < n_cl; ++i) ok = (cl_ptr[i + 1] - 1) / rows - cl_ptr[i] / rows + 1 <= 7;
            std::vector<int> wg_cl((size_t)5 * std::max(h->fused_blocks, 1), 0);
            for (int b = 0; ok && b * rows < n_cam && b < h->fused_blocks; ++b) {
                int* list = &wg_cl[(size_t)5 * b + 1];
                int n = 0;
                for (int r = b * rows; ok && r < std::min(n_cam, (b + 1) * rows); ++r)
                    for (int e = bsr_ptr[r]; ok && e < bsr_ptr[r + 1]; ++e) {
                        const int J = cam_cl[bsr_col[e]];
                        bool seen = false;
                        for (int q = 0; q < n; ++q) seen = seen || list[q] == J;
                        if (!seen) { if (n < 4) list[n++] = J; else ok = false; }
                    }
                wg_cl[(size_t)5 * b] = n;
            }
            if (ok) {
                h->wg_cl.upload(wg_cl, st);
                HIP_TRY(hipStreamSynchronize(st));
                if (h->cd != 6) h->czhat.alloc((size_t)n_cam * kCoarseVecs * h->cd);
            }
            if (const char* e = getenv("SFMBA_PCG_FUSED_COARSE")) ok = ok && atoi(e) != 0;
            h->fused_coarse_ok = ok;
            h->fused_coarse_from = 12;
            if (const char* e = getenv("SFMBA_PCG_FUSED_COARSE_FROM")) h->fused_coarse_from = std::max(1, atoi(e));
            // the launch-per-iteration solver, which a failed single launch falls back to, stays
            // block-Jacobi: its own use of these buffers would collide with the set-up on the side stream
            h->coarse_forced = false;
            h->coarse_from = 1 << 30;
        }
        if (const char* e = h->fused_pcg ? nullptr : getenv("SFMBA_PCG_COARSE_FROM")) {     // tuning: the switch-on rule with any cluster size
            h->coarse_from = std::max(1, atoi(e));
            h->coarse_forced = false;
        }
        tr.lap("schur: coarse space");
    }
    h->dense_n = 0;                 // (sized when first needed: ensure_dense)
    if (want_dense) {
        ensure_dense(h);
        h->use_dense = true;
    }
}

namespace {

// Argument checks and the shape of the problem, common to both set-up paths.
void begin_problem(H* h, int64_t n_cam, int64_t n_pts, int64_t n_obs, const void* camera_params, const void* points_3d,
                   const void* cam_idx, const void* pt_idx, const void* points_2d, const double* K) {
    if (n_cam <= 0 || n_pts <= 0 || n_obs < 0) fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_set_problem: sizes must be positive");
    const int model = h->opt.camera_model;
    if (model != SFMBA_CAMERA_PINHOLE && model != SFMBA_CAMERA_BAL)
        fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_set_problem: unknown camera model %d", model);
    if (!camera_params || !points_3d || (!K && model == SFMBA_CAMERA_PINHOLE) || (n_obs > 0 && (!cam_idx || !pt_idx || !points_2d)))
        fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_set_problem: null array");
    if (n_obs > 2000000000LL || n_pts > 2000000000LL || n_cam > 50000000LL)
        fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_set_problem: sizes exceed 32-bit indexing");
    HIP_TRY(hipSetDevice(h->device));
    h->has_problem = false;
    h->n_cam = n_cam; h->n_pts_all = n_pts; h->n_obs_all = n_obs;
    h->model = model == SFMBA_CAMERA_BAL ? MODEL_BAL : MODEL_PINHOLE;
    h->cd = cam_dim(h->model); h->cv = cam_sums(h->model); h->cw = cam_words(h->model);
    h->dev.model = h->model;
    h->dev.K = K ? Intr{K[0], K[4], K[2], K[5]} : Intr{1.0, 1.0, 0.0, 0.0};
    if (const char* e = getenv("SFMBA_CHUNK")) { int v = atoi(e); if (v >= 64) h->chunk = v; }
    h->nb_cam = (int)((n_cam + kBlock - 1) / kBlock);
}

void alloc_work_buffers(H* h);

// The structure of the problem built on host threads (counting sorts, pair lists), then uploaded.
// Kept as the reference layout the device path (sfmba_setup.hip) must reproduce.
void host_setup(H* h, const int64_t* cam_idx, const int64_t* pt_idx, const double* points_2d) {
    const int64_t n_cam = h->n_cam, n_pts = h->n_pts_all, n_obs = h->n_obs_all;
    SetupTrace tr;
    // The sorts below run on several host threads.  Every pass gives thread t a range of the KEYS
    // (points or cameras); each thread scans all observations in order and handles those whose key
    // is its own, so the result is the stable sort a single thread produces, whatever the count.
    int nt = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* e = getenv("SFMBA_THREADS")) nt = std::max(1, atoi(e));
    if (n_obs < 200000) nt = 1;
    auto run_threads = [&](auto&& body) {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(body, t);
        body(0);
        for (auto& x : th) x.join();
    };
    auto share = [&](int64_t n, int t) { return n * t / nt; };     // thread t owns keys [share(n, t), share(n, t + 1))
    {   // the first observation with an index out of range (threads split the observations here)
        std::vector<int64_t> bad((size_t)nt, n_obs);
        run_threads([&](int t) {
            for (int64_t i = share(n_obs, t); i < share(n_obs, t + 1); ++i)
                if (cam_idx[i] < 0 || cam_idx[i] >= n_cam || pt_idx[i] < 0 || pt_idx[i] >= n_pts) { bad[t] = i; break; }
        });
        const int64_t i = *std::min_element(bad.begin(), bad.end());
        if (i < n_obs) {
            if (cam_idx[i] < 0 || cam_idx[i] >= n_cam)
                fail(SFMBA_ERR_INDEX_RANGE, "index %lld is out of bounds for axis 0 with size %lld", (long long)cam_idx[i], (long long)n_cam);
            fail(SFMBA_ERR_INDEX_RANGE, "index %lld is out of bounds for axis 0 with size %lld", (long long)pt_idx[i], (long long)n_pts);
        }
    }
    // observations per point, then the owned point range: contiguous, equal observation share
    std::vector<int> cnt((size_t)n_pts + 1, 0);
    run_threads([&](int t) {
        const int64_t lo = share(n_pts, t), hi = share(n_pts, t + 1);
        for (int64_t i = 0; i < n_obs; ++i) { const int64_t p = pt_idx[i]; if (p >= lo && p < hi) cnt[p + 1]++; }
    });
    std::vector<int64_t> pre((size_t)n_pts + 1, 0);
    for (int64_t p = 0; p < n_pts; ++p) pre[p + 1] = pre[p] + cnt[p + 1];
    const int nr = std::max(1, h->opt.nranks), rk = nr > 1 ? h->opt.rank : 0;
    auto boundary = [&](int r) -> int64_t {
        if (r <= 0) return 0;
        if (r >= nr) return n_pts;
        int64_t target = (int64_t)((double)n_obs * r / nr);
        return std::lower_bound(pre.begin(), pre.end(), target) - pre.begin();
    };
    h->pt_lo = std::min(boundary(rk), n_pts);
    h->pt_hi = std::min(std::max(boundary(rk + 1), h->pt_lo), n_pts);
    h->n_pts = (int)(h->pt_hi - h->pt_lo);
    h->n_obs = (int)(pre[h->pt_hi] - pre[h->pt_lo]);
    const int np = h->n_pts, no = h->n_obs;

    // Renumber the owned points by their smallest observing camera (counting sort, stable in the
    // caller's point id).  Points seen by the same cameras then sit next to each other in HBM:
    // the per-camera gathers of point data and the per-point gathers of camera data stay in L2.
    std::vector<int> min_cam((size_t)np, (int)n_cam);   // n_cam = "unobserved", sorted last
    run_threads([&](int t) {
        const int64_t lo = h->pt_lo + share(np, t), hi = h->pt_lo + share(np, t + 1);
        for (int64_t i = 0; i < n_obs; ++i) {
            const int64_t p = pt_idx[i];
            if (p < lo || p >= hi) continue;
            int& m = min_cam[p - h->pt_lo];
            if ((int)cam_idx[i] < m) m = (int)cam_idx[i];
        }
    });
    std::vector<int> key_ptr((size_t)n_cam + 2, 0);
    for (int p = 0; p < np; ++p) key_ptr[min_cam[p] + 1]++;
    for (int64_t c = 0; c <= n_cam; ++c) key_ptr[c + 1] += key_ptr[c];
    std::vector<int> old2new((size_t)np);
    h->new2old.assign((size_t)np, 0);
    for (int p = 0; p < np; ++p) {
        int q = key_ptr[min_cam[p]]++;
        old2new[p] = q;
        h->new2old[q] = p;
    }
    // point-major order of the owned observations (stable counting sort by renumbered point)
    std::vector<int> pt_ptr((size_t)np + 1, 0);
    for (int p = 0; p < np; ++p) pt_ptr[old2new[p] + 1] = (int)(pre[h->pt_lo + p + 1] - pre[h->pt_lo + p]);
    for (int p = 0; p < np; ++p) pt_ptr[p + 1] += pt_ptr[p];
    std::vector<int> fill(pt_ptr.begin(), pt_ptr.end() - 1);
    std::vector<int> pm_cam(no), pm_pt(no), pm_orig(no);
    std::vector<double2> pm_uv(no);
    run_threads([&](int t) {
        const int64_t lo = h->pt_lo + share(np, t), hi = h->pt_lo + share(np, t + 1);
        for (int64_t i = 0; i < n_obs; ++i) {
            const int64_t p = pt_idx[i];
            if (p < lo || p >= hi) continue;
            const int q = old2new[p - h->pt_lo];
            const int j = fill[q]++;
            pm_cam[j] = (int)cam_idx[i];
            pm_pt[j] = q;
            pm_orig[j] = (int)i;
            pm_uv[j] = make_double2(points_2d[2 * i], points_2d[2 * i + 1]);
        }
    });
    tr.lap("validate + point-major sort");
    // camera-major order (stable counting sort of the point-major list by camera)
    std::vector<int> cam_ptr((size_t)n_cam + 1, 0);
    run_threads([&](int t) {
        const int lo = (int)share(n_cam, t), hi = (int)share(n_cam, t + 1);
        for (int j = 0; j < no; ++j) { const int c = pm_cam[j]; if (c >= lo && c < hi) cam_ptr[c + 1]++; }
    });
    for (int64_t c = 0; c < n_cam; ++c) cam_ptr[c + 1] += cam_ptr[c];
    std::vector<int> cfill(cam_ptr.begin(), cam_ptr.end() - 1);
    std::vector<int> cm_pt(no), pm2cm(no);
    std::vector<double2> cm_uv(no);
    run_threads([&](int t) {
        const int lo = (int)share(n_cam, t), hi = (int)share(n_cam, t + 1);
        for (int j = 0; j < no; ++j) {
            const int c = pm_cam[j];
            if (c < lo || c >= hi) continue;
            const int k = cfill[c]++;
            pm2cm[j] = k;
            cm_pt[k] = pm_pt[j];
            cm_uv[k] = pm_uv[j];
        }
    });
    // chunks: pieces of at most `chunk` observations of one camera
    std::vector<int> chunk_cam, chunk_beg, cam_chunk_ptr((size_t)n_cam + 1, 0);
    for (int64_t c = 0; c < n_cam; ++c) {
        cam_chunk_ptr[c] = (int)chunk_cam.size();
        for (int b = cam_ptr[c]; b < cam_ptr[c + 1]; b += h->chunk) { chunk_cam.push_back((int)c); chunk_beg.push_back(b); }
    }
    cam_chunk_ptr[n_cam] = (int)chunk_cam.size();
    chunk_beg.push_back(no);
    h->n_chunks = (int)chunk_cam.size();

    tr.lap("camera-major sort + chunks");
    hipStream_t st = h->stream;
    h->pt_ptr.upload(pt_ptr, st); h->pm_cam.upload(pm_cam, st); h->pm_pt.upload(pm_pt, st);
    h->pm_orig.upload(pm_orig, st); h->pm_uv.upload(pm_uv, st);
    h->cm_pt.upload(cm_pt, st); h->cm_uv.upload(cm_uv, st);
    h->chunk_cam.upload(chunk_cam, st); h->chunk_beg.upload(chunk_beg, st); h->cam_chunk_ptr.upload(cam_chunk_ptr, st);
    {   // wave tiles of the point-major order: whole points, <= 64 observations (or one long point)
        std::vector<int> wt;
        int beg = 0;
        for (int p = 0; p < np; ++p) {
            const int k = pt_ptr[p + 1] - pt_ptr[p];
            if (k == 0) continue;
            if (pt_ptr[p + 1] - beg > 64 && pt_ptr[p] > beg) { wt.push_back(beg); beg = pt_ptr[p]; }
            if (k > 64) { wt.push_back(beg); beg = pt_ptr[p + 1]; }
        }
        if (beg < no) wt.push_back(beg);
        h->n_wtiles = (int)wt.size();
        wt.push_back(no);
        h->wtile_beg.upload(wt, st);
    }
    h->new2old_dev.upload(h->new2old, st);
    h->device_built = false;
    tr.lap("uploads");
    build_schur_structure(h, n_obs, cam_idx, pt_idx, pre, pt_ptr, pm_cam);
    HIP_TRY(hipStreamSynchronize(st));      // the uploads read host vectors that end here
    tr.lap("schur structure (total)");
}

// Arrays whose sizes follow from the counts of the structure (either path).
void alloc_work_buffers(H* h) {
    const int np = h->n_pts, no = h->n_obs;
    const int64_t n_cam = h->n_cam;
    hipStream_t st = h->stream;
    if ((np + kBlock - 1) / kBlock > kMaxScalBlocks) fail(SFMBA_ERR_INVALID_ARGUMENT, "partition has too many points for one GPU");
    if ((h->n_wtiles + 3) / 4 > kMaxScalBlocks || (no + kBlock - 1) / kBlock > kMaxScalBlocks)
        fail(SFMBA_ERR_INVALID_ARGUMENT, "partition has too many observations for one GPU");
    for (DBuf<double>* b : {&h->cm_sx, &h->cm_sy, &h->cm_siz}) b->alloc((size_t)no);
    h->V.alloc((size_t)np * 6); h->gp.alloc((size_t)np * 4); h->Vinv.alloc((size_t)np * 6);
    h->V.zero(st); h->gp.zero(st);      // points without observations are never written
    h->s.alloc((size_t)np); h->s0.alloc((size_t)np);
    const size_t cd = (size_t)h->cd, cv = (size_t)h->cv;
    h->chunk_part.alloc((size_t)std::max(h->n_chunks, 1) * std::max<int>((int)cv, lin_sdiag_values(h->model)));
    h->UG.alloc((size_t)n_cam * cv + (size_t)std::max(1, h->opt.nranks)); h->T.alloc((size_t)n_cam * cv);
    h->UG.zero(h->stream);
    h->Minv.alloc((size_t)n_cam * cd * cd); h->rhs.alloc((size_t)n_cam * cd);
    h->cam_seen.alloc((size_t)n_cam);
    for (DBuf<double>* b : {&h->px, &h->pr, &h->pz, &h->pp[0], &h->pp[1], &h->pq, &h->qS}) b->alloc((size_t)n_cam * cd);
    h->nb_bsr = bsr_blocks_host((int)n_cam, h->model);
    {
        Dev t{};
        t.n_cam = (int)n_cam;
        t.model = h->model;
        h->dot_off = std::max(h->nb_cam, pcg_spmv_blocks(t));       // p'q partials, then r'z partials
        h->pcg.alloc(2); h->dot_part.alloc((size_t)h->dot_off + pcg_update_blocks(t));
    }
    if (h->explicit_schur) { h->pm_seed.release(); h->pm_rv.release(); }     // only the matrix-free product reads them
    else { h->pm_seed.alloc((size_t)no); h->pm_rv.alloc((size_t)no); }
    h->scal_part.alloc((size_t)SLOT_COUNT * kMaxScalBlocks); h->scalars.alloc(SLOT_COUNT);
    h->scal_part.zero(st); h->scalars.zero(st); h->chunk_part.zero(st);
    h->tmp_b.alloc((size_t)std::max<int64_t>(64, n_cam * (int64_t)(cd * cd)));
    HIP_TRY(hipStreamSynchronize(st));
}

// How the structure is built: on the GPU unless the host path is asked for (options.setup_mode or
// SFMBA_SETUP=host; it is the layout the device path is tested against).
bool use_device_setup(const H* h) {
    if (const char* e = getenv("SFMBA_SETUP")) return strcmp(e, "host") != 0;
    return h->opt.setup_mode != 1;
}

void do_set_problem(H* h, int64_t n_cam, int64_t n_pts, int64_t n_obs, const double* camera_params,
                    const double* points_3d, const int64_t* cam_idx, const int64_t* pt_idx,
                    const double* points_2d, const double* K) {
    begin_problem(h, n_cam, n_pts, n_obs, camera_params, points_3d, cam_idx, pt_idx, points_2d, K);
    SetupTrace tr;
    bool points_uploaded = false;      // all of points_3d is in in_pts already
    if (use_device_setup(h)) {
        hipStream_t st = h->stream;
        h->in_cam.alloc((size_t)n_obs); h->in_pt.alloc((size_t)n_obs); h->in_uv.alloc((size_t)n_obs);
        h->in_pts.alloc(3 * (size_t)n_pts);
        if (n_obs) {
            HIP_TRY(hipMemcpyAsync(h->in_pt.p, pt_idx, (size_t)n_obs * sizeof(int64_t), hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(h->in_cam.p, cam_idx, (size_t)n_obs * sizeof(int64_t), hipMemcpyHostToDevice, st));
        }
        tr.lap("upload of the index arrays");
        // The image points (half of the bytes) and the 3-D points go up from a second host thread
        // on a stream of their own while the index work runs on the GPU; the structure needs them
        // only at its very end (device_place_uv, do_set_params).
        ensure_copy_stream(h);
        hipError_t copy_err = hipSuccess;
        std::thread copier([&] {
            copy_err = hipSetDevice(h->device);
            if (copy_err == hipSuccess && n_obs)
                copy_err = hipMemcpyAsync(h->in_uv.p, points_2d, (size_t)n_obs * sizeof(double2), hipMemcpyHostToDevice, h->copy_stream);
            if (copy_err == hipSuccess)
                copy_err = hipMemcpyAsync(h->in_pts.p, points_3d, 3 * (size_t)n_pts * sizeof(double), hipMemcpyHostToDevice, h->copy_stream);
            if (copy_err == hipSuccess) copy_err = hipStreamSynchronize(h->copy_stream);
        });
        try {
            device_setup(h, h->in_cam.p, h->in_pt.p);
        } catch (...) {
            copier.join();
            throw;
        }
        copier.join();
        if (copy_err != hipSuccess) throw HipError{copy_err, "upload of points_2d / points_3d"};
        tr.lap("structure on the device");
        device_place_uv(h, h->in_uv.p);
        points_uploaded = true;
    } else {
        host_setup(h, cam_idx, pt_idx, points_2d);
    }
    alloc_work_buffers(h);
    tr.lap("work buffers");
    do_set_params(h, camera_params, points_3d, points_uploaded);
    tr.lap("parameters");
    h->has_problem = true;
}

// The same with the observations already in device memory (reference layout: int64 indices,
// n_obs x 2 doubles); cameras and points are small and still come from the host.
// A pointer the kernels of this handle may dereference: device memory of the handle's GPU, managed
// memory, or pinned host memory.  Anything else (an ordinary host array passed by mistake) is
// refused here: a kernel that touched it would fault the GPU.
void require_device_pointer(H* h, const void* ptr, const char* what) {
    hipPointerAttribute_t attr{};
    const hipError_t e = hipPointerGetAttributes(&attr, ptr);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_set_problem_device: %s is not a device pointer (%s)", what, hipGetErrorString(e));
    }
    if (attr.type == hipMemoryTypeDevice && attr.device != h->device)
        fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_set_problem_device: %s lives on GPU %d, the handle on GPU %d", what, attr.device, h->device);
    if (attr.type != hipMemoryTypeDevice && attr.type != hipMemoryTypeManaged && attr.type != hipMemoryTypeHost)
        fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_set_problem_device: %s is not a device pointer", what);
}

void do_set_problem_device(H* h, int64_t n_cam, int64_t n_pts, int64_t n_obs, const double* camera_params,
                           const double* points_3d, const int64_t* cam_idx_dev, const int64_t* pt_idx_dev,
                           const double* points_2d_dev, const double* K) {
    if (n_obs > 0) {
        if (!cam_idx_dev || !pt_idx_dev || !points_2d_dev) fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_set_problem_device: null array");
        HIP_TRY(hipSetDevice(h->device));
        require_device_pointer(h, cam_idx_dev, "camera_indices");
        require_device_pointer(h, pt_idx_dev, "point_indices");
        require_device_pointer(h, points_2d_dev, "points_2d");
    }
    begin_problem(h, n_cam, n_pts, n_obs, camera_params, points_3d, cam_idx_dev, pt_idx_dev, points_2d_dev, K);
    SetupTrace tr;
    device_setup(h, cam_idx_dev, pt_idx_dev);
    device_place_uv(h, reinterpret_cast<const double2*>(points_2d_dev));
    tr.lap("structure on the device");
    alloc_work_buffers(h);
    do_set_params(h, camera_params, points_3d);
    tr.lap("work buffers + parameters");
    h->has_problem = true;
}

template <class F>
int guarded(sfmba_handle* h, F&& f) {
    if (!h) return SFMBA_ERR_INVALID_ARGUMENT;
    int code = SFMBA_OK;
    try {
        f();
        return SFMBA_OK;
    } catch (const ApiError& e) {
        h->err = e.msg;
        code = e.code;
    } catch (const HipError& e) {
        h->err = std::string(e.what) + ": " + hipGetErrorString(e.e);
        code = SFMBA_ERR_HIP;
    } catch (const std::bad_alloc&) {
        h->err = "out of host memory";
        code = SFMBA_ERR_INVALID_ARGUMENT;
    } catch (...) {
        h->err = "unexpected exception";
        code = SFMBA_ERR_INVALID_ARGUMENT;
    }
    // a member of an in-process group that fails would leave the others waiting in their next
    // collective for ever: wake them, they fail with SFMBA_ERR_COMM
    if (h->group) h->group->abort();
    return code;
}

// The host copy of the point permutation (diagnostic entry points): the device path leaves it
// on the GPU until somebody asks.
void fetch_new2old(H* h) {
    if (h->new2old.size() == (size_t)h->n_pts) return;
    h->new2old.resize((size_t)h->n_pts);
    if (h->n_pts) HIP_TRY(hipMemcpy(h->new2old.data(), h->new2old_dev.p, (size_t)h->n_pts * sizeof(int), hipMemcpyDeviceToHost));
}

void need_problem(H* h, const char* who) {
    if (!h->has_problem) fail(SFMBA_ERR_INVALID_ARGUMENT, "%s: no problem set", who);
    HIP_TRY(hipSetDevice(h->device));
}
void need_linearized(H* h, const char* who) {
    need_problem(h, who);
    if (!h->linearized) fail(SFMBA_ERR_INVALID_ARGUMENT, "%s: call sfmba_linearize first", who);
}

}  // namespace

// ---------------------------------------------------------------------------------- C ABI
extern "C" {

void sfmba_default_options(sfmba_options* o) {
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->ftol = 1e-4; o->xtol = 1e-12; o->gtol = 1e-8; o->max_nfev = 0;
    o->max_iterations = 200; o->initial_lambda = 0.0; o->pcg_rtol = 0.0; o->pcg_max_iter = 1000;
    o->linear_solver = SFMBA_SOLVER_AUTO; o->dense_max_dim = 1536; o->verbose = 0; o->profile = 0; o->device = -1;
    o->rank = 0; o->nranks = 1;
    o->pcg_mode = 0; o->pcg_blocks = 0;
    o->camera_model = SFMBA_CAMERA_PINHOLE;
    o->pcg_coarse = 0;
    o->setup_mode = 0;
    o->matrix_split = 0;
}

int sfmba_create(const sfmba_options* opts, sfmba_handle** out) {
    if (!out) return SFMBA_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    sfmba_handle* h = new (std::nothrow) sfmba_handle();
    if (!h) return SFMBA_ERR_INVALID_ARGUMENT;
    if (opts) h->opt = *opts; else sfmba_default_options(&h->opt);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { delete h; return SFMBA_ERR_NO_DEVICE; }
    int dev = h->opt.device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev >= n) { delete h; return SFMBA_ERR_NO_DEVICE; }
    h->device = dev;
    if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return SFMBA_ERR_HIP;
    }
    // this device's copy of the scratch array of sfmba_triangulate_noalloc (once per device).  A
    // handle that never takes that entry point does not need it: a failure here (return code
    // checked in scratch_ptr) leaves the pointer null and is reported by that entry point alone.
    if (scratch_ptr(dev, &h->scratch) != hipSuccess) { h->scratch = nullptr; (void)hipGetLastError(); }
    if (hipHostMalloc((void**)&h->pcg_host, 2 * sizeof(PcgState), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&h->scalars_host, SLOT_COUNT * sizeof(double), hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&h->pcg_ev[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->pcg_ev[1], hipEventDisableTiming) != hipSuccess) {
        delete h;
        return SFMBA_ERR_HIP;
    }
    *out = h;
    return SFMBA_OK;
}

int sfmba_destroy(sfmba_handle* h) {
    if (!h) return SFMBA_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(h->device);
    delete h;
    return SFMBA_OK;
}

int sfmba_set_print(sfmba_handle* h, void (*sink)(const char* text, void* user), void* user) {
    if (!h) return SFMBA_ERR_INVALID_ARGUMENT;
    h->print_cb = sink;
    h->print_user = user;
    return SFMBA_OK;
}

const char* sfmba_last_error(const sfmba_handle* h) { return h ? h->err.c_str() : "null handle"; }

int sfmba_set_problem(sfmba_handle* h, int64_t n_cameras, int64_t n_points, int64_t n_obs,
                      const double* camera_params, const double* points_3d,
                      const int64_t* camera_indices, const int64_t* point_indices,
                      const double* points_2d, const double* K) {
    return guarded(h, [&] {
        do_set_problem(h, n_cameras, n_points, n_obs, camera_params, points_3d, camera_indices, point_indices, points_2d, K);
    });
}

int sfmba_set_problem_device(sfmba_handle* h, int64_t n_cameras, int64_t n_points, int64_t n_obs,
                             const double* camera_params, const double* points_3d,
                             const int64_t* camera_indices_dev, const int64_t* point_indices_dev,
                             const double* points_2d_dev, const double* K) {
    return guarded(h, [&] {
        do_set_problem_device(h, n_cameras, n_points, n_obs, camera_params, points_3d, camera_indices_dev, point_indices_dev, points_2d_dev, K);
    });
}

int sfmba_set_params(sfmba_handle* h, const double* camera_params, const double* points_3d) {
    return guarded(h, [&] {
        need_problem(h, "sfmba_set_params");
        if (!camera_params || !points_3d) fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_set_params: null array");
        do_set_params(h, camera_params, points_3d);
    });
}

int sfmba_reset_params(sfmba_handle* h) {
    return guarded(h, [&] {
        need_problem(h, "sfmba_reset_params");
        // No copy: the loaded parameters (cams0, X0) become the current point as they are; they are
        // never written - the first trial step goes into buffer 0, which takes over when it is
        // accepted.  (The points without observations have held their loaded values in both buffers
        // since the problem was set, and no step rewrites them.)
        h->cur = 0;
        h->at_start = true;
        h->linearized = false;
        refresh_dev(h);
    });
}

int sfmba_residuals(sfmba_handle* h, double* out) {
    return guarded(h, [&] {
        need_problem(h, "sfmba_residuals");
        if (!out) fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_residuals: null output");
        refresh_dev(h);
        const size_t n = 2 * (size_t)h->n_obs_all;
        if (h->tmp_a.n < n) h->tmp_a.alloc(n);
        const bool whole = h->n_obs == h->n_obs_all;
        if (!whole) HIP_TRY(hipMemcpyAsync(h->tmp_a.p, out, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        launch_residuals(h->dev, h->tmp_a.p, h->stream);
        HIP_TRY(hipMemcpyAsync(out, h->tmp_a.p, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    });
}

int sfmba_jacobian_blocks(sfmba_handle* h, double* jac_cam, double* jac_pt) {
    return guarded(h, [&] {
        need_problem(h, "sfmba_jacobian_blocks");
        if (!jac_cam || !jac_pt) fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_jacobian_blocks: null output");
        refresh_dev(h);
        const size_t n = (size_t)h->n_obs_all;
        DBuf<double> jc, jp;
        const size_t jw = 2 * (size_t)h->cd;
        jc.alloc(jw * n); jp.alloc(6 * n);
        if (h->n_obs != h->n_obs_all) {
            HIP_TRY(hipMemcpyAsync(jc.p, jac_cam, jw * n * sizeof(double), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(hipMemcpyAsync(jp.p, jac_pt, 6 * n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        }
        launch_jacobian_blocks(h->dev, jc.p, jp.p, h->stream);
        HIP_TRY(hipMemcpyAsync(jac_cam, jc.p, jw * n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(jac_pt, jp.p, 6 * n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    });
}

int sfmba_solve(sfmba_handle* h, sfmba_stats* stats) {
    return guarded(h, [&] { do_solve(h, stats); });
}

int sfmba_get_params(sfmba_handle* h, double* camera_params, double* points_3d) {
    return guarded(h, [&] {
        need_problem(h, "sfmba_get_params");
        refresh_dev(h);
        if (camera_params && h->model == MODEL_BAL) {
            std::vector<double> rec((size_t)h->n_cam * 16);
            HIP_TRY(hipMemcpyAsync(rec.data(), h->dev.cams, rec.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            for (int64_t c = 0; c < h->n_cam; ++c) {
                const double* r = rec.data() + 16 * c;
                double* o = camera_params + 9 * c;
                vector_of_rotation(r, o);
                for (int i = 0; i < 6; ++i) o[3 + i] = r[9 + i];
            }
        } else if (camera_params) {
            HIP_TRY(hipMemcpyAsync(camera_params, h->dev.cams, (size_t)h->n_cam * 12 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        }
        if (points_3d) {
            // back into the caller's numbering on the device; with several ranks every rank returns
            // all points: the sum of zero-padded full-length arrays
            const size_t full = (size_t)h->n_pts_all * 3;
            if (h->tmp_a.n < full) h->tmp_a.alloc(full);
            if (h->has_comm()) HIP_TRY(hipMemsetAsync(h->tmp_a.p, 0, full * sizeof(double), h->stream));
            double* dst = h->tmp_a.p + (h->has_comm() ? 3 * (size_t)h->pt_lo : 0);
            if (h->n_pts) launch_points_out(h->dev.X, h->new2old_dev.p, h->n_pts, dst, h->stream);
            if (h->has_comm()) {
                allreduce(h, h->tmp_a.p, full);
                HIP_TRY(hipMemcpyAsync(points_3d, h->tmp_a.p, full * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            } else if (h->n_pts) {
                HIP_TRY(hipMemcpyAsync(points_3d + 3 * h->pt_lo, dst, 3 * (size_t)h->n_pts * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            }
        }
        HIP_TRY(hipStreamSynchronize(h->stream));
    });
}

int sfmba_reprojection_errors(sfmba_handle* h, double* mean_error, int64_t* n_observations) {
    return guarded(h, [&] {
        need_problem(h, "sfmba_reprojection_errors");
        if (!mean_error) fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_reprojection_errors: null output");
        refresh_dev(h);
        const size_t n = (size_t)h->n_cam * 2;
        launch_reproj_cm(h->dev, h->tmp_b.p, h->stream);      // tmp_b holds >= 2 n_cam doubles
        allreduce(h, h->tmp_b.p, n);
        std::vector<double> v(n);
        HIP_TRY(hipMemcpyAsync(v.data(), h->tmp_b.p, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        for (int64_t c = 0; c < h->n_cam; ++c) {
            const double cnt = v[2 * c + 1];
            mean_error[c] = cnt > 0 ? v[2 * c] / cnt : std::numeric_limits<double>::quiet_NaN();   // np.mean([]) is nan
            if (n_observations) n_observations[c] = (int64_t)cnt;
        }
        h->linearized = false;      // chunk_part was reused
    });
}

int sfmba_triangulate(sfmba_handle* h, int64_t n, const double* P1, const double* P2,
                      const double* x1, const double* x2, double* X, double* err1, double* err2) {
    return guarded(h, [&] {
        if (n < 0 || n > 2000000000LL || !P1 || !P2 || (n > 0 && (!x1 || !x2 || !X)))
            fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_triangulate: bad arguments");
        if (n == 0) return;
        HIP_TRY(hipSetDevice(h->device));
        DBuf<double2> d1, d2;
        DBuf<double> dX, de;
        d1.alloc((size_t)n); d2.alloc((size_t)n); dX.alloc((size_t)3 * n); de.alloc((size_t)2 * n);
        HIP_TRY(hipMemcpyAsync(d1.p, x1, (size_t)n * sizeof(double2), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(d2.p, x2, (size_t)n * sizeof(double2), hipMemcpyHostToDevice, h->stream));
        launch_triangulate(P1, P2, (int)n, d1.p, d2.p, dX.p, de.p, de.p + n, h->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(X, dX.p, (size_t)3 * n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (err1) HIP_TRY(hipMemcpyAsync(err1, de.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (err2) HIP_TRY(hipMemcpyAsync(err2, de.p + n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    });
}

int sfmba_triangulate_noalloc(sfmba_handle* h, int64_t n, const double* P1, const double* P2,
                              const double* x1, const double* x2, double* X, double* err1, double* err2) {
    return guarded(h, [&] {
        if (n < 0 || !P1 || !P2 || (n > 0 && (!x1 || !x2 || !X)))
            fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_triangulate_noalloc: bad arguments");
        if (n > kNoallocMax)
            fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_triangulate_noalloc: %lld correspondences exceed the limit of %lld "
                 "(SFMBA_TRIANGULATE_NOALLOC_MAX); use sfmba_triangulate", (long long)n, (long long)kNoallocMax);
        if (n == 0) return;
        HIP_TRY(hipSetDevice(h->device));
        if (!h->scratch) {                     // not resolved in sfmba_create: once more, and say why not
            const hipError_t e = scratch_ptr(h->device, &h->scratch);
            if (e != hipSuccess || !h->scratch)
                fail(SFMBA_ERR_HIP, "sfmba_triangulate_noalloc: the library's scratch array has no address on device %d (%s)",
                     h->device, hipGetErrorString(e));
        }
        unsigned char* base = h->scratch;      // this device's address of the array, never the array's name
        double2* d1 = reinterpret_cast<double2*>(base + kScrX1);
        double2* d2 = reinterpret_cast<double2*>(base + kScrX2);
        double* dX = reinterpret_cast<double*>(base + kScrX);
        double* de = reinterpret_cast<double*>(base + kScrErr);
        // (every word the kernel reads or the copies below return is written first in this call:
        // nothing of an earlier call's contents is used, nothing has to start at zero)
        HIP_TRY(hipMemcpyAsync(d1, x1, (size_t)n * sizeof(double2), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(d2, x2, (size_t)n * sizeof(double2), hipMemcpyHostToDevice, h->stream));
        launch_triangulate(P1, P2, (int)n, d1, d2, dX, de, de + kNoallocMax, h->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(X, dX, (size_t)3 * n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (err1) HIP_TRY(hipMemcpyAsync(err1, de, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (err2) HIP_TRY(hipMemcpyAsync(err2, de + kNoallocMax, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    });
}

int sfmba_point_errors(sfmba_handle* h, int64_t n, const double* P, const double* X,
                       const double* x, double* err) {
    return guarded(h, [&] {
        if (n < 0 || n > 2000000000LL || !P || (n > 0 && (!X || !x || !err)))
            fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_point_errors: bad arguments");
        if (n == 0) return;
        HIP_TRY(hipSetDevice(h->device));
        DBuf<double2> dx;
        DBuf<double> dX, de;
        dx.alloc((size_t)n); dX.alloc((size_t)3 * n); de.alloc((size_t)n);
        HIP_TRY(hipMemcpyAsync(dx.p, x, (size_t)n * sizeof(double2), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(dX.p, X, (size_t)3 * n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        launch_reprojection_errors(P, (int)n, dX.p, dx.p, de.p, h->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(err, de.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    });
}

int sfmba_linearize(sfmba_handle* h, double lambda) {
    return guarded(h, [&] {
        need_problem(h, "sfmba_linearize");
        phase_linearize(h);
        phase_damp(h, lambda);
        HIP_TRY(hipStreamSynchronize(h->stream));
    });
}

int sfmba_get_blocks(sfmba_handle* h, double* U, double* g_cam, double* V, double* g_pt) {
    return guarded(h, [&] {
        need_linearized(h, "sfmba_get_blocks");
        fetch_new2old(h);
        // (inside a solve the camera sums are formed in one pass and U_c is not kept on its own)
        if (h->cm_lambda >= 0.0 && (U || g_cam)) fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_get_blocks: call sfmba_linearize first");
        // (nor are V_p and g_p stored when the points were damped in the linearisation pass)
        if (!h->have_v && (V || g_pt)) fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_get_blocks: call sfmba_linearize first");
        if (U || g_cam) {
            const int cd = h->cd, cv = h->cv;
            std::vector<double> ug((size_t)h->n_cam * cv);
            HIP_TRY(hipMemcpy(ug.data(), h->UG.p, ug.size() * sizeof(double), hipMemcpyDeviceToHost));
            for (int64_t c = 0; c < h->n_cam; ++c) {
                const double* u = ug.data() + (size_t)cv * c;
                if (U) {
                    int n = 0;
                    double* o = U + (size_t)cd * cd * c;
                    for (int i = 0; i < cd; ++i)
                        for (int j = i; j < cd; ++j) { o[cd * i + j] = u[n]; o[cd * j + i] = u[n]; ++n; }
                }
                if (g_cam) for (int i = 0; i < cd; ++i) g_cam[(size_t)cd * c + i] = u[cv - cd + i];
            }
        }
        if (V) {
            std::vector<double> v((size_t)h->n_pts * 6);
            if (h->n_pts) HIP_TRY(hipMemcpy(v.data(), h->V.p, v.size() * sizeof(double), hipMemcpyDeviceToHost));
            for (int p = 0; p < h->n_pts; ++p) {
                const double* a = v.data() + 6 * (size_t)p;
                double* o = V + 9 * (h->pt_lo + h->new2old[p]);
                o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[1]; o[4] = a[3]; o[5] = a[4]; o[6] = a[2]; o[7] = a[4]; o[8] = a[5];
            }
        }
        if (g_pt) {
            std::vector<double> g((size_t)h->n_pts * 4);
            if (h->n_pts) HIP_TRY(hipMemcpy(g.data(), h->gp.p, g.size() * sizeof(double), hipMemcpyDeviceToHost));
            for (int p = 0; p < h->n_pts; ++p)
                for (int i = 0; i < 3; ++i) g_pt[3 * (h->pt_lo + h->new2old[p]) + i] = g[4 * (size_t)p + i];
        }
    });
}

int sfmba_schur_matvec(sfmba_handle* h, const double* x, double* y) {
    return guarded(h, [&] {
        need_linearized(h, "sfmba_schur_matvec");
        if (!x || !y) fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_schur_matvec: null array");
        const size_t n = (size_t)h->n_cam * h->cd;
        HIP_TRY(hipMemcpyAsync(h->pz.p, x, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        if (h->explicit_schur) {
            launch_bsr_spmv(h->dev, h->pz.p, h->pq.p, h->stream);
            if (h->split && h->has_comm()) allreduce(h, h->pq.p, n);     // this rank's share of S x
        } else {
            schur_product(h, h->pz.p, nullptr);
            launch_schur_apply(h->dev, h->pz.p, h->lin_lambda, h->pq.p, h->stream);
        }
        HIP_TRY(hipMemcpyAsync(y, h->pq.p, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    });
}

int sfmba_get_reduced(sfmba_handle* h, double* rhs, double* precond_inv) {
    return guarded(h, [&] {
        need_linearized(h, "sfmba_get_reduced");
        if (rhs) HIP_TRY(hipMemcpy(rhs, h->rhs.p, (size_t)h->n_cam * h->cd * sizeof(double), hipMemcpyDeviceToHost));
        if (precond_inv) HIP_TRY(hipMemcpy(precond_inv, h->Minv.p, (size_t)h->n_cam * h->cd * h->cd * sizeof(double), hipMemcpyDeviceToHost));
    });
}

int sfmba_precondition(sfmba_handle* h, const double* r, double* z, int32_t* coarse_dim) {
    return guarded(h, [&] {
        need_linearized(h, "sfmba_precondition");
        if (!r || !z) fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_precondition: null array");
        const size_t n = (size_t)h->n_cam * h->cd;
        // the first kernel of a PCG solve does exactly this for the right-hand side: r -> pr,
        // M_J^-1 r -> pz, state 0 armed
        HIP_TRY(hipMemcpyAsync(h->pq.p, r, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        Dev d = h->dev;
        d.rhs = h->pq.p;
        launch_pcg_init(d, 0.1, 1, h->stream);
        const bool two_level = h->explicit_schur && h->coarse_built;
        if (two_level) {
            if (!h->coarse_valid) {
                h->coarse.Einv = coarse_setup_here(h);
                h->coarse_valid = true;
            }
            launch_coarse_apply(h->dev, h->coarse, 0, h->stream);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(z, h->pz.p, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (coarse_dim) *coarse_dim = two_level ? h->coarse.n : 0;
    });
}

int sfmba_solve_step(sfmba_handle* h, double* step_cam, double* step_pt, int32_t* pcg_iters) {
    return guarded(h, [&] {
        need_linearized(h, "sfmba_solve_step");
        fetch_new2old(h);
        int it = phase_linear_solve(h, h->lin_lambda);
        if (it < 0) fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_solve_step: the reduced matrix is not positive definite at this damping");
        if (pcg_iters) *pcg_iters = it;
        DBuf<double> dp;
        dp.alloc((size_t)std::max(h->n_pts, 1) * 3);
        dp.zero(h->stream);                 // points without observations do not move
        phase_trial(h, dp.p);
        if (step_cam) HIP_TRY(hipMemcpyAsync(step_cam, h->px.p, (size_t)h->n_cam * h->cd * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        std::vector<double> hdp((size_t)h->n_pts * 3);
        if (step_pt && h->n_pts)
            HIP_TRY(hipMemcpyAsync(hdp.data(), dp.p, hdp.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (step_pt)
            for (int p = 0; p < h->n_pts; ++p)
                for (int i = 0; i < 3; ++i) step_pt[3 * (h->pt_lo + h->new2old[p]) + i] = hdp[3 * (size_t)p + i];
    });
}

int sfmba_get_structure(sfmba_handle* h, const char* name, int32_t* out, int64_t capacity, int64_t* count) {
    return guarded(h, [&] {
        need_problem(h, "sfmba_get_structure");
        if (!name || !count) fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_get_structure: null argument");
        struct Entry { const char* name; const DBuf<int>* buf; };
        const Entry table[] = {
            {"pt_ptr", &h->pt_ptr}, {"pm_cam", &h->pm_cam}, {"pm_pt", &h->pm_pt}, {"pm_orig", &h->pm_orig}, {"cm_pt", &h->cm_pt},
            {"chunk_cam", &h->chunk_cam}, {"chunk_beg", &h->chunk_beg}, {"cam_chunk_ptr", &h->cam_chunk_ptr},
            {"wtile_beg", &h->wtile_beg}, {"new2old", &h->new2old_dev},
            {"ublk_r", &h->ublk_r}, {"ublk_c", &h->ublk_c}, {"ublk_chunk_ptr", &h->ublk_chunk_ptr}, {"ublk_pos", &h->ublk_pos},
            {"ublk_pos_t", &h->ublk_pos_t}, {"ublk_diag", &h->ublk_diag}, {"pchunk_blk", &h->pchunk_blk}, {"pchunk_beg", &h->pchunk_beg},
            {"pair_p", &h->pair_p},
            {"bsr_ptr", &h->bsr_ptr}, {"bsr_col", &h->bsr_col}, {"bsr_diag", &h->bsr_diag}};
        const DBuf<int>* b = nullptr;
        for (const Entry& e : table) if (strcmp(e.name, name) == 0) b = e.buf;
        if (!b) fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_get_structure: no array named %s", name);
        const bool schur = strncmp(name, "ublk", 4) == 0 || strncmp(name, "pchunk", 6) == 0 || strncmp(name, "pair", 4) == 0 || strncmp(name, "bsr", 3) == 0;
        const int64_t n = schur && !h->explicit_schur ? 0 : (int64_t)b->n;
        *count = n;
        if (!out) return;
        if (capacity < n) fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_get_structure: %s holds %lld entries", name, (long long)n);
        if (n) HIP_TRY(hipMemcpy(out, b->p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    });
}

int sfmba_comm_unique_id(void* out128) {
    if (!out128) return SFMBA_ERR_INVALID_ARGUMENT;
    std::string err;
    if (!g_rccl.load(err)) return SFMBA_ERR_COMM;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return SFMBA_ERR_COMM;
    memcpy(out128, &id, sizeof id);
    return SFMBA_OK;
}

int sfmba_comm_init(sfmba_handle* h, const void* unique_id128) {
    return guarded(h, [&] {
        if (!unique_id128) fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_comm_init: null id");
        if (h->opt.nranks < 1 || h->opt.rank < 0 || h->opt.rank >= h->opt.nranks)
            fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_comm_init: rank %d of %d", h->opt.rank, h->opt.nranks);
        std::string err;
        if (!g_rccl.load(err)) fail(SFMBA_ERR_COMM, "%s", err.c_str());
        HIP_TRY(hipSetDevice(h->device));
        ncclUniqueId id;
        memcpy(&id, unique_id128, sizeof id);
        ncclResult_t r = g_rccl.CommInitRank(&h->comm, h->opt.nranks, id, h->opt.rank);
        if (r != ncclSuccess) { h->comm = nullptr; fail(SFMBA_ERR_COMM, "ncclCommInitRank: %s", g_rccl.GetErrorString(r)); }
        h->comm_destroy = g_rccl.CommDestroy;
    });
}

int sfmba_local_group_create(int32_t nranks, void** out) {
    if (!out || nranks < 1 || nranks > 64) return SFMBA_ERR_INVALID_ARGUMENT;
    LocalGroup* g = new (std::nothrow) LocalGroup();
    if (!g) return SFMBA_ERR_INVALID_ARGUMENT;
    g->n = nranks;
    g->bufs.assign(nranks, nullptr);
    g->counts.assign(nranks, 0);
    g->devices.assign(nranks, -1);
    *out = g;
    return SFMBA_OK;
}

int sfmba_local_group_destroy(void* group) {
    if (!group) return SFMBA_ERR_INVALID_ARGUMENT;
    delete static_cast<LocalGroup*>(group);
    return SFMBA_OK;
}

int sfmba_comm_init_local(sfmba_handle* h, void* group) {
    return guarded(h, [&] {
        LocalGroup* g = static_cast<LocalGroup*>(group);
        if (!g) fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_comm_init_local: null group");
        if (h->opt.nranks != g->n || h->opt.rank < 0 || h->opt.rank >= g->n)
            fail(SFMBA_ERR_INVALID_ARGUMENT, "sfmba_comm_init_local: rank %d of %d does not fit a group of %d", h->opt.rank, h->opt.nranks, g->n);
        {
            std::lock_guard<std::mutex> lk(g->m);
            g->devices[h->opt.rank] = h->device;
        }
        if (!g->barrier()) fail(SFMBA_ERR_COMM, "in-process group aborted: another member failed");
        // every pair of devices in the group must be able to address each other's memory
        for (int r = 0; r < g->n; ++r) {
            if (g->devices[r] == h->device) continue;
            int ok = 0;
            HIP_TRY(hipDeviceCanAccessPeer(&ok, h->device, g->devices[r]));
            if (!ok) fail(SFMBA_ERR_COMM, "device %d cannot access device %d", h->device, g->devices[r]);
            hipError_t e = hipDeviceEnablePeerAccess(g->devices[r], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) throw HipError{e, "hipDeviceEnablePeerAccess"};
            (void)hipGetLastError();
        }
        h->group = g;
    });
}

}  // extern "C"
